// Micro-benchmark (round 4): what does kernel-argument PRELOAD buy a latency-bound launch?
// gfx940+ can have the command processor place the first dwords of the kernarg segment in user SGPRs at dispatch
// (`-mllvm -amdgpu-kernarg-preload-count=N`; scalar / pointer parameters only - a by-value struct is not preloaded), so the
// wave does not start with an s_load from a cold scalar cache.  The decode step's projection launches spend ~1000 cycles between
// kernel entry and their first loads, about half of it that fetch.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=14 tools/kernarg_preload_bench.hip -o /tmp/kpb && /tmp/kpb
// Prints us per kernel of a 200-launch dependent chain in a replayed graph for: the arguments in a by-value struct (two cache
// lines touched), the same as leading scalar parameters (preloaded), and the in-kernel cycles from entry to "loads issued".
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Args {                 // like GemmArgs: what the entry needs in line 0, the epilogue's fields in line 1, padding to 320 B
    const float4* w; const float4* x; int n, k, m, pad0; long long pad1[4];
    float* out; int ldo; int pad2; long long pad3[6];
    unsigned long long* stamps; long long pad4[7];
    long long pad5[16];
};

__device__ inline void body(const float4* w, const float4* x, int n, int k, int m, float* out, int ldo, unsigned long long* stamps,
                            unsigned long long t0) {
    const int tid = threadIdx.x, b = blockIdx.x;
    // 8 independent 16-byte loads per lane, addresses from the arguments (an L2-resident 8 MB region)
    float4 acc = make_float4(0, 0, 0, 0);
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = w[((size_t)b * 8 + u) * 256 + tid + (size_t)(n & 1)];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    const float4 xv = x[tid & (k - 1)];
    out[(size_t)b * ldo + tid] = acc.x + acc.y + acc.z + acc.w + xv.x * (float)m;
    if (tid == 0 && stamps) atomicAdd(&stamps[b & 7], t1 - t0);
}

__global__ __launch_bounds__(256) void k_struct(Args a) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    body(a.w, a.x, a.n, a.k, a.m, a.out, a.ldo, a.stamps, t0);
}
__global__ __launch_bounds__(256) void k_scalar(const float4* w, const float4* x, int n, int k, int m, float* out, int ldo,
                                                unsigned long long* stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    body(w, x, n, k, m, out, ldo, stamps, t0);
}

template <typename F> double run(hipStream_t s, int n, int reps, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipGraphExecDestroy(ge);
    return us / (double(n) * reps);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int WG = 256;
    float4 *w, *x; float* out; unsigned long long* st;
    CK(hipMalloc(&w, (size_t)WG * 8 * 257 * 16)); CK(hipMalloc(&x, 4096 * 16)); CK(hipMalloc(&out, (size_t)WG * 256 * 4)); CK(hipMalloc(&st, 64));
    CK(hipMemset(w, 0, (size_t)WG * 8 * 257 * 16)); CK(hipMemset(x, 0, 4096 * 16)); CK(hipMemset(st, 0, 64));
    Args a = {}; a.w = w; a.x = x; a.n = 2; a.k = 256; a.m = 3; a.out = out; a.ldo = 256; a.stamps = st;
    const int n = 200, reps = 50;
    for (int round = 0; round < 2; ++round) {
        double ts = run(s, n, reps, [&](hipStream_t q) { hipLaunchKernelGGL(k_struct, dim3(WG), dim3(256), 0, q, a); });
        unsigned long long h[8]; CK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost)); CK(hipMemset(st, 0, 64));
        unsigned long long cs = 0; for (int i = 0; i < 8; ++i) cs += h[i];
        double tp = run(s, n, reps, [&](hipStream_t q) { hipLaunchKernelGGL(k_scalar, dim3(WG), dim3(256), 0, q, (const float4*)w, (const float4*)x, 2, 256, 3, out, 256, st); });
        CK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost)); CK(hipMemset(st, 0, 64));
        unsigned long long cp = 0; for (int i = 0; i < 8; ++i) cp += h[i];
        const double launches = (double)n * (reps + 3) * WG;
        printf("round %d: struct args %.3f us/kernel (entry -> loads issued %.0f cycles) | scalar (preloadable) args %.3f us/kernel (%.0f cycles)\n",
               round, ts, cs / launches, tp, cp / launches);
    }
    return 0;
}
