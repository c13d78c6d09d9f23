// Micro-benchmark (round 4): does the 256 MB Infinity Cache shorten a LATENCY-bound projection launch?  The decoder's weights (725 MB in
// bf16) cycle through it once per step, so nothing survives; but if cached weights were worth ~1 us per launch (as L2-resident ones are,
// tools/cu_load_path_bench.hip) one could keep a third of the layers' weights on the default cache policy and stream the rest non-temporally.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mall_latency_bench.hip -o /tmp/mallat && /tmp/mallat
// The bare loader of a QKV / fc1 workgroup (32 KB of weights + 64 KB of shared activations per workgroup, 256 workgroups, chain of dependent
// launches in a replayed graph) over a ROTATING set of weight buffers of 8 MB each: 1 buffer (L2-resident), 8 / 16 / 24 (64-192 MB: fits
// the Infinity Cache, not the L2s), 64 / 128 (0.5-1 GB: HBM), with default-policy and with non-temporal weight loads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NW = 8, WT = 4, AT = 8;

template <bool NT>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ w, const u32x4* __restrict__ a, unsigned* __restrict__ out) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32x4* wp = w + ((size_t)blockIdx.x * NW + wv) * WT * 64 + lane;
    const u32x4* ap = a + (size_t)wv * AT * 64 + lane;
    u32x4 acc = {0, 0, 0, 0}, va[AT], vw[WT];
#pragma unroll
    for (int t = 0; t < AT; ++t) va[t] = ap[t * 64];
#pragma unroll
    for (int t = 0; t < WT; ++t) vw[t] = NT ? __builtin_nontemporal_load(wp + t * 64) : wp[t * 64];
#pragma unroll
    for (int t = 0; t < AT; ++t) acc ^= va[t];
#pragma unroll
    for (int t = 0; t < WT; ++t) acc ^= vw[t];
    const unsigned r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[blockIdx.x * 512 + tid] = r;
    if (tid == 0) out[blockIdx.x] = r;
}

template <typename F> double run(hipStream_t s, int n, int reps, F launch) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) launch(s, i);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    hipGraphExecDestroy(ge);
    return us / (double(n) * reps);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int WG = 256, NBMAX = 128;
    const size_t wbytes = (size_t)WG * NW * WT * 1024;          // 8 MB per launch
    u32x4 *w, *a; unsigned* out;
    CK(hipMalloc(&w, wbytes * NBMAX)); CK(hipMalloc(&a, (size_t)NW * AT * 1024)); CK(hipMalloc(&out, (size_t)WG * 512 * 4));
    CK(hipMemset(w, 0, wbytes * NBMAX)); CK(hipMemset(a, 0, (size_t)NW * AT * 1024));
    const int n = 384, reps = 20;
    for (int nb : {1, 8, 16, 24, 64, 128}) {
        const double tp = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<false>, dim3(WG), dim3(512), 0, q, w + (size_t)(i % nb) * (wbytes / 16), a, out); });
        const double tn = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<true>, dim3(WG), dim3(512), 0, q, w + (size_t)(i % nb) * (wbytes / 16), a, out); });
        printf("rotating over %3d x 8 MB = %4d MB of weights: default-policy loads %.3f us/launch | non-temporal loads %.3f us/launch\n", nb, nb * 8, tp, tn);
    }
    return 0;
}
