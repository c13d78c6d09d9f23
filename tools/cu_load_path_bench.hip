// Micro-benchmark (round 4): the decode projections are bound by the bytes one workgroup pulls through its CU (DESIGN.md section 7:
// t = ~1.6 us + bytes / ~62 GB/s).  Does LDS-DMA (global_load_lds_dwordx4 + ds_read_b128) move a projection workgroup's 96 KB -
// 32 KB of once-read weights from HBM + 64 KB of activations every workgroup reads (L2) - faster than plain 16-byte loads into VGPRs?
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/cu_load_path_bench.hip -o /tmp/culp && /tmp/culp
// 256 workgroups x 8 waves; wave w loads 4 weight tiles + 8 activation tiles of 1 KB (the QKV / fc1 instance's burst), reduces them
// and stores one word; a chain of 200 dependent launches in a replayed graph; the weight buffer rotates over 16 x 8 MB so that it is
// never cache-resident.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NW = 8, WT = 4, AT = 8;

// `spin` > 0: a dependent integer chain of that many steps behind the loads (stands for the ~1.7 us of MFMAs, K-split hand-over and
// epilogue a product launch spends after its bytes have landed: the window in which a prefetch for the next launch could travel)
__device__ inline unsigned tail_work(unsigned r, int spin) {
    for (int i = 0; i < spin; ++i) r = r * 1664525u + 1013904223u;
    return r;
}

template <bool DMA>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ w, const u32x4* __restrict__ a, unsigned* __restrict__ out, int spin = 0) {
    __shared__ u32x4 lds[NW * (WT + AT) * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32x4* wp = w + ((size_t)blockIdx.x * NW + wv) * WT * 64 + lane;
    const u32x4* ap = a + (size_t)wv * AT * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    if (DMA) {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        u32x4* base = lds + wv * (WT + AT) * 64;
#pragma unroll
        for (int t = 0; t < AT; ++t) __builtin_amdgcn_global_load_lds((const void*)(ap + t * 64), (lds_void*)(base + t * 64), 16, 0, 0);
#pragma unroll
        for (int t = 0; t < WT; ++t) __builtin_amdgcn_global_load_lds((const void*)(wp + t * 64), (lds_void*)(base + (AT + t) * 64), 16, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < WT + AT; ++t) { const u32x4 v = base[t * 64 + lane]; acc ^= v; }
#endif
    } else {
        u32x4 va[AT], vw[WT];
#pragma unroll
        for (int t = 0; t < AT; ++t) va[t] = ap[t * 64];
#pragma unroll
        for (int t = 0; t < WT; ++t) vw[t] = __builtin_nontemporal_load(wp + t * 64);
#pragma unroll
        for (int t = 0; t < AT; ++t) acc ^= va[t];
#pragma unroll
        for (int t = 0; t < WT; ++t) acc ^= vw[t];
    }
    const unsigned r = tail_work(acc.x ^ acc.y ^ acc.z ^ acc.w, spin);
    if (r == 0x12345678u) out[blockIdx.x * 512 + tid] = r;
    if (tid == 0) out[blockIdx.x] = r;         // one word: the next launch depends on nothing but the boundary
}

// Variant P: as the VGPR-load kernel, plus - right behind its own loads - a SPARSE touch (4 bytes per 128-byte line) of the NEXT launch's
// weights: workgroup j touches the 32 KB that workgroup j of the next launch will read (same blockIdx % 8 = the same XCD's L2 under
// round-robin placement), i.e. 256 lines = 4 wave-instructions per workgroup.  The lines travel HBM -> L2 while this launch multiplies.
__global__ __launch_bounds__(512) void k_pf(const u32x4* __restrict__ w, const u32x4* __restrict__ a, const unsigned* __restrict__ wnext,
                                             unsigned* __restrict__ out, int spin) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32x4* wp = w + ((size_t)blockIdx.x * NW + wv) * WT * 64 + lane;
    const u32x4* ap = a + (size_t)wv * AT * 64 + lane;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 va[AT], vw[WT];
#pragma unroll
    for (int t = 0; t < AT; ++t) va[t] = ap[t * 64];
#pragma unroll
    for (int t = 0; t < WT; ++t) vw[t] = __builtin_nontemporal_load(wp + t * 64);
    unsigned pf = 0;
    if (wv < 4)      // 4 waves x 64 lanes x one dword per 128-byte line = the next launch's 32 KB for this block id
        pf = wnext[((size_t)blockIdx.x * NW * WT * 1024 + (size_t)(wv * 64 + lane) * 128) / 4];
#pragma unroll
    for (int t = 0; t < AT; ++t) acc ^= va[t];
#pragma unroll
    for (int t = 0; t < WT; ++t) acc ^= vw[t];
    const unsigned r = tail_work(acc.x ^ acc.y ^ acc.z ^ acc.w, spin) ^ pf;      // the prefetched words are needed only at the very end
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
    const int WG = 256, NB = 16;
    const size_t wbytes = (size_t)WG * NW * WT * 1024;          // 8 MB per launch
    u32x4 *w, *a; unsigned* out;
    CK(hipMalloc(&w, wbytes * NB)); CK(hipMalloc(&a, (size_t)NW * AT * 1024)); CK(hipMalloc(&out, (size_t)WG * 512 * 4));
    CK(hipMemset(w, 0, wbytes * NB)); CK(hipMemset(a, 0, (size_t)NW * AT * 1024));
    const int n = 192, reps = 40;
    for (int round = 0; round < 3; ++round) {
        const double tv = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<false>, dim3(WG), dim3(512), 0, q, w + (size_t)(i % NB) * (wbytes / 16), a, out, 0); });
        const double td = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<true>, dim3(WG), dim3(512), 0, q, w + (size_t)(i % NB) * (wbytes / 16), a, out, 0); });
        printf("round %d: 96 KB per workgroup (32 KB HBM + 64 KB L2) - VGPR loads %.3f us/launch | LDS-DMA + ds_read %.3f us/launch\n", round, tv, td);
        // the same with the weights cache-resident (one 8 MB buffer every launch: each XCD's L2 keeps its eighth): what a launch would
        // gain if something had pulled its weights in ahead of it
        const double tr = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<false>, dim3(WG), dim3(512), 0, q, w, a, out, 0); });
        printf("         weights cache-resident (same 8 MB every launch) - VGPR loads %.3f us/launch\n", tr);
        const double tp = run(s, n, reps, [&](hipStream_t q, int i) {
            hipLaunchKernelGGL(k_pf, dim3(WG), dim3(512), 0, q, w + (size_t)(i % NB) * (wbytes / 16), a,
                               (const unsigned*)(w + (size_t)((i + 1) % NB) * (wbytes / 16)), out, 0); });
        printf("         rotating weights, each launch sparse-touches the NEXT launch's 32 KB per workgroup - %.3f us/launch\n", tp);
        for (int spin : {400, 800}) {
            const double c = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<false>, dim3(WG), dim3(512), 0, q, w + (size_t)(i % NB) * (wbytes / 16), a, out, spin); });
            const double h = run(s, n, reps, [&](hipStream_t q, int i) { hipLaunchKernelGGL(k<false>, dim3(WG), dim3(512), 0, q, w, a, out, spin); });
            const double f = run(s, n, reps, [&](hipStream_t q, int i) {
                hipLaunchKernelGGL(k_pf, dim3(WG), dim3(512), 0, q, w + (size_t)(i % NB) * (wbytes / 16), a,
                                   (const unsigned*)(w + (size_t)((i + 1) % NB) * (wbytes / 16)), out, spin); });
            printf("         with a %d-step dependent tail behind the loads: cold weights %.3f | resident %.3f | cold + prefetch of the next launch's %.3f us/launch\n",
                   spin, c, h, f);
        }
    }
    return 0;
}
