// Micro-benchmark (round 4): does a buffer that was just streamed once come back FASTER than from HBM when it is read again a few
// tens of microseconds later - i.e. can the 256 MB Infinity Cache (MALL) serve a self-attention launch's K / V stream (131 KB x S per
// layer: 171 MB at the run-average context) if something pulled it in while the latency-bound projection chain had the HBM idle?
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mall_reread_bench.hip -o /tmp/mall && /tmp/mall
// Per size: `cold` = the consumer read of a buffer nothing touched since 3 other buffers of that size went by; `warm` = the same read
// 1 launch after a first read of the SAME buffer (+ a 30 MB filler read in between, the chain's weights), for every combination of
// cache policy of the first (prefetching) read and of the consumer read (default / non-temporal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(512) void read_kernel(const f32x4* __restrict__ src, size_t n16, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n16; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += src[i];
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x] = s;
}

static void launch(bool nt, const f32x4* p, size_t bytes, float* out, hipStream_t st) {
    if (nt) read_kernel<true><<<2048, 512, 0, st>>>(p, bytes / 16, out);
    else read_kernel<false><<<2048, 512, 0, st>>>(p, bytes / 16, out);
}

int main() {
    const size_t MB = 1000 * 1000;
    const size_t sizes[] = {32 * MB, 64 * MB, 128 * MB, 171 * MB, 200 * MB, 256 * MB, 340 * MB};
    float* out; CK(hipMalloc(&out, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    f32x4* filler; CK(hipMalloc(&filler, 30 * MB)); CK(hipMemsetAsync(filler, 0, 30 * MB, st));
    const int NBUF = 4, reps = 24;
    hipEvent_t e0[reps], e1[reps];
    for (int r = 0; r < reps; ++r) { CK(hipEventCreate(&e0[r])); CK(hipEventCreate(&e1[r])); }
    for (size_t bytes : sizes) {
        std::vector<f32x4*> bufs(NBUF);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemsetAsync(b, 0, bytes, st)); }
        CK(hipStreamSynchronize(st));
        for (int mode = 0; mode < 6; ++mode) {
            // mode 0 / 1: cold consumer (nt / default).  2..5: warm: prefetch policy = (mode - 2) >> 1, consumer policy = (mode - 2) & 1  (1 = nt)
            const bool warm = mode >= 2;
            const bool pf_nt = warm && (((mode - 2) >> 1) & 1), cons_nt = warm ? ((mode - 2) & 1) : (mode == 0);
            for (int r = 0; r < reps + 2; ++r) {
                const f32x4* a = bufs[r % NBUF];
                if (warm) launch(pf_nt, a, bytes, out, st);
                launch(true, filler, 30 * MB, out, st);
                if (r >= 2) CK(hipEventRecord(e0[r - 2], st));
                launch(cons_nt, a, bytes, out, st);
                if (r >= 2) CK(hipEventRecord(e1[r - 2], st));
            }
            CK(hipStreamSynchronize(st));
            double us = 0;
            for (int r = 0; r < reps; ++r) { float ms; CK(hipEventElapsedTime(&ms, e0[r], e1[r])); us += ms * 1e3; }
            us /= reps;
            printf("%4zu MB  %-4s first read %-7s consumer read %-7s : %8.2f us  %7.1f GB/s\n", bytes / MB, warm ? "warm" : "cold",
                   warm ? (pf_nt ? "nt" : "default") : "-", cons_nt ? "nt" : "default", us, bytes / us * 1e-3);
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
