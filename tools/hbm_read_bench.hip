// Practical HBM streaming-read ceiling on MI355X, for reading the attention kernel's roofline fraction against:
// a grid-strided 16-byte-per-lane read of `bytes` with a trivial reduction, at the footprints the self-attention
// kernel touches (170 MB = mean step, 340 MB = last step) and one far larger than the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bench.hip -o /tmp/hbm && /tmp/hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int UNR>
__global__ __launch_bounds__(512) void read_kernel(const f32x4* __restrict__ src, size_t n16, float* __restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + (UNR - 1) * stride < n16; i += UNR * stride) {
        f32x4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u];
    }
    for (; i < n16; i += stride) acc += src[i];
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x] = s;      // keeps the loads alive, never true for zero-filled memory
}

int main() {
    const size_t sizes[] = {170000384ull, 340000768ull, 2048ull << 20};
    const int grids[] = {512, 1024, 2048, 4096, 8192};
    float* out; CK(hipMalloc(&out, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t bytes : sizes) {
        // 8 distinct buffers rotated so that consecutive launches never re-read what the previous one left in cache
        const int NBUF = bytes > (1ull << 30) ? 2 : 8;
        std::vector<f32x4*> bufs(NBUF);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemsetAsync(b, 0, bytes, st)); }
        CK(hipStreamSynchronize(st));
        for (int g : grids) {
            const int reps = 40;
            for (int w = 0; w < 4; ++w) read_kernel<4><<<g, 512, 0, st>>>(bufs[w % NBUF], bytes / 16, out);
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) read_kernel<4><<<g, 512, 0, st>>>(bufs[r % NBUF], bytes / 16, out);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double us = ms * 1e3 / reps;
            printf("bytes %11zu  grid %5d x512  %8.2f us/launch  %7.1f GB/s\n", bytes, g, us, bytes / us * 1e-3);
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
