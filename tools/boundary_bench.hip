// Micro-benchmark: cost of a dependent kernel boundary on this stack (hipGraph replay vs eager stream).
//   hipcc --offload-arch=gfx950 -O3 tools/boundary_bench.hip -o /tmp/boundary_bench && /tmp/boundary_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_chain(int* p) { int v = p[0]; p[1] = v + 1; }            // one dependent load + store
__global__ void k_wide(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.0f; }   // 256 WGs, RMW

template <typename F> double run(hipStream_t s, int n, int reps, bool graph, F launch) {
    hipGraphExec_t ge = nullptr;
    if (graph) {
        hipGraph_t g;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch(s);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
    }
    for (int w = 0; w < 2; ++w) { if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n; ++i) launch(s); }
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n; ++i) launch(s); }
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (ge) hipGraphExecDestroy(ge);
    return us / (double(n) * reps);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* p; float* q; CK(hipMalloc(&p, 64)); CK(hipMalloc(&q, 256 * 256 * 4)); CK(hipMemset(p, 0, 64)); CK(hipMemset(q, 0, 256 * 256 * 4));
    const int n = 200, reps = 50;
    printf("us per kernel (chain of %d dependent launches)\n", n);
    printf("  empty   graph %.2f   eager %.2f\n", run(s, n, reps, true, [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }),
           run(s, n, reps, false, [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }));
    printf("  chain   graph %.2f   eager %.2f\n", run(s, n, reps, true, [&](hipStream_t st) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(1), 0, st, p); }),
           run(s, n, reps, false, [&](hipStream_t st) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(1), 0, st, p); }));
    printf("  wide    graph %.2f   eager %.2f\n", run(s, n, reps, true, [&](hipStream_t st) { hipLaunchKernelGGL(k_wide, dim3(256), dim3(256), 0, st, q); }),
           run(s, n, reps, false, [&](hipStream_t st) { hipLaunchKernelGGL(k_wide, dim3(256), dim3(256), 0, st, q); }));
    return 0;
}
