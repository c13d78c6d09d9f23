// Census (round 5): how many workgroups of one launch are RESIDENT on the chip at once?  The fused QKV + self-attention launch (704
// workgroups of 512 threads, 72 VGPRs, 18 KB of LDS, barriers) showed exactly 512 resident - two per CU - although registers, LDS and
// wave slots allow three (profiles/r05_stamps_decode_step.txt).  Every workgroup of a grid announces itself, then spins until all have
// announced or ~1 ms has passed; the number that had announced when the first workgroup gave up is the residency.
//   hipcc --offload-arch=gfx950 -O3 tools/wg_residency_bench.hip -o /tmp/wgres && /tmp/wgres
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS, int VG, bool BARRIER>
__global__ __launch_bounds__(THREADS) void census(unsigned* ctl, int nwg) {
    extern __shared__ unsigned lds[];
    // hold VG live registers across the spin
    unsigned r[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) r[i] = threadIdx.x * (i + 1) + ctl[8 + (i & 7)];
#pragma unroll
    for (int i = 0; i < VG; ++i) asm volatile("" : "+v"(r[i]));         // pin every value in a register of its own across the spin
    if (BARRIER) __syncthreads();
    if (threadIdx.x == 0) {
        lds[0] = 1;
        atomicAdd(&ctl[0], 1u);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg) {
            if (wall_clock64() - t0 > 100000ull) {       // 1 ms at 100 MHz
                unsigned seen = __hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicMin(&ctl[1], seen);
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    if (BARRIER) __syncthreads();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < VG; ++i) asm volatile("" : "+v"(r[i]));
#pragma unroll
    for (int i = 0; i < VG; ++i) acc ^= r[i];
    if (acc == 0x12345u) ctl[2] = acc;
}

template <int THREADS, int VG, bool BARRIER> void run(unsigned* ctl, int nwg, size_t ldsb, const char* what) {
    unsigned h[16] = {0, 0xffffffffu};
    CK(hipMemcpy(ctl, h, sizeof(h), hipMemcpyHostToDevice));
    hipLaunchKernelGGL((census<THREADS, VG, BARRIER>), dim3(nwg), dim3(THREADS), ldsb, 0, ctl, nwg);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, ctl, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-58s grid %4d: resident at once %4u%s\n", what, nwg, h[1] == 0xffffffffu ? h[0] : h[1], h[1] == 0xffffffffu ? " (all)" : "");
}

int main() {
    unsigned* ctl; CK(hipMalloc(&ctl, 64));
    run<512, 56, true>(ctl, 1024, 18432, "512 threads, VG 56, 18 KB LDS, barrier");
    run<512, 64, true>(ctl, 1024, 18432, "512 threads, VG 64, 18 KB LDS, barrier");
    run<512, 72, true>(ctl, 1024, 18432, "512 threads, VG 72, 18 KB LDS, barrier");
    run<512, 88, true>(ctl, 1024, 18432, "512 threads, VG 88, 18 KB LDS, barrier");
    run<512, 104, true>(ctl, 1024, 18432, "512 threads, VG 104, 18 KB LDS, barrier");
    run<512, 120, true>(ctl, 1024, 18432, "512 threads, VG 120, 18 KB LDS, barrier");
    run<512, 64, true>(ctl, 1024, 1024, "512 threads, VG 64, 1 KB LDS, barrier");
    run<512, 64, false>(ctl, 1024, 0, "512 threads, VG 64, no LDS, no barrier");
    run<256, 64, true>(ctl, 2048, 9216, "256 threads, VG 64, 9 KB LDS, barrier");
    run<1024, 64, true>(ctl, 512, 36864, "1024 threads, VG 64, 36 KB LDS, barrier");
    return 0;
}
