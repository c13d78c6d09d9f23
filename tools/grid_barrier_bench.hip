// What does a phase boundary cost inside ONE persistent kernel on MI355X, compared with the 1.55 us kernel boundary
// of a hipGraph (tools/boundary_bench.hip)?  Every wave has a bounded spin, so the grid always drains.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_bench.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LD_ACQ(p) __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
#define LD_RLX(p) __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ADD_REL(p, v) __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
#define ST_REL(p, v) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)

// variant 1: arrivals add to a counter nobody spins on; the last arriver publishes the generation in a separate line.
// variant 2: 8 shard counters (blockIdx % 8, i.e. one per XCD) feeding a top counter, same release flag.
// ctr layout (unsigned, 64-byte lines): [0] arrivals, [16] flag, [32 + 16*s] shard s, [160] top
__device__ __forceinline__ void grid_barrier_v(unsigned* ctr, unsigned gen, unsigned* err, int variant, int sleep) {
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last;
        if (variant == 1) last = ADD_REL(ctr, 1u) == gen * gridDim.x - 1;
        else {
            unsigned s = blockIdx.x & 7, per = gridDim.x >> 3;
            last = false;
            if (ADD_REL(ctr + 32 + 16 * s, 1u) == gen * per - 1) last = ADD_REL(ctr + 160, 1u) == gen * 8 - 1;
        }
        if (last) ST_REL(ctr + 16, gen);
        else {
            unsigned spins = 0;
            while (LD_RLX(ctr + 16) < gen) {
                if (sleep) __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 22)) { *err = 1; break; }
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);                 // agent scope: writes of this WG are visible first
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *err = 1; break; }             // exit condition every wave reaches
        }
    }
    __syncthreads();
    return true;
}

// mode 0: barrier only.  mode 1: each phase every WG writes 16 B/lane (a [32 x 1024] bf16 activation = 64 KB in total is
// written by WGs 0..7), then after the barrier every WG reads the whole 64 KB (what a decode GEMM's prologue does).
__global__ __launch_bounds__(512) void persist_kernel(unsigned* ctr, unsigned* err, f32x4* act, float* out, int phases, int mode, int variant, int sleep) {
    f32x4 acc = {0, 0, 0, 0};
    for (int p = 0; p < phases; ++p) {
        if (mode == 1) {
            f32x4* buf = act + (size_t)(p & 1) * 4096;
            if (blockIdx.x < 8) buf[blockIdx.x * 512 + threadIdx.x] = f32x4{(float)p, 1.f, 2.f, 3.f};
            if (variant) grid_barrier_v(ctr, p + 1, err, variant, sleep); else grid_barrier(ctr, (unsigned)(p + 1) * gridDim.x, err);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += buf[i * 512 + threadIdx.x];
        } else {
            if (variant) grid_barrier_v(ctr, p + 1, err, variant, sleep); else grid_barrier(ctr, (unsigned)(p + 1) * gridDim.x, err);
        }
    }
    if (acc.x == -1.f) out[0] = acc.y;
    if (mode == 1 && blockIdx.x == 0 && threadIdx.x == 0) out[1] = acc.x;     // = 8 * sum(p) when every read saw fresh data
}

int main() {
    unsigned *ctr, *err; f32x4* act; float* out;
    CK(hipMalloc(&ctr, 1024)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&act, 2 * 65536)); CK(hipMalloc(&out, 64));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 2000;
    for (int variant = 0; variant < 3; ++variant) for (int sleep = 0; sleep < 2; ++sleep)
    for (int mode = 0; mode < 2; ++mode)
        for (int g : {64, 128, 256, 512}) {
            if (variant == 0 && sleep == 0) continue;
            CK(hipMemsetAsync(ctr, 0, 1024, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(out, 0, 64, st));
            CK(hipEventRecord(e0, st));
            persist_kernel<<<g, 512, 0, st>>>(ctr, err, act, out, phases, mode, variant, sleep);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned herr; float hout[2];
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hout, out, 8, hipMemcpyDeviceToHost));
            double expect = 8.0 * (double)(phases - 1) * phases / 2;
            printf("variant %d sleep %d  mode %d  grid %4d x512  %6.3f us/phase  timeout %u  check %s\n", variant, sleep, mode, g, ms * 1e3 / phases, herr,
                   mode == 0 ? "-" : (hout[1] == (float)expect ? "fresh" : "STALE"));
        }
    return 0;
}
