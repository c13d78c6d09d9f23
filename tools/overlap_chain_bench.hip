// Micro-benchmark (round 5): can the projection chain of a decode layer - six dependent launches of 256 workgroups,
// each ~3.5-4.7 us + a 1.2-1.5 us kernel boundary - run as OVERLAPPED launches?  Launch i + 1 starts while launch i still runs (the two
// alternate between two streams = two hardware queues), issues its once-read weight loads at once and only then waits for launch i's
// output through an in-kernel hand-off (sharded arrival counter, write-through stores, L1-bypassing loads: the third row of the table in
// MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire").  What it could hide: the boundary, the ~0.45 us
// between kernel entry and the first load, and the HBM latency + CU-load-path time of the weights.  What it adds: the hand-off.
//
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_chain_bench.hip -o /tmp/ovl && /tmp/ovl
//
// Model of one launch (shapes of the product's decode projections at 32 rows, bf16): 256 workgroups x 8 waves; a wave loads WT weight
// tiles of 1 KB (WT = 4: QKV / fc1, 32 KB per workgroup; WT = 16: fc2, 128 KB) non-temporally from a rotating 16-launch buffer (never
// cache-resident) and AT = 8 activation tiles of 1 KB (64 KB per workgroup, the same 64 KB for every workgroup: the previous launch's
// output, 256 B from each of its 256 workgroups), reduces them, runs a dependent integer tail (the MFMAs, K-split hand-over and epilogue
// of a product launch) and writes its 256 B of output.  Every word a launch reads from its predecessor is CHECKED against the tag that
// predecessor must have written in this replay (stale or torn data counts as an error); every spin is bounded and reports.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NW = 8, AT = 8, WG = 256, SHARDS = 8;
constexpr int SPIN_LIMIT = 400000;          // ~0.1-0.4 s of polling: a missing producer ends in an error report, never in a hang

struct Ctl {                                // one per benchmark, zeroed before every variant
    unsigned epoch;                         // replay number, bumped by the first node of the graph
    unsigned errors;                        // words that did not carry the expected tag
    unsigned timeouts;                      // polls that hit SPIN_LIMIT
    unsigned pad[29];
    unsigned cnt[64][SHARDS][32];           // arrival counters: launch i, shard = blockIdx % 8, one 128-byte line each
};

__global__ void bump_kernel(Ctl* c) { if (threadIdx.x == 0) c->epoch += 1; }

__device__ inline unsigned tail_work(unsigned r, int spin) {
    for (int i = 0; i < spin; ++i) r = r * 1664525u + 1013904223u;
    return r;
}
__device__ inline u32x4 load_sc1(const u32x4* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline void store_sc1(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ inline unsigned load_dword_sc1(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// MODE 0: the product's form - everything is ordered by the kernel boundary, plain loads and stores.
// MODE 1: weights first, then wait for launch `idx - 1` of this replay, then its output through L1-bypassing (`sc1`) loads.
// MODE 2: as 1, but the polling wave runs ONE agent-scope acquire (buffer_inv sc1) behind its poll and the output is read by plain loads.
template <int WT, int MODE>
__global__ __launch_bounds__(512) void proj(const u32x4* __restrict__ w, const u32x4* act_in, u32x4* act_out, Ctl* c, int idx, int spin,
                                            unsigned* __restrict__ sink) {
    constexpr bool OVL = MODE != 0;
    __shared__ unsigned s_epoch;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32x4* wp = w + ((size_t)blockIdx.x * NW + wv) * WT * 64 + lane;
    const u32x4* ap = act_in + (size_t)wv * AT * 64 + lane;
    u32x4 vw[WT], va[AT];
#pragma unroll
    for (int t = 0; t < WT; ++t) vw[t] = __builtin_nontemporal_load(wp + t * 64);
    unsigned epoch;
    if (OVL) {
        if (wv == 0) {
            // epoch was written by bump_kernel, an EARLIER dispatch in front of the fork: a plain (scalar) read is current
            const unsigned e = c->epoch;
            if (idx > 0) {
                const unsigned want = e * (WG / SHARDS);
                int it = 0;
                for (;; ++it) {
                    unsigned v = want;
                    if (lane < SHARDS) v = load_dword_sc1(&c->cnt[idx - 1][lane][0]);
                    if (__all(v >= want)) break;
                    if (it >= SPIN_LIMIT) { if (lane == 0) atomicAdd(&c->timeouts, 1u); break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            if (MODE == 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the barrier below is held until the invalidate has completed
            }
            if (lane == 0) s_epoch = e;
        }
        __syncthreads();                                        // the barrier between the poll and EVERY load of the bytes
        epoch = s_epoch;
#pragma unroll
        for (int t = 0; t < AT; ++t) va[t] = MODE == 1 ? load_sc1(ap + t * 64) : ap[t * 64];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        epoch = c->epoch;
#pragma unroll
        for (int t = 0; t < AT; ++t) va[t] = ap[t * 64];
    }
    u32x4 acc = {0, 0, 0, 0};
    unsigned bad = 0;
    const unsigned want_tag = (epoch << 8) | (unsigned)((idx - 1) & 0xff);
#pragma unroll
    for (int t = 0; t < AT; ++t) {
        if (idx > 0) bad += (va[t].x != want_tag) + (va[t].y != want_tag) + (va[t].z != want_tag) + (va[t].w != want_tag);
        acc ^= va[t];
    }
#pragma unroll
    for (int t = 0; t < WT; ++t) acc ^= vw[t];
    if (bad) atomicAdd(&c->errors, bad);
    const unsigned r = tail_work(acc.x ^ acc.y ^ acc.z ^ acc.w, spin);
    if (r == 0x12345678u) sink[blockIdx.x * 512 + tid] = r;      // keeps the loads and the tail alive
    // epilogue: this workgroup's 256 B of the next launch's activations (two whole 128-byte lines, one store instruction)
    const unsigned tag = (epoch << 8) | (unsigned)(idx & 0xff);
    if (wv == 0 && lane < 16) {
        const u32x4 o = {tag, tag, tag, tag};
        if (OVL) store_sc1(act_out + blockIdx.x * 16 + lane, o);
        else act_out[blockIdx.x * 16 + lane] = o;
    }
    if (OVL) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its stores ...
        __syncthreads();                                          // ... before the one lane that signals for all of them
        if (tid == 0) __hip_atomic_fetch_add(&c->cnt[idx][blockIdx.x % SHARDS][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct Bufs { u32x4* w; u32x4* act[2]; unsigned* sink; Ctl* ctl; size_t wbytes; };

template <int WT, int OVL> void launch_one(const Bufs& b, hipStream_t q, int i, int spin) {
    const size_t per = (size_t)WG * NW * WT * 1024 / 16;
    hipLaunchKernelGGL((proj<WT, OVL>), dim3(WG), dim3(512), 0, q, b.w + (size_t)(i % 16) * per, b.act[(i + 1) & 1], b.act[i & 1], b.ctl, i, spin, b.sink);
}

// n dependent launches per replay; returns us per launch
template <int WT, int OVL> double run(const Bufs& b, hipStream_t s0, hipStream_t s1, int n, int reps, int spin, bool quiet = false) {
    CK(hipMemset(b.ctl, 0, sizeof(Ctl)));
    CK(hipDeviceSynchronize());              // the memset runs on the NULL stream: the replays run on non-blocking ones
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s0, b.ctl);
    if (OVL) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
    for (int i = 0; i < n; ++i) launch_one<WT, OVL>(b, (OVL && (i & 1)) ? s1 : s0, i, spin);
    if (OVL) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    Ctl h; CK(hipMemcpy(&h, b.ctl, 64, hipMemcpyDeviceToHost));
    if (!quiet && (h.errors || h.timeouts || h.epoch != (unsigned)(reps + 3)))
        printf("      !! epoch %u (want %d), stale/torn words %u, poll timeouts %u\n", h.epoch, reps + 3, h.errors, h.timeouts);
    CK(hipGraphExecDestroy(ge)); CK(hipEventDestroy(fork)); CK(hipEventDestroy(join));
    return us / (double(n) * reps);
}

// Does hipExtAnyOrderLaunch (barrier bit cleared on ONE queue) overlap two launches on gfx950?  hip_ext.h says "not supported on AMD
// GFX9xx boards": measured here so that the record says what happens.  Kernel A spins ~20 us and stamps its end, kernel B stamps its start.
__global__ void stamp_kernel(unsigned long long* t, int slot, int spin) {
    if (threadIdx.x == 0 && blockIdx.x == 0) t[slot] = wall_clock64();
    unsigned r = tail_work(threadIdx.x, spin);
    if (r == 0x12345678u) t[7] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) t[slot + 1] = wall_clock64();
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    Bufs b;
    b.wbytes = (size_t)WG * NW * 16 * 1024 * 16;                 // 16 launches x 32 MB (WT = 16)
    CK(hipMalloc(&b.w, b.wbytes)); CK(hipMemset(b.w, 0, b.wbytes));
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&b.act[i], 64 * 1024)); CK(hipMemset(b.act[i], 0, 64 * 1024)); }
    CK(hipMalloc(&b.sink, (size_t)WG * 512 * 4)); CK(hipMalloc(&b.ctl, sizeof(Ctl)));

    {   // any-order launch probe
        unsigned long long* t; CK(hipMalloc(&t, 64)); CK(hipMemset(t, 0, 64));
        for (int flags = 0; flags < 2; ++flags) {
            hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s0, nullptr, nullptr, flags, t, 0, 40000);
            hipError_t e1 = hipGetLastError();
            hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s0, nullptr, nullptr, flags, t, 2, 10);
            hipError_t e2 = hipGetLastError();
            CK(hipStreamSynchronize(s0));
            unsigned long long h[4]; CK(hipMemcpy(h, t, 32, hipMemcpyDeviceToHost));
            printf("any-order probe, flags = %d (%s / %s): A runs %.1f us; B starts %.1f us after A's start (%s)\n", flags, hipGetErrorString(e1),
                   hipGetErrorString(e2), (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0,
                   h[2] < h[1] ? "OVERLAPPED" : "serialised");
        }
        CK(hipFree(t));
    }

    const int reps = 60;
    for (int round = 0; round < 2; ++round) {
        for (int spin : {0, 300, 600}) {
            const double a4 = run<4, 0>(b, s0, s1, 48, reps, spin);
            const double o4 = run<4, 1>(b, s0, s1, 48, reps, spin);
            const double q4 = run<4, 2>(b, s0, s1, 48, reps, spin);
            const double a16 = run<16, 0>(b, s0, s1, 48, reps, spin);
            const double o16 = run<16, 1>(b, s0, s1, 48, reps, spin);
            const double q16 = run<16, 2>(b, s0, s1, 48, reps, spin);
            printf("round %d tail %3d | 32 KB weights + 64 KB activations per workgroup: boundary chain %.3f us/launch, overlapped sc1-loads %.3f (%+.1f %%), "
                   "overlapped acquire %.3f (%+.1f %%) | 128 KB + 64 KB: boundary %.3f, sc1-loads %.3f (%+.1f %%), acquire %.3f (%+.1f %%)\n", round, spin,
                   a4, o4, (o4 / a4 - 1) * 100, q4, (q4 / a4 - 1) * 100, a16, o16, (o16 / a16 - 1) * 100, q16, (q16 / a16 - 1) * 100);
            fflush(stdout);
        }
    }
    return 0;
}
