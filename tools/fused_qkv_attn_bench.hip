// Micro-benchmark (round 5): QKV projection + self-attention of a decode layer as ONE launch.
//
// Today (per layer at the run-average context): [gap 1.21][QKV 3.57][gap 1.49][self-attention 25.86] us.  The
// attention launch is HBM-bound (171 MB at 6.6 TB/s inside the launch) while the QKV launch in front of it is latency-bound and leaves
// the HBM idle (6.3 MB in 3.6 us).  In a fused launch the 512 attention workgroups are dispatched right behind the 192 projection
// workgroups (same grid, higher block ids), issue their first K bursts at once - 64 KB per workgroup in registers, 32 MB chip-wide -
// and wait for THEIR head's q / k / v (12 of the 192 projection workgroups) through a per-head arrival counter.  What that could
// hide: the boundary between the two launches, the attention's own start-up, and ~20 % of its bytes under the projection.
// What it costs: the hand-off (write-through stores, one atomic per producing workgroup, a polling wave per consumer), and the
// projection's loads queueing behind 32 MB of prefetch.  This file measures the pair both ways before anything is built.
//
//   hipcc --offload-arch=gfx950 -O3 tools/fused_qkv_attn_bench.hip -o /tmp/fqa && /tmp/fqa
//
// Model (bf16 sizes of mini_v0.1 at batch 32): projection workgroup = 8 waves x 4 weight tiles of 1 KB (rotating buffers, non-temporal)
// + 64 KB of activations every workgroup reads, a dependent tail, 2 KB of output (fp32 q for 16 features x 32 rows).  Attention
// workgroup (head, utterance) = 8 waves streaming S rows of 128 B of K, then of V, in bursts of 4 x 1 KB per wave, double-buffered
// (the product's loop shape), scores through LDS, a barrier, a second pass, 64 outputs.  Every q word an attention workgroup reads
// is checked against the tag its producers must have written in this replay; spins are bounded.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NW = 8, WT = 4, AT = 8, NPROJ = 192, H = 16, B = 32, UNR = 4, LAYERS = 24, NBUF = 12;
constexpr int SPIN_LIMIT = 400000;

struct Ctl {
    unsigned epoch, errors, timeouts, pad[29];
    unsigned cnt[LAYERS][H][32];            // per (layer, head): arrivals of the 12 projection workgroups that produce its q / k / v
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

struct Args {
    const u32x4* w;        // this layer's projection weights (192 x 8 x 4 KB)
    const u32x4* x;        // 64 KB of activations
    u32x4* q;              // [B][H][16 lanes of 16 B] : 256 B per (utterance, head), written by the projection, read by the attention
    const u32x4* kv;       // this layer's cache: [B][H][2][S][128 B]
    float* out;            // [B][H][64]
    Ctl* c;
    unsigned* sink;
    int layer, S, spin;
};

// projection workgroup n of 192: n / 64 = q | k | v, head = (n % 64) / 4.  Only the q blocks' output is read back (k / v rows of
// the step go to the cache in the product; their arrival still counts).
template <bool FUSED> __device__ inline void proj_body(const Args& a, int n, unsigned epoch) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32x4* wp = a.w + ((size_t)n * NW + wv) * WT * 64 + lane;
    const u32x4* ap = a.x + (size_t)wv * AT * 64 + lane;
    u32x4 vw[WT], va[AT];
#pragma unroll
    for (int t = 0; t < WT; ++t) vw[t] = __builtin_nontemporal_load(wp + t * 64);
#pragma unroll
    for (int t = 0; t < AT; ++t) va[t] = ap[t * 64];
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < AT; ++t) acc ^= va[t];
#pragma unroll
    for (int t = 0; t < WT; ++t) acc ^= vw[t];
    const unsigned r = tail_work(acc.x ^ acc.y ^ acc.z ^ acc.w, a.spin);
    if (r == 0x12345678u) a.sink[n * 512 + tid] = r;
    const int head = (n % 64) / 4, part = n / 64, quarter = n % 4;
    // 16 features x 32 rows of fp32 = 2 KB: for the q part, 64 B of each utterance's 256-B q row of this head
    const unsigned tag = (epoch << 8) | (unsigned)a.layer;
    if (part == 0 && wv < 2 ) {
        // lane -> (utterance = wv * 16 + lane / 4, 16-byte piece = lane % 4) of this quarter
        const int bu = wv * 16 + (lane >> 2), piece = quarter * 4 + (lane & 3);
        u32x4* dst = a.q + ((size_t)bu * H + head) * 16 + piece;
        const u32x4 o = {tag, tag, tag, tag};
        if (FUSED) store_sc1(dst, o); else *dst = o;
    }
    if (FUSED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&a.c->cnt[a.layer][head][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <bool FUSED> __device__ inline void attn_body(const Args& a, int h, int b, unsigned epoch_in, float* sc) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tt = lane >> 3, cch = lane & 7;                   // token in tile, 16-byte chunk of the 128-B row
    const int S = a.S, ntiles = (S + 7) / 8, STRIDE = NW * UNR;
    const u32x4* kbase = a.kv + (((size_t)b * H + h) * 2) * (size_t)S * 8;
    const u32x4* vbase = kbase + (size_t)S * 8;
    u32x4 bufA[UNR], bufB[UNR];
    auto load_burst = [&](u32x4 (&buf)[UNR], int t0, const u32x4* base) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            int tok = (t0 + u * NW) * 8 + tt;
            if (tok > S - 1) tok = S - 1;
            buf[u] = __builtin_nontemporal_load(base + (size_t)tok * 8 + cch);
        }
    };
    // early bursts: addresses depend on nothing but kernel arguments
    load_burst(bufA, w, kbase);
    unsigned epoch = epoch_in;
    u32x4 qv;
    __shared__ unsigned s_epoch;
    if (FUSED) {
        if (STRIDE + w < ntiles) load_burst(bufB, w + STRIDE, kbase);       // second burst in flight too: 64 KB per workgroup
        if (w == 0) {
            const unsigned e = a.c->epoch;
            const unsigned want = e * 12u;
            int it = 0;
            for (;; ++it) {
                const unsigned v = load_dword_sc1(&a.c->cnt[a.layer][h][0]);
                if (v >= want) break;
                if (it >= SPIN_LIMIT) { if (lane == 0) atomicAdd(&a.c->timeouts, 1u); break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (lane == 0) s_epoch = e;
        }
        __syncthreads();
        epoch = s_epoch;
        qv = load_sc1(a.q + ((size_t)b * H + h) * 16 + (lane & 15));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        qv = a.q[((size_t)b * H + h) * 16 + (lane & 15)];
    }
    const unsigned want_tag = (epoch << 8) | (unsigned)a.layer;
    const unsigned bad = (qv.x != want_tag) + (qv.y != want_tag) + (qv.z != want_tag) + (qv.w != want_tag);
    if (bad) atomicAdd(&a.c->errors, bad);
    const float qf = (float)(qv.x & 0xff) * 1e-3f + 0.5f;
    auto score_burst = [&](const u32x4 (&buf)[UNR], int t0) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int tok = (t0 + u * NW) * 8 + tt;
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d = fmaf(qf, __builtin_bit_cast(float, buf[u][i] << 16), d);
                d = fmaf(qf, __builtin_bit_cast(float, buf[u][i] & 0xffff0000u), d);
            }
            d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
            if (cch == 0 && tok < S) sc[tok] = d;
        }
    };
    {   // pass 1
        int t0 = w;
        bool haveB = FUSED;
        while (true) {
            bool more = t0 + STRIDE < ntiles;
            if (more && !haveB) load_burst(bufB, t0 + STRIDE, kbase);
            haveB = false;
            score_burst(bufA, t0);
            if (!more) break;
            t0 += STRIDE;
            more = t0 + STRIDE < ntiles;
            if (more) load_burst(bufA, t0 + STRIDE, kbase);
            score_burst(bufB, t0);
            if (!more) break;
            t0 += STRIDE;
        }
    }
    load_burst(bufA, w, vbase);
    __syncthreads();
    float m = 0.f;
    for (int t = tid; t < S; t += NW * 64) m += sc[t];
    for (int o = 32; o; o >>= 1) m += __shfl_xor(m, o);
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto pv_burst = [&](const u32x4 (&buf)[UNR], int t0) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int tok = (t0 + u * NW) * 8 + tt;
            const float p = (tok < S) ? sc[tok] + m : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = fmaf(p, __builtin_bit_cast(float, buf[u][i] << 16), acc[2 * i]);
                acc[2 * i + 1] = fmaf(p, __builtin_bit_cast(float, buf[u][i] & 0xffff0000u), acc[2 * i + 1]);
            }
        }
    };
    {   // pass 2
        int t0 = w;
        while (true) {
            bool more = t0 + STRIDE < ntiles;
            if (more) load_burst(bufB, t0 + STRIDE, vbase);
            pv_burst(bufA, t0);
            if (!more) break;
            t0 += STRIDE;
            more = t0 + STRIDE < ntiles;
            if (more) load_burst(bufA, t0 + STRIDE, vbase);
            pv_burst(bufB, t0);
            if (!more) break;
            t0 += STRIDE;
        }
    }
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o += acc[j];
    for (int s = 32; s; s >>= 1) o += __shfl_xor(o, s);
    if (lane == 0) a.out[((size_t)b * H + h) * 64 + w] = o;
}

__global__ __launch_bounds__(512) void proj_kernel(Args a) { proj_body<false>(a, blockIdx.x, a.c->epoch); }
__global__ __launch_bounds__(512) void attn_kernel(Args a) {
    extern __shared__ float sc[];
    attn_body<false>(a, blockIdx.x % H, blockIdx.x / H, a.c->epoch, sc);
}
__global__ __launch_bounds__(512) void fused_kernel(Args a) {
    extern __shared__ float sc[];
    if (blockIdx.x < NPROJ) proj_body<true>(a, blockIdx.x, a.c->epoch);
    else attn_body<true>(a, (blockIdx.x - NPROJ) % H, (blockIdx.x - NPROJ) / H, 0u, sc);
}

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int Smax = 2600;
    const size_t wbytes = (size_t)NPROJ * NW * WT * 1024, kvbytes = (size_t)B * H * 2 * Smax * 128;
    u32x4 *w, *x, *q, *kv; float* out; unsigned* sink; Ctl* ctl;
    CK(hipMalloc(&w, wbytes * NBUF)); CK(hipMalloc(&x, 64 * 1024)); CK(hipMalloc(&q, (size_t)B * H * 256)); CK(hipMalloc(&kv, kvbytes * NBUF));
    CK(hipMalloc(&out, (size_t)B * H * 64 * 4)); CK(hipMalloc(&sink, (size_t)NPROJ * 512 * 4)); CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMemset(w, 0, wbytes * NBUF)); CK(hipMemset(x, 0, 64 * 1024)); CK(hipMemset(kv, 0x11, kvbytes * NBUF)); CK(hipMemset(q, 0, (size_t)B * H * 256));
    const int reps = 30;
    for (int S : {128, 700, 1300, 2500}) {
        for (int spin : {60, 110}) {
            double t[2];
            for (int fused = 0; fused < 2; ++fused) {
                CK(hipMemset(ctl, 0, sizeof(Ctl)));
                CK(hipDeviceSynchronize());          // the memset runs on the NULL stream: the replays below run on a non-blocking one
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, s, ctl);
                for (int l = 0; l < LAYERS; ++l) {
                    Args a{w + (size_t)(l % NBUF) * (wbytes / 16), x, q, kv + (size_t)(l % NBUF) * (kvbytes / 16), out, ctl, sink, l, S, spin};
                    const size_t lds = (size_t)Smax * 4;
                    if (fused) hipLaunchKernelGGL(fused_kernel, dim3(NPROJ + H * B), dim3(512), lds, s, a);
                    else {
                        hipLaunchKernelGGL(proj_kernel, dim3(NPROJ), dim3(512), 0, s, a);
                        hipLaunchKernelGGL(attn_kernel, dim3(H * B), dim3(512), lds, s, a);
                    }
                }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphDestroy(g));
                for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                t[fused] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (double(LAYERS) * reps);
                Ctl hc; CK(hipMemcpy(&hc, ctl, 64, hipMemcpyDeviceToHost));
                if (hc.errors || hc.timeouts || hc.epoch != (unsigned)(reps + 3))
                    printf("   !! fused=%d: epoch %u (want %d), wrong q words %u, poll timeouts %u\n", fused, hc.epoch, reps + 3, hc.errors, hc.timeouts);
                CK(hipGraphExecDestroy(ge));
            }
            const double mb = (double)B * H * 2 * S * 128 / 1e6;
            printf("S %4d (%.0f MB of K/V per layer) tail %3d | projection + attention as two launches %.2f us per layer | fused launch %.2f us (%+.2f us, %+.1f %%)\n",
                   S, mb, spin, t[0], t[1], t[1] - t[0], (t[1] / t[0] - 1) * 100);
            fflush(stdout);
        }
    }
    return 0;
}
