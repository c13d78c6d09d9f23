// QKV projection + self-attention of a decode layer as ONE launch (round 5).
//
// Per layer the decode step ran [boundary 1.2 us][QKV projection 3.6 us][boundary 1.5 us][self-attention 25.9 us]: the attention
// launch is HBM-bound (171 MB at 6.6 TB/s inside the launch) while the projection in front of it is latency-bound and leaves the
// HBM idle.  Here ONE grid holds both: blocks 0 .. n_gemm - 1 are the projection workgroups (gemm_device.inc, the folded-LayerNorm
// QKV instance, unchanged arithmetic), the H x M blocks behind them are the attention workgroups (attn_device.inc).  Workgroups
// are dispatched in block order, so every projection workgroup is resident before the first attention workgroup starts; the
// attention workgroups issue their first two K bursts at once (rows this launch does not write: 64 KB per workgroup in registers,
// 32 MB chip-wide, under the projection) and then wait for THEIR head's q / k / v - twelve of the projection workgroups - through
// per-(layer, m-tile, head) arrival tags.  Hand-off form: MI355X_MICROARCH.md "Valid forms" (write-through payload stores, every
// storing wave drained, workgroup barrier, one lane signals; consumer: one wave polls with L1-bypassing loads, workgroup barrier,
// then loads).  tools/fused_qkv_attn_bench.hip measured the pair before it was built (profiles/r05_fused_qkv_attn_bench.txt:
// -1.9 .. -2.5 us per layer at the run-average context).  Results are BIT-IDENTICAL to the two launches (tests): the same
// products in the same order on both sides, only the store / load instructions of the handed-over rows differ.
//
// Deadlock freedom: an attention workgroup spins only on tags of projection workgroups of the same grid with LOWER block ids,
// which the dispatcher has already placed (it walks a grid in block order and never skips); the spin is bounded anyway and
// reports through `err` (engine.hip turns that into an error of the call).
#include "ptts_common.h"
#include <hip/hip_ext.h>
#include "fused.h"
#include <cstdlib>

namespace {

// 8- and 16-byte write-through (agent-scope) stores: global_store_dwordx2 ... sc1
__device__ inline void fused_store8_wt(void* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void fused_store16_wt(void* p, float4 v) {
    fused_store8_wt(p, make_uint2(__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)));
    fused_store8_wt(reinterpret_cast<char*>(p) + 8, make_uint2(__builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)));
}
// L1-bypassing loads (the compiler does not count inline-asm loads: each statement waits for its own)
__device__ inline int fused_load_dword_sc1(const int* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int EPL> __device__ inline void fused_load_q_sc1(const float* p, float (&q)[EPL]);
template <> __device__ inline void fused_load_q_sc1<4>(const float* p, float (&q)[4]) {
    f32x4 a;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(p) : "memory");
    q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = a[3];
}
template <> __device__ inline void fused_load_q_sc1<8>(const float* p, float (&q)[8]) {
    f32x4 a, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
    q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; q[3] = a[3]; q[4] = b[0]; q[5] = b[1]; q[6] = b[2]; q[7] = b[3];
}

#define PTTS_GEMM_FUSED 1
#define PTTS_GEMM_SIGNATURE __device__ __forceinline__ void gemm_body(GemmArgs a, const int gemm_bx, const int gemm_by, int* fused_arrive, \
                                                                      const int fused_mtiles, const int fused_heads)
#define PTTS_GEMM_BX gemm_bx
#define PTTS_GEMM_BY gemm_by
#include "gemm_device.inc"

#define PTTS_ATTN_FUSED 1
#define PTTS_FUSED_GIVE_UP_TICKS 200000000ull      // s_memrealtime ticks (100 MHz): 2 s
#define PTTS_ATTN_SIGNATURE __device__ __forceinline__ void attn_body(AttnArgs a, const int attn_h, const int attn_b, const int* fused_arrive, \
                                                                      const int fused_mtiles, int* fused_err, const int fused_pre)
#define PTTS_ATTN_H attn_h
#define PTTS_ATTN_B attn_b
#include "attn_device.inc"

struct FusedArgs {
    GemmArgs g;
    AttnArgs a;
    int* arrive;
    int* arrive_prod;                  // where the projection half stores its tags: = arrive (a test hook points it elsewhere: the attention half then times out)
    int* err;
    int n_gemm, nblk, mtiles, heads;
    int pre;                           // early K bursts per attention workgroup (0 .. 2)
};

// diagnostic build only: the in-kernel stamps cost 16 registers (84 instead of 68: two workgroups per CU instead of three, i.e. a third
// of the attention workgroups would start only when the projection workgroups have left - not the product's timeline); hold the
// stamped kernel to the product's occupancy class
#ifdef PTTS_STAMPS
#define PTTS_FUSED_OCC __attribute__((amdgpu_waves_per_eu(6)))
#else
#define PTTS_FUSED_OCC
#endif
template <typename P, int TILES>
__global__ __launch_bounds__(512) PTTS_FUSED_OCC void fused_qkv_attn_kernel(FusedArgs f) {
    const int i = blockIdx.x;
    if (i < f.n_gemm) {
        gemm_body<P, 8, 1, false, 1, 2, true, EPI_QKV_SELF, ACT_NONE, 0, TILES, false>(f.g, i % f.nblk, i / f.nblk, f.arrive_prod, f.mtiles, f.heads);
    } else {
        const int j = i - f.n_gemm;
        attn_body<P, P, true>(f.a, j % f.heads, j / f.heads, f.arrive, f.mtiles, f.err, f.pre);
    }
}

// exact-fit tile bursts (K / KSTEP = 8 waves x TILES): 4 = mini_v0.1 (D = 1024, bf16), 6 = BASELINE configs[4] (D = 1536), 8 = D = 2048 (or
// D = 1024 in the fp32 mode); 0 = the guarded generic instance for every other width (134 registers: the attention workgroups
// then join a CU only when its projection workgroup has left - correct, less overlap)
template <typename P> int tiles_for(const GemmArgs& g) {
    const int ksteps = g.K / P::KSTEP;
    if (ksteps == 32) return 4;
    if (ksteps == 48) return 6;
    if (ksteps == 64) return 8;
    return 0;
}

}  // namespace

bool fused_qkv_attn_supported(const GemmArgs& g, const AttnArgs& a, int dtype) {
    // (PTTS_NO_FUSED_QKV_ATTN=1 - A/B and the bit-identity tests - is read once, at engine creation: ptts_engine::fuse_qkv_attn)
    // PTTS_GEMM_GENERIC=1 sends the unfused projection to the generic instance: the pair is then two launches as well, so that the
    // switch changes every projection of the step and not all but this one
    // (read per call like gemm.hip's own test of it - eager launches and graph capture only, never a replayed step - because the
    // bit-identity test flips it inside one process)
    if (getenv("PTTS_GEMM_GENERIC")) return false;
    if (!g.fold || g.W2 || g.bias || g.rowmask || g.force_whole || g.rope_k || g.npos != 0 || g.epi != EPI_QKV_SELF || g.pro != PRO_FRAG) return false;
    if (g.N != 3 * g.D || g.D % 64 != 0 || g.M < 1) return false;
    if (a.pool.block_table || a.lnfix_qkv || !a.len_from_state || a.rows_per_utt != 1 || a.npos != 0) return false;
    // the unfused dispatch (gemm.hip: launch_typed) runs this projection on the 8-wave whole-row folded instance exactly when
    // 16 < K / KSTEP <= 64 and the half-split grid is not chosen; only there is the fused half the SAME arithmetic (bit-identical)
    const int kstep = dtype == PTTS_BF16 ? PBF16::KSTEP : PF32::KSTEP;
    if (g.K % kstep != 0) return false;
    const int ksteps = g.K / kstep;
    const bool halves = g.N <= 2048 && g.M > 16;
    return ksteps > 16 && ksteps <= 64 && !halves;
}

hipError_t fused_qkv_attn_launch(const GemmArgs& g, const AttnArgs& a, int M, int heads, int dtype, int* arrive, int* err,
                                 hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, int* arrive_prod) {
    FusedArgs f;
    f.g = g; f.a = a; f.arrive = arrive; f.err = err; f.arrive_prod = arrive_prod ? arrive_prod : arrive;
    f.nblk = g.N / 16; f.mtiles = (M + 31) / 32; f.heads = heads;
    f.n_gemm = f.nblk * f.mtiles;
    static const int pre_env = getenv("PTTS_FUSED_PRE") ? atoi(getenv("PTTS_FUSED_PRE")) : 2;          // experiment switches
    f.pre = pre_env;
    // dynamic LDS: the larger of the two halves (K-split hand-over + statistics of the projection; scores + partial outputs of the attention)
    const size_t lds_g = (size_t)8 * 1 * 2 * 64 * 16 + (size_t)8 * 2 * 2 * 16 * 4;
    const size_t lds_a = ((size_t)a.s_cap + 8 * 64 + 8 + a.pool.max_blocks + 64) * sizeof(float);
    const size_t lds = lds_g > lds_a ? lds_g : lds_a;
    dim3 grid(f.n_gemm + heads * M), block(512);
#define PTTS_FUSED_LAUNCH(P_, T_)                                                                                            \
    do {                                                                                                                      \
        if (ev_start) hipExtLaunchKernelGGL((fused_qkv_attn_kernel<P_, T_>), grid, block, (uint32_t)lds, stream, ev_start, ev_stop, 0, f); \
        else hipLaunchKernelGGL((fused_qkv_attn_kernel<P_, T_>), grid, block, lds, stream, f);                                \
    } while (0)
    if (dtype == PTTS_BF16) {
        switch (tiles_for<PBF16>(g)) {
            case 4: PTTS_FUSED_LAUNCH(PBF16, 4); break;
            case 6: PTTS_FUSED_LAUNCH(PBF16, 6); break;
            case 8: PTTS_FUSED_LAUNCH(PBF16, 8); break;
            default: PTTS_FUSED_LAUNCH(PBF16, 0);
        }
    } else {
        switch (tiles_for<PF32>(g)) {
            case 4: PTTS_FUSED_LAUNCH(PF32, 4); break;
            case 6: PTTS_FUSED_LAUNCH(PF32, 6); break;
            case 8: PTTS_FUSED_LAUNCH(PF32, 8); break;
            default: PTTS_FUSED_LAUNCH(PF32, 0);
        }
    }
#undef PTTS_FUSED_LAUNCH
    return hipGetLastError();
}
