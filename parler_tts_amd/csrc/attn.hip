// Single-query attention over the paged KV pool (decode step) and small full attention (T5 encoder).
//
// attn_decode: one workgroup per (head, utterance); the query is one token (reference M:264-382 with
// tgt_len = 1; q was already scaled by hd^-0.5 in the projection epilogue, M:282).  HBM-bound: the
// workgroup streams S x 64 K values and S x 64 V values exactly once.
//   * K/V rows are 64 elements (128 B bf16 / 256 B fp32).  A wave64 `global_load_dwordx4` covers
//     TPW = 8 (bf16) / 4 (fp32) consecutive tokens = 1 KiB contiguous inside a 64-token page;
//     UNR such loads are issued back-to-back per wave before the first use.  The loads are
//     non-temporal (the cache is read once per step) and the pool keeps one slab per (layer, K/V, head),
//     so a workgroup with consecutive page ids streams one contiguous region.
//   * pass 1: scores s_t = q.k_t (+ additive mask, 0 or finfo.min as the reference adds it, M:339)
//     -> LDS; lanes of a token reduce with wave shuffles
//   * softmax over the whole row in fp32 exactly as torch does it (max, exp, sum, divide; M:342)
//   * pass 2: o = sum_t p_t v_t accumulated per lane, reduced across token-lanes and waves (LDS)
//   * output is written in MFMA B-fragment order for the out_proj GEMM (gemm.hip PRO_FRAG)
// The number of cached tokens comes from device memory (StepState.pos) so that one captured
// hipGraph serves every step.
// attn_page_kernel: contexts of <= 64 keys (cross-attention over a short description): one wave per
// (utterance, head), no LDS, no barrier; it also finishes the look-ahead query projection (LayerNorm
// correction from per-block row statistics).
#include "ptts_common.h"
#include <hip/hip_ext.h>
#include "attn.h"
#include <cstdlib>

namespace {
#define PTTS_ATTN_SIGNATURE __global__ __launch_bounds__(NWAVES * 64) void attn_decode_kernel(AttnArgs a)
#define PTTS_ATTN_H blockIdx.x
#define PTTS_ATTN_B blockIdx.y
#include "attn_device.inc"

// ------------------------------------------------------------------------------------------------
// Single-page variant (S <= 64 keys known on the host: cross-attention over a short description).
// One WAVE per (utterance, head), four waves per workgroup, no LDS and no barrier: every K and V tile
// of the page is requested up front (S/TPW loads each), the softmax runs on DPP/shuffle reductions
// across the token-lanes, and the 64 outputs leave as eight 16-byte fragment stores.  The general
// kernel above spends ~6.7 us on a 40-key context (8 waves, 8 barriers); this one has a single
// dependent global access (page id -> tiles).
// ------------------------------------------------------------------------------------------------
#define PTTS_PAGE_SIGNATURE __global__ __launch_bounds__(256) void attn_page_kernel(AttnArgs a, int H)
#define PTTS_PAGE_H (blockIdx.x * 4 + (threadIdx.x >> 6))
#define PTTS_PAGE_B blockIdx.y
#include "attn_page_device.inc"

// ------------------------------------------------------------------------------------------------
// Causal self-attention of a batched prefill chunk (reference M:264-382 with tgt_len = npos under the 4-D causal mask
// of M:870-872): the chunk's K / V rows are already in the paged cache (QKV projection epilogue), so query row
// (utterance b, position pos0 + i) is a single-query attention over the keys 0 .. pos0 + i - the decode kernel's
// arithmetic with ONE WAVE per query instead of one workgroup: tiles of TPW keys, scores to the wave's LDS row, exact
// fp32 softmax (max, exp, sum, divide), P.V, output in fragment order.  A one-off phase: P + prefix positions.
// ------------------------------------------------------------------------------------------------
template <typename P, typename OP>
__global__ __launch_bounds__(NWAVES * 64) void attn_causal_kernel(AttnArgs a, int s_max) {
    typedef typename P::elem_t elem_t;
    typedef KVLoad<P> L;
    typedef typename L::raw_t raw_t;
    constexpr int LPT = L::LPT, EPL = L::EPL, TPW = 64 / LPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = blockIdx.z * NWAVES + w;                       // position inside the chunk (uniform per wave)
    float* sc = reinterpret_cast<float*>(smem) + (size_t)w * s_max;
    if (i >= a.npos) return;                                     // whole wave; no workgroup barrier below
    const int S = a.pos0 + i + 1;
    const int row = b * a.npos + i;
    const int tt = lane / LPT, c = lane % LPT;
    const elem_t* pool = reinterpret_cast<const elem_t*>(a.pool.base);
    const int* table = a.pool.block_table ? a.pool.block_table + (size_t)b * a.pool.max_blocks : nullptr;
    const int page0 = b * a.pool.max_blocks;
    const size_t page_elems = a.pool.page_elems();
    const size_t head_off = (size_t)a.layer * 2 * a.pool.kv_elems() + (size_t)h * a.pool.head_elems() + c * EPL;
    const size_t v_off = a.pool.kv_elems();
    const float* bias = a.bias ? a.bias + (size_t)b * a.ld_bias : nullptr;
    float q[EPL];
    {
        const float* qp = a.q + (size_t)row * a.ldq + h * PTTS_HEAD_DIM + c * EPL;
#pragma unroll
        for (int j = 0; j < EPL; ++j) q[j] = qp[j];
    }
    const int ntiles = (S + TPW - 1) / TPW;
    auto kv_ptr = [&](int tok, size_t add) {
        const int pg = table ? table[tok >> 6] : page0 + (tok >> 6);
        return reinterpret_cast<const raw_t*>(pool + (size_t)pg * page_elems + head_off + add +
                                              (size_t)(tok & 63) * PTTS_HEAD_DIM);
    };
    float m = -INFINITY;
    for (int t = 0; t < ntiles; ++t) {
        int tok = t * TPW + tt;
        const bool live = tok < S;
        if (!live) tok = S - 1;
        float kf[EPL];
        L::unpack(*kv_ptr(tok, 0), kf);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < EPL; ++j) d = fmaf(q[j], kf[j], d);
        d = (LPT == 8) ? sum_oct(d) : sum_row16(d);
        if (bias) d += bias[tok];
        if (c == 0 && live) sc[tok] = d;
        if (live) m = fmaxf(m, d);
    }
    m = wave_max(m);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float sum = 0.f;
    for (int t = lane; t < S; t += 64) {
        const float e = expf(sc[t] - m);
        sc[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float acc[EPL];
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc[j] = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        int tok = t * TPW + tt;
        const bool live = tok < S;
        if (!live) tok = S - 1;
        const float p = live ? sc[tok] / sum : 0.f;
        float vf[EPL];
        L::unpack(*kv_ptr(tok, v_off), vf);
#pragma unroll
        for (int j = 0; j < EPL; ++j) acc[j] = fmaf(p, vf[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
        if (LPT == 8) acc[j] += dpp_f<0x128>(acc[j]);
        acc[j] += __shfl_xor(acc[j], 16);
        acc[j] += __shfl_xor(acc[j], 32);
    }
    if (tt == 0) {
        typename OP::elem_t* of = reinterpret_cast<typename OP::elem_t*>(a.ofrag) + (size_t)(row >> 5) * 32 * a.D;
#pragma unroll
        for (int j = 0; j < EPL; ++j) of[frag_off<OP>(row & 31, h * PTTS_HEAD_DIM + c * EPL + j)] = to_elem<OP>(acc[j]);
    }
}

// ------------------------------------------------------------------------------------------------
// Full (bidirectional) attention for the T5 encoder: one workgroup per (head, b); any T the LDS score rows hold (thousands).
// scores = q.k (no scaling) + rel-pos bias[h][bucket(j - i)] + (1 - mask_j) * finfo.min; softmax fp32.
// (HF:models/t5/modeling_t5.py:144-173, 217-280.)  q/k/v row-major fp32 [B*T][3*inner] as written by
// the QKV projection; output in fragment order for the `o` projection.
// ------------------------------------------------------------------------------------------------
// LDSKV = true: K and V of the (head, utterance) are parked in LDS (fits up to T ~ 300 at d_kv 64); false: they are read
// from the fp32 QKV rows in global memory (L2-resident: T x 3 x inner floats per utterance), for descriptions up to the
// 512 tokens T5 accepts.  Same arithmetic and summation order either way.
// DK > 0: the head width is a compile-time constant (64 = every T5 this engine has met), so the score and P.V loops unroll and
// their LDS reads go out in batches.  The round-1 form (run-time dk, one dependent LDS round trip per multiply-add, the softmax
// division repeated inside the P.V loop) took 64 us per launch at 32 x 40 tokens - 0.77 ms of the 12-layer encoder, more than its
// GEMMs once those ran on the tile kernel.  Same arithmetic in the same order: p = e / sum is computed once per (query, key)
// instead of once per (query, key, feature).
template <typename OP, bool LDSKV, int DK = 0>
__global__ __launch_bounds__(256) void attn_full_kernel(FullAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = a.T, dk = DK > 0 ? DK : a.dk;
    float* ks = reinterpret_cast<float*>(smem);          // [T][dk+1]   (LDSKV only)
    float* vs = ks + (LDSKV ? T * (dk + 1) : 0);         // [T][dk+1]
    float* pr = vs + (LDSKV ? T * (dk + 1) : 0);         // [4 waves][T] probabilities of the wave's query
    float* qs = pr + 4 * T;                              // [4][dk]
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int inner = a.H * dk;
    const float* base = a.qkv + (size_t)b * T * a.ld;
    const float* kg = base + inner + h * dk;             // K row j at kg + j * ld
    const float* vg = base + 2 * inner + h * dk;
    if (LDSKV) {
        for (int i = tid; i < T * dk; i += 256) {
            int t = i / dk, d = i % dk;
            ks[t * (dk + 1) + d] = kg[(size_t)t * a.ld + d];
            vs[t * (dk + 1) + d] = vg[(size_t)t * a.ld + d];
        }
        __syncthreads();
    }
    const int* mask = a.mask ? a.mask + (size_t)b * T : nullptr;
    for (int qi = w; qi < T; qi += 4) {
        for (int d = lane; d < dk; d += 64) qs[w * dk + d] = base[(size_t)qi * a.ld + h * dk + d];
        __builtin_amdgcn_wave_barrier();
        float mx = -INFINITY;
        for (int j = lane; j < T; j += 64) {
            float s = 0.f;
            if (LDSKV) {
#pragma unroll 16
                for (int d = 0; d < dk; ++d) s = fmaf(qs[w * dk + d], ks[j * (dk + 1) + d], s);
            } else {
                const float4* kr = reinterpret_cast<const float4*>(kg + (size_t)j * a.ld);
                for (int d = 0; d < dk; d += 4) {
                    const float4 kv = kr[d >> 2];
                    s = fmaf(qs[w * dk + d], kv.x, s); s = fmaf(qs[w * dk + d + 1], kv.y, s);
                    s = fmaf(qs[w * dk + d + 2], kv.z, s); s = fmaf(qs[w * dk + d + 3], kv.w, s);
                }
            }
            s = s * a.scale;
            s = s + a.rel_bias[(size_t)h * a.n_rel + (j - qi + T - 1)];
            if (mask) s = s + (1.0f - (float)mask[j]) * -3.4028234663852886e38f;
            pr[w * T + j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) {
            float e = expf(pr[w * T + j] - mx);
            pr[w * T + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int j = lane; j < T; j += 64) pr[w * T + j] = pr[w * T + j] / sum;      // the lane's own entries: no barrier needed in between
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < dk; d += 64) {
            float o = 0.f;
            if (LDSKV) {
#pragma unroll 8
                for (int j = 0; j < T; ++j) o = fmaf(pr[w * T + j], vs[j * (dk + 1) + d], o);
            } else {
#pragma unroll 4
                for (int j = 0; j < T; ++j) o = fmaf(pr[w * T + j], vg[(size_t)j * a.ld + d], o);   // lanes read one row: coalesced
            }
            int row = b * T + qi;
            typename OP::elem_t* of = reinterpret_cast<typename OP::elem_t*>(a.ofrag) + (size_t)(row >> 5) * 32 * inner;
            of[frag_off<OP>(row & 31, h * dk + d)] = to_elem<OP>(o);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

hipError_t attn_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<PBF16, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    const void* causal[4] = {reinterpret_cast<const void*>(attn_causal_kernel<PBF16, PBF16>),
                             reinterpret_cast<const void*>(attn_causal_kernel<PBF16, PF32>),
                             reinterpret_cast<const void*>(attn_causal_kernel<PF32, PBF16>),
                             reinterpret_cast<const void*>(attn_causal_kernel<PF32, PF32>)};
    for (const void* k : causal) {
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<PBF16, true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<PF32, true, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_full_kernel<PF32, true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t attn_decode_launch(const AttnArgs& a, int B, int H, int cache_dtype, int out_dtype, hipStream_t stream, hipEvent_t ev_start,
                              hipEvent_t ev_stop) {
    if (!a.no_page_kernel && !a.len_from_state && a.len <= PTTS_PAGE_TOKENS) {
        dim3 pgrid((H + 3) / 4, B), pblock(256);
        // tiles of the page the context reaches (8 keys per bf16 tile, 4 per fp32 tile); PTTS_ATTN_PAGE_FULL=1: always the whole page (A/B)
        static const bool full_page = getenv("PTTS_ATTN_PAGE_FULL") != nullptr;
        const int tiles = full_page ? 64 : (a.len + (cache_dtype == PTTS_BF16 ? 7 : 3)) / (cache_dtype == PTTS_BF16 ? 8 : 4);
#define PTTS_PAGE_LAUNCH(P_, O_, N1, N2)                                                                                   \
        do {                                                                                                                 \
            if (tiles <= N1) hipLaunchKernelGGL((attn_page_kernel<P_, O_, N1>), pgrid, pblock, 0, stream, a, H);             \
            else if (tiles <= N2) hipLaunchKernelGGL((attn_page_kernel<P_, O_, N2>), pgrid, pblock, 0, stream, a, H);        \
            else hipLaunchKernelGGL((attn_page_kernel<P_, O_, 0>), pgrid, pblock, 0, stream, a, H);                          \
        } while (0)
        if (cache_dtype == PTTS_BF16) {
            if (out_dtype == PTTS_BF16) PTTS_PAGE_LAUNCH(PBF16, PBF16, 3, 5);
            else PTTS_PAGE_LAUNCH(PBF16, PF32, 3, 5);
        } else {
            if (out_dtype == PTTS_BF16) PTTS_PAGE_LAUNCH(PF32, PBF16, 5, 10);
            else PTTS_PAGE_LAUNCH(PF32, PF32, 5, 10);
        }
#undef PTTS_PAGE_LAUNCH
        return hipGetLastError();
    }
    size_t lds = ((size_t)a.s_cap + NWAVES * 64 + NWAVES + a.pool.max_blocks + 64) * sizeof(float);   // + 64: look-ahead q
    dim3 grid(H, B), block(NWAVES * 64);
    // (a compile-time specialised instance - identity table, no look-ahead prologue - measured SLOWER in round 4: decode pass
    // 3586 -> 3609 ms, 29.0 -> 29.3 us per launch; deleted in round 6)
#define PTTS_ATTN_LAUNCH(P_, O_)                                                                             \
    do {                                                                                                     \
        /* profiled launches: events bound to THIS dispatch (its own start / end timestamps, no barrier packets) */ \
        if (ev_start && a.len_from_state)                                                               \
            hipExtLaunchKernelGGL((attn_decode_kernel<P_, O_, true>), grid, block, (uint32_t)lds, stream, ev_start, ev_stop, 0, a); \
        else if (ev_start)                                                                                   \
            hipExtLaunchKernelGGL((attn_decode_kernel<P_, O_, false>), grid, block, (uint32_t)lds, stream, ev_start, ev_stop, 0, a); \
        else if (a.len_from_state) hipLaunchKernelGGL((attn_decode_kernel<P_, O_, true>), grid, block, lds, stream, a); \
        else hipLaunchKernelGGL((attn_decode_kernel<P_, O_, false>), grid, block, lds, stream, a);            \
    } while (0)
    if (cache_dtype == PTTS_BF16) {
        if (out_dtype == PTTS_BF16) PTTS_ATTN_LAUNCH(PBF16, PBF16);
        else PTTS_ATTN_LAUNCH(PBF16, PF32);
    } else {
        if (out_dtype == PTTS_BF16) PTTS_ATTN_LAUNCH(PF32, PBF16);
        else PTTS_ATTN_LAUNCH(PF32, PF32);
    }
#undef PTTS_ATTN_LAUNCH
    return hipGetLastError();
}

hipError_t attn_causal_launch(const AttnArgs& a, int B, int H, int cache_dtype, int out_dtype, hipStream_t stream) {
    if (a.npos <= 0) return hipErrorInvalidValue;
    const int s_max = (a.pos0 + a.npos + 3) & ~3;
    const size_t lds = (size_t)NWAVES * s_max * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(H, B, (a.npos + NWAVES - 1) / NWAVES), block(NWAVES * 64);
    if (cache_dtype == PTTS_BF16) {
        if (out_dtype == PTTS_BF16) hipLaunchKernelGGL((attn_causal_kernel<PBF16, PBF16>), grid, block, lds, stream, a, s_max);
        else hipLaunchKernelGGL((attn_causal_kernel<PBF16, PF32>), grid, block, lds, stream, a, s_max);
    } else {
        if (out_dtype == PTTS_BF16) hipLaunchKernelGGL((attn_causal_kernel<PF32, PBF16>), grid, block, lds, stream, a, s_max);
        else hipLaunchKernelGGL((attn_causal_kernel<PF32, PF32>), grid, block, lds, stream, a, s_max);
    }
    return hipGetLastError();
}

hipError_t attn_full_launch(const FullAttnArgs& a, int B, int out_dtype, hipStream_t stream) {
    size_t lds = ((size_t)2 * a.T * (a.dk + 1) + 4 * a.T + 4 * a.dk) * sizeof(float);
    dim3 grid(a.H, B), block(256);
    const bool force_global = getenv("PTTS_T5_GLOBAL_KV") != nullptr;      // A/B and tests of the long-description path
    if (lds <= 160 * 1024 && !force_global) {
        if (a.dk == 64) {
            if (out_dtype == PTTS_BF16) hipLaunchKernelGGL((attn_full_kernel<PBF16, true, 64>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((attn_full_kernel<PF32, true, 64>), grid, block, lds, stream, a);
        } else {
            if (out_dtype == PTTS_BF16) hipLaunchKernelGGL((attn_full_kernel<PBF16, true>), grid, block, lds, stream, a);
            else hipLaunchKernelGGL((attn_full_kernel<PF32, true>), grid, block, lds, stream, a);
        }
        return hipGetLastError();
    }
    // K / V of this (head, utterance) exceed the LDS (T > ~300 at d_kv = 64): read them from the QKV rows in global memory
    if (a.dk % 4 != 0 || a.ld % 4 != 0) return hipErrorInvalidValue;
    lds = ((size_t)4 * a.T + 4 * a.dk) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (out_dtype == PTTS_BF16) hipLaunchKernelGGL((attn_full_kernel<PBF16, false>), grid, block, lds, stream, a);
    else hipLaunchKernelGGL((attn_full_kernel<PF32, false>), grid, block, lds, stream, a);
    return hipGetLastError();
}
