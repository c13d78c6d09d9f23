// MFMA tile GEMM for the ONE-OFF phases of generate(): the T5 encoder (modeling_parler_tts.py:2229), the batched prefill (decoder
// forward over all prompt positions, M:794-978) and the cross-attention K / V projections (M:295-298).  bf16 engine mode only, every row count.
//
// Why a second GEMM kernel.  gemm_kernel (gemm.hip) is built for the decode step: a workgroup owns ONE 16-feature block of ONE
// 32-row m-tile and its waves split K - at 32 rows the weights are read once and the launch is a latency chain.  The one-off
// phases run the same kernel with M / 32 m-tiles in gridDim.y: at M = 1280 (32 descriptions x 40 tokens) every weight block is
// re-read 40 times and every activation m-tile N / 16 times through L2 - 420 MB of L2 traffic for a 4.5-GFLOP product, 37-47 TFLOP/s,
// 0.026 of the matrix peak over T5 + prefill.  Here a workgroup owns a 64-feature x 64-row tile:
//
//   * operands travel by LDS-DMA (`global_load_lds_dwordx4`) - both are ALREADY stored as 1-KiB tiles in MFMA operand order, lane-major
//     (packed weights = A operand, activation fragments = B operand: ptts_common.h), so one wave-instruction moves one tile, the
//     LDS image is the fragment image and every fragment read is a lane-linear `ds_read_b128` (conflict free, no swizzle needed);
//   * the workgroup tile is 64 features x 64 rows; a stage = 2 k-steps (K = 64) of 4 weight blocks + 4 row tiles = 16 tiles = 16 KiB;
//     a RING of 4 stages in LDS (64 KiB: two workgroups per CU), three stages in flight while one is multiplied: counted
//     `s_waitcnt vmcnt(8)` + ONE raw `s_barrier` per stage (never __syncthreads(): it would drain the DMA - cdna_hip_programming.md
//     "Pipelining across barriers").  How it got there (profiles/r06_tile_gemm_probe.txt): the first build had 128 x 128 tiles and
//     two stages and measured 2.4 us PER STAGE - these products are L2 / Infinity-Cache resident and latency-bound, a workgroup
//     moves (bytes in flight) / (round trip ~2 us); a ring of 4 brought 0.67 us per stage = DMA issue + fragment reads + MFMAs in
//     series, because one workgroup per CU is one wave per SIMD marching from barrier to barrier; 64 x 64 tiles with the same ring
//     put two workgroups on a CU (they overlap each other's phases) and 4x the workgroups on the grid (the batched prefill at 512
//     rows x 1024 features is 32 tiles of 128 x 128 on 256 CUs) - fastest or within 4 % on all seven shapes of the benchmark;
//   * 4 waves as 2 (features) x 2 (rows), each 32 x 32 outputs = 2 x 2 accumulator tiles of v_mfma_f32_16x16x32_bf16; no K split,
//     so no hand-over through LDS and every dot product is ONE k-ordered fp32 accumulation;
//   * weights re-read M / 64 times and activations N / 64 times: 3x less L2 traffic than the 32-row form at M = 1280;
//   * the epilogues are gemm_device.inc's (same code, same rounding points): row-major (+ bias / mask / activation), residual
//     (+ fragment copy + per-block LayerNorm partials for the look-ahead query), GELU / gated-GELU fragments, q-scale, KV-cache
//     append (self: per-row cache positions of a prefill chunk; cross).  The folded LayerNorm (decoder) takes its row statistics
//     from the matrix cores as the decode instances do (ones . x and the diagonal of x . x^T, two more MFMAs per fragment).
//   * the look-ahead role of the out-projection launch (second role of gemm_kernel) is a second launch of this kernel with two
//     fragment sources along K.
//
// Numerics: same operands, same products; the fp32 summation ORDER differs from gemm_kernel's 8-wave K split, so results agree to
// fp32 rounding, not bit for bit (tests: against torch at the GEMM level, T5 / prefill against the fixtures at the same bounds).
// `PTTS_PREFILL_M128=0` keeps every one-off GEMM on gemm_kernel (A/B).
#include "ptts_common.h"
#include "gemm.h"
#include <cstdlib>

namespace {

// the epilogues of the decode GEMM (never instantiated as a kernel here)
#define PTTS_GEMM_SIGNATURE __device__ __forceinline__ void gemm_body_unused(GemmArgs a, const int gemm_bx, const int gemm_by)
#define PTTS_GEMM_BX gemm_bx
#define PTTS_GEMM_BY gemm_by
#include "gemm_device.inc"

constexpr int TILE_KS = 2;                                              // k-steps (of 32) per stage
// (no row threshold: a description encoded alone must get the same bits as inside a batch of 32 - tests hold generate() to that -
// so EVERY launch of a one-off phase takes this kernel in the bf16 mode, whatever its row count)
// T8 = weight blocks = row tiles (of 16) per workgroup tile: 4 (64 x 64) is the instance in use (8 = 128 x 128 was measured); S = stages in the LDS ring
template <int T8, int S> struct TileCfg {
    static constexpr int STAGE_TILES = 2 * T8 * TILE_KS;                // 1-KiB tiles per stage: T8 weight blocks + T8 row tiles, x 2 k-steps
    static constexpr int STAGE_BYTES = STAGE_TILES * 1024;
    static constexpr int DMA_PER_WAVE = STAGE_TILES / 4;                // LDS-DMA instructions per wave and stage (= T8)
    static constexpr int LDS_BYTES = S * STAGE_BYTES;
    static constexpr int W = T8 / 2;                                    // accumulator tiles per wave and direction
};

template <int N> __device__ inline void tile_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void tile_barrier() {        // own LDS reads retired, then the workgroup barrier; does NOT drain vmcnt (pending LDS-DMA)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int T8, int S, bool FOLD>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmArgs a) {
    typedef TileCfg<T8, S> C;
    constexpr int STAGE_BYTES = C::STAGE_BYTES, DMA_PER_WAVE = C::DMA_PER_WAVE, W = C::W;
    static_assert((S & (S - 1)) == 0 && S >= 4, "ring depth: a power of two, at least 4");
    extern __shared__ __attribute__((aligned(16))) char smem[];         // [S stages][STAGE_TILES][1 KiB]: the only LDS object
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w & 1, wm = w >> 1;
    const int nblocks = a.N / 16, ksteps = a.K / 32;
    const int nb0 = blockIdx.x * T8;
    const int rt0 = blockIdx.y * T8;                                    // first 16-row tile of this workgroup
    const int rt_valid = 2 * ((a.M + 31) / 32);                         // 16-row tiles the fragment buffers hold (32-row m-tiles)
    const int nst = (ksteps + TILE_KS - 1) / TILE_KS;
    // second fragment source along K (look-ahead role): k-steps < ksplit come from xfrag_lo (row length ksplit * 32)
    const int ksplit = a.xfrag_lo ? a.ksplit : 0;

    // ---- this wave's share of a stage: waves 0 / 1 move the lower / upper half of the weight blocks, waves 2 / 3 of the row tiles (x 2 k-steps) ----
    // src0[b] serves k-steps < ksplit, src1[b] the rest (already offset so that the k-step index applies unchanged); indices past the
    // matrix are clamped to its last block / tile: every wave issues the same DMA count at every stage (the vmcnt arithmetic
    // needs that, and a branch around a DMA makes hipcc drain the queue), the clamped copies are never multiplied into a kept result
    const uint16_t* src0[W];
    const uint16_t* src1[W];
    int kstride = 512;                                                  // elements per k-step in this wave's source
    if (w < 2) {
        kstride = 512;
#pragma unroll
        for (int b = 0; b < W; ++b) {
            int nb = nb0 + w * W + b;
            nb = nb < nblocks ? nb : nblocks - 1;
            src0[b] = src1[b] = reinterpret_cast<const uint16_t*>(a.W) + (size_t)nb * ksteps * 512 + lane * 8;
        }
    } else {
        kstride = 1024;
        const int klo = ksplit * 32, khi = a.K - klo;
#pragma unroll
        for (int b = 0; b < W; ++b) {
            int rt = rt0 + (w - 2) * W + b;
            rt = rt < rt_valid ? rt : rt_valid - 1;
            const size_t mt = (size_t)(rt >> 1), h = rt & 1;
            src1[b] = reinterpret_cast<const uint16_t*>(a.xfrag) + mt * 32 * khi + h * 512 + lane * 8 - (ptrdiff_t)ksplit * 1024;
            src0[b] = ksplit ? reinterpret_cast<const uint16_t*>(a.xfrag_lo) + mt * 32 * klo + h * 512 + lane * 8 : src1[b];
        }
    }
    auto dma_stage = [&](int st, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)            // the builtin does not exist in the host pass
        typedef __attribute__((address_space(3))) void lds_void;
        char* dst = smem + buf * STAGE_BYTES + w * (DMA_PER_WAVE * 1024);
#pragma unroll
        for (int q = 0; q < DMA_PER_WAVE; ++q) {
            int ks = st * TILE_KS + (q & 1);
            ks = ks < ksteps ? ks : ksteps - 1;
            const uint16_t* src = (ks < ksplit ? src0[q >> 1] : src1[q >> 1]) + (size_t)ks * kstride;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src), (lds_void*)(dst + q * 1024), 16, 0, 0);
        }
#else
        (void)st; (void)buf; (void)kstride;
#endif
    };

    f32x4 acc[W][W];
#pragma unroll
    for (int i = 0; i < W; ++i)
#pragma unroll
        for (int j = 0; j < W; ++j) acc[i][j] = {0, 0, 0, 0};
    // folded LayerNorm: row statistics on the matrix cores, as gemm_kernel's folded instances take them (MSTAT)
    f32x4 ms_sum[W], ms_sq[W];
    bf16x8 ones_frag;
    if (FOLD) {
#pragma unroll
        for (int j = 0; j < W; ++j) { ms_sum[j] = {0, 0, 0, 0}; ms_sq[j] = {0, 0, 0, 0}; }
        const u32x4 o = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};       // eight bf16 1.0
        ones_frag = __builtin_bit_cast(bf16x8, o);
    }
    auto multiply = [&](int st, int buf) {
        const char* base = smem + buf * STAGE_BYTES + lane * 16;
#pragma unroll
        for (int kk = 0; kk < TILE_KS; ++kk) {
            if (st * TILE_KS + kk < ksteps) {                           // wave-uniform (scalar) guard: MFMA ignores EXEC
                bf16x8 af[W], bf[W];
#pragma unroll
                for (int i = 0; i < W; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + ((wn * W + i) * TILE_KS + kk) * 1024);
#pragma unroll
                for (int j = 0; j < W; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(base + (T8 * TILE_KS + (wm * W + j) * TILE_KS + kk) * 1024);
#pragma unroll
                for (int i = 0; i < W; ++i)
#pragma unroll
                    for (int j = 0; j < W; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (FOLD) {
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        ms_sum[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_frag, bf[j], ms_sum[j], 0, 0, 0);
                        ms_sq[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], bf[j], ms_sq[j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- main loop: S - 1 stages travel while one is multiplied.  Stage numbers past the last are clamped (a copy of the last stage
    // into a buffer nobody reads any more): every iteration issues the same DMA count, so the counted wait is a constant ----
    auto dma_clamped = [&](int st) { dma_stage(st < nst ? st : nst - 1, st & (S - 1)); };
#pragma unroll
    for (int p = 0; p < S - 1; ++p) dma_clamped(p);
    for (int st = 0; st < nst; ++st) {
        tile_wait_vmcnt<DMA_PER_WAVE * (S - 2)>();    // own pieces of stage st have landed (the S - 2 younger stages stay in flight)
        tile_barrier();                               // ... and everyone else's; and every wave has left stage st - 1's buffer
        dma_clamped(st + S - 1);                      // into that buffer: (st + S - 1) mod S = (st - 1) mod S
        multiply(st, st & (S - 1));
    }
    tile_wait_vmcnt<0>();                             // the clamped tail copies: nothing may be in flight when the wave ends

    // ---- epilogue: lane l of accumulator tile (i, j) holds features n .. n + 3 (n = 16 block + 4 (l >> 4)) of row 16 tile + (l & 15) ----
    const int epi = a.epi;
    const bool gated = epi == EPI_GATED_FRAG;
    const float invK = 1.0f / (float)a.K;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int rt = rt0 + wm * W + j;
        const int row = rt * 16 + (lane & 15);
        float mean = 0.f, rstd = 1.f;
        if (FOLD) {
            // ones . x: every accumulator row holds the column sums, so each lane already has the sum of ITS row (column l & 15);
            // x . x^T: element (r, r) sits in lane 16 (r >> 2) + r, slot r & 3
            const float sx = ms_sum[j][0];
            const int r = lane & 15, m = r & 3;
            const f32x4 q4 = ms_sq[j];
            const float d = m == 0 ? q4[0] : (m == 1 ? q4[1] : (m == 2 ? q4[2] : q4[3]));
            const float sq = __shfl(d, 16 * (r >> 2) + r);
            mean = a.rms ? 0.f : sx * invK;
            const float var = a.rms ? sq * invK : fmaxf(sq * invK - mean * mean, 0.f);
            rstd = 1.0f / sqrtf(var + a.eps);
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int nb = nb0 + wn * W + i;
            if (nb >= nblocks) continue;                                 // wave-uniform: blocks past the matrix (clamped copies)
            if (gated && (i & 1)) continue;                              // the odd block of a pair is the gate's multiplier
            f32x4 v = acc[i][j], v1 = {0, 0, 0, 0};
            int n = nb * 16 + 4 * (lane >> 4);
            if (gated) { v1 = acc[i | 1][j]; n = (nb >> 1) * 16 + 4 * (lane >> 4); }
            if (FOLD) {
                const float4 c1v = *reinterpret_cast<const float4*>(a.c1 + n), c2v = *reinterpret_cast<const float4*>(a.c2 + n);
                v[0] = rstd * (v[0] - mean * c1v.x) + c2v.x;
                v[1] = rstd * (v[1] - mean * c1v.y) + c2v.y;
                v[2] = rstd * (v[2] - mean * c1v.z) + c2v.z;
                v[3] = rstd * (v[3] - mean * c1v.w) + c2v.w;
            }
            if (epi == EPI_RESIDUAL) {
                // gemm_device.inc's residual epilogue, written out because the per-block LayerNorm partials need the updated row
                float4 x = make_float4(0, 0, 0, 0);
                const bool live = row < a.M;
                if (live) {
                    float4* o = reinterpret_cast<float4*>(a.out + (size_t)row * a.ldo + n);
                    x = *o;
                    x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
                    *o = x;
                    if (a.xfrag_out) {
                        uint16_t* of = reinterpret_cast<uint16_t*>(a.xfrag_out) + (size_t)(row >> 5) * 32 * a.ldo;
                        const float xv[4] = {x.x, x.y, x.z, x.w};
                        store_frag4<PBF16>(of, row & 31, n, xv);
                    }
                }
                if (a.xstat) {      // (sum, sum of squares) of this row over the block's 16 columns, on the values bf16 will hold
                    float ps = 0.f, pq = 0.f;
                    if (live) {
                        const float xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float t = from_elem<PBF16>(to_elem<PBF16>(xv[r]));
                            ps += t;
                            pq = fmaf(t, t, pq);
                        }
                    }
                    ps += __shfl_xor(ps, 16); pq += __shfl_xor(pq, 16);
                    ps += __shfl_xor(ps, 32); pq += __shfl_xor(pq, 32);
                    if (lane < 16 && live) {
                        float* st2 = a.xstat + (((size_t)(rt >> 1) * nblocks + nb) * 32 + 16 * (rt & 1) + lane) * 2;
                        st2[0] = ps;
                        st2[1] = pq;
                    }
                }
            } else if (epi == EPI_ROWMAJOR && !a.bias && !a.rowmask && a.act == ACT_NONE && (a.ldo & 3) == 0) {
                // plain fp32 rows (T5 q / k / v, the raw look-ahead query): ONE 16-byte store per lane - the four 16-lane groups of the
                // wave complete 64 contiguous bytes of each of the 16 rows (the generic epilogue's four 4-byte stores per lane took
                // 25 of the 37 us of the T5 QKV launch: tools/tile_gemm_probe.py, K = 64 against K = 768)
                if (row < a.M) *reinterpret_cast<float4*>(a.out + (size_t)row * a.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
            } else if (row < a.M) {
                epilogue<PBF16, -1, -1>(a, epi, row, n, v, v1, make_float4(0, 0, 0, 0), false);
            }
        }
    }
}

// fp32 rows -> (RMSNorm | LayerNorm | nothing) -> bf16 fragments: the staged prologue of gemm_kernel as a pass of its own, so that
// the tile kernel sees fragments only (one wave per row; same arithmetic and the same rounding point as the staged prologue)
__global__ __launch_bounds__(256) void rows_norm_to_frag_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, int pro, uint16_t* xf,
                                                                int M, int K) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = X + (size_t)row * ldx;
    float mean = 0.f, rstd = 1.f;
    const bool norm = pro == PRO_LAYERNORM || pro == PRO_RMSNORM;
    if (norm) {
        const float invK = 1.0f / (float)K;
        if (pro == PRO_LAYERNORM) {
            float s = 0.f;
            for (int k = 4 * lane; k < K; k += 256) { const float4 v = *reinterpret_cast<const float4*>(xr + k); s += (v.x + v.y) + (v.z + v.w); }
            mean = wave_sum(s) * invK;
        }
        float q = 0.f;
        for (int k = 4 * lane; k < K; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
            const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
            q = fmaf(d0, d0, q); q = fmaf(d1, d1, q); q = fmaf(d2, d2, q); q = fmaf(d3, d3, q);
        }
        rstd = 1.0f / sqrtf(wave_sum(q) * invK + eps);
    }
    uint16_t* of = xf + (size_t)(row >> 5) * 32 * K;
    for (int k = 4 * lane; k < K; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + k);
        float y[4] = {v.x, v.y, v.z, v.w};
        if (norm) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + k);
            if (pro == PRO_LAYERNORM) {        // (x * rstd - rstd * mean) * gamma + beta: ATen's form, as the staged prologue
                const float4 b = *reinterpret_cast<const float4*>(beta + k);
                const float sh = -rstd * mean;
                y[0] = (y[0] * rstd + sh) * g.x + b.x; y[1] = (y[1] * rstd + sh) * g.y + b.y;
                y[2] = (y[2] * rstd + sh) * g.z + b.z; y[3] = (y[3] * rstd + sh) * g.w + b.w;
            } else {                           // T5 RMSNorm: w * (x * rsqrt(mean(x^2) + eps))
                y[0] = g.x * (y[0] * rstd); y[1] = g.y * (y[1] * rstd); y[2] = g.z * (y[2] * rstd); y[3] = g.w * (y[3] * rstd);
            }
        }
        store_frag4<PBF16>(of, row & 31, k, y);
    }
}

}  // namespace

bool gemm_tile_enabled() {      // read per call (one-off phases only, never a replayed step): the A/B test flips it inside one process
    const char* v = getenv("PTTS_PREFILL_M128");
    return !(v && atoi(v) == 0);
}

bool gemm_tile_supported(const GemmArgs& a, int dtype) {
    if (dtype != PTTS_BF16 || !a.tile_ok || !gemm_tile_enabled()) return false;
    if (a.M < 1 || a.pro != PRO_FRAG || !a.xfrag || a.force_whole) return false;
    if (a.K % 32 != 0 || a.N % 16 != 0) return false;
    if (a.epi == EPI_GATED_FRAG && a.N % 32 != 0) return false;
    if (a.W2 && (a.K2 % 32 != 0 || a.N2 % 16 != 0 || !a.xfrag_lo || !a.out2 || a.ksplit <= 0 || a.ksplit >= a.K2 / 32)) return false;
    return true;
}

template <int T8, int S, bool FOLD>
hipError_t tile_launch_one(const GemmArgs& g, hipStream_t stream) {
    typedef TileCfg<T8, S> C;
    static bool opted = false;        // > 64 KiB of dynamic LDS is opted into per kernel (one-off phases: never inside a stream capture)
    if (!opted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tile_kernel<T8, S, FOLD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           C::LDS_BYTES);
        if (e != hipSuccess) return e;
        opted = true;
    }
    dim3 grid((g.N / 16 + T8 - 1) / T8, (g.M + 16 * T8 - 1) / (16 * T8)), block(256);
    hipLaunchKernelGGL((gemm_tile_kernel<T8, S, FOLD>), grid, block, C::LDS_BYTES, stream, g);
    return hipGetLastError();
}

hipError_t gemm_tile_launch(const GemmArgs& a, hipStream_t stream) {
    auto one = [&](const GemmArgs& g) -> hipError_t {
        // ONE tile shape: 64 x 64 with a ring of 4 stages (64 KiB: two workgroups per CU).  Measured against 128 x 128 / ring of 4
        // (128 KiB) and 64 x 64 / ring of 8 (128 KiB) on the seven product shapes of the benchmark's T5 + prefill
        // (profiles/r06_tile_gemm_probe.txt): fastest or within 4 % of the fastest on every one - two co-resident workgroups overlap
        // each other's DMA issue, fragment reads and MFMAs, which one workgroup with one wave per SIMD cannot
        return g.fold ? tile_launch_one<4, 4, true>(g, stream) : tile_launch_one<4, 4, false>(g, stream);
    };
    GemmArgs g = a;
    g.W2 = nullptr;
    if (!a.W2) { g.xfrag_lo = nullptr; g.ksplit = 0; }
    else { g.xfrag_lo = nullptr; g.ksplit = 0; }     // the first role reads `xfrag` alone
    hipError_t he = one(g);
    if (he != hipSuccess) return he;
    if (a.W2) {
        // second role of the out-projection launch (look-ahead query): out2 = [xfrag_lo | xfrag] . W2^T, raw, row-major.  It reads the
        // rows this pair of launches STARTS from (xfrag_lo, the other fragment buffer than xfrag_out) and the first role's input:
        // nothing the first launch writes, so the order of the two does not matter
        GemmArgs r = a;
        r.W = a.W2; r.N = a.N2; r.K = a.K2; r.W2 = nullptr;
        r.epi = EPI_ROWMAJOR; r.out = a.out2; r.ldo = a.ldo2;
        r.bias = nullptr; r.rowmask = nullptr; r.act = ACT_NONE; r.xfrag_out = nullptr; r.xstat = nullptr; r.fold = 0;
        he = one(r);
    }
    return he;
}

hipError_t rows_norm_to_frag_launch(const float* X, int ldx, const float* gamma, const float* beta, float eps, int pro, void* xfrag,
                                    int M, int K, hipStream_t stream) {
    if (K % 4 != 0 || ldx % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rows_norm_to_frag_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, X, ldx, gamma, beta, eps, pro,
                       (uint16_t*)xfrag, M, K);
    return hipGetLastError();
}
