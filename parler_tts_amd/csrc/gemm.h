// Host-side interface of gemm.hip (internal to the library; the public C ABI is include/parler_tts_hip.h).
#pragma once
#include "ptts_common.h"
#include <cstddef>

enum { PTTS_F32 = 0, PTTS_BF16 = 1 };
enum { PRO_FRAG = 0, PRO_ROWMAJOR = 1, PRO_LAYERNORM = 2, PRO_RMSNORM = 3 };
enum { EPI_ROWMAJOR = 0, EPI_RESIDUAL = 1, EPI_ACT_FRAG = 2, EPI_QKV_SELF = 3, EPI_KV_CROSS = 4, EPI_Q_SCALE = 5,
       EPI_GATED_FRAG = 6 };
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_GELU_NEW = 2, ACT_RELU = 3 };

// Field ORDER matters for speed, not for meaning: the struct is the kernel-argument block, hipcc fetches it through the scalar
// cache piecemeal (each field right before its first use), and that cache starts cold at every launch - every 64-byte line of
// the block costs one miss (~0.2 us) the first time it is touched.  So what the decode-step instances read while they issue
// their loads sits in line 0, what their epilogues read in line 1 (first touched early, by the fold constants / residual slice),
// the KV pool and the second role in line 2, the staged-prologue / one-off fields behind them.
struct GemmArgs {
    // ---- line 0: load phase ----
    const void* W = nullptr;      // packed weights [N/16][K/KSTEP][64 lanes][16 B]
    const void* xfrag = nullptr;  // frag-major [(M+31)/32][32*K]
    int N = 0, K = 0, M = 0;
    int epi = EPI_ROWMAJOR;
    float* out = nullptr;
    int ldo = 0;                  // row stride of `out`, or feature count of the frag output
    // folded LayerNorm (PRO_FRAG only): `xfrag` holds the RAW residual rows, W is diag(gamma).W, and the
    // normalisation is applied after the MFMA: y = rstd * (x.W' - mean * c1) + c2   (see fold_layernorm_launch)
    int fold = 0;
    const float* c1 = nullptr;    // [N] sum_k W'[n][k]
    const float* c2 = nullptr;    // [N] sum_k beta[k] * W[n][k]
    // ---- line 1: epilogues of the decode-step instances ----
    void* xfrag_out = nullptr;    // EPI_RESIDUAL: also store the updated rows as fragments [(M+31)/32][32*ldo]
    float* xstat = nullptr;       // EPI_RESIDUAL: per 16-column block (sum, sum of squares) of the updated rows as the
                                  // weight dtype rounds them: [((m-tile * N/16) + block) * 32 + row][2]
    void* ofrag = nullptr;
    const StepState* state = nullptr;
    float eps = 1e-5f;
    float scale = 1.0f;
    int layer = 0;
    int D = 0;                    // model width (segment size of the fused QKV / KV projection)
    // EPI_QKV_SELF in a batched prefill chunk: row = b * npos + i is cache position pos0 + i of utterance b
    // (npos = 0: decode, row = utterance, position = state->pos)
    int npos = 0, pos0 = 0;
    int act = ACT_NONE;
    // decoder-op variants (extensions, configuration.py; no counterpart in the reference):
    int rms = 0;                  // folded path: RMSNorm instead of LayerNorm - y = rsqrt(mean(x^2) + eps) * (x.W') (c2 is zero)
    // ---- line 2: KV pool, second role ----
    KVPool pool = {};
    // second role in the same launch (fragment input, half-split grid only; "look-ahead projection"):
    // workgroups past the first role's grid compute  out2 = [xfrag_lo | xfrag] . W2^T  (raw, row-major), i.e. a
    // projection of the rows this launch is about to produce, from the rows it starts from and its own input
    const void* W2 = nullptr;     // packed [N2][K2]
    const void* xfrag_lo = nullptr;   // fragments of the first ksplit k-steps; the rest come from `xfrag`
    float* out2 = nullptr;
    int N2 = 0, K2 = 0;
    // ---- the rest: staged prologues, one-off phases, switches ----
    int ksplit = 0;
    int ldo2 = 0;
    int pro = PRO_FRAG;
    int ldx = 0;
    const float* x = nullptr;  // row-major fp32 [M][ldx]
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const float* bias = nullptr;
    const int* rowmask = nullptr;
    int T = 1;                    // cross rows per utterance (row = b*T + t)
    int force_whole = 0;          // test hook only: keep the folded GEMM on the two-half workgroup where the half-split grid would be chosen
    int wide2 = 0;                // with W2: 32-row workgroups for BOTH roles (N/16 + N2/16 workgroups per m-tile) instead of the
                                  // half-split grid: fewer, fatter workgroups so that the launch is one round of one workgroup per CU
    int rope_k = 0;               // EPI_QKV_SELF: the k segment goes, unrotated and in fp32, to out[row][D + n] (ldo >= 2 D) instead
                                  // of the cache; rope_launch (misc.h) rotates q in place and appends the rotated k
    int tile_ok = 0;              // set by the ONE-OFF phases only (T5 encoder, batched prefill, cross K / V): in the bf16 mode the
                                  // launch goes to the 64 x 64 MFMA tile kernel (gemm_tile.hip: other fp32 summation order), at
                                  // every row count.  Decode steps never set it: their instances are pinned bit for bit by tests
#ifdef PTTS_STAMPS
    StampRef stamps = {nullptr, -1};
#endif
};
static_assert(offsetof(GemmArgs, xfrag_out) == 64 && offsetof(GemmArgs, pool) == 128, "GemmArgs: the hot fields fill kernel-argument lines 0 and 1");

hipError_t gemm_init();
int gemm_validate(const GemmArgs& a, int dtype);
hipError_t gemm_launch(const GemmArgs& a, int dtype, hipStream_t stream);
hipError_t pack_weight_launch(const float* W, void* out, int N, int K, int n_off, int n_stride, int dtype,
                              hipStream_t stream);
// Wf = diag(gamma).W re-rounded to the weight dtype (both in packed layout), c1[n] = sum_k Wf[n][k],
// c2[n] = sum_k beta[k] * W[n][k]; N is the padded row count of the packed matrix
hipError_t fold_layernorm_launch(const void* W, void* Wf, const float* gamma, const float* beta, float* c1, float* c2,
                                 int N, int K, int dtype, hipStream_t stream);
// row-major fp32 [M][K] -> fragment layout of the weight dtype (test hook for the folded path)
hipError_t rows_to_frag_launch(const float* X, void* xfrag, int M, int K, int dtype, hipStream_t stream);
// look-ahead weights (packed [N][2K]): columns 0..K-1 = Wf (the gamma-folded projection, packed [N][K]); columns K..2K-1 =
// Wf . Wo, Wo packed [K][K] = the residual projection whose output the folded projection would have read
// out [N][K1 + K2] = [Wf | Wf . Wo] with Wf [N][K1], Wo [K1][K2], all in packed tile order
hipError_t lookahead_build_launch(const void* Wf, const void* Wo, void* out, int N, int K1, int K2, int dtype, hipStream_t stream);
// number of elements of `a` that differ bitwise from `b`; *first = linear index of one of them (diagnostics)
hipError_t compare_launch(const float* a, const float* b, size_t n, unsigned* count, unsigned* first, hipStream_t stream);
hipError_t convert_launch(const float* src, void* dst, size_t n, int dtype, hipStream_t stream);

// ---- gemm_tile.hip: 64 x 64 MFMA tile kernel for the one-off phases (bf16 mode, fragment input) ----
bool gemm_tile_enabled();                                   // PTTS_PREFILL_M128=0 switches it off (A/B)
bool gemm_tile_supported(const GemmArgs& a, int dtype);     // gemm_launch asks this first
hipError_t gemm_tile_launch(const GemmArgs& a, hipStream_t stream);
// fp32 rows [M][ldx] -> (pro = PRO_ROWMAJOR: nothing | PRO_LAYERNORM | PRO_RMSNORM) -> bf16 fragments [(M + 31) / 32][32 K]: the staged
// prologue of gemm_kernel as a pass of its own (one conversion then serves every launch that reads the rows)
hipError_t rows_norm_to_frag_launch(const float* X, int ldx, const float* gamma, const float* beta, float eps, int pro, void* xfrag,
                                    int M, int K, hipStream_t stream);
