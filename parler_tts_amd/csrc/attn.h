// Host-side interface of attn.hip (internal).
#pragma once
#include "ptts_common.h"
#include <cstddef>
#include "gemm.h"

// Field order: see GemmArgs (gemm.h) - the struct is the kernel-argument block and every 64-byte line of it is one cold miss
// in the scalar cache; what the decode kernels' prologues read sits in lines 0 and 1.
struct AttnArgs {
    // ---- line 0 ----
    const float* q = nullptr;     // [B][ldq] fp32, already scaled
    KVPool pool = {};
    const StepState* state = nullptr;
    // decode, self-attention only: an utterance whose K codebook rows have all latched EOS only produces pad tokens from here on
    // (select_kernel), whatever its logits are: its workgroups attend to key 0 alone instead of streaming the cache.  The
    // reference keeps computing finished rows ("extension"); tokens and waveforms are unchanged.
    const int* unfinished = nullptr;   // [B][alive_k] (select_kernel's latch), or null: never skip
    int ldq = 0;
    int layer = 0;
    // ---- line 1 ----
    const float* bias = nullptr;  // additive mask [B][ld_bias] (0 / finfo.min) or null
    const int* bias_flag = nullptr;  // single-page kernel: device word, 0 = `bias` holds no finfo.min at all (an all-ones mask): the kernel
                                     // then skips the mask loads (a quarter of its load instructions); null = always load
    void* ofrag = nullptr;        // output in B-fragment order, feature count D
    int len = 0;                  // number of keys when !len_from_state
    int len_from_state = 0;       // 1: keys = state->pos + 1 (self attention, after this step's append)
    int ld_bias = 0;
    int s_cap = 0;                // LDS score capacity (>= max keys, multiple of 4)
    int D = 0;
    // batched prefill: query row r belongs to utterance r / rows_per_utt (its KV pages, its mask row); decode: 1
    int rows_per_utt = 1;
    int alive_k = 0, alive_b = 0;      // codebooks per utterance; utterances (query row r belongs to utterance r % alive_b: guidance halves)
    // general kernel, self-attention only (len_from_state): `q` holds the RAW look-ahead product [q | k | v] (ldq = 3 D) of the
    // folded QKV projection; the kernel finishes the LayerNorm of its head's three 64-wide slices (c1 / c2 are [3 D]), scales
    // q, and appends the k / v rows to the cache at state->pos itself before it streams the cache
    int lnfix_qkv = 0;       // 1: always use the general kernel (A/B and equivalence tests)
    int lnfix_blocks = 0;                // 16-column blocks per row (K / 16)
    // ---- line 2: look-ahead query ----
    // single-page kernel only: `q` holds the RAW look-ahead product x.W'^T; the LayerNorm of the folded projection
    // is applied here from per-block row statistics: q = scale * (rstd * (q - mean * c1) + c2)
    const float* lnfix_stat = nullptr;   // [((b / 32) * blocks + block) * 32 + b % 32][2] = (sum, sum of squares)
    const float* lnfix_c1 = nullptr;
    const float* lnfix_c2 = nullptr;
    float lnfix_eps = 1e-5f, lnfix_scale = 1.0f;
    int lnfix_rms = 0;                   // RMSNorm variant of the folded projection: no centring (mean = 0, var = mean(x^2))
    int no_page_kernel = 0;
    // causal prefill kernel only: the launch covers positions pos0 .. pos0 + npos - 1 of every utterance; query row
    // b * npos + i sits at cache position pos0 + i and attends the keys 0 .. pos0 + i (M:870-872 causal mask)
    int npos = 0, pos0 = 0;
#ifdef PTTS_STAMPS
    StampRef stamps = {nullptr, -1};
#endif
};
static_assert(offsetof(AttnArgs, bias) == 64 && offsetof(AttnArgs, lnfix_stat) == 128, "AttnArgs: the hot fields fill kernel-argument lines 0 and 1");

struct FullAttnArgs {
    const float* qkv = nullptr;   // [B*T][ld] rows = [q | k | v], each H*dk wide
    int ld = 0;
    int T = 0, H = 0, dk = 0;
    float scale = 1.0f;
    const float* rel_bias = nullptr;  // [H][n_rel], index (j - i + T - 1)
    int n_rel = 0;
    const int* mask = nullptr;        // [B][T] 1 = attend
    void* ofrag = nullptr;
};

hipError_t attn_init();
// ev_start / ev_stop (both or neither): bound to the dispatch itself through hipExtLaunchKernel, so their elapsed time is the
// kernel's own begin -> end (what a profiler reports), without barrier packets around it
hipError_t attn_decode_launch(const AttnArgs& a, int B, int H, int cache_dtype, int out_dtype, hipStream_t stream,
                              hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// causal self-attention of a prefill chunk (a.npos positions per utterance, B utterances): one wave per query row
hipError_t attn_causal_launch(const AttnArgs& a, int B, int H, int cache_dtype, int out_dtype, hipStream_t stream);
hipError_t attn_full_launch(const FullAttnArgs& a, int B, int out_dtype, hipStream_t stream);
