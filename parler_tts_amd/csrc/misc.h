// Host-side interface of misc.hip (internal).
#pragma once
#include "ptts_common.h"
#include "gemm.h"

struct EmbedArgs {
    const void* tables = nullptr;        // [K][rows][D] codebook embeddings (weight dtype)
    int rows = 0, K = 0, D = 0;
    const void* pos_table = nullptr;     // [max_positions][D]
    const int* tok_in = nullptr;         // [B*K] ids to feed (delay pattern already applied)
    const void* prompt_table = nullptr;  // [prompt_vocab][D]
    const long long* prompt_ids = nullptr;  // [B][P]
    const float* prompt_embeds = nullptr;   // [B][P][D] fp32 pre-embedded prompts (override the table lookup), or null
    int P = 0;
    StepState* state = nullptr;          // [2]: the loop state and the counters the last selection wrote for the next step (ptts_common.h)
    const int* unfinished = nullptr;     // [n_latch] EOS latches: embed_codebooks_launch settles the loop state from them (null: it does not)
    int n_latch = 0;
    float* x = nullptr;                  // [B][D] residual stream
    void* xfrag = nullptr;               // optional fragment-layout copy of x (weight dtype) for the folded LayerNorm path
};

struct SelectArgs {
    const float* logits = nullptr;       // [rows (x2 under guidance)][V]
    int rows = 0, V = 0, vocab = 0, L = 0;   // V = row stride of logits/probs, vocab = valid entries
    StepState* state = nullptr;
    long long* raw_ids = nullptr;        // [rows][L]
    const long long* pattern = nullptr;  // [rows][L]
    int* unfinished = nullptr;           // [rows]
    int* tok_in = nullptr;               // [rows (x2)]
    const long long* force = nullptr;    // [rows] or null
    const long long* force_seq = nullptr;   // [rows][L] or null: column cur_len overrides the choice where >= 0
    long long* own = nullptr;               // [rows][L] or null: the step's own choice goes to column cur_len
    int eos_free_until = 0;                 // score[eos] = -inf while cur_len < this (MinLength / MinNewTokens processors)
    float* probs = nullptr;              // [rows][V] or null
    int pad = 0, eos = 0;
    int do_sample = 0, top_k = 0;
    float temperature = 1.0f, guidance = 1.0f, top_p = 1.0f;
    unsigned long long seed = 0;
};

hipError_t embed_codebooks_launch(const EmbedArgs& a, int B, int dtype, hipStream_t s);
hipError_t embed_prompt_launch(const EmbedArgs& a, int B, int dtype, hipStream_t s);
hipError_t advance_prompt_launch(StepState* st, hipStream_t s);
// *st = v and (unfinished != NULL) unfinished[0 .. n_flags) = 1, on the device: the loop state of a new call without a host copy
hipError_t reset_state_launch(StepState* st, const StepState& v, int* unfinished, int n_flags, hipStream_t s);
// batched prefill: rows b * npos + i of x (+ fragments) for the chunk's positions pos0 .. pos0 + npos - 1.
// prompt rows: embed_prompts[prompt_ids[b][pos]] (or the pre-embedded row) + pos_table[pos];
// prefix rows (decoder_input_ids columns): sum_k embed_tokens[k][masked id of column col0 + i] + pos_table[pos0 + i]
hipError_t embed_prompt_rows_launch(const EmbedArgs& a, int B, int npos, int pos0, int dtype, hipStream_t s);
hipError_t embed_prefix_rows_launch(const EmbedArgs& a, int B, int npos, int pos0, int col0, const long long* raw_ids,
                                    const long long* pattern, int rows_src, int L, int dtype, hipStream_t s);
hipError_t select_launch(const SelectArgs& a, hipStream_t s);
// settles st[0] from st[1] (the counters select_launch left for the next step) and the latches; idempotent, and the first
// workgroup past the utterances of embed_codebooks_launch does the same thing at the head of the next step
hipError_t advance_launch(StepState* st, const int* unfinished, int rows, int L, hipStream_t s);
hipError_t init_tokens_launch(const long long* raw_ids, const long long* pattern, int* tok_in, int rows, int rows_eff,
                              int L, int col, hipStream_t s);
// any_masked (optional device word): set to 1 iff the mask holds a zero, i.e. the bias holds a finfo.min (else 0)
hipError_t build_bias_launch(const int* mask, float* bias, int B, int n_mask, int ld_bias, hipStream_t s, int* any_masked = nullptr);
hipError_t build_bias_segment_launch(const int* mask, float* bias, int B, int n, int ld_bias, int offset, hipStream_t s);
hipError_t rowmask_mul_launch(const float* x, const int* rowmask, float* y, int rows, int D, hipStream_t s);
hipError_t t5_embed_launch(const long long* ids, const void* table, float* x, int rows, int D, int dtype, hipStream_t s);
hipError_t rmsnorm_rows_launch(const float* x, const float* w, float* y, int rows, int D, float eps, hipStream_t s);

// ---- decoder-op variants (extensions: configuration.py norm_type / position_embedding / mlp_type; nothing in the reference) ----
// Rotary positions on the self-attention: qk [rows][ld] fp32 holds the scaled q of every row in columns [0, D) and its
// unrotated k in [D, 2 D) (EPI_QKV_SELF with rope_k); pairs (i, i + 32) of every 64-wide head are rotated by the angle of
// the row's cache position (decode: state->pos; prefill chunk: pos0 + row % npos), q in place, k into the paged cache.
// cos / sin: [max_positions][32] fp32 (built by the host with the oracle's formula, so the tables are bit-identical).
struct RopeArgs {
    float* qk = nullptr;
    int ld = 0, D = 0, H = 0, rows = 0;
    KVPool pool = {};
    int layer = 0;
    const StepState* state = nullptr;
    int npos = 0, pos0 = 0;
    const float* cos_t = nullptr;
    const float* sin_t = nullptr;
};
hipError_t rope_launch(const RopeArgs& a, int dtype, hipStream_t s);
// gated MLP: raw [rows][2 F] fp32 = [gate | up] pre-activations -> silu(gate) * up as B fragments [(rows + 31) / 32][32 * F]
hipError_t swiglu_frag_launch(const float* raw, void* ofrag, int rows, int F, int dtype, hipStream_t s);
