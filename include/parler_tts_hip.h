/*
 * parler_tts_hip.h - C ABI of the MI355X (gfx950) Parler-TTS generate() engine.
 *
 * The reference (danlyth/parler-tts) has no FFI layer: its boundary is the Python API
 * `ParlerTTSForConditionalGeneration.generate()` (parler_tts/modeling_parler_tts.py:2342-2686).
 * `parler_tts_amd/modeling.py` keeps that Python API and drives the hot path through the entry points
 * below (ctypes; see INTEGRATION.md for the binding a maintainer of the reference would add).
 * Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in _host
 *   - tensors are caller-owned; the engine owns only its packed weights, the paged KV pools and
 *     workspaces (allocated in ptts_create / ptts_prepare, never in the per-step calls)
 *   - every call that launches work is asynchronous on `stream` (a hipStream_t passed as void*).
 *     NULL (the legacy default stream, which cannot be captured into a hipGraph) is accepted too: the
 *     work then runs on a stream the engine owns and the call is host-synchronous (it waits for the
 *     NULL stream before starting and returns when its own work has finished)
 *   - token ids index embedding tables directly: the caller guarantees 0 <= id < rows (the Python layer raises IndexError
 *     first, as the reference's nn.Embedding does); tensor extents follow from the size arguments and cannot be checked here
 *   - return value: 0 = ok, negative = PTTS_E_* ; ptts_last_error() gives the text (HIP error
 *     strings included); no exceptions cross the ABI
 *   - one engine per device; not thread-safe per engine, re-entrant across engines
 */
#ifndef PARLER_TTS_HIP_H
#define PARLER_TTS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTTS_ABI_VERSION 8

enum { PTTS_DTYPE_F32 = 0, PTTS_DTYPE_BF16 = 1 };

enum {
    PTTS_OK = 0,
    PTTS_E_INVALID = -1,    /* bad argument / shape */
    PTTS_E_STATE = -2,      /* call order (weights missing, prefill not run, ...) */
    PTTS_E_HIP = -3,        /* a HIP runtime call failed */
    PTTS_E_NOMEM = -4,
    PTTS_E_UNSUPPORTED = -5
};

/* Architecture numbers: ParlerTTSDecoderConfig (configuration_parler_tts.py:82-127), the T5 encoder
 * config, DACConfig (dac_wrapper/configuration_dac.py:8-25). */
typedef struct ptts_config {
    int32_t abi_version;        /* PTTS_ABI_VERSION */
    int32_t dtype;              /* weights + KV cache storage: PTTS_DTYPE_*; accumulation is always fp32 */
    /* decoder */
    int32_t hidden, layers, heads, ffn, vocab, codebooks, max_positions;
    int32_t pad_token_id, eos_token_id, bos_token_id;
    /* composite */
    int32_t prompt_vocab;       /* rows of embed_prompts */
    int32_t enc_hidden;         /* T5 d_model (enc_to_dec_proj exists iff enc_hidden != hidden) */
    /* T5 encoder */
    int32_t t5_layers, t5_heads, t5_dkv, t5_dff, t5_vocab, t5_gated, t5_act; /* act: 1 gelu, 2 gelu_new, 3 relu */
    float t5_eps;
    /* limits used to size pools/workspaces */
    int32_t max_batch;          /* utterances per call (x2 internally under classifier-free guidance) */
    int32_t max_desc_len;       /* T */
    int32_t max_prompt_len;     /* P */
    int32_t max_length;         /* generation columns L (<= max_positions - P) */
    /* codec decoder (DACConfig + the descript package defaults); dac_hidden = 0: engine without the codec */
    int32_t dac_codebook_size, dac_codebook_dim, dac_latent, dac_hidden, dac_n_up;
    int32_t dac_up[8];          /* upsampling strides, e.g. 8,8,4,2 */
    /* codec encoder (audio prompts): descript `encoder_dim` (64); its strides are dac_up reversed; 0 = engine without it */
    int32_t dac_enc_hidden;
    /* decoder-op variants for random-init scaled-decoder studies (EXTENSIONS: the reference's decoder has LayerNorm with bias,
     * additive sinusoidal positions and a 2-matrix GELU MLP, modeling_parler_tts.py:402-413, :181-223, :488-490, and no switch):
     * bit 0 RMSNorm (x * rsqrt(mean(x^2) + 1e-5) * weight; the *.layer_norm.bias tensors are accepted and ignored),
     * bit 1 rotary positions on the self-attention (pairs (i, i + 32) per head; tables decoder.rope.cos / decoder.rope.sin
     *       [max_positions][32] fp32 arrive through ptts_load_tensor; no additive position table),
     * bit 2 SwiGLU MLP (fc2(silu(fc1(x)) * fc1_up(x)); extra tensor decoder.model.decoder.layers.{l}.fc1_up.weight [ffn][hidden]).
     * 0 = the reference's ops. */
    int32_t dec_variant;
} ptts_config;

typedef struct ptts_engine ptts_engine;

const char* ptts_last_error(void);
int ptts_abi_version(void);

/* ---- lifetime ---- */
int ptts_create(const ptts_config* cfg, ptts_engine** out);
int ptts_destroy(ptts_engine* e);

/* Hand one fp32 tensor of the reference state dict (key layout: SURVEY.md 8b) to the engine, which
 * re-tiles / converts it into its own HBM layout (packed MFMA tiles, bf16 tables).  `data` may be
 * freed after the stream has drained.  Replaces nn.Module parameter ownership
 * (modeling_parler_tts.py:760-781, 1069-1071, 1635-1642). */
int ptts_load_tensor(ptts_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim,
                     void* stream);
/* 0 when every tensor of the decoder (+T5 when t5_layers > 0) has arrived; otherwise PTTS_E_STATE and the
 * first missing names in ptts_last_error(). */
int ptts_weights_ready(ptts_engine* e);

/* ---- text encoder: T5EncoderModel(input_ids, attention_mask).last_hidden_state
 *      (called at modeling_parler_tts.py:2229).  rel_bias [t5_heads][2T-1] is the bucketed relative
 *      position bias gathered by the host (index j - i + T - 1).  out: [B][T][enc_hidden] fp32. ---- */
int ptts_t5_encode(ptts_engine* e, int32_t B, int32_t T, const int64_t* input_ids, const int32_t* attention_mask,
                   const float* rel_bias, float* out, void* stream);

/* Pre-embedded prompts (`prompt_hidden_states`, modeling_parler_tts.py:839-840; takes precedence over
 * `prompt_input_ids` as at :2476): embeds [B_eff][P][hidden] fp32, consumed by the NEXT ptts_prefill instead of
 * embed_prompts[prompt_ids] (prompt_ids may then be NULL) and kept alive by the caller until that prefill has
 * completed; NULL restores the id lookup. */
int ptts_set_prompt_embeds(ptts_engine* e, const float* embeds);

/* `decoder_attention_mask` (modeling_parler_tts.py:845-846: concatenated behind prompt_attention_mask, then the 4-D causal
 * mask of :870-872; the third-party loop appends a column of ones per generated token): mask [B_eff][start_len] int32
 * over the prefix columns (start column included), consumed by the NEXT ptts_prefill, which gives the cache positions
 * P .. P+start_len-1 of masked columns the additive finfo.min bias; kept alive by the caller until that prefill has
 * completed; NULL = none. */
int ptts_set_decoder_mask(ptts_engine* e, const int32_t* mask);

/* ---- prefill: everything generate() does before the token loop (modeling_parler_tts.py:2467-2537)
 *      plus the first forward pass over the prompt (decoder forward M:794-978 with P+1 positions):
 *      enc_to_dec_proj + mask (M:2010-2017), cross K/V of every layer (M:295-298), prompt embedding
 *      (M:2243-2245), the P prompt positions through the self-attention cache.
 *      B_eff rows are fed (= 2B under classifier-free guidance: the caller passes the zeroed "null"
 *      half exactly as M:2231-2237 builds it).
 *        enc_states   [B_eff][T][enc_hidden] fp32      attention_mask [B_eff][T] int32 (never NULL)
 *        prompt_ids   [B_eff][P] int64 (P may be 0)    prompt_mask [B_eff][P] int32 or NULL
 *        pattern      [B*K][L] int64 delay pattern (build_delay_pattern_mask, M:77-139)
 *        raw_ids      [B*K][L] int64 caller-owned output; column 0.. are filled by decode steps
 *      start_len > 1 = a decoder prefix (user decoder_input_ids, modeling_parler_tts.py:2150-2195): columns
 *      0..start_len-2 of raw_ids (as masked by `pattern`) pass through the cache like prompt positions.
 *      After it, StepState = {pos = P + start_len - 1, cur_len = start_len = first free column}. ---- */
int ptts_prefill(ptts_engine* e, int32_t B, int32_t B_eff, int32_t T, int32_t P, int32_t L,
                 const float* enc_states, const int32_t* attention_mask, const int64_t* prompt_ids,
                 const int32_t* prompt_mask, const int64_t* pattern, int64_t* raw_ids, int32_t start_len,
                 void* stream);

/* ---- codec: DACModel.decode (dac_wrapper/modeling_dac.py:98-134) = quantizer.from_codes + DAC.decode.
 *      codes [B][K][F] int64 (every id < codebook_size), out [B][F * hop] fp32 waveform.  Conv stack =
 *      implicit-GEMM MFMA kernels with fused Snake / bias / residual / tanh (csrc/conv.hip).  The first
 *      call (and any call needing a larger workspace) allocates two ping-pong activation buffers. ---- */
int ptts_dac_decode(ptts_engine* e, int32_t B, int32_t F, const int64_t* codes, float* out, void* stream);

/* ---- codec, encode direction: DACModel.encode (dac_wrapper/modeling_dac.py:25-96) as generate() calls it for audio
 *      prompts (`input_values`, modeling_parler_tts.py:2247-2292, :2483-2487) = the descript DAC encoder conv stack +
 *      the residual vector quantiser.  audio [B][T] fp32 (one channel), codes [B][K][T / hop] int64 (floor: the
 *      reference cuts the preprocess padding off again before it encodes, :85-86, and every strided conv floors).
 *      Same conv kernels as the decode side; a stride-s conv with kernel 2s is a 3-tap conv over the [T/s][s*C] view of
 *      its channels-last input.  Shares the workspace of ptts_dac_decode. ---- */
int ptts_dac_encode(ptts_engine* e, int32_t B, int32_t T, const float* audio, int64_t* codes, void* stream);

/* Sampling parameters of the third-party GenerationMixin step the reference calls at
 * modeling_parler_tts.py:2582 (greedy) / :2608 (sample).  */
typedef struct ptts_sampling {
    int32_t do_sample;       /* 0: argmax */
    float temperature;       /* scores / temperature */
    int32_t top_k;           /* 0 = off */
    float guidance_scale;    /* > 1: rows [0,B*K) conditional, [B*K,2B*K) unconditional (M:2556-2558) */
    uint64_t seed;           /* Philox key; draws are (seed, row, step) so runs are reproducible */
    float top_p;             /* nucleus warper applied after top-k; <= 0 or >= 1 = off */
    /* generation_config.eos_token_id / pad_token_id of THIS call (the third-party loop pads finished rows with the
     * generation config's pad id and latches on its eos id, not on the decoder config's); -1 = the ptts_config values,
     * eos_token_id <= -2 = no EOS criterion at all (generation_config.eos_token_id = None: rows never finish early) */
    int32_t eos_token_id, pad_token_id;
    /* MinLengthLogitsProcessor / MinNewTokensLengthLogitsProcessor of the third-party `_get_logits_processor` the
     * reference calls at modeling_parler_tts.py:2561-2567: score[eos] = -inf while the id matrix has fewer than
     * eos_free_until columns ( = max(min_length, start_len + min_new_tokens) ); 0 = off */
    int32_t eos_free_until;
} ptts_sampling;

/* ---- decode: n_steps iterations of {prepare_inputs_for_generation (M:2081-2148), forward
 *      (M:1927-2079 -> M:794-978 -> M:1096-1185), token selection, pad-out of finished rows, EOS
 *      latch}.  State stays on the device; steps are replayed from a captured hipGraph.
 *      profile_every > 0: every profile_every-th step is launched eagerly with HIP events around the
 *      self-attention kernels (read back with ptts_profile_read).
 *      REQUIRED after the last ptts_decode_steps / ptts_select_step of a call and BEFORE raw_ids (or anything derived from
 *      them) is used: one read of the loop state - ptts_get_state, or a ptts_state_snapshot taken behind the last step and
 *      polled.  That read is where an error of the work already queued surfaces: these calls only enqueue, and a decode layer's
 *      QKV projection + self-attention run as ONE launch whose attention half waits for the projection half through in-grid
 *      arrival tags with a bounded (2 s of device wall time), sticky wait - if it ever gives up, the step used a stale query,
 *      the error word is set, and ptts_get_state / ptts_state_poll return PTTS_E_HIP ("gave up waiting ... results invalid").
 *      Invariant behind those tags (enforced here, stated for completeness): a tag is the cache position + 1, so every pass over
 *      the layers must run at a position of its own within one ptts_prefill .. decode sequence - ptts_decode_steps clamps n_steps
 *      to the columns left, ptts_forward_step refuses a second forward half before its ptts_select_step and at max_length, and
 *      ptts_prefill clears the tags.
 *      The loop state a read returns is settled at every return of these calls: inside a run of replayed steps the counters and the
 *      unfinished-row count of a step are written by the NEXT step's first launch (no launch of their own), and the call ends with
 *      one settling launch behind its last replay. ---- */
int ptts_decode_steps(ptts_engine* e, int32_t n_steps, const ptts_sampling* sp, int32_t profile_every, void* stream);

/* One decode step in two halves, for callers that edit the scores in between (custom LogitsProcessor lists,
 * modeling_parler_tts.py:2561-2567, applied inside the third-party greedy_search / sample before the warpers):
 *   ptts_forward_step : embed -> decoder layers -> LM heads; the logits [B_eff*K][Vp] (Vp = vocab rounded up to 16)
 *                       are the device buffer ptts_last_logits returns and may be overwritten in place
 *   ptts_select_step  : the selection half of ptts_decode_steps on those logits (guidance combine, temperature /
 *                       top-k / top-p, argmax or draw, pad-out, EOS latch, delay-pattern feed, state advance)
 * Under guidance the buffer holds the conditional rows first, then the unconditional ones; writing the same
 * processed scores to both halves makes the built-in combine return them unchanged. */
int ptts_forward_step(ptts_engine* e, void* stream);
int ptts_select_step(ptts_engine* e, const ptts_sampling* sp, void* stream);

/* Loop state (host copy; synchronises `stream`): cur_len, unfinished rows, the column count at which
 * every row had finished (0 = not yet), steps run.  Also the point where a failed in-grid hand-off of the queued steps is
 * reported (PTTS_E_HIP; see ptts_decode_steps): every caller reads it once behind its last step. */
typedef struct ptts_state {
    int32_t pos, cur_len, prompt_len, n_unfinished, done_len, step_count;
} ptts_state;
int ptts_get_state(ptts_engine* e, ptts_state* out_host, void* stream);

/* The same without stopping the stream (ABI v8).  The third-party loop the reference calls at M:2582 / M:2608 tests
 * `unfinished_sequences.max() == 0` after EVERY step, on the host; here the loop state lives on the device and a blocking read
 * every few steps would drain the queue of replayed steps.  ptts_state_snapshot enqueues, behind the work already on `stream`, a
 * one-thread kernel that copies the loop state into pinned host-mapped slot `slot` (0 .. PTTS_STATE_SLOTS-1) and stamps it;
 * ptts_state_poll watches THAT stamp from the host (no runtime call on the stream: work enqueued after the snapshot keeps
 * running) and returns the copy.  A caller that enqueues chunk i + 1
 * before it polls chunk i's snapshot learns of the end of the run at most two chunks late and never lets the stream run dry.
 * ptts_state_poll on a slot with no snapshot taken since the last poll is PTTS_E_STATE. */
#define PTTS_STATE_SLOTS 4
int ptts_state_snapshot(ptts_engine* e, int32_t slot, void* stream);
int ptts_state_poll(ptts_engine* e, int32_t slot, ptts_state* out_host);

/* logits of the last executed step [B_eff*K][vocab] fp32 (device pointer owned by the engine), for
 * parity tests; probabilities after warping when sampling. */
int ptts_last_logits(ptts_engine* e, const float** logits, const float** probs);

/* Override the identity page table of the self-attention KV pool: table [max_batch_eff][max_blocks]
 * int32 of distinct page ids < n_pages (tests shuffle it; results must not change). */
int ptts_set_block_table(ptts_engine* e, const int32_t* table_host, int32_t rows, int32_t max_blocks);
int ptts_pool_info(ptts_engine* e, int32_t* n_pages, int32_t* max_blocks, int64_t* page_bytes);

/* Test hook (teacher forcing): tokens_host [B*K] int64; a value >= 0 replaces the next steps' selected
 * token of that row before the pad-out / EOS latch, -1 leaves the row alone; NULL clears.  While set,
 * steps are launched eagerly. */
int ptts_set_forced_tokens(ptts_engine* e, const int64_t* tokens_host, int32_t rows);

/* Teacher forcing along a whole id matrix kept in DEVICE memory (parity tests at depth: thousands of steps under graph
 * replay): the step that fills column c of raw_ids takes forced_ids[row][c] when that is >= 0, and writes the token it
 * would have chosen itself to own_choice[row][c] (own_choice may be NULL).  Both [B*K][L] int64, caller-owned, alive until
 * cleared; NULL forced_ids clears both.  Takes effect with the next ptts_prefill / ptts_decode_steps. */
int ptts_set_forced_sequence(ptts_engine* e, const int64_t* forced_ids, int64_t* own_choice);

/* Debug hooks for kernel-level parity tests: stop issuing kernels after n launches of each following
 * prefill / decode call (-1 = off; while on, steps run eagerly), and expose engine-owned buffers
 * ("x", "q", "attn_frag", "mlp_frag", "enc_h", "self_pool", "cross_pool", "self_bias", "cross_bias", "tok_in",
 * "codec_latent" = the encoder output [b][F][latent] of the last ptts_dac_encode chunk). */
int ptts_debug_stop_after(ptts_engine* e, int32_t n_kernels);
int ptts_debug_buffer(ptts_engine* e, const char* name, void** ptr, int64_t* bytes);

/* Test hook: how many conv launches of the codec (ptts_dac_decode / ptts_dac_encode, any engine of this process) went to each
 * kernel family since the last reset, so that a parity test at production size can assert that the production kernels -
 * not a fallback - produced the waveform it compared.  counts_host[i], i < n:
 *   0 conv_last_kernel  1 / 2 conv1x1_stream_kernel<96 / 192>  3 / 4 fused residual unit (conv_kernel_h3r<.., true>) 96 / 192 wide
 *   5 conv_kernel_s1  6 / 7 / 8 conv_kernel_h3r<96, 64, taps 7 / 2 (transposed) / 1>  9 / 10 / 11 conv_kernel_h3r<192, 32, taps 7 / 2 / 1>
 *   12 conv_kernel_h3 (register-staged)  13 conv_kernel (fp32 MFMA).   PTTS_DAC_KERNEL_KINDS entries exist. */
#define PTTS_DAC_KERNEL_KINDS 14
int ptts_dac_kernel_counts(int64_t* counts_host, int32_t n, int32_t reset);

/* HIP-event profile of the dominant kernel (self-attention decode, or - ABI v8 - the fused QKV projection + self-attention launch
 * where a layer runs it): sums over the sampled launches. */
typedef struct ptts_profile {
    int64_t launches;        /* sampled launches of the dominant kernel */
    double total_ms;         /* sum of their durations */
    double algo_bytes;       /* sum of their algorithmic bytes: 2 * B_eff * S * hidden * sizeof(cache elem), + 3 * hidden^2 *
                                sizeof(weight) of QKV weights for a fused launch */
    int64_t steps;           /* sampled steps */
    double step_ms;          /* sum of the sampled steps' durations (eager) */
    int64_t fused_launches;  /* how many of `launches` were the fused QKV + self-attention launch (csrc/fused.hip) */
} ptts_profile;
int ptts_profile_read(ptts_engine* e, ptts_profile* out_host, int32_t reset);

/* Stand-alone kernels exposed for unit tests (same code paths the engine uses).  prologue: 0 fragments are not
 * reachable here, 1 = plain rows, 2 = LayerNorm (staged), 3 = T5 RMSNorm, 4 = LayerNorm through the folded path
 * (gamma folded into the weights, row statistics from the activation fragments, normalisation in the epilogue), 5 = plain
 * rows through the fragment-input path, 6 = 4 kept on the two-half workgroup where the dispatch would pick the
 * half-split grid (N <= 2048, M > 16; A/B of the two folded instances), 7 / 8 = 5 / 4 launched as the one-off phases launch
 * them: in the bf16 mode the product runs on the 64 x 64 MFMA tile kernel (csrc/gemm_tile.hip) that the T5 encoder, the batched
 * prefill and the cross K / V projections use.  With PTTS_TEST_GEMM_ITERS=n in
 * the environment the GEMM is relaunched n times and every result compared bitwise with the first (stderr report). */
int ptts_test_gemm(int32_t dtype, int32_t M, int32_t N, int32_t K, const float* W, const float* X, int32_t prologue,
                   const float* gamma, const float* beta, float eps, int32_t act, const float* bias, float* Y,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARLER_TTS_HIP_H */
