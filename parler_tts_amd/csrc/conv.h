// Host-side interface of conv.hip (internal).
#pragma once
#include "ptts_common.h"

struct ConvArgs {
    const float* in = nullptr;     // [B][Tin][Cin] channels-last
    float* out = nullptr;          // [B][Tout][Cout]
    const float* w = nullptr;      // packed [taps][N][Cin]
    const void* w_hi = nullptr;    // the same weights split into fp16 hi / lo halves (fast path), or null
    const void* w_lo = nullptr;
    const void* w_units = nullptr; // the split weights as LDS-DMA units [N / unit_tn][Cin / 32][taps][hi | lo][unit_tn][32] (conv_units_launch), or null
    int unit_tn = 0;
    // fused residual unit (conv_kernel_h3r, C = unit_tn = 96): out = in + bias2 + W2 . snake_alpha2(bias + conv7(snake_alpha(in)));
    // w2_units = the 1x1 weights as A fragments in the accumulator's k order (conv_units2_launch); `res` stays null, out != in
    const void* w2_units = nullptr;
    const float* bias2 = nullptr;
    const float* alpha2 = nullptr;
    const float* bias = nullptr;   // [Cout]
    const float* alpha = nullptr;  // [Cin] Snake on the input, or null
    const float* res = nullptr;    // residual, same shape as out, or null (may alias out)
    int B = 0, Tin = 0, Cin = 0, Tout = 0, Cout = 0;
    int N = 0;                     // GEMM outputs per row: Cout (conv) or stride*Cout (transposed)
    int U = 0;                     // GEMM rows: Tout (conv) or Tin + 1 (transposed)
    int taps = 1, tstep = 1, toff = 0;
    int stride = 1, pad_t = 0;     // transposed: t = stride*u - pad_t + n / Cout
    int act_out = 0;               // 1 = tanh
    // Packed activations between a block's last layer and the NEXT block's transposed conv (round 6; decided per pair by
    // conv_can_pack).  A transposed conv re-stages its input window - global load, Snake, fp16 hi / lo split, ds_write, barrier - once
    // per 192-wide output block (32 / 16 / 4 / 1 blocks) for two taps of MFMA work each time.  With `alpha_out` set the PRODUCER applies
    // that Snake and the split once per element in its epilogue and stores, where the fp32 values of an 8-channel group stood, the
    // 32 bytes [hi x 8 | lo x 8] (conv.hip: store_packed8); with `in_packed` set the CONSUMER is convt_dma_kernel, whose whole
    // activation window arrives by LDS-DMA a chunk ahead - no register staging, no arithmetic, no ds_write, no stash barrier (its
    // `alpha` is then null).  Same values as before: the waveform is bit-identical (test).
    const float* alpha_out = nullptr;   // producer: Snake alpha [Cout] of the consuming transposed conv, or null (fp32 output)
    int in_packed = 0;                  // consumer (2-tap ring kernels only): `in` holds packed quads
    size_t in_bs = 0, out_bs = 0;  // floats between utterances of in / out (and res); 0 = dense (Tin*Cin, Tout*Cout)
};

// kernel families conv_launch dispatches to (indices of ptts_dac_kernel_counts, include/parler_tts_hip.h)
enum {
    CONV_KIND_LAST = 0,        // conv_last_kernel (one output channel)
    CONV_KIND_STREAM_96 = 1,   // conv1x1_stream_kernel<96>
    CONV_KIND_STREAM_192 = 2,  // conv1x1_stream_kernel<192>
    CONV_KIND_FUSED_96 = 3,    // conv_kernel_h3r<96, 64, 7, 4, true>   fused residual unit
    CONV_KIND_FUSED_192 = 4,   // conv_kernel_h3r<192, 32, 7, 4, true>
    CONV_KIND_S1 = 5,          // conv_kernel_s1 (windowless wide 1x1)
    CONV_KIND_RING_96_T7 = 6,  // conv_kernel_h3r<96, 64, 7 | 2 | 1, 4, false>: 6, 7 (transposed), 8
    CONV_KIND_RING_96_T2 = 7,
    CONV_KIND_RING_96_T1 = 8,
    CONV_KIND_RING_192_T7 = 9, // conv_kernel_h3r<192, 64, 7, 4, false, 2> (64 x 96 wave tiles) | <192, 32, 2 | 1, 4, false>: 9, 10 (transposed: convt_dma_kernel on packed input), 11
    CONV_KIND_RING_192_T2 = 10,
    CONV_KIND_RING_192_T1 = 11,
    CONV_KIND_H3 = 12,         // conv_kernel_h3 (register-staged fp16 x3 kernel)
    CONV_KIND_F32 = 13,        // conv_kernel (fp32 MFMA)
    CONV_KIND_COUNT = 14
};
void conv_kernel_counts(long long* out, int n, bool reset);

hipError_t conv_init();
hipError_t conv_launch(const ConvArgs& a, hipStream_t s);
// the CONV_KIND_* conv_launch would dispatch `a` to (nothing is launched), -1 if it would refuse
int conv_kind_of(const ConvArgs& a);
// true when `producer` (its output = `consumer`'s input, consumer = a transposed conv with a Snake) may write packed quads and
// `consumer` read them: both on kernel families that implement it; PTTS_DAC_NO_PRESPLIT=1 says no (A/B)
bool conv_can_pack(const ConvArgs& producer, const ConvArgs& consumer);
hipError_t dac_latent_launch(const long long* codes, const float* table, float* z, int B, int K, int F, int C, int ncodes,
                             hipStream_t s);
hipError_t dac_table_launch(const float* codebook, const float* proj_w, const float* proj_b, float* table, int ncodes, int dim,
                            int C, hipStream_t s);
// same bytes as w_hi + w_lo, regrouped so that one tap of one 32-channel chunk of one TN-wide output block is contiguous and
// already in the swizzled order the kernel reads from LDS
hipError_t conv_units_launch(const void* hi, const void* lo, void* out, int taps, int N, int Cin, int TN, hipStream_t s);
// the split [C][C] weights of a 1x1 conv as stage-2 units of the fused residual unit
hipError_t conv_units2_launch(const void* hi, const void* lo, void* out, int C, hipStream_t s);
hipError_t split_halves_launch(const float* src, void* hi, void* lo, size_t n, hipStream_t s);
hipError_t conv_pack_launch(const float* w, float* out, int Cout, int Cin, int k, int transposed, int stride, int N,
                            hipStream_t s);

// ---- codec ENCODE side (audio prompts, dac_wrapper/modeling_dac.py:25-96) ----
// first conv of the encoder: one input channel, k = 7, zero padding 3: out[b][t][c] = bias[c] + sum_k w[c][k] * x[b][t+k-3]
hipError_t enc_first_conv_launch(const float* audio, const float* w, const float* bias, float* out, int B, int T, int C,
                                 size_t out_bs, hipStream_t s);
// buf[b*bs + i] = 0 for i in [from, to): the rows a [T/s][s*C] view reads past the true length must be zero (Snake(0) = 0)
hipError_t zero_tail_launch(float* buf, int B, size_t bs, size_t from, size_t to, hipStream_t s);
// Conv1d(C -> N, kernel 2s, stride s, padding ceil(s/2)) as a 3-tap stride-1 conv over the [T/s][s*C] view of its input:
// out[tap][n][j*C + c] = w[n][c][(tap-1)*s + j + pad] (0 outside the kernel), taps at rows t-1, t, t+1
hipError_t conv_pack_strided_launch(const float* w, float* out, int N, int C, int stride, hipStream_t s);
hipError_t tile_alpha_launch(const float* alpha, float* out, int C, int reps, hipStream_t s);
// L2-normalised codebook rows (F.normalize, eps 1e-12) and their squared norms
hipError_t cb_normalize_launch(const float* cb, float* cbn, float* cbn2, int entries, int dim, hipStream_t s);
struct RvqArgs {
    const float* z = nullptr;            // [B][F][D] latent, channels-last
    int B = 0, F = 0, D = 0, K = 0, entries = 0, dim = 0;
    const float* const* in_w = nullptr;  // [K] -> [dim][D]
    const float* const* in_b = nullptr;  // [K] -> [dim]
    const float* const* cb = nullptr;    // [K] -> [entries][dim] raw codebook
    const float* const* cbn = nullptr;   // [K] -> normalised rows
    const float* const* cbn2 = nullptr;  // [K] -> [entries]
    const float* const* out_w = nullptr; // [K] -> [D][dim]
    const float* const* out_b = nullptr; // [K] -> [D]
    long long* codes = nullptr;          // [B][K][F]
};
hipError_t rvq_encode_launch(const RvqArgs& a, hipStream_t s);
