// Host-side interface of conv.hip (internal).
#pragma once
#include "ptts_common.h"

struct ConvArgs {
    const float* in = nullptr;     // [B][Tin][Cin] channels-last
    float* out = nullptr;          // [B][Tout][Cout]
    const float* w = nullptr;      // packed [taps][N][Cin]
    const void* w_hi = nullptr;    // the same weights split into fp16 hi / lo halves (fast path), or null
    const void* w_lo = nullptr;
    const float* bias = nullptr;   // [Cout]
    const float* alpha = nullptr;  // [Cin] Snake on the input, or null
    const float* res = nullptr;    // residual, same shape as out, or null (may alias out)
    int B = 0, Tin = 0, Cin = 0, Tout = 0, Cout = 0;
    int N = 0;                     // GEMM outputs per row: Cout (conv) or stride*Cout (transposed)
    int U = 0;                     // GEMM rows: Tout (conv) or Tin + 1 (transposed)
    int taps = 1, tstep = 1, toff = 0;
    int stride = 1, pad_t = 0;     // transposed: t = stride*u - pad_t + n / Cout
    int act_out = 0;               // 1 = tanh
};

hipError_t conv_init();
hipError_t conv_launch(const ConvArgs& a, hipStream_t s);
hipError_t dac_latent_launch(const long long* codes, const float* table, float* z, int B, int K, int F, int C, int ncodes,
                             hipStream_t s);
hipError_t dac_table_launch(const float* codebook, const float* proj_w, const float* proj_b, float* table, int ncodes, int dim,
                            int C, hipStream_t s);
hipError_t split_halves_launch(const float* src, void* hi, void* lo, size_t n, hipStream_t s);
hipError_t conv_pack_launch(const float* w, float* out, int Cout, int Cin, int k, int transposed, int stride, int N,
                            hipStream_t s);
