// DAC decoder convolutions as implicit GEMM on MFMA (gfx950), channels-last activations.
//
// Reference arithmetic: DACModel.decode (dac_wrapper/modeling_dac.py:98-134) -> descript DAC decoder;
// structure per HF models/dac/modeling_dac.py:86-100 (Snake), 175-210 (residual unit), 236-266 (decoder
// block), 407-442 (decoder):  conv7 -> R x [snake, convT(k=2s, stride s, pad s/2), 3 x {snake, conv7(dil),
// snake, conv1, +x}] -> snake -> conv7 -> tanh.
//
// One kernel covers every layer:
//     out[b][u][n] = epi( bias[n % Cout] + sum_tap sum_ci  A(in[b][u + tap*tstep + toff][ci]) * W[tap][n][ci] )
//   Conv1d          : taps = k, tstep = dilation, toff = -pad, n = co, output row u
//   ConvTranspose1d : a stride-s transposed conv with kernel 2s and padding s/2 is s interleaved 2-tap
//                     convolutions (output phase r = (t + pad) mod s uses taps r and r+s):
//                     taps = 2, tstep = -1, toff = 0, n = r*Cout + co, output time t = s*u - pad + r,
//                     u in [0, Tin]; W[tap][r*Cout+co][ci] = w[ci][co][r + tap*s] (repacked at load)
//   A() = Snake  x + sin^2(alpha x) / (alpha + 1e-9)  applied ONCE per staged input element (or identity)
//   epi  = + residual (same shape as out) | tanh
//
// Tiling: workgroup = TM(128) time rows x TN(64|96) outputs of one utterance, 4 waves (one per SIMD), each
// wave 32 rows x TN; per 16-channel chunk the input window (TM + halo rows, Snake applied) and the
// weights of all taps are staged in LDS; fp32 operands, v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain:
// the waveform must match the CPU reference to 1e-4 rms, which rules out plain bf16 operands).
//
// conv_kernel_h3 is the fast path for Cin % 32 == 0: every fp32 operand is split x = hi + lo with hi, lo
// in fp16 (11-bit mantissas: hi + lo carries 22 bits) and a product is hi*hi + hi*lo + lo*hi on
// v_mfma_f32_16x16x32_f16 with fp32 accumulation: fp32-class accuracy (dropped term ~2^-22 relative)
// at 3/16 of the matrix-pipe time of the fp32 MFMA.  Weights are split once at load, activations
// while they are staged (after Snake).  256 time rows x TN outputs per 8-wave workgroup, 32-channel chunks.
#include "ptts_common.h"
#include "conv.h"
#ifndef PTTS_CONV_ABLATE
#define PTTS_CONV_ABLATE 0      // timing-only ablations of conv_kernel_h3r (tools/gpu_calls.md); never in the product build
#endif
#include <atomic>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int TM = 128;      // time rows per workgroup
constexpr int KC = 16;       // input channels per staged chunk
constexpr int LDA = KC + 4;  // padded row strides (floats)
constexpr int MAX_TAPS = 7;

__device__ inline float snake_f(float x, float alpha, float inv) {
    // hardware sine (v_sin_f32 after a 1/2pi scale): the staging pass is VALU-bound in the narrow late blocks,
    // where the full-range software sinf costs more than the MFMAs it feeds.  |alpha*x| stays O(10) in the
    // codec, where the hardware sine is accurate to ~1e-6 absolute (waveform parity test: rms < 1e-5).
    const float s = __sinf(alpha * x);
    return x + inv * (s * s);
}

template <int TN>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = TN / 16;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * TN;
    const int u0 = blockIdx.y * TM;
    const int b = blockIdx.z;
    // window of input rows touched by this tile
    const int lo = a.tstep >= 0 ? a.toff : a.toff + (a.taps - 1) * a.tstep;
    const int hi = a.tstep >= 0 ? a.toff + (a.taps - 1) * a.tstep : a.toff;
    const int win = TM + (hi - lo);
    float* xs = reinterpret_cast<float*>(smem);                 // [win][LDA]
    float* ws = xs + (size_t)win * LDA;                         // [taps][TN][LDA]

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0, 0, 0, 0};

    const float* inb = a.in + (size_t)b * (a.in_bs ? a.in_bs : (size_t)a.Tin * a.Cin);
    for (int c0 = 0; c0 < a.Cin; c0 += KC) {
        // ---- stage the input window (Snake applied once per element) ----
        for (int i = tid; i < win * (KC / 4); i += 256) {
            const int r = i / (KC / 4), q = i % (KC / 4);
            const int t = u0 + lo + r;
            float4 v = make_float4(0, 0, 0, 0);
            if (t >= 0 && t < a.Tin) {
                v = *reinterpret_cast<const float4*>(inb + (size_t)t * a.Cin + c0 + 4 * q);
                if (a.alpha) {
                    const float4 al = *reinterpret_cast<const float4*>(a.alpha + c0 + 4 * q);
                    v.x = snake_f(v.x, al.x, 1.0f / (al.x + 1e-9f));
                    v.y = snake_f(v.y, al.y, 1.0f / (al.y + 1e-9f));
                    v.z = snake_f(v.z, al.z, 1.0f / (al.z + 1e-9f));
                    v.w = snake_f(v.w, al.w, 1.0f / (al.w + 1e-9f));
                }
            }
            *reinterpret_cast<float4*>(xs + (size_t)r * LDA + 4 * q) = v;
        }
        // ---- stage the weights of every tap for this channel chunk: W[tap][n][ci] ----
        for (int i = tid; i < a.taps * TN * (KC / 4); i += 256) {
            const int q = i % (KC / 4), n = (i / (KC / 4)) % TN, tap = i / ((KC / 4) * TN);
            float4 v = make_float4(0, 0, 0, 0);
            if (n0 + n < a.N)
                v = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.N + n0 + n) * a.Cin + c0 + 4 * q);
            *reinterpret_cast<float4*>(ws + ((size_t)tap * TN + n) * LDA + 4 * q) = v;
        }
        __syncthreads();
        for (int tap = 0; tap < a.taps; ++tap) {
            const int roff = tap * a.tstep + a.toff - lo;        // window row of output row 0 for this tap
            const float* xa = xs + (size_t)(w * 32 + roff + (lane & 15)) * LDA + (lane >> 4);
            const float* wb = ws + ((size_t)tap * TN + (lane & 15)) * LDA + (lane >> 4);
#pragma unroll
            for (int kk = 0; kk < KC; kk += 4) {
                const float a0 = xa[kk], a1 = xa[16 * LDA + kk];
                float bf[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[j] = wb[(size_t)j * 16 * LDA + kk];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    // weights as the A operand (rows = outputs), activations as B (cols = time):
                    // D[n = 4*(l>>4)+r][t = l&15] -> a lane owns 4 consecutive outputs of one time row
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], a0, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], a1, acc[1][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: bias, residual, tanh; channels-last stores of 4 consecutive outputs ----
    const size_t obs = a.out_bs ? a.out_bs : (size_t)a.Tout * a.Cout;
    float* outb = a.out + (size_t)b * obs;
    const float* resb = a.res ? a.res + (size_t)b * obs : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = u0 + w * 32 + i * 16 + (lane & 15);
        if (u >= a.U) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 16 + 4 * (lane >> 4);
            if (n >= a.N) continue;
            int t, co;
            if (a.stride == 1) { t = u; co = n; }
            else { const int r = n / a.Cout; co = n - r * a.Cout; t = a.stride * u - a.pad_t + r; }
            if (t < 0 || t >= a.Tout) continue;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = acc[i][j][e] + (co + e < a.Cout ? a.bias[co + e] : 0.f);
            float* o = outb + (size_t)t * a.Cout + co;
            if (co + 3 < a.Cout) {
                if (resb) {
                    const float4 rv = *reinterpret_cast<const float4*>(resb + (size_t)t * a.Cout + co);
                    y[0] += rv.x; y[1] += rv.y; y[2] += rv.z; y[3] += rv.w;
                }
                if (a.act_out == 1) { y[0] = tanhf(y[0]); y[1] = tanhf(y[1]); y[2] = tanhf(y[2]); y[3] = tanhf(y[3]); }
                *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
                for (int e = 0; e < 4 && co + e < a.Cout; ++e) {
                    float v = y[e];
                    if (resb) v += resb[(size_t)t * a.Cout + co + e];
                    if (a.act_out == 1) v = tanhf(v);
                    o[e] = v;
                }
            }
        }
    }
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#ifdef PTTS_CONV_LIFE
// diagnostic build (-DPTTS_CONV_LIFE; no stamp inside the chunk loop): what a ring-kernel workgroup spends before its first tap, in
// its chunk loop and behind it, in s_memrealtime ticks (10 ns): [0] prologue [1] chunk loop [2] stage 2 + epilogue [3] workgroups
__device__ unsigned long long g_conv_life[4];
#define LIFE(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#else
#define LIFE(var) do { } while (0)
#endif
#ifdef PTTS_CONV_STAMPS
// diagnostic build (make CXXFLAGS+=-DPTTS_CONV_STAMPS): wave 0 of every workgroup adds the cycles of each phase of the chunk loop
// [0] fetch issue  [1] MFMA loop  [2] barrier after it  [3] stash  [4] barrier before the next chunk  [5] chunks  [6] whole kernel
__device__ unsigned long long g_conv_stamps[16];
__device__ unsigned long long g_conv_wave[8][8];      // ring kernel, per wave of the workgroup: the six phases summed over workgroups
#define STAMP(i, t0) do { const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); tacc[i] += t1_ - t0; t0 = t1_; } while (0)
#else
#define STAMP(i, t0) do { } while (0)
#endif
constexpr int KC3 = 32;       // channels per chunk = one fp16 MFMA k-step
constexpr int h3_xa(int RW) { return RW == 32 ? 6 : 9; }                       // window float4 per thread the register staging holds
constexpr int h3_max_span(int RW) { return h3_xa(RW) * 512 / (KC3 / 4) - 8 * RW; } // largest (taps - 1) * |tstep|: 128 | 64
constexpr int LDH = KC3;      // row stride in halves: 64-byte rows, NO padding; conflicts are removed by an XOR swizzle instead
// ds_read_b128 is served in four 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... MI355X_MICROARCH.md, LDS) over 64
// banks of 4 B: with a fragment read of 16 rows x 4 columns of 16 B, an 80-byte padded stride puts two lanes of EVERY group on
// one bank quad (2-way).  Unpadded 64-byte rows with the 16-byte column index XORed by 2 * bit 2 of the row are conflict-free
// for every group and every starting row (the taps shift it), and the image is 20 % smaller.
__device__ inline int swz16(int row, int col16) { return col16 ^ (((row >> 2) & 1) << 1); }

__device__ inline void split_h(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Round 6, "packed" activations (ConvArgs::alpha_out / in_packed, conv.h): the producer of a transposed conv's input applies that
// conv's Snake and the fp16 hi / lo split ONCE per element in its epilogue and stores, where the fp32 values of an 8-channel group
// (t, 8 k .. 8 k + 7) stood, the 32 bytes [hi x 8 | lo x 8] - each 16-byte half is then one MFMA fragment half, which is what lets the
// consumer (convt_dma_kernel) fetch its whole activation window by LDS-DMA, with no register staging, no arithmetic and no ds_write.
// A lane holds one channel QUAD (4 g .. 4 g + 3 of a 16-channel block): it writes its four hi values into the hi half and its four lo
// values into the lo half of the group it shares with lane g ^ 1 (same row, same wave-instruction).  Same Snake, same split, same
// values as the consumer's own stash computed from the fp32 quad: the MFMA operands, and therefore the waveform, are bit-identical.
// In-place residual epilogues stay safe: the two lanes of a group overwrite each other's fp32 quad, but both belong to ONE wave
// instruction stream - their residual loads have both completed before either store issues.
// (al, inv) = the consumer's Snake alpha and 1 / (alpha + 1e-9) of the quad's channels: from an LDS table where the epilogue has one
// (four IEEE divisions per quad and row were half of what the packing cost the producers), else computed here - the same fp32
// division either way
__device__ inline void store_packed8(float* __restrict__ row, int co, float4 y, float4 al, float4 inv) {
    _Float16 h[4], l[4];
    split_h(snake_f(y.x, al.x, inv.x), h[0], l[0]); split_h(snake_f(y.y, al.y, inv.y), h[1], l[1]);
    split_h(snake_f(y.z, al.z, inv.z), h[2], l[2]); split_h(snake_f(y.w, al.w, inv.w), h[3], l[3]);
    char* grp = reinterpret_cast<char*>(row + (co & ~7));           // 32 bytes of the 8-channel group
    const int half = (co >> 2) & 1;                                  // this quad: channels 0-3 or 4-7 of the group
    *reinterpret_cast<uint2*>(grp + 8 * half) = *reinterpret_cast<const uint2*>(h);
    *reinterpret_cast<uint2*>(grp + 16 + 8 * half) = *reinterpret_cast<const uint2*>(l);
}
__device__ inline void store_packed8(float* __restrict__ row, int co, float4 y, const float* __restrict__ alpha4) {
    const float4 al = *reinterpret_cast<const float4*>(alpha4);
    store_packed8(row, co, y, al, make_float4(1.0f / (al.x + 1e-9f), 1.0f / (al.y + 1e-9f), 1.0f / (al.z + 1e-9f), 1.0f / (al.w + 1e-9f)));
}

// RW = time rows per wave (32 | 64): with 64 a wave reads 8 activation + 2 NT weight fragments per tap for 12 NT MFMAs instead
// of 4 + 2 NT for 6 NT (the MFMA loop runs the LDS pipe at ~87 % of the matrix pipe's time with RW = 32), and a staged
// chunk feeds twice the rows
template <int TN, int RW>
__global__ __launch_bounds__(512) void conv_kernel_h3(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = TN / 16, NI = RW / 16, TMW = 8 * RW;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n0 = blockIdx.x * TN;
    const int u0 = blockIdx.y * TMW;
    const int b = blockIdx.z;
    const int lo_r = a.tstep >= 0 ? a.toff : a.toff + (a.taps - 1) * a.tstep;
    const int hi_r = a.tstep >= 0 ? a.toff + (a.taps - 1) * a.tstep : a.toff;
    const int win = TMW + (hi_r - lo_r);
    _Float16* xh = reinterpret_cast<_Float16*>(smem);            // [win][LDH] hi
    _Float16* xl = xh + (size_t)win * LDH;                       // [win][LDH] lo
    _Float16* wh = xl + (size_t)win * LDH;                       // [taps][TN][LDH] hi
    _Float16* wl = wh + (size_t)a.taps * TN * LDH;               // [taps][TN][LDH] lo

    f32x4 acc[NI][NT];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0, 0, 0, 0};

    const float* inb = a.in + (size_t)b * (a.in_bs ? a.in_bs : (size_t)a.Tin * a.Cin);
    const _Float16* gwh = reinterpret_cast<const _Float16*>(a.w_hi);
    const _Float16* gwl = reinterpret_cast<const _Float16*>(a.w_lo);

    // Software pipeline over the 32-channel chunks: the NEXT chunk's input window and tap weights are fetched
    // into registers before this chunk's MFMA loop and written to LDS (Snake + fp16 split on the way) after it,
    // so global/L2 latency hides under the matrix work instead of sitting between two barriers (the kernel
    // holds one workgroup per CU: nothing else would cover it).  Measured on the 384-channel 7-tap layers:
    // MFMA time was ~1/3 of the single-buffered kernel.
    constexpr int XA = h3_xa(RW);                                     // window float4 per thread (win <= TMW + h3_max_span(RW))
    constexpr int WR = (MAX_TAPS * TN * (KC3 / 8) + 511) / 512;       // weight uint4 pairs per thread
    const int nx = win * (KC3 / 4), nw = a.taps * TN * (KC3 / 8);
    float4 xr[XA];
    uint4 wrh[WR], wrl[WR];
    float4 alr = make_float4(0, 0, 0, 0);
    auto fetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < XA; ++k) {
            const int i = tid + 512 * k;
            xr[k] = make_float4(0, 0, 0, 0);
            if (i < nx) {
                const int r = i / (KC3 / 4), q = i % (KC3 / 4);
                const int t = u0 + lo_r + r;
                if (t >= 0 && t < a.Tin) xr[k] = *reinterpret_cast<const float4*>(inb + (size_t)t * a.Cin + c0 + 4 * q);
            }
        }
        if (a.alpha) alr = *reinterpret_cast<const float4*>(a.alpha + c0 + 4 * (tid % (KC3 / 4)));   // q is the same for every k
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            const int i = tid + 512 * k;
            wrh[k] = make_uint4(0, 0, 0, 0);
            wrl[k] = make_uint4(0, 0, 0, 0);
            if (i < nw) {
                const int q = i % (KC3 / 8), n = (i / (KC3 / 8)) % TN, tap = i / ((KC3 / 8) * TN);
                if (n0 + n < a.N) {
                    const size_t g = ((size_t)tap * a.N + n0 + n) * a.Cin + c0 + 8 * q;
                    wrh[k] = *reinterpret_cast<const uint4*>(gwh + g);
                    wrl[k] = *reinterpret_cast<const uint4*>(gwl + g);
                }
            }
        }
    };
    auto stash = [&]() {
        const bool snake = a.alpha != nullptr;
        const float4 inv = make_float4(1.0f / (alr.x + 1e-9f), 1.0f / (alr.y + 1e-9f), 1.0f / (alr.z + 1e-9f), 1.0f / (alr.w + 1e-9f));
#pragma unroll
        for (int k = 0; k < XA; ++k) {
            const int i = tid + 512 * k;
            if (i < nx) {
                const int r = i / (KC3 / 4), q = i % (KC3 / 4);
                float4 v = xr[k];
                if (snake) {   // Snake(0) = 0, so the zero padding needs no special case
                    v.x = snake_f(v.x, alr.x, inv.x); v.y = snake_f(v.y, alr.y, inv.y);
                    v.z = snake_f(v.z, alr.z, inv.z); v.w = snake_f(v.w, alr.w, inv.w);
                }
                _Float16 h[4], l[4];
                split_h(v.x, h[0], l[0]); split_h(v.y, h[1], l[1]); split_h(v.z, h[2], l[2]); split_h(v.w, h[3], l[3]);
                const int xo = r * LDH + 8 * swz16(r, q >> 1) + 4 * (q & 1);
                *reinterpret_cast<uint2*>(xh + xo) = *reinterpret_cast<uint2*>(h);
                *reinterpret_cast<uint2*>(xl + xo) = *reinterpret_cast<uint2*>(l);
            }
        }
#pragma unroll
        for (int k = 0; k < WR; ++k) {
            const int i = tid + 512 * k;
            if (i < nw) {
                const int q = i % (KC3 / 8), n = (i / (KC3 / 8)) % TN, tap = i / ((KC3 / 8) * TN);
                const int wo = (tap * TN + n) * LDH + 8 * swz16(n, q);      // tap * TN is a multiple of 8: the row's bit 2 is n's
                *reinterpret_cast<uint4*>(wh + wo) = wrh[k];
                *reinterpret_cast<uint4*>(wl + wo) = wrl[k];
            }
        }
    };

#ifdef PTTS_CONV_STAMPS
    unsigned long long ts = __builtin_amdgcn_s_memtime();
    const unsigned long long ts_begin = ts;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};
#endif
    fetch(0);
    stash();
    for (int c0 = 0; c0 < a.Cin; c0 += KC3) {
        __syncthreads();                                  // chunk c0 is in LDS
        STAMP(4, ts);
        const bool more = c0 + KC3 < a.Cin;
        if (more) fetch(c0 + KC3);
        asm volatile("" ::: "memory");                    // keep the fetch above the MFMA loop (hipcc sinks loads to their use)
        STAMP(0, ts);
        for (int tap = 0; tap < a.taps; ++tap) {
            const int roff = tap * a.tstep + a.toff - lo_r;
            const int xrow = w * RW + roff + (lane & 15);
            const size_t xo = (size_t)xrow * LDH + 8 * swz16(xrow, lane >> 4);     // row + 16 has the same bit 2
            half8 xfh[NI], xfl[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                xfh[i] = *reinterpret_cast<const half8*>(xh + xo + (size_t)i * 16 * LDH);
                xfl[i] = *reinterpret_cast<const half8*>(xl + xo + (size_t)i * 16 * LDH);
            }
            const size_t wo = ((size_t)tap * TN + (lane & 15)) * LDH + 8 * swz16(lane & 15, lane >> 4);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const half8 bh = *reinterpret_cast<const half8*>(wh + wo + (size_t)j * 16 * LDH);
                const half8 bl = *reinterpret_cast<const half8*>(wl + wo + (size_t)j * 16 * LDH);
                // small terms first, then the leading term
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, xfh[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, xfl[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, xfh[i], acc[i][j], 0, 0, 0);
                }
            }
        }
        STAMP(1, ts);
        __syncthreads();                                  // every wave is done reading chunk c0
        STAMP(2, ts);
        if (more) stash();
        STAMP(3, ts);
#ifdef PTTS_CONV_STAMPS
        tacc[5] += 1;
#endif
    }
#ifdef PTTS_CONV_STAMPS
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_stamps[i], tacc[i]);
        atomicAdd(&g_conv_stamps[6], __builtin_amdgcn_s_memtime() - ts_begin);
    }
#endif

    // ---- epilogue (same as the fp32 kernel) ----
    const size_t obs = a.out_bs ? a.out_bs : (size_t)a.Tout * a.Cout;
    float* outb = a.out + (size_t)b * obs;
    const float* resb = a.res ? a.res + (size_t)b * obs : nullptr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = u0 + w * RW + i * 16 + (lane & 15);
        if (u >= a.U) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 16 + 4 * (lane >> 4);
            if (n >= a.N) continue;
            int t, co;
            if (a.stride == 1) { t = u; co = n; }
            else { const int r = n / a.Cout; co = n - r * a.Cout; t = a.stride * u - a.pad_t + r; }
            if (t < 0 || t >= a.Tout) continue;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = acc[i][j][e] + (co + e < a.Cout ? a.bias[co + e] : 0.f);
            float* o = outb + (size_t)t * a.Cout + co;
            if (co + 3 < a.Cout) {
                if (resb) {
                    const float4 rv = *reinterpret_cast<const float4*>(resb + (size_t)t * a.Cout + co);
                    y[0] += rv.x; y[1] += rv.y; y[2] += rv.z; y[3] += rv.w;
                }
                if (a.act_out == 1) { y[0] = tanhf(y[0]); y[1] = tanhf(y[1]); y[2] = tanhf(y[2]); y[3] = tanhf(y[3]); }
                *reinterpret_cast<float4*>(o) = make_float4(y[0], y[1], y[2], y[3]);
            } else {
                for (int e = 0; e < 4 && co + e < a.Cout; ++e) {
                    float v = y[e];
                    if (resb) v += resb[(size_t)t * a.Cout + co + e];
                    if (a.act_out == 1) v = tanhf(v);
                    o[e] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_kernel_h3r: the same arithmetic as conv_kernel_h3 (same products, same order: bit-identical results) with the weights
// streamed by LDS-DMA instead of through registers.
//
// What bounded conv_kernel_h3 (in-kernel s_memtime stamps of a 768-channel 7-tap layer, cycles per 32-channel chunk of wave 0:
// issuing the 18 staging loads 2.8 k, MFMA loop 7.3 k + 3.5 k at the barrier behind it for the slower waves, Snake / split /
// ds_write 2.1 k, barrier 1.5 k; the matrix pipe has 8.1 k of work in those 17.2 k): (i) the 86 KB of tap weights go through 48
// staging registers per lane, which is what keeps a wave at 32 rows: with 64 rows a wave reads 8 + 12 fragments per tap for 72
// MFMAs instead of 4 + 12 for 36 (the MFMA loop holds the LDS pipe at 87 % of the matrix pipe's time at 32 rows), and a staged
// window feeds twice the rows; (ii) all the loads of a chunk are issued in one burst the texture addresser takes 2.3 k cycles
// to accept.  Here the weights of ONE tap of one chunk are a 12 KB unit (hi + lo image of TN x 32 halves, stored in exactly
// the swizzled LDS order in global memory, so the copy is linear: two 12-byte global_load_lds per lane at TN = 96), units
// travel through a ring of RS slots, D = RS - 1 units ahead of the MFMAs, and the activation loads of the next chunk are
// spread over the taps.  Per unit: wait for the own DMA pieces of unit u (counted vmcnt: younger pieces stay in flight) ->
// barrier (everyone's pieces of u are in LDS, everyone is done reading u - 1) -> issue the DMA of unit u + D into the slot of
// u - 1 -> this tap's share of the activation loads -> fragments + MFMAs.  The activation window keeps its register staging
// (Snake and the fp16 split need the VALU anyway) with one barrier on each side of the stash, as before.
//
// vmcnt bookkeeping (loads, LDS-DMA count together, in issue order): every step issues the same instructions on every wave -
// the DMA of a unit past the end re-fetches the last unit into a free slot, activation loads are never branched around (out of
// range lanes read row 0 and are zeroed in the stash) - so the number of operations younger than unit u's pieces at the wait is
// a compile-time function of the tap (tables n_steady / n_last below; the first D units wait for everything but the pieces).
// ------------------------------------------------------------------------------------------------
template <int T, int TEND, class F> __device__ inline void static_for(F&& f) {
    if constexpr (T < TEND) { f(std::integral_constant<int, T>{}); static_for<T + 1, TEND>(f); }
}
template <int N> __device__ inline void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ inline void lds_barrier() {      // own LDS writes done, then the workgroup barrier; does NOT drain vmcnt (pending LDS-DMA)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int TN, int RW, int TAPS, int RS, int WC = 1, int NW = 8>
struct H3R {
    // WC = 2 (round 6): the eight waves tile the workgroup's TMW x TN outputs as 4 row groups x 2 column halves instead of 8 x 1
    // NW = 4 (round 6): four-wave workgroups, two of them on a CU (fused unit at C = 96)
    static constexpr int NT = TN / 16 / WC, NI = RW / 16, TMW = (NW / WC) * RW, D = RS - 1;
    static constexpr int UB = TN * 128;                    // bytes of one weight unit
    static constexpr int GS = 16;                          // bytes per lane and LDS-DMA instruction (12-byte pieces do not land lane x 12 apart)
    static constexpr int PI = UB / (64 * GS);              // 1 KB wave-instructions a unit needs: 8 | 12 | 16
    static constexpr int G = (PI + NW - 1) / NW;           // issued per wave and unit - the SAME on every wave (the vmcnt tables need that,
                                                           // and a branch around a DMA makes hipcc drain vmcnt at the next load): at TN = 96
                                                           // waves 4 .. 7 have no second piece and copy their first one again
    static constexpr int XA = NW == 8 ? h3_xa(TMW / 8) : (TMW + 64) * (KC3 / 4) / (NW * 64);      // NW = 4: span <= 64
    static constexpr int xs(int t) {                       // activation loads issued at tap t (+ the alpha row at tap 0)
        int n = t == 0 ? 1 : 0;
        for (int k = 0; k < XA; ++k) n += (k % TAPS == t) ? 1 : 0;
        return n;
    }
    static constexpr int xrel(int j, bool more) {          // step j relative to this chunk's tap 0; j < 0: an earlier chunk
        return j >= 0 ? (more ? xs(j % TAPS) : 0) : xs(((j % TAPS) + TAPS) % TAPS);
    }
    static constexpr int younger(int t, bool more) {       // operations issued after unit u's pieces, before its wait (u >= D)
        int n = xrel(t - D, more);
        for (int j = t - D + 1; j < t; ++j) n += G + xrel(j, more);
        return n;
    }
    static constexpr int n_min = G * (D - 1);              // only the DMA pieces of the units in between: always a lower bound
};

// FUSE (a residual unit whose channels fit one output block, C = TN = 96: HF modeling_dac.py:175-210 x + conv1(snake(conv7(snake(x))))):
// the 7-tap conv's tile never leaves the registers.  The accumulators (lane = (channel quad g, row t): channels 16 j + 4 g .. + 3)
// get the bias, the 1x1 conv's Snake and the fp16 split and ARE the B operand of the second MFMA stage: a k-step of 32 channels
// takes the lane's quads of blocks 2 m and 2 m + 1, i.e. k position 8 g + e holds channel 32 m + 4 g + e (e < 4) or
// 32 m + 16 + 4 g + e - 4 - a permutation of the step's channels, so the 1x1 weights are stored as A fragments in that k order
// (conv_units2_kernel) and nothing moves between lanes.  The fragments arrive as KS more units through the same ring (instead
// of the clamped tail units).  h is neither written (8 GB per layer at C = 96) nor read back, and one launch goes away.
// The sum over a k-step runs in another channel order than in the separate 1x1 kernel: equal to ~1e-7 relative, not bit for bit.
//
// WC = 2 (round 6, the wide 7-tap convs): a wave computes 64 rows x 96 outputs instead of 32 x 192 - the same workgroup tile, ring
// units, window and MFMA count, the same products in the same order per output (bit-identical), but 8 activation + 12 weight
// fragment reads per tap and wave instead of 4 + 24: the CU's LDS pipe, 78 % as busy as the matrix pipe at 28 reads per 72 MFMAs,
// gets 29 % less to do (profiles/r06_dac_pipelined_stash.txt: that stream, not the stash or the barriers, is what holds the family).
// Not for FUSE: the 1x1 stage needs all TN channels of a row in one wave.
template <int TN, int RW, int TAPS, int RS, bool FUSE, int WC = 1, int NW = 8>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_kernel_h3r(ConvArgs a) {
    using K = H3R<TN, RW, TAPS, RS, WC, NW>;
    constexpr int NTH = 64 * NW, RPP = NTH / (KC3 / 4);       // threads; window rows one staging piece covers
    static_assert(NW == 8 || (NW == 4 && WC == 1 && K::PI % NW == 0), "four-wave workgroups: every wave copies PI / 4 pieces of a unit");
    static_assert(!FUSE || (TN % 32 == 0 && RW % 32 == 0), "fused 1x1 stage: k-steps of 32 channels, row tiles in pairs");
    static_assert(WC == 1 || (WC == 2 && !FUSE && (TN / 16) % 2 == 0), "column halves: plain conv instances only");
    constexpr int NT = K::NT, NI = K::NI, TMW = K::TMW, D = K::D, UB = K::UB, GS = K::GS, XA = K::XA;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    LIFE(life0);
    const int wr = WC == 1 ? w : w % (NW / WC), wc = WC == 1 ? 0 : w / (NW / WC);      // the wave's row group / column part
    const int n0 = blockIdx.x * TN;
    const int u0 = blockIdx.y * TMW;
    const int b = blockIdx.z;
    const int lo_r = a.tstep >= 0 ? a.toff : a.toff + (TAPS - 1) * a.tstep;
    const int hi_r = a.tstep >= 0 ? a.toff + (TAPS - 1) * a.tstep : a.toff;
    const int win = TMW + (hi_r - lo_r);
    char* ring = smem;                                           // [RS][hi: TN x 64 B | lo: TN x 64 B]; first: LDS-DMA bases stay below 64 KB
    _Float16* xh = reinterpret_cast<_Float16*>(smem + RS * UB);  // [win][LDH] hi
    _Float16* xl = xh + (size_t)win * LDH;                       // [win][LDH] lo

    f32x4 acc[NI][NT];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0, 0, 0, 0};

    const float* inb = a.in + (size_t)b * (a.in_bs ? a.in_bs : (size_t)a.Tin * a.Cin);
    const int nchunks = a.Cin / KC3, nunits = nchunks * TAPS;
    const char* units = reinterpret_cast<const char*>(a.w_units) + (size_t)blockIdx.x * nunits * UB;   // [chunk][tap] of this block
    const bool snake = a.alpha != nullptr;
    const float* alpha_src = snake ? a.alpha + 4 * (tid % (KC3 / 4)) : reinterpret_cast<const float*>(a.w_units);

    // activation staging: the lane's piece k sits 64 rows below piece k - 1; lanes outside the window / the utterance read row 0
    const int nx = win * (KC3 / 4);
    float4 xr[XA];
    float4 alr = make_float4(0, 0, 0, 0);
    const int xq = tid % (KC3 / 4), xrow0 = tid / (KC3 / 4);
    const unsigned xbase = (unsigned)(u0 + lo_r + xrow0) * (unsigned)a.Cin + 4 * xq;     // wraps for rows < 0: masked by xvalid
    const unsigned xstep = (unsigned)RPP * (unsigned)a.Cin;
    unsigned xvalid = 0;
#pragma unroll
    for (int k = 0; k < XA; ++k) {
        const int t = u0 + lo_r + xrow0 + RPP * k;
        xvalid |= (tid + NTH * k < nx && t >= 0 && t < a.Tin) ? 1u << k : 0u;
    }
    auto xload = [&](int k, int c0) {
        const unsigned off = (xvalid >> k) & 1 ? xbase + k * xstep : 0u;
        xr[k] = *reinterpret_cast<const float4*>(inb + off + c0);
    };
    auto aload = [&](int c0) { alr = *reinterpret_cast<const float4*>(alpha_src + (snake ? c0 : 0)); };
    auto stash = [&]() {
        const float4 inv = make_float4(1.0f / (alr.x + 1e-9f), 1.0f / (alr.y + 1e-9f), 1.0f / (alr.z + 1e-9f), 1.0f / (alr.w + 1e-9f));
#pragma unroll
        for (int k = 0; k < XA; ++k) {
            if (tid + NTH * k < nx) {
                const int r = xrow0 + RPP * k;
                float4 v = (xvalid >> k) & 1 ? xr[k] : make_float4(0, 0, 0, 0);
#if PTTS_CONV_ABLATE != 3       // 3 = timing-only ablation (WRONG results): the stash without Snake
                if (snake) {   // Snake(0) = 0, so the zero padding needs no special case
                    v.x = snake_f(v.x, alr.x, inv.x); v.y = snake_f(v.y, alr.y, inv.y);
                    v.z = snake_f(v.z, alr.z, inv.z); v.w = snake_f(v.w, alr.w, inv.w);
                }
#endif
                _Float16 h[4], l[4];
                split_h(v.x, h[0], l[0]); split_h(v.y, h[1], l[1]); split_h(v.z, h[2], l[2]); split_h(v.w, h[3], l[3]);
                const int xo = r * LDH + 8 * swz16(r, xq >> 1) + 4 * (xq & 1);
                *reinterpret_cast<uint2*>(xh + xo) = *reinterpret_cast<uint2*>(h);
                *reinterpret_cast<uint2*>(xl + xo) = *reinterpret_cast<uint2*>(l);
            }
        }
    };
    // LDS-DMA of unit u into ring slot `slot`: wave-instruction m copies pieces 64 m .. 64 m + 63 (16 B each) of the linear image;
    // wave w issues m = w, w + 8, .. (m = w again where the unit has no such piece)
    auto dma = [&](int u, int slot) {
        const int uc = u < nunits ? u : nunits - 1;
        const char* src = units + (size_t)uc * UB + (size_t)tid * GS;
        if constexpr (FUSE)                               // units nunits .. nunits + KS - 1: the A fragments of the fused 1x1 stage
            if (u >= nunits) src = reinterpret_cast<const char*>(a.w2_units) + (size_t)(u - nunits < TN / 32 ? u - nunits : 0) * UB + (size_t)tid * GS;
        char* dst = ring + slot * UB + w * (64 * GS);
#if defined(PTTS_RING_DEBUG_COPY)         // diagnostic build: the same pieces moved through registers (tells a layout bug from a DMA one)
#pragma unroll
        for (int j = 0; j < K::G; ++j) {
            const int o = w + NW * j < K::PI ? j * NTH * GS : 0;
            *reinterpret_cast<uint4*>(dst + o + lane * GS) = *reinterpret_cast<const uint4*>(src + o);
        }
#elif defined(__HIP_DEVICE_COMPILE__)      // the builtin does not exist in the host pass, which would then drop the kernel's launch stub
        typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
        for (int j = 0; j < K::G; ++j) {
            const int o = w + NW * j < K::PI ? j * NTH * GS : 0;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src + o), (lds_void*)(dst + o), 16, 0, 0);
        }
#else
        (void)src; (void)dst;
#endif
    };

#pragma unroll
    for (int k = 0; k < XA; ++k) xload(k, 0);
    aload(0);
    stash();
#pragma unroll
    for (int v = 0; v < D; ++v) dma(v, v);

#ifdef PTTS_CONV_STAMPS
    unsigned long long ts = __builtin_amdgcn_s_memtime();
    const unsigned long long ts_begin = ts;
    unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0};       // [0] DMA wait [1] tap barrier [2] issue [3] MFMA [4] chunk barrier [5] stash
    unsigned long long nchunk_done = 0;
#endif
    LIFE(life1);
    int slot = 0;                                         // ring slot of the unit about to be consumed
    // one chunk: MORE = the next chunk exists (its activation loads ride on the taps); the last chunk is peeled so that the loop
    // body has no conditional loads (hipcc would otherwise merge "loads may be pending" into the loop head and drain vmcnt)
    auto chunk = [&](int c, auto more_c) {
        constexpr bool more = decltype(more_c)::value;
        const int cn = (c + 1) * KC3;
        static_for<0, TAPS>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            // ---- unit u = c * TAPS + t: its pieces have landed (own wait, then everyone's through the barrier) ----
            if (c * TAPS + t < D) wait_vmcnt<K::n_min>();
            else wait_vmcnt<K::younger(t, more)>();
            STAMP(0, ts);
#if PTTS_CONV_ABLATE == 5       // timing-only ablation (WRONG results, may read half-landed units): the tap without its workgroup barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            lds_barrier();                                // at t = 0 also: the stashed window of chunk c is visible
#endif
            STAMP(1, ts);
            {
                int ps = slot + D; if (ps >= RS) ps -= RS;
                dma(c * TAPS + t + D, ps);                // into the slot unit u - 1 was read from
            }
            asm volatile("" ::: "memory");                // the tables assume the DMA pieces are issued BEFORE this tap's activation loads
            if constexpr (more) {
#pragma unroll
                for (int k = 0; k < XA; ++k) if (k % TAPS == t) xload(k, cn);
                if (t == 0) aload(cn);
            }
            asm volatile("" ::: "memory");                // keep the loads above the MFMAs (hipcc sinks loads to their use)
            STAMP(2, ts);
            const _Float16* wh = reinterpret_cast<const _Float16*>(ring + slot * UB);
            const _Float16* wl = wh + TN * LDH;
            const int roff = t * a.tstep + a.toff - lo_r;
            const int xrow = wr * RW + roff + (lane & 15);
            const size_t xo = (size_t)xrow * LDH + 8 * swz16(xrow, lane >> 4);     // row + 16 has the same bit 2
            half8 xfh[NI], xfl[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                xfh[i] = *reinterpret_cast<const half8*>(xh + xo + (size_t)i * 16 * LDH);
                xfl[i] = *reinterpret_cast<const half8*>(xl + xo + (size_t)i * 16 * LDH);
            }
            const size_t wo = (size_t)(wc * NT * 16 + (lane & 15)) * LDH + 8 * swz16(lane & 15, lane >> 4);   // wc * NT * 16 is a multiple of 8: the row's bit 2 is the lane's
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#if PTTS_CONV_ABLATE == 1       // timing-only ablation (WRONG results): one weight fragment pair per tap instead of NT
                const half8 bh = *reinterpret_cast<const half8*>(wh + wo), bl = *reinterpret_cast<const half8*>(wl + wo);
#else
                const half8 bh = *reinterpret_cast<const half8*>(wh + wo + (size_t)j * 16 * LDH);
                const half8 bl = *reinterpret_cast<const half8*>(wl + wo + (size_t)j * 16 * LDH);
#endif
#pragma unroll
                for (int i = 0; i < NI; ++i) {            // small terms first, then the leading term
#if PTTS_CONV_ABLATE == 2       // timing-only ablation (WRONG results): the fragment reads stay, one MFMA instead of three
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh + bl, xfh[i] + xfl[i], acc[i][j], 0, 0, 0);
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, xfh[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, xfl[i], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, xfh[i], acc[i][j], 0, 0, 0);
#endif
                }
            }
            if (++slot == RS) slot = 0;
            STAMP(3, ts);
        });
        lds_barrier();                                    // every wave is done reading the window of chunk c
        STAMP(4, ts);
        if constexpr (more) stash();
        STAMP(5, ts);
#ifdef PTTS_CONV_STAMPS
        nchunk_done += 1;
#endif
    };
    for (int c = 0; c + 1 < nchunks; ++c) chunk(c, std::true_type{});
    chunk(nchunks - 1, std::false_type{});
    LIFE(life2);
#ifdef PTTS_CONV_LIFE
    auto life_end = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left: what a successor workgroup on this CU waits for at least
        const unsigned long long life3 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            atomicAdd(&g_conv_life[0], life1 - life0); atomicAdd(&g_conv_life[1], life2 - life1);
            atomicAdd(&g_conv_life[2], life3 - life2); atomicAdd(&g_conv_life[3], 1ull);
        }
    };
#endif
#ifdef PTTS_CONV_STAMPS
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_stamps[8 + i], tacc[i]);
        atomicAdd(&g_conv_stamps[5], nchunk_done);
    }
    if (lane == 0) {      // per wave: waves 0-3 and 4-7 share the SIMDs pairwise and do NOT spend their time alike (profiles/r04_dac_per_wave_stamps.txt)
        for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_wave[w][i], tacc[i]);
    }
    if (tid == 0) {
        atomicAdd(&g_conv_stamps[6], __builtin_amdgcn_s_memtime() - ts_begin);
    }
#endif
    if constexpr (!FUSE) wait_vmcnt<0>();                 // the D clamped tail units: nothing may be in flight when the wave ends

    if constexpr (FUSE) {
        constexpr int KS = TN / 32;
        // the 1x1 conv's Snake table over the dead activation window
        lds_barrier();                                    // everyone is out of the last chunk: the window is dead
        float* als = reinterpret_cast<float*>(xh);        // [TN] alpha, [TN] 1 / (alpha + 1e-9)
        float* ivs = als + TN;
        float* als_o = ivs + TN;                          // [TN] / [TN]: the same pair for the CONSUMER's Snake when the output goes out packed
        float* ivs_o = als_o + TN;
        for (int i = tid; i < TN; i += NTH) {
            const float al = a.alpha2[i]; als[i] = al; ivs[i] = 1.0f / (al + 1e-9f);
            if (a.alpha_out) { const float ao = a.alpha_out[i]; als_o[i] = ao; ivs_o[i] = 1.0f / (ao + 1e-9f); }
        }
        const int g = lane >> 4;
        const size_t obs2 = (size_t)a.Tout * a.Cout;
        float* outb2 = a.out + (size_t)b * obs2;
        // B fragments of k-step m for row tile i: the lane's quads of blocks 2 m and 2 m + 1 (+ bias, Snake, split)
        auto bfrag = [&](int i, int m, half8& bh, half8& bl) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                const int j = 2 * m + hb, ch = 16 * j + 4 * g;
                const float4 b1 = *reinterpret_cast<const float4*>(a.bias + ch);
                const f32x4 al = *reinterpret_cast<const f32x4*>(als + ch), iv = *reinterpret_cast<const f32x4*>(ivs + ch);
                const float bq[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hval = acc[i][j][r] + bq[r];                 // h as the 7-tap conv would have stored it
                    _Float16 h, l;
                    split_h(snake_f(hval, al[r], iv[r]), h, l);
                    bh[4 * hb + r] = h;
                    bl[4 * hb + r] = l;
                }
            }
        };
        auto store_tile = [&](int i, const f32x4 (&o)[NT]) {
            const int u = u0 + w * RW + i * 16 + (lane & 15);
            if (u >= a.U) return;
#pragma unroll
            for (int jb = 0; jb < NT; ++jb) {
                const int co = 16 * jb + 4 * g;
                const float4 b2 = *reinterpret_cast<const float4*>(a.bias2 + co);
                const float4 rv = *reinterpret_cast<const float4*>(inb + (size_t)u * a.Cin + co);      // the unit's input x
                float4 y;
                y.x = (o[jb][0] + b2.x) + rv.x; y.y = (o[jb][1] + b2.y) + rv.y;
                y.z = (o[jb][2] + b2.z) + rv.z; y.w = (o[jb][3] + b2.w) + rv.w;
                if (a.alpha_out)      // the next layer is a transposed conv (conv.h)
                    store_packed8(outb2 + (size_t)u * a.Cout, co, y, *reinterpret_cast<const float4*>(als_o + co), *reinterpret_cast<const float4*>(ivs_o + co));
                else *reinterpret_cast<float4*>(outb2 + (size_t)u * a.Cout + co) = y;
            }
        };
        if constexpr (KS <= D) {
            // every stage-2 unit is already in the ring (the tail prefetch fetched them): two row tiles at a time, so that an
            // A fragment read feeds 6 MFMAs and the second accumulator set stays at 2 NT registers quads
            wait_vmcnt<0>();
            lds_barrier();                                // everyone's DMA pieces and the Snake table are in LDS
            int s2[KS];
#pragma unroll
            for (int m = 0; m < KS; ++m) { int q = slot + m; if (q >= RS) q -= RS; s2[m] = q; }
#pragma unroll
            for (int ip = 0; ip < NI; ip += 2) {
                half8 bh[2][KS], bl[2][KS];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int m = 0; m < KS; ++m) bfrag(ip + ii, m, bh[ii][m], bl[ii][m]);
                f32x4 acc2[2][NT];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int jb = 0; jb < NT; ++jb) acc2[ii][jb] = {0, 0, 0, 0};
#pragma unroll
                for (int m = 0; m < KS; ++m) {
                    const half8* fh = reinterpret_cast<const half8*>(ring + s2[m] * UB);   // [jb][lane] hi, then [jb][lane] lo
                    const half8* fl = fh + NT * 64;
#pragma unroll
                    for (int jb = 0; jb < NT; ++jb) {
                        const half8 ah = fh[jb * 64 + lane], alo = fl[jb * 64 + lane];
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {  // same product order as the separate kernels: lo.hi, hi.lo, hi.hi
                            acc2[ii][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh[ii][m], acc2[ii][jb], 0, 0, 0);
                            acc2[ii][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ii][m], acc2[ii][jb], 0, 0, 0);
                            acc2[ii][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ii][m], acc2[ii][jb], 0, 0, 0);
                        }
                    }
                }
                store_tile(ip, acc2[0]);
                store_tile(ip + 1, acc2[1]);
            }
        } else {
            // more stage-2 units than the ring holds: they keep streaming, one k-step per ring step, every row tile accumulates
            static_assert(NI == 2 || NW == 4, "streamed stage 2 keeps a second accumulator set for every row tile");
            f32x4 acc2[NI][NT];
#pragma unroll
            for (int ii = 0; ii < NI; ++ii)
#pragma unroll
                for (int jb = 0; jb < NT; ++jb) acc2[ii][jb] = {0, 0, 0, 0};
            static_for<0, KS>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                wait_vmcnt<K::n_min>();                   // only DMA pieces are in flight: the D - 1 younger units stay
                lds_barrier();                            // (m = 0: also publishes the Snake table)
                {
                    int ps = slot + D; if (ps >= RS) ps -= RS;
                    dma(nunits + m + D, ps);
                }
                asm volatile("" ::: "memory");
                half8 bh[NI], bl[NI];
#pragma unroll
                for (int ii = 0; ii < NI; ++ii) bfrag(ii, m, bh[ii], bl[ii]);
                const half8* fh = reinterpret_cast<const half8*>(ring + slot * UB);
                const half8* fl = fh + NT * 64;
#pragma unroll
                for (int jb = 0; jb < NT; ++jb) {
                    const half8 ah = fh[jb * 64 + lane], alo = fl[jb * 64 + lane];
#pragma unroll
                    for (int ii = 0; ii < NI; ++ii) {
                        acc2[ii][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh[ii], acc2[ii][jb], 0, 0, 0);
                        acc2[ii][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ii], acc2[ii][jb], 0, 0, 0);
                        acc2[ii][jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ii], acc2[ii][jb], 0, 0, 0);
                    }
                }
                if (++slot == RS) slot = 0;
            });
            wait_vmcnt<0>();
#pragma unroll
            for (int ii = 0; ii < NI; ++ii) store_tile(ii, acc2[ii]);
        }
#ifdef PTTS_CONV_LIFE
        life_end();
#endif
        return;
    }

    // ---- epilogue (same as conv_kernel_h3) ----
    const size_t obs = a.out_bs ? a.out_bs : (size_t)a.Tout * a.Cout;
    float* outb = a.out + (size_t)b * obs;
    const float* resb = a.res ? a.res + (size_t)b * obs : nullptr;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = u0 + wr * RW + i * 16 + (lane & 15);
        if (u >= a.U) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + (wc * NT + j) * 16 + 4 * (lane >> 4);
            int t, co;
            if (a.stride == 1) { t = u; co = n; }
            else { const int r = n / a.Cout; co = n - r * a.Cout; t = a.stride * u - a.pad_t + r; }
            if (t < 0 || t >= a.Tout) continue;
            const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);      // N % TN == 0 and Cout % 16 == 0: whole quads
            float4 y = make_float4(acc[i][j][0] + bs.x, acc[i][j][1] + bs.y, acc[i][j][2] + bs.z, acc[i][j][3] + bs.w);
            if (resb) {
                const float4 rv = *reinterpret_cast<const float4*>(resb + (size_t)t * a.Cout + co);
                y.x += rv.x; y.y += rv.y; y.z += rv.z; y.w += rv.w;
            }
            if (a.act_out == 1) { y.x = tanhf(y.x); y.y = tanhf(y.y); y.z = tanhf(y.z); y.w = tanhf(y.w); }
            if (a.alpha_out) store_packed8(outb + (size_t)t * a.Cout, co, y, a.alpha_out + co);          // the next layer is a transposed conv (conv.h)
            else *reinterpret_cast<float4*>(outb + (size_t)t * a.Cout + co) = y;
        }
    }
#ifdef PTTS_CONV_LIFE
    life_end();
#endif
}

// ------------------------------------------------------------------------------------------------
// convt_dma_kernel (round 6): the transposed convs (two taps: rows u and u - 1) on PACKED input (conv.h: alpha_out / in_packed).
//
// conv_kernel_h3r's two-tap instance spends a 32-channel chunk as ~4.6 k cycles of matrix-pipe work and ~5.7 k of staging: global
// loads into registers, Snake, fp16 split, ds_write, a barrier on each side - once per 192-wide output block (32 / 16 / 4 / 1 of them
// per layer), for two taps of MFMAs each time.  Removing the arithmetic alone bought 8 % (profiles/r06_dac_packed_handover.txt): the
// cost is the load -> ds_write -> barrier round trip.  Here the producer has already stored Snake(x) split into fp16 halves as
// [hi x 8 | lo x 8] per 8-channel group, i.e. every 16 bytes of the input ARE one fragment half, so the activation window travels
// exactly like the weights: `global_load_lds_dwordx4` straight into the swizzled LDS image (the per-lane SOURCE address carries the
// swizzle and the row clamp; rows outside the utterance read a zero line), one chunk ahead, into the other of two window buffers.
// No staging registers, no VALU, no ds_write, no stash barrier: per tap one counted wait, one barrier (the weight ring's), the DMAs of
// the units / window to come, fragments + MFMAs - the same products in the same order as the ring kernel (bit-identical, test).
// LDS: weight ring 3 x 24 KB + 2 windows x (hi + lo) x 272 rows x 64 B = 72 + 69.6 = 141.6 KB.
// vmcnt (everything is LDS-DMA, in issue order; per wave: U = 3 pieces of a weight unit, W = 5 pieces of a window):
//   prologue U[0] W[0] U[1];  step (c, tap 0): wait for U[2c], W[c] -> younger: U[2c+1] = 3;  then issue U[2c+2], W[c+1];
//   step (c, tap 1): wait for U[2c+1] -> younger: U[2c+2], W[c+1] = 8;  then issue U[2c+3].   Past the end every index is clamped
//   (a re-copy into a buffer nobody reads any more), so the counts are constants.
// ------------------------------------------------------------------------------------------------
__device__ const uint4 g_convt_zero_line[4] = {};      // what a window row outside [0, Tin) reads

__global__ __launch_bounds__(512) void convt_dma_kernel(ConvArgs a) {
    constexpr int TN = 192, NT = TN / 16, NI = 2, RW = 32, TMW = 8 * RW, RS = 3, UB = TN * 128, GU = UB / 8192;      // ring of 3: two units ahead
    constexpr int WROWS = 272, WPIECES = 2 * (WROWS / 16), GW = (WPIECES + 7) / 8;      // window: 17 x 1 KB per plane, 34 pieces, 5 per wave
    constexpr int WBYTES = 2 * WROWS * 64;                                             // one window buffer: hi plane, lo plane
    static_assert(GU == 3 && GW == 5, "the vmcnt constants below");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                    // [RS][hi: TN x 64 B | lo: TN x 64 B]
    char* win = smem + RS * UB;                           // [2][hi: WROWS x 64 B | lo: WROWS x 64 B]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * TN;
    const int u0 = blockIdx.y * TMW;
    const int b = blockIdx.z;
    const int nchunks = a.Cin / KC3, nunits = nchunks * 2;
    const char* units = reinterpret_cast<const char*>(a.w_units) + (size_t)blockIdx.x * nunits * UB;
    const char* inb = reinterpret_cast<const char*>(a.in + (size_t)b * a.Tin * a.Cin);

    auto dma_unit = [&](int u) {                          // unit u = (chunk u / 2, tap u % 2) into slot u % RS
        const int uc = u < nunits ? u : nunits - 1;
        const char* src = units + (size_t)uc * UB + (size_t)tid * 16;
        char* dst = ring + (u % RS) * UB + w * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
        for (int j = 0; j < GU; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src + j * 8192), (lds_void*)(dst + j * 8192), 16, 0, 0);
#else
        (void)src; (void)dst;
#endif
    };
    // window of chunk c: rows t = u0 - 1 + r, r = 0 .. 256 (tap 1 reads row u - 1), 32 channels = four 8-channel groups per row.
    // Piece m (16 rows x 64 B of one plane) -> lane l: row 16 (m mod 17) + l / 4, LDS column slot l % 4 = swz16(row, group).
    auto dma_window_piece = [&](int c, int j) {            // piece w + 8 j of window c (j < GW)
        const int cc = c < nchunks ? c : nchunks - 1;
        char* dstb = win + (c & 1) * WBYTES;
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
        int m = w + 8 * j;
        m = m < WPIECES ? m : WPIECES - 1;                // waves past the last piece copy it again (same bytes, same place)
        const int plane = m >= WROWS / 16 ? 1 : 0, mp = m - plane * (WROWS / 16);
        const int r = 16 * mp + (lane >> 2);
        const int grp = swz16(r, lane & 3);               // the XOR is its own inverse: slot -> channel group
        const int t = u0 - 1 + r;
        const char* src = (r <= TMW && t >= 0 && t < a.Tin)
                              ? inb + ((size_t)t * a.Cin + (size_t)cc * KC3 + 8 * grp) * 4 + 16 * plane
                              : reinterpret_cast<const char*>(g_convt_zero_line);
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src), (lds_void*)(dstb + plane * (WROWS * 64) + mp * 1024), 16, 0, 0);
#else
        (void)cc; (void)dstb; (void)inb; (void)g_convt_zero_line; (void)j;
#endif
    };
    auto dma_window = [&](int c) {
#pragma unroll
        for (int j = 0; j < GW; ++j) dma_window_piece(c, j);
    };

    f32x4 acc[NI][NT];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0, 0, 0, 0};

    // Fragment reads as inline asm with hand-counted waits.  hipcc waits `lgkmcnt(0)` in front of an MFMA whose operand came from LDS
    // however many younger reads are in flight (round 4 saw the same in conv_kernel_h3r: "still waits lgkmcnt(0) every second
    // pair"), so a compiler-visible prefetch never gets more than a few MFMAs ahead.  LDS reads return in order: with the weight
    // fragments read TWO output blocks ahead (three register pairs), block j's MFMAs wait until at most the four younger reads are
    // outstanding.  The wait statement takes the registers it validates as "+v" operands, so no consumer can be scheduled above it.
    // `spread` (tap 0): the five window pieces of chunk c + 1 go out one per two output blocks INSIDE the MFMA loop instead of in one
    // burst behind the barrier - eight waves x eight LDS-DMA instructions in a row hold every wave's first MFMA back behind the
    // texture addresser's queue
    half8 xfh[NI] = {}, xfl[NI] = {}, bh[3] = {}, bl[3] = {};      // fragment registers of multiply(): declared (and zeroed) once per kernel
    auto multiply = [&](int c, auto tc, int slot) {
        constexpr int t = decltype(tc)::value;
        typedef __attribute__((address_space(3))) const char lds_cchar;
        const int xrow = w * RW + (1 - t) + (lane & 15);                          // tap 0: row u (window row u - u0 + 1), tap 1: row u - 1
        const unsigned xbase = (unsigned)(uintptr_t)(lds_cchar*)(win + (c & 1) * WBYTES) +
                               (unsigned)(xrow * LDH + 8 * swz16(xrow, lane >> 4)) * 2u;       // row + 16 has the same bit 2
        const unsigned wbase = (unsigned)(uintptr_t)(lds_cchar*)(ring + slot * UB) +
                               (unsigned)((lane & 15) * LDH + 8 * swz16(lane & 15, lane >> 4)) * 2u;
        constexpr int XL = WROWS * 64, WL = TN * LDH * 2;                        // byte distance hi plane -> lo plane
        static_assert(NI == 2, "two interleaved accumulator chains");
        // "+v": the fragment registers stay LIVE between a block's MFMAs and the read that refills them, so hipcc cannot hand one of
        // them to an MFMA as a temporary destination in between - a matrix-pipe write still retiring into a register that an
        // inline-asm load (which the hazard recognizer does not model) is about to fill (tools/check_mfma_asm_hazard.py)
#define PTTS_LDS_READ(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(base), "n"(off))
        PTTS_LDS_READ(xfh[0], xbase, 0);
        PTTS_LDS_READ(xfl[0], xbase, XL);
        PTTS_LDS_READ(xfh[1], xbase, 1024);
        PTTS_LDS_READ(xfl[1], xbase, XL + 1024);
        PTTS_LDS_READ(bh[0], wbase, 0);
        PTTS_LDS_READ(bl[0], wbase, WL);
        PTTS_LDS_READ(bh[1], wbase, 1024);
        PTTS_LDS_READ(bl[1], wbase, WL + 1024);
        static_for<0, NT>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (j + 2 < NT) {
                PTTS_LDS_READ(bh[(j + 2) % 3], wbase, (j + 2) * 1024);
                PTTS_LDS_READ(bl[(j + 2) % 3], wbase, WL + (j + 2) * 1024);
            }
            if constexpr (t == 0 && (j & 1) == 1 && j / 2 < GW) dma_window_piece(c + 1, j / 2);
            constexpr int younger = 2 * ((NT - 1 - j) < 2 ? (NT - 1 - j) : 2);        // reads issued behind block j's pair
            if constexpr (j == 0)
                asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(xfh[0]), "+v"(xfl[0]), "+v"(xfh[1]), "+v"(xfl[1]), "+v"(bh[0]), "+v"(bl[0]) : "n"(younger));
            else
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(bh[j % 3]), "+v"(bl[j % 3]) : "n"(younger));
            // product order per accumulator: lo.hi, hi.lo, hi.hi (as conv_kernel_h3 / h3r); the two row tiles' chains interleaved
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j % 3], xfh[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j % 3], xfh[1], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j % 3], xfl[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j % 3], xfl[1], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j % 3], xfh[0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j % 3], xfh[1], acc[1][j], 0, 0, 0);
        });
#undef PTTS_LDS_READ
    };

    dma_unit(0);
    dma_window(0);
    dma_unit(1);
    int slot = 0;
    for (int c = 0; c < nchunks; ++c) {
        wait_vmcnt<GU>();                                 // own pieces of unit 2c and of window c landed (unit 2c + 1 may still fly)
        lds_barrier();                                    // everyone's; and every wave has left unit 2c - 1 and window c - 1
        dma_unit(2 * c + 2);
        asm volatile("" ::: "memory");
        multiply(c, std::integral_constant<int, 0>{}, slot);        // issues window c + 1 on the way
        if (++slot == RS) slot = 0;
        wait_vmcnt<GU + GW>();                            // own pieces of unit 2c + 1 landed
        lds_barrier();
        dma_unit(2 * c + 3);
        asm volatile("" ::: "memory");
        multiply(c, std::integral_constant<int, 1>{}, slot);
        if (++slot == RS) slot = 0;
    }
    wait_vmcnt<0>();                                      // the clamped tail copies: nothing may be in flight when the wave ends

    // ---- epilogue: conv_kernel_h3r's, transposed branch (t = stride u - pad + n / Cout) ----
    const size_t obs = (size_t)a.Tout * a.Cout;
    float* outb = a.out + (size_t)b * obs;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = u0 + w * RW + i * 16 + (lane & 15);
        if (u >= a.U) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n0 + j * 16 + 4 * (lane >> 4);
            const int r = n / a.Cout, co = n - r * a.Cout, t = a.stride * u - a.pad_t + r;
            if (t < 0 || t >= a.Tout) continue;
            const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
            *reinterpret_cast<float4*>(outb + (size_t)t * a.Cout + co) =
                make_float4(acc[i][j][0] + bs.x, acc[i][j][1] + bs.y, acc[i][j][2] + bs.z, acc[i][j][3] + bs.w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_kernel_s1: wide 1x1 convs (C = 384 / 768 residual units) without an activation window.
//
// On conv_kernel_h3r a 1x1 layer spends a chunk's 7.7 k cycles as 2.3 k of matrix-pipe work, 2.1 k of Snake / split / LDS writes
// and 3.3 k of barriers and waits around them: one tap per staged window.  A 1x1 conv has no halo, so nothing is shared between
// waves but the weights: every wave loads ITS 32 rows x 32 channels straight into MFMA B-fragment order (lane = (channel group g,
// row t): 8 consecutive channels = 32 contiguous bytes, the four groups of a row one 128-byte line), two chunks ahead in two
// register sets, applies Snake + the fp16 split in registers while the other wave of its SIMD multiplies, and the weight units
// travel through the LDS-DMA ring as before.  One barrier per chunk (the ring's), no LDS traffic for activations.  The fragments
// of chunk c + 1 are prepared BEHIND the MFMAs of chunk c: the plain loads they need were issued a whole MFMA phase earlier, so
// the vmcnt drain hipcc puts in front of them (it does not count LDS-DMA exactly) finds everything landed.
// Same products in the same order as conv_kernel_h3: bit-identical.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void conv_kernel_s1(ConvArgs a) {
    constexpr int TN = 192, NT = TN / 16, NI = 2, RW = 32, TMW = 8 * RW, RS = 4, D = 3, UB = TN * 128, G = UB / 8192;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                            // [RS][hi: TN x 64 B | lo: TN x 64 B]
    float* als = reinterpret_cast<float*>(smem + RS * UB);        // [Cin] Snake alpha, [Cin] 1 / (alpha + 1e-9)
    float* ivs = als + a.Cin;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t16 = lane & 15;
    // (an XCD-aware block map that puts the output blocks of one row tile on one XCD, so that the rows come from HBM once, measured
    // slower: 1.73 vs 1.62 ms at C = 768 - each XCD then cycles through all the layer's weight units instead of one block's)
    const int nb = blockIdx.x;
    const int n0 = nb * TN;
    const int u0 = blockIdx.y * TMW;
    const int b = blockIdx.z;
    const int nchunks = a.Cin / KC3;
    const bool snake = a.alpha != nullptr;
    const char* units = reinterpret_cast<const char*>(a.w_units) + (size_t)nb * nchunks * UB;
    const float* inb = a.in + (size_t)b * a.Tin * a.Cin;

    auto dma = [&](int u) {                               // unit (= chunk) u into slot u % RS; past the end: the last one again
        const int uc = u < nchunks ? u : nchunks - 1;
        const char* src = units + (size_t)uc * UB + (size_t)tid * 16;
        char* dst = ring + (u % RS) * UB + w * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
        for (int j = 0; j < G; ++j)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(src + j * 8192), (lds_void*)(dst + j * 8192), 16, 0, 0);
#else
        (void)src; (void)dst;
#endif
    };
    // the lane's two rows (clamped: rows past the end are computed on the last row and never stored)
    const float* xrow[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int u = u0 + w * RW + i * 16 + t16;
        if (u > a.Tin - 1) u = a.Tin - 1;
        xrow[i] = inb + (size_t)u * a.Cin + 8 * g;
    }
    // Round 6: the activation loads are inline asm with counted waits.  As compiler-visible loads the first use of a loaded register
    // got `s_waitcnt vmcnt(0)` (hipcc does not count through LDS-DMA instructions): at the END of every step the wave drained the
    // loads AND the weight units it had issued at the START of that same step - a full memory round trip inside every 72-MFMA step,
    // waves parked 60 % of their cycles, matrix pipe busy 0.23-0.33 (profiles/r06_pmc_dac.txt).  Now chunk c + 2's rows are requested
    // at the start of step c and awaited at the end of step c + 1 with `vmcnt(XL + G)`: only the unit and the rows issued since stay in
    // flight.  ("+v": the registers stay live, no spill / copy between the load and its wait; the kernel has no scratch.)
    constexpr int XL = 2 * NI;                             // load instructions per xload
    f32x4 xs[2][NI][2] = {};                               // [register set][row tile][two quads = 8 channels]
    auto xload = [&](auto setc, int c) {
        constexpr int S = decltype(setc)::value;
        const int cc = c < nchunks ? c : nchunks - 1;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float* p = xrow[i] + cc * KC3;
            f32x4 &q0 = xs[S][i][0], &q1 = xs[S][i][1];   // (named first: an asm operand inside a generic lambda does not capture)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(q0) : "v"(p) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "+v"(q1) : "v"(p) : "memory");
        }
    };
    half8 fh[NI], fl[NI];                                  // B fragments of the chunk about to be multiplied
    auto prepare = [&](auto setc, int c, auto youngerc) {  // Snake + split of chunk c (register set S) into fh / fl
        constexpr int S = decltype(setc)::value;
        static_assert(NI == 2, "the wait below names the set's four registers");
        // the set's loads have landed when at most `younger` operations issued after them are outstanding (loads and LDS-DMA retire in order)
        f32x4 &w0 = xs[S][0][0], &w1 = xs[S][0][1], &w2 = xs[S][1][0], &w3 = xs[S][1][1];
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "n"(decltype(youngerc)::value));
        const int cc = c < nchunks ? c : nchunks - 1;
        f32x4 al0 = {0, 0, 0, 0}, al1 = {0, 0, 0, 0}, iv0 = {0, 0, 0, 0}, iv1 = {0, 0, 0, 0};
        if (snake) {
            al0 = *reinterpret_cast<const f32x4*>(als + cc * KC3 + 8 * g); al1 = *reinterpret_cast<const f32x4*>(als + cc * KC3 + 8 * g + 4);
            iv0 = *reinterpret_cast<const f32x4*>(ivs + cc * KC3 + 8 * g); iv1 = *reinterpret_cast<const f32x4*>(ivs + cc * KC3 + 8 * g + 4);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float x8[8] = {xs[S][i][0][0], xs[S][i][0][1], xs[S][i][0][2], xs[S][i][0][3],
                                 xs[S][i][1][0], xs[S][i][1][1], xs[S][i][1][2], xs[S][i][1][3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = snake ? snake_f(x8[e], e < 4 ? al0[e] : al1[e - 4], e < 4 ? iv0[e] : iv1[e - 4]) : x8[e];
                _Float16 h, l;
                split_h(y, h, l);
                fh[i][e] = h;
                fl[i][e] = l;
            }
        }
    };

    f32x4 acc[NI][NT];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = {0, 0, 0, 0};

    // prologue: Snake table, chunk 0 prepared, chunk 1 in flight, units 0 .. D - 1 in flight
    if (snake)
        for (int i = tid; i < a.Cin; i += 512) { const float al = a.alpha[i]; als[i] = al; ivs[i] = 1.0f / (al + 1e-9f); }
    xload(std::integral_constant<int, 0>{}, 0);
    xload(std::integral_constant<int, 1>{}, 1);
    lds_barrier();                                        // the table
    prepare(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, XL>{});      // chunk 1's rows may still fly
#pragma unroll
    for (int v = 0; v < D; ++v) dma(v);

    int slot = 0;
    auto step = [&](int c, auto parc) {
        constexpr int P = decltype(parc)::value;           // c % 2: chunk c was loaded into set P, chunk c + 1 sits in set 1 - P
        // own pieces of unit c: step 0 waits for them itself (younger: units 1 .. D - 1); from step 1 on the wait at the end of the
        // step before (vmcnt <= XL + G, for rows issued BEHIND unit c) has already covered them
        if (c == 0) wait_vmcnt<G * (D - 1)>();
        lds_barrier();                                    // everyone's pieces of unit c are in LDS, everyone is done reading unit c - 1
        dma(c + D);
        xload(std::integral_constant<int, P>{}, c + 2);   // set P is free: chunk c is in fh / fl
        asm volatile("" ::: "memory");
        const _Float16* wh = reinterpret_cast<const _Float16*>(ring + slot * UB);
        const _Float16* wl = wh + TN * LDH;
        const size_t wo = (size_t)t16 * LDH + 8 * swz16(t16, g);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const half8 bh = *reinterpret_cast<const half8*>(wh + wo + (size_t)j * 16 * LDH);
            const half8 bl = *reinterpret_cast<const half8*>(wl + wo + (size_t)j * 16 * LDH);
#pragma unroll
            for (int i = 0; i < NI; ++i) {                // small terms first, then the leading term
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, fh[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fl[i], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, fh[i], acc[i][j], 0, 0, 0);
            }
        }
        if (++slot == RS) slot = 0;
        // behind the MFMAs: the next chunk's fragments.  Its rows went out at the start of step c - 1 (step 0: in the prologue, which is
        // further back still); issued since: this step's unit and rows
        prepare(std::integral_constant<int, 1 - P>{}, c + 1, std::integral_constant<int, XL + G>{});
    };
    for (int c = 0; c < nchunks; c += 2) {
        step(c, std::integral_constant<int, 0>{});
        if (c + 1 < nchunks) step(c + 1, std::integral_constant<int, 1>{});
    }
    wait_vmcnt<0>();

    const size_t obs = (size_t)a.Tout * a.Cout;
    float* outb = a.out + (size_t)b * obs;
    const float* resb = a.res ? a.res + (size_t)b * obs : nullptr;
    if (a.alpha_out) {      // packed output: the consumer's (alpha, 1 / (alpha + 1e-9)) of this block's channels over the dead input tables (Cin >= TN)
        lds_barrier();                                    // every wave is done with the input tables
        for (int i = tid; i < TN; i += 512) { const float ao = a.alpha_out[n0 + i]; als[i] = ao; ivs[i] = 1.0f / (ao + 1e-9f); }
        lds_barrier();
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = u0 + w * RW + i * 16 + t16;
        if (u >= a.U) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = n0 + j * 16 + 4 * g;
            const float4 bs = *reinterpret_cast<const float4*>(a.bias + co);
            float4 y = make_float4(acc[i][j][0] + bs.x, acc[i][j][1] + bs.y, acc[i][j][2] + bs.z, acc[i][j][3] + bs.w);
            if (resb) {
                const float4 rv = *reinterpret_cast<const float4*>(resb + (size_t)u * a.Cout + co);
                y.x += rv.x; y.y += rv.y; y.z += rv.z; y.w += rv.w;
            }
            // the next layer is a transposed conv (conv.h): Snake + split here, once (in place: see store_packed8)
            if (a.alpha_out)
                store_packed8(outb + (size_t)u * a.Cout, co, y, *reinterpret_cast<const float4*>(als + co - n0), *reinterpret_cast<const float4*>(ivs + co - n0));
            else *reinterpret_cast<float4*>(outb + (size_t)u * a.Cout + co) = y;
        }
    }
}

// A fragments of a fused 1x1 stage (C -> C): unit m = k-step m: { hi: [C / 16 blocks][64 lanes] x 16 B, lo: same }; lane (g = l >> 4,
// n = l & 15) of block jb holds W[16 jb + n][channel of k position 8 g + e], e = 0 .. 7, with the accumulator's channel order
// channel = 32 m + 4 g + e (e < 4), 32 m + 16 + 4 g + (e - 4) (e >= 4)
__global__ void conv_units2_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, uint4* __restrict__ out, int C) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte fragment
    const int NB = C / 16;
    const size_t total = (size_t)(C / 32) * 2 * NB * 64;
    if (idx >= total) return;
    const int l = (int)(idx % 64), jb = (int)((idx / 64) % NB), part = (int)((idx / (64 * (size_t)NB)) % 2), m = (int)(idx / (128 * (size_t)NB));
    const int g = l >> 4, n = 16 * jb + (l & 15);
    const _Float16* row = (part ? lo : hi) + (size_t)n * C + 32 * m;
    _Float16 v[8];
    for (int e = 0; e < 4; ++e) { v[e] = row[4 * g + e]; v[4 + e] = row[16 + 4 * g + e]; }
    out[idx] = *reinterpret_cast<uint4*>(v);
}

// unit image of the split weights: [N / TN][Cin / 32][taps] x { hi: TN rows x 64 B, lo: TN rows x 64 B }, 16-byte columns swizzled
// as the LDS reads expect (swz16 of the row)
__global__ void conv_units_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo, uint4* __restrict__ out, int taps,
                                  int N, int Cin, int TN) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte piece
    const int nchunks = Cin / 32;
    const size_t total = (size_t)(N / TN) * nchunks * taps * TN * 8;
    if (idx >= total) return;
    const int q = (int)(idx % 4);                 // stored column
    const int n = (int)((idx / 4) % TN);
    const int part = (int)((idx / (4 * (size_t)TN)) % 2);
    const size_t unit = idx / (8 * (size_t)TN);
    const int t = (int)(unit % taps), c = (int)((unit / taps) % nchunks), nb = (int)(unit / ((size_t)taps * nchunks));
    const int qs = swz16(n, q);                   // the source column that lands in stored column q (the swizzle is an involution)
    const _Float16* src = (part ? lo : hi) + ((size_t)t * N + (size_t)nb * TN + n) * Cin + c * 32 + 8 * qs;
    out[idx] = *reinterpret_cast<const uint4*>(src);
}

// ------------------------------------------------------------------------------------------------
// Streaming 1x1 convolution of a residual unit (HF modeling_dac.py:175-210: x + conv1(snake(h))) for the narrow, long
// blocks (C = 96 / 192 channels, T up to 1.3 M rows per utterance), where the layer is HBM-bound: 3 tensor passes (read h,
// read x, write x') against 2 C^2 flops per row.  A 1x1 conv has no halo and no tap reuse, so staging the activations in
// LDS (conv_kernel_h3: two barriers per 32-channel chunk, one workgroup per CU) only costs: here every wave streams its
// own 16-row tiles straight from global memory into MFMA B fragments (lane = (channel group g, row t): 8 consecutive
// channels = 32 contiguous bytes, the four groups of a row one 128-byte line), applies Snake and the fp16 hi / lo split
// in registers, and multiplies with the weight fragments that sit in LDS for the whole launch (A-operand order, hi + lo:
// 36 KB at C = 96, 144 KB at C = 192).  No barrier after the weights are parked; the epilogue adds bias and the residual
// and stores 4 consecutive channels per lane.  Same products in the same order as conv_kernel_h3 (bl.xh, bh.xl, bh.xh per
// 32-channel step, ascending): results are bit-identical to it.
// ------------------------------------------------------------------------------------------------
template <int C, int NWV>
__global__ __launch_bounds__(NWV * 64) void conv1x1_stream_kernel(ConvArgs a, long long rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NB = C / 16, KS = C / 32;
    half8* wh = reinterpret_cast<half8*>(smem);                  // [NB][KS][64 lanes]
    half8* wl = wh + NB * KS * 64;
    float* als = reinterpret_cast<float*>(wl + NB * KS * 64);    // [C] Snake alpha, [C] 1 / (alpha + 1e-9)
    float* ivs = als + C;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, t16 = lane & 15;
    {   // park the weights: tile (jb, m) lane l = W[16 jb + (l & 15)][32 m + 8 (l >> 4) .. + 7]
        const _Float16* gh = reinterpret_cast<const _Float16*>(a.w_hi);
        const _Float16* gl = reinterpret_cast<const _Float16*>(a.w_lo);
        for (int i = tid; i < NB * KS * 64; i += NWV * 64) {
            const int l = i & 63, m = (i >> 6) % KS, jb = (i >> 6) / KS;
            const size_t off = (size_t)(16 * jb + (l & 15)) * C + 32 * m + 8 * (l >> 4);
            wh[i] = *reinterpret_cast<const half8*>(gh + off);
            wl[i] = *reinterpret_cast<const half8*>(gl + off);
        }
        for (int i = tid; i < C; i += NWV * 64) { const float al = a.alpha[i]; als[i] = al; ivs[i] = 1.0f / (al + 1e-9f); }
    }
    __syncthreads();
    const long long ntiles = (rows + 15) / 16;
    for (long long tile = (long long)blockIdx.x * NWV + w; tile < ntiles; tile += (long long)gridDim.x * NWV) {
        const long long row = tile * 16 + t16;
        const bool live = row < rows;
        const size_t rbase = (size_t)(live ? row : rows - 1) * C;
        const float* xin = a.in + rbase + 8 * g;
        f32x4 acc[NB];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) acc[jb] = {0, 0, 0, 0};
        float4 v0 = *reinterpret_cast<const float4*>(xin), v1 = *reinterpret_cast<const float4*>(xin + 4);
#pragma unroll
        for (int m = 0; m < KS; ++m) {
            const float4 c0 = v0, c1 = v1;
            if (m + 1 < KS) {                      // next k-step's 32 bytes are requested before this step's matrix work
                v0 = *reinterpret_cast<const float4*>(xin + 32 * (m + 1));
                v1 = *reinterpret_cast<const float4*>(xin + 32 * (m + 1) + 4);
            }
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(als + 32 * m + 8 * g), a1 = *reinterpret_cast<const f32x4*>(als + 32 * m + 8 * g + 4);
            const f32x4 i0 = *reinterpret_cast<const f32x4*>(ivs + 32 * m + 8 * g), i1 = *reinterpret_cast<const f32x4*>(ivs + 32 * m + 8 * g + 4);
            const float x8[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float al8[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            const float iv8[8] = {i0[0], i0[1], i0[2], i0[3], i1[0], i1[1], i1[2], i1[3]};
            half8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = snake_f(x8[j], al8[j], iv8[j]);
                _Float16 h, l;
                split_h(y, h, l);
                bh[j] = h;
                bl[j] = l;
            }
            // C = 96: the 36 weight fragments are loop invariant and hipcc keeps them in registers (no LDS read per tile);
            // C = 192 would need 576 registers for that: the clobber pins the reads inside the loop
            if (C > 96) asm volatile("" ::: "memory");
#pragma unroll
            for (int jb = 0; jb < NB; ++jb) {
                const half8 ah = wh[(jb * KS + m) * 64 + lane];
                const half8 alo = wl[(jb * KS + m) * 64 + lane];
                acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh, acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[jb], 0, 0, 0);
                acc[jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[jb], 0, 0, 0);
            }
        }
        const float* resr = a.res + rbase + 4 * g;
        float* outr = a.out + rbase + 4 * g;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const float4 rv = *reinterpret_cast<const float4*>(resr + 16 * jb);
            const float4 bs = *reinterpret_cast<const float4*>(a.bias + 16 * jb + 4 * g);
            float4 y;
            y.x = (acc[jb][0] + bs.x) + rv.x; y.y = (acc[jb][1] + bs.y) + rv.y;
            y.z = (acc[jb][2] + bs.z) + rv.z; y.w = (acc[jb][3] + bs.w) + rv.w;
            if (live) *reinterpret_cast<float4*>(outr + 16 * jb) = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Last layer of the codec decoder: snake -> Conv1d(C -> 1, k = 7) -> tanh (HF modeling_dac.py:436-442).  One output
// channel makes it a memory-bound reduction (672 MACs per sample against 384 bytes read), not a GEMM: on the 64-wide MFMA
// tiles of conv_kernel_h3 63 of 64 output columns are padding.  Here a thread owns one output sample; the input window
// (256 + 6 rows, Snake applied once per element) goes through LDS in 32-channel chunks with a 33-float row stride
// (consecutive threads read consecutive rows: conflict free), the 7 x C weights are wave-uniform scalar operands, the sum runs
// in exact fp32 FMAs in (chunk, tap, channel) order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_last_kernel(ConvArgs a) {
    constexpr int TR = 256, HALO = 6, CH = 32, LD = CH + 1;
    __shared__ float xs[(TR + HALO) * LD];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * TR, b = blockIdx.y;
    const float* inb = a.in + (size_t)b * (a.in_bs ? a.in_bs : (size_t)a.Tin * a.Cin);
    float acc = 0.f;
    for (int c0 = 0; c0 < a.Cin; c0 += CH) {
        __syncthreads();
        for (int i = tid; i < (TR + HALO) * (CH / 4); i += 256) {
            const int r = i / (CH / 4), q = i % (CH / 4);
            const int t = t0 - 3 + r;
            float4 v = make_float4(0, 0, 0, 0);
            if (t >= 0 && t < a.Tin) {
                v = *reinterpret_cast<const float4*>(inb + (size_t)t * a.Cin + c0 + 4 * q);
                const float4 al = *reinterpret_cast<const float4*>(a.alpha + c0 + 4 * q);
                v.x = snake_f(v.x, al.x, 1.0f / (al.x + 1e-9f)); v.y = snake_f(v.y, al.y, 1.0f / (al.y + 1e-9f));
                v.z = snake_f(v.z, al.z, 1.0f / (al.z + 1e-9f)); v.w = snake_f(v.w, al.w, 1.0f / (al.w + 1e-9f));
            }
            float* d = xs + r * LD + 4 * q;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float* xr = xs + (tid + k) * LD;
            const float* wk = a.w + (size_t)k * a.Cin + c0;       // packed [tap][N = 1][Cin]: uniform -> scalar loads
#pragma unroll
            for (int c = 0; c < CH; ++c) acc = fmaf(wk[c], xr[c], acc);
        }
    }
    const int t = t0 + tid;
    if (t < a.Tout) {
        float y = acc + a.bias[0];
        if (a.act_out == 1) y = tanhf(y);
        a.out[(size_t)b * (a.out_bs ? a.out_bs : (size_t)a.Tout) + t] = y;
    }
}

__global__ void split_halves_kernel(const float* src, _Float16* hi, _Float16* lo, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    _Float16 h, l;
    split_h(src[i], h, l);
    hi[i] = h; lo[i] = l;
}

// z[b][f][c] = sum_k table[k][codes[b][k][f]][c]   (left to right over codebooks, as from_codes adds them)
__global__ void dac_latent_kernel(const long long* codes, const float* table, float* z, int K, int F, int C, int ncodes) {
    const int f = blockIdx.x, b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const long long code = codes[((size_t)b * K + k) * F + f];
            // precondition (checked by the host layer): 0 <= code < ncodes; an id outside contributes nothing instead of
            // reading past the table
            if ((unsigned long long)code < (unsigned long long)ncodes) s = s + table[((size_t)k * ncodes + code) * C + c];
        }
        z[((size_t)b * F + f) * C + c] = s;
    }
}

// table[k][code][c] = bias_k[c] + sum_d out_proj_k[c][d] * codebook_k[code][d]   (1x1 conv of the embedding)
__global__ void dac_table_kernel(const float* codebook, const float* proj_w, const float* proj_b, float* table, int ncodes,
                                 int dim, int C) {
    const int code = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int d = 0; d < dim; ++d) s = fmaf(proj_w[(size_t)c * dim + d], codebook[(size_t)code * dim + d], s);
        table[(size_t)code * C + c] = s + proj_b[c];
    }
}

// Conv1d weight [Cout][Cin][k] -> [k][Npad][Cin];  ConvTranspose1d weight [Cin][Cout][2s] -> [2][s*Cout][Cin]
__global__ void conv_pack_kernel(const float* w, float* out, int Cout, int Cin, int k, int transposed, int s, int N) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)(transposed ? 2 : k) * N * Cin;
    if (idx >= total) return;
    const int ci = (int)(idx % Cin);
    const int n = (int)((idx / Cin) % N);
    const int tap = (int)(idx / ((size_t)Cin * N));
    float v = 0.f;
    if (!transposed) {
        if (n < Cout) v = w[((size_t)n * Cin + ci) * k + tap];
    } else {
        const int r = n / Cout, co = n % Cout;
        if (r < s) v = w[((size_t)ci * Cout + co) * (2 * s) + r + tap * s];
    }
    out[idx] = v;
}


// ---- codec ENCODE side ----------------------------------------------------------------------------------------------
__global__ void enc_first_conv_kernel(const float* __restrict__ audio, const float* __restrict__ w, const float* __restrict__ bias,
                                      float* __restrict__ out, int T, int C, size_t out_bs) {
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;    // (t, c)
    if (i >= (size_t)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    const float* x = audio + (size_t)b * T;
    float acc = bias[c];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int tt = t + k - 3;
        if (tt >= 0 && tt < T) acc = fmaf(w[c * 7 + k], x[tt], acc);
    }
    out[(size_t)b * out_bs + (size_t)t * C + c] = acc;
}

__global__ void zero_tail_kernel(float* buf, size_t bs, size_t from, size_t to) {
    const size_t i = from + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) buf[(size_t)blockIdx.y * bs + i] = 0.f;
}

__global__ void conv_pack_strided_kernel(const float* __restrict__ w, float* __restrict__ out, int N, int C, int stride) {
    const int Ce = stride * C, pad = (stride + 1) / 2;
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)3 * N * Ce) return;
    const int ce = (int)(idx % Ce), n = (int)((idx / Ce) % N), tap = (int)(idx / ((size_t)Ce * N));
    const int j = ce / C, c = ce % C;
    const int k = (tap - 1) * stride + j + pad;
    out[idx] = (k >= 0 && k < 2 * stride) ? w[((size_t)n * C + c) * (2 * stride) + k] : 0.f;
}

__global__ void tile_alpha_kernel(const float* __restrict__ alpha, float* __restrict__ out, int C, int reps) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * reps) out[i] = alpha[i % C];
}

__global__ void cb_normalize_kernel(const float* __restrict__ cb, float* __restrict__ cbn, float* __restrict__ cbn2, int entries, int dim) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= entries) return;
    float ss = 0.f;
    for (int d = 0; d < dim; ++d) ss += cb[j * dim + d] * cb[j * dim + d];
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float s2 = 0.f;
    for (int d = 0; d < dim; ++d) { const float v = cb[j * dim + d] / nrm; cbn[j * dim + d] = v; s2 += v * v; }
    cbn2[j] = s2;
}

// Residual vector quantisation of one frame per workgroup (256 threads): for every codebook project the residual to
// `dim` (in_proj), L2-normalise, take the nearest normalised codebook entry (first maximum of
// -(|e|^2 - 2 e.c) + |c|^2, as DacVectorQuantize.decode_latents computes it), subtract out_proj(codebook[idx]).
__global__ __launch_bounds__(256) void rvq_encode_kernel(RvqArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* r = reinterpret_cast<float*>(smem);             // [D] residual
    __shared__ float part[4][16];
    __shared__ float ze[16];
    __shared__ float bestv[4];
    __shared__ int besti[4];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int D = a.D, dim = a.dim;
    for (int d = tid; d < D; d += 256) r[d] = a.z[((size_t)b * a.F + f) * D + d];
    __syncthreads();
    for (int k = 0; k < a.K; ++k) {
        // in_proj: dim dot products of length D
        const float* W = a.in_w[k];
        for (int o = 0; o < dim; ++o) {
            float acc = 0.f;
            for (int d = tid; d < D; d += 256) acc = fmaf(W[(size_t)o * D + d], r[d], acc);
            acc = wave_sum(acc);
            if (lane == 0) part[w][o] = acc;
        }
        __syncthreads();
        if (tid < dim) ze[tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + a.in_b[k][tid];
        __syncthreads();
        float e[16], l2 = 0.f, ss = 0.f;
        for (int o = 0; o < dim; ++o) ss += ze[o] * ze[o];
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        for (int o = 0; o < dim; ++o) { e[o] = ze[o] / nrm; l2 += e[o] * e[o]; }
        // nearest entry: first index of the maximum
        float bv = -INFINITY; int bi = 0x7fffffff;
        const float* cn = a.cbn[k];
        for (int j = tid; j < a.entries; j += 256) {
            float dot = 0.f;
            for (int o = 0; o < dim; ++o) dot = fmaf(e[o], cn[(size_t)j * dim + o], dot);
            const float dist = -(l2 - 2.f * dot) + a.cbn2[k][j];
            if (dist > bv || (dist == bv && j < bi)) { bv = dist; bi = j; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { bestv[w] = bv; besti[w] = bi; }
        __syncthreads();
        bv = bestv[0]; bi = besti[0];
        for (int i = 1; i < 4; ++i) if (bestv[i] > bv || (bestv[i] == bv && besti[i] < bi)) { bv = bestv[i]; bi = besti[i]; }
        if (tid == 0) a.codes[((size_t)b * a.K + k) * a.F + f] = bi;
        // residual -= out_proj(codebook[idx]) (+ bias)
        const float* cq = a.cb[k] + (size_t)bi * dim;
        const float* Wo = a.out_w[k];
        for (int d = tid; d < D; d += 256) {
            float acc = a.out_b[k][d];
            for (int o = 0; o < dim; ++o) acc = fmaf(Wo[(size_t)d * dim + o], cq[o], acc);
            r[d] -= acc;
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t conv_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3<64, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3<96, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3<64, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3<96, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<96, 64, 7, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<96, 64, 2, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<96, 64, 1, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_s1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(convt_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<96, 64, 7, 3, true, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<96, 64, 7, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<192, 32, 7, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<192, 32, 7, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<192, 64, 7, 4, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<192, 32, 2, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_kernel_h3r<192, 32, 1, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_stream_kernel<192, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_stream_kernel<192, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024);
}

size_t conv_h3_lds(const ConvArgs& a, int TN, int RW = 32) {
    const int lo = a.tstep >= 0 ? a.toff : a.toff + (a.taps - 1) * a.tstep;
    const int hi = a.tstep >= 0 ? a.toff + (a.taps - 1) * a.tstep : a.toff;
    const int win = 8 * RW + (hi - lo);
    return ((size_t)2 * win * LDH + (size_t)2 * a.taps * TN * LDH) * sizeof(_Float16);
}

// which kernel each conv_launch call dispatched (host-side tallies: tests assert that a decode at production size ran on
// the production kernels and not on a fallback; include/parler_tts_hip.h: ptts_dac_kernel_counts)
static std::atomic<long long> g_conv_counts[CONV_KIND_COUNT];
static inline void tally(int kind) { g_conv_counts[kind].fetch_add(1, std::memory_order_relaxed); }
// conv_kind_of(): the dispatch below run as a plan - the kernel family it WOULD launch is written here and nothing is launched
static thread_local int* g_conv_plan = nullptr;
#define PTTS_CONV_PICK(kind) do { if (g_conv_plan) { *g_conv_plan = (kind); return hipSuccess; } } while (0)
// packed activations (conv.h): which kernel families write them / read them
static inline bool conv_kind_packs(int k) { return k == CONV_KIND_FUSED_96 || k == CONV_KIND_FUSED_192 || k == CONV_KIND_S1 ||
                                                   k == CONV_KIND_RING_96_T7 || k == CONV_KIND_RING_192_T7 || k == CONV_KIND_RING_96_T1 || k == CONV_KIND_RING_192_T1; }
static inline bool conv_kind_unpacks(int k) { return k == CONV_KIND_RING_192_T2; }      // convt_dma_kernel takes over from the 192-wide two-tap ring instance
#define PTTS_CONV_PACK_GUARD(kind) do { if ((a.alpha_out && !conv_kind_packs(kind)) || (a.in_packed && !conv_kind_unpacks(kind))) return hipErrorInvalidValue; } while (0)
void conv_kernel_counts(long long* out, int n, bool reset) {
    for (int i = 0; i < n && i < CONV_KIND_COUNT; ++i) out[i] = g_conv_counts[i].load(std::memory_order_relaxed);
    for (int i = CONV_KIND_COUNT; i < n; ++i) out[i] = 0;
    if (reset) for (int i = 0; i < CONV_KIND_COUNT; ++i) g_conv_counts[i].store(0, std::memory_order_relaxed);
}

#ifdef PTTS_CONV_LIFE
static hipError_t conv_launch_inner(const ConvArgs& a, hipStream_t s);
hipError_t conv_launch(const ConvArgs& a, hipStream_t s) {
    if (g_conv_plan) return conv_launch_inner(a, s);
    unsigned long long z[4] = {0, 0, 0, 0};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_life), z, sizeof(z));
    (void)hipEventRecord(e0, s);
    const hipError_t e = conv_launch_inner(a, s);
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_conv_life), sizeof(z));
    if (z[3]) {
        const double n = (double)z[3], life = (z[0] + z[1] + z[2]) / n * 0.01, slot_us = ms * 1e3 * (n < 256 ? n : 256.0) / n;
        fprintf(stderr, "ring life Cin=%d N=%d taps=%d dil=%d U=%d fused=%d: %.0f workgroups, kernel %.3f ms = %.1f us per workgroup slot | prologue %.2f us, "
                "chunk loop %.2f, stage 2 + epilogue + store drain %.2f, life %.2f -> between workgroups %.2f us\n", a.Cin, a.N, a.taps,
                a.tstep, a.U, a.w2_units ? 1 : 0, n, ms, slot_us, z[0] / n * 0.01, z[1] / n * 0.01, z[2] / n * 0.01, life, slot_us - life);
    }
    return e;
}
static hipError_t conv_launch_inner(const ConvArgs& a, hipStream_t s) {
#elif defined(PTTS_CONV_STAMPS)
static hipError_t conv_launch_inner(const ConvArgs& a, hipStream_t s);
hipError_t conv_launch(const ConvArgs& a, hipStream_t s) {
    unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), z, sizeof(z));
    const hipError_t e = conv_launch_inner(a, s);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_conv_stamps), sizeof(z));
    unsigned long long zw[8][8];
    (void)hipMemcpyFromSymbol(zw, HIP_SYMBOL(g_conv_wave), sizeof(zw));
    if (z[5] && z[8]) {
        for (int wv = 0; wv < 8; ++wv)
            fprintf(stderr, "  wave %d: dma-wait %.0f tap-barrier %.0f issue %.0f mfma %.0f chunk-barrier %.0f stash %.0f\n", wv, (double)zw[wv][0] / z[5],
                    (double)zw[wv][1] / z[5], (double)zw[wv][2] / z[5], (double)zw[wv][3] / z[5], (double)zw[wv][4] / z[5], (double)zw[wv][5] / z[5]);
    }
    static const unsigned long long zero_w[8][8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_wave), zero_w, sizeof(zero_w));
    if (z[5] && z[8])
        fprintf(stderr, "ring stamps Cin=%d N=%d taps=%d dil=%d U=%d: per chunk (cycles) dma-wait %.1f tap-barrier %.1f issue %.1f mfma %.1f chunk-barrier %.1f stash %.1f | chunks %llu kernel/wg %.0f\n",
                a.Cin, a.N, a.taps, a.tstep, a.U, (double)z[8] / z[5], (double)z[9] / z[5], (double)z[10] / z[5], (double)z[11] / z[5],
                (double)z[12] / z[5], (double)z[13] / z[5], z[5], (double)z[6] / ((double)z[5] / (a.Cin / 32)));
    else if (z[5])
        fprintf(stderr, "conv stamps Cin=%d N=%d taps=%d dil=%d U=%d: per chunk (s_memtime ticks) fetch %.1f mfma %.1f bar %.1f stash %.1f bar %.1f | chunks %llu kernel/wg %.0f\n",
                a.Cin, a.N, a.taps, a.tstep, a.U, (double)z[0] / z[5], (double)z[1] / z[5], (double)z[2] / z[5], (double)z[3] / z[5],
                (double)z[4] / z[5], z[5], (double)z[6] / ((double)z[5] / (a.Cin / 32)));
    return e;
}
static hipError_t conv_launch_inner(const ConvArgs& a, hipStream_t s) {
#else
hipError_t conv_launch(const ConvArgs& a, hipStream_t s) {
#endif
    if (a.taps > MAX_TAPS || a.Cin % KC != 0) return hipErrorInvalidValue;
    // last layer (one output channel, 7 taps): the reduction kernel
    if (getenv("PTTS_DAC_NO_LAST_KERNEL") == nullptr && a.Cout == 1 && a.N == 1 && a.taps == 7 && a.tstep == 1 && a.toff == -3 &&
        a.stride == 1 && a.Cin % 32 == 0 && a.alpha && !a.res && a.Tin == a.Tout) {
        PTTS_CONV_PICK(CONV_KIND_LAST);
        PTTS_CONV_PACK_GUARD(CONV_KIND_LAST);
        hipLaunchKernelGGL(conv_last_kernel, dim3((a.Tout + 255) / 256, a.B), dim3(256), 0, s, a);
        tally(CONV_KIND_LAST);
        return hipGetLastError();
    }
    // residual 1x1 conv of the narrow, long blocks: the streaming kernel (dense rows, Snake on the input, residual add)
    const bool no_stream = getenv("PTTS_DAC_NO_STREAM_1X1") != nullptr;      // A/B and the equivalence test
    if (!no_stream && a.taps == 1 && a.stride == 1 && a.w_hi && a.w_lo && a.alpha && a.res && a.act_out == 0 && a.Cin == a.Cout &&
        a.N == a.Cout && a.in_bs == 0 && a.out_bs == 0 && a.Tin == a.Tout && a.U == a.Tout && (a.Cin == 96 || a.Cin == 192)) {
        const long long rows = (long long)a.B * a.Tout;
        const long long tiles = (rows + 15) / 16;
        PTTS_CONV_PICK(a.Cin == 96 ? CONV_KIND_STREAM_96 : CONV_KIND_STREAM_192);
        PTTS_CONV_PACK_GUARD(CONV_KIND_STREAM_96);
        if (a.Cin == 96) {
            const size_t lds = (size_t)6 * 3 * 64 * 16 * 2 + 2 * 96 * sizeof(float);
            long long g = (tiles + 3) / 4; if (g > 256 * 4) g = 256 * 4;
            hipLaunchKernelGGL((conv1x1_stream_kernel<96, 4>), dim3((unsigned)g), dim3(256), lds, s, a, rows);
            tally(CONV_KIND_STREAM_96);
        } else {
            const size_t lds = (size_t)12 * 6 * 64 * 16 * 2 + 2 * 192 * sizeof(float);
            if (getenv("PTTS_DAC_1X1_WAVES8")) {
                long long g = (tiles + 7) / 8; if (g > 256) g = 256;
                hipLaunchKernelGGL((conv1x1_stream_kernel<192, 8>), dim3((unsigned)g), dim3(512), lds, s, a, rows);
            } else {
                long long g = (tiles + 15) / 16; if (g > 256) g = 256;
                hipLaunchKernelGGL((conv1x1_stream_kernel<192, 16>), dim3((unsigned)g), dim3(1024), lds, s, a, rows);
            }
            tally(CONV_KIND_STREAM_192);
        }
        return hipGetLastError();
    }
    // LDS-DMA ring kernel: 64 rows per wave, unit image present, long enough launches
    const bool no_ring = getenv("PTTS_DAC_NO_RING") != nullptr;          // A/B and the equivalence test
    const int ring_taps = getenv("PTTS_DAC_RING_TAPS") ? atoi(getenv("PTTS_DAC_RING_TAPS")) : 7;   // bit 0: 7-tap, 1: 1x1, 2: transposed
    const int ring_min_u = getenv("PTTS_DAC_RING_MIN_U") ? atoi(getenv("PTTS_DAC_RING_MIN_U")) : 1024;   // tests: 1
    if (a.w2_units && (no_ring || !a.w_units)) return hipErrorInvalidValue;          // the fused unit exists on the ring kernel only
    if (!no_ring && a.w_units && (a.unit_tn == 96 || a.unit_tn == 192) && a.N % a.unit_tn == 0 && a.Cin % KC3 == 0 && a.Cout % 4 == 0 &&
        (a.U >= ring_min_u || a.w2_units) && (a.taps == 7 || a.taps == 2 || a.taps == 1) && ((ring_taps >> (a.taps == 7 ? 0 : a.taps)) & 1)) {
        const int span = (a.taps - 1) * (a.tstep >= 0 ? a.tstep : -a.tstep);
        constexpr int RS = 4;
        // 96 outputs x 64 rows per wave; few-tap layers (transposed, wide 1x1: the window is staged once per 1 - 2 units of MFMA
        // work) take 192 outputs x 32 rows per wave, which halves the Snake / split work per MFMA
        const int rw = a.unit_tn == 96 ? 64 : 32;
        const size_t lds = (size_t)(8 * rw + span) * 128 + (size_t)RS * a.unit_tn * 128;
        if (span <= h3_max_span(rw) && lds <= 160 * 1024) {
            dim3 grid(a.N / a.unit_tn, (a.U + 8 * rw - 1) / (8 * rw), a.B);
            if (a.w2_units) {     // fused residual unit: one output block holds every channel, dense tensors, distinct output
                if (a.taps != 7 || a.N != a.unit_tn || a.Cin != a.unit_tn || a.Cout != a.unit_tn || !a.alpha2 || !a.bias2 || a.res || a.in_bs ||
                    a.out_bs || a.out == a.in || a.stride != 1 || a.act_out)
                    return hipErrorInvalidValue;
                PTTS_CONV_PICK(a.unit_tn == 96 ? CONV_KIND_FUSED_96 : CONV_KIND_FUSED_192);
                PTTS_CONV_PACK_GUARD(CONV_KIND_FUSED_96);
                // C = 96: four-wave workgroups over 256 rows, ring of 3 + window = 76 KB: TWO on a CU, so that one's prologue, second
                // stage and epilogue (39 % of a CU slot as one eight-wave workgroup, profiles/r06_dac_workgroup_life.txt) run beside the
                // other's chunk loop.  PTTS_DAC_FUSED96_WAVES=8: the eight-wave workgroup over 512 rows (rounds 3-5; A/B and test)
                const char* f96 = getenv("PTTS_DAC_FUSED96_WAVES");
                if (a.unit_tn == 96 && span <= 64 && !(f96 && atoi(f96) == 8)) {
                    const size_t lds4 = (size_t)(256 + span) * 128 + (size_t)3 * 96 * 128;
                    dim3 grid4(1, (a.U + 255) / 256, a.B);
                    hipLaunchKernelGGL((conv_kernel_h3r<96, 64, 7, 3, true, 1, 4>), grid4, dim3(256), lds4, s, a);
                } else if (a.unit_tn == 96) hipLaunchKernelGGL((conv_kernel_h3r<96, 64, 7, RS, true>), grid, dim3(512), lds, s, a);
                else hipLaunchKernelGGL((conv_kernel_h3r<192, 32, 7, RS, true>), grid, dim3(512), lds, s, a);
                tally(a.unit_tn == 96 ? CONV_KIND_FUSED_96 : CONV_KIND_FUSED_192);
                return hipGetLastError();
            }
            if (a.unit_tn == 192 && a.taps == 1 && a.stride == 1 && !a.in_bs && !a.out_bs && a.act_out == 0 && a.Tin == a.Tout &&
                a.U == a.Tout && a.Cin <= 2048 && getenv("PTTS_DAC_NO_S1") == nullptr) {
                // wide 1x1 conv: activations straight into B fragments, no window
                const size_t lds_s1 = (size_t)4 * 192 * 128 + (size_t)2 * a.Cin * sizeof(float);
                PTTS_CONV_PICK(CONV_KIND_S1);
                PTTS_CONV_PACK_GUARD(CONV_KIND_S1);
                hipLaunchKernelGGL(conv_kernel_s1, grid, dim3(512), lds_s1, s, a);
                tally(CONV_KIND_S1);
                return hipGetLastError();
            }
            if (a.in_packed) {    // transposed conv on packed input: the LDS-DMA window kernel (conv_can_pack has checked the pair)
                if (a.unit_tn != 192 || a.taps != 2 || a.tstep != -1 || a.toff != 0 || a.stride < 2 || a.in_bs || a.out_bs || a.res || a.act_out ||
                    a.alpha || a.alpha_out || a.U != a.Tin + 1 || a.Cin % 8 != 0)
                    return hipErrorInvalidValue;
                PTTS_CONV_PICK(CONV_KIND_RING_192_T2);
                const size_t lds_t = (size_t)3 * 192 * 128 + (size_t)2 * 2 * 272 * 64;
                hipLaunchKernelGGL(convt_dma_kernel, grid, dim3(512), lds_t, s, a);
                tally(CONV_KIND_RING_192_T2);
                return hipGetLastError();
            }
            {
                const int kind = (a.unit_tn == 96 ? CONV_KIND_RING_96_T7 : CONV_KIND_RING_192_T7) + (a.taps == 7 ? 0 : (a.taps == 2 ? 1 : 2));
                PTTS_CONV_PICK(kind);
                PTTS_CONV_PACK_GUARD(kind);
                // a packing producer writes whole quads at (t, co): a strided (transposed) or tanh-ed output is never packed
                if (a.alpha_out && (a.stride != 1 || a.act_out)) return hipErrorInvalidValue;
            }
            if (a.unit_tn == 96) {
                if (a.taps == 7) hipLaunchKernelGGL((conv_kernel_h3r<96, 64, 7, RS, false>), grid, dim3(512), lds, s, a);
                else if (a.taps == 2) hipLaunchKernelGGL((conv_kernel_h3r<96, 64, 2, RS, false>), grid, dim3(512), lds, s, a);
                else hipLaunchKernelGGL((conv_kernel_h3r<96, 64, 1, RS, false>), grid, dim3(512), lds, s, a);
            } else {
                // 7 taps: 64 x 96 wave tiles (same workgroup tile, grid and LDS); PTTS_DAC_WAVE_TILE=32x192: the 32 x 192 instance (A/B, test)
                const char* wt_now = getenv("PTTS_DAC_WAVE_TILE");
                if (a.taps == 7 && !(wt_now && strcmp(wt_now, "32x192") == 0))
                    hipLaunchKernelGGL((conv_kernel_h3r<192, 64, 7, RS, false, 2>), grid, dim3(512), lds, s, a);
                else if (a.taps == 7) hipLaunchKernelGGL((conv_kernel_h3r<192, 32, 7, RS, false>), grid, dim3(512), lds, s, a);
                else if (a.taps == 2) hipLaunchKernelGGL((conv_kernel_h3r<192, 32, 2, RS, false>), grid, dim3(512), lds, s, a);
                else hipLaunchKernelGGL((conv_kernel_h3r<192, 32, 1, RS, false>), grid, dim3(512), lds, s, a);
            }
            tally((a.unit_tn == 96 ? CONV_KIND_RING_96_T7 : CONV_KIND_RING_192_T7) + (a.taps == 7 ? 0 : (a.taps == 2 ? 1 : 2)));
            return hipGetLastError();
        }
    }
    // the fused residual unit exists on the ring kernel only: a guard above that failed must not fall through to a kernel
    // that would silently drop the unit's 1x1 conv and residual
    if (a.w2_units) return hipErrorInvalidValue;
    if (a.w_hi && a.w_lo && a.Cin % KC3 == 0) {
        // TN = 96 when it divides N and fits LDS, else 64
        int TN = (a.N % 96 == 0 && conv_h3_lds(a, 96) <= 160 * 1024) ? 96 : 64;
        if (a.N % 64 != 0 && a.N % 96 != 0) TN = a.N <= 64 ? 64 : 96;
        const int span = (a.taps - 1) * (a.tstep >= 0 ? a.tstep : -a.tstep);
        // 64 rows per wave (512-row workgroups) where the launch is long enough and the window fits
        const char* tile = getenv("PTTS_DAC_TILE");          // "96x32" | "64x32" | "64x64" | "96x64": A/B and tests
        int RW = 32;
        if (tile) {
            int tn = 0, rw = 0;
            if (sscanf(tile, "%dx%d", &tn, &rw) == 2 && (tn == 64 || tn == 96) && (rw == 32 || rw == 64)) {
                if (a.N % tn == 0) TN = tn;
                RW = rw;
            }
        }
        if (RW == 64 && (span > h3_max_span(64) || conv_h3_lds(a, TN, 64) > 160 * 1024 || a.U < 1024)) RW = 32;
        const size_t lds3 = conv_h3_lds(a, TN, RW);
        if (lds3 <= 160 * 1024 && span <= h3_max_span(RW)) {
            const int tmw = 8 * RW;
            dim3 grid((a.N + TN - 1) / TN, (a.U + tmw - 1) / tmw, a.B);
            PTTS_CONV_PICK(CONV_KIND_H3);
            PTTS_CONV_PACK_GUARD(CONV_KIND_H3);
            if (TN == 64 && RW == 32) hipLaunchKernelGGL((conv_kernel_h3<64, 32>), grid, dim3(512), lds3, s, a);
            else if (TN == 96 && RW == 32) hipLaunchKernelGGL((conv_kernel_h3<96, 32>), grid, dim3(512), lds3, s, a);
            else if (TN == 64) hipLaunchKernelGGL((conv_kernel_h3<64, 64>), grid, dim3(512), lds3, s, a);
            else hipLaunchKernelGGL((conv_kernel_h3<96, 64>), grid, dim3(512), lds3, s, a);
            tally(CONV_KIND_H3);
            return hipGetLastError();
        }
    }
    const int lo = a.tstep >= 0 ? a.toff : a.toff + (a.taps - 1) * a.tstep;
    const int hi = a.tstep >= 0 ? a.toff + (a.taps - 1) * a.tstep : a.toff;
    const int win = TM + (hi - lo);
    const int TN = (a.N % 64 == 0) ? 64 : 96;
    const size_t lds = ((size_t)win * LDA + (size_t)a.taps * TN * LDA) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid((a.N + TN - 1) / TN, (a.U + TM - 1) / TM, a.B);
    PTTS_CONV_PICK(CONV_KIND_F32);
    PTTS_CONV_PACK_GUARD(CONV_KIND_F32);
    if (TN == 64) hipLaunchKernelGGL(conv_kernel<64>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv_kernel<96>, grid, dim3(256), lds, s, a);
    tally(CONV_KIND_F32);
    return hipGetLastError();
}

int conv_kind_of(const ConvArgs& a) {
    int kind = -1;
    g_conv_plan = &kind;
#if defined(PTTS_CONV_STAMPS) || defined(PTTS_CONV_LIFE)
    (void)conv_launch_inner(a, nullptr);
#else
    (void)conv_launch(a, nullptr);
#endif
    g_conv_plan = nullptr;
    return kind;
}

bool conv_can_pack(const ConvArgs& producer, const ConvArgs& consumer) {
    if (getenv("PTTS_DAC_NO_PRESPLIT")) return false;             // A/B and the bit-identity test
    if (!consumer.alpha || consumer.taps != 2 || consumer.Cin != producer.Cout || producer.stride != 1 || producer.act_out ||
        producer.out_bs || consumer.in_bs || consumer.out_bs || producer.Cout % 8 != 0 || consumer.unit_tn != 192 || consumer.tstep != -1 ||
        consumer.toff != 0 || consumer.stride < 2 || consumer.res || consumer.act_out || consumer.U != consumer.Tin + 1)
        return false;
    return conv_kind_packs(conv_kind_of(producer)) && conv_kind_unpacks(conv_kind_of(consumer));
}

hipError_t dac_latent_launch(const long long* codes, const float* table, float* z, int B, int K, int F, int C, int ncodes,
                             hipStream_t s) {
    hipLaunchKernelGGL(dac_latent_kernel, dim3(F, B), dim3(256), 0, s, codes, table, z, K, F, C, ncodes);
    return hipGetLastError();
}

hipError_t dac_table_launch(const float* codebook, const float* proj_w, const float* proj_b, float* table, int ncodes, int dim,
                            int C, hipStream_t s) {
    hipLaunchKernelGGL(dac_table_kernel, dim3(ncodes), dim3(256), 0, s, codebook, proj_w, proj_b, table, ncodes, dim, C);
    return hipGetLastError();
}

hipError_t conv_pack_launch(const float* w, float* out, int Cout, int Cin, int k, int transposed, int stride, int N,
                            hipStream_t s) {
    size_t total = (size_t)(transposed ? 2 : k) * N * Cin;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out, Cout, Cin, k,
                       transposed, stride, N);
    return hipGetLastError();
}

hipError_t conv_units2_launch(const void* hi, const void* lo, void* out, int C, hipStream_t s) {
    if (C % 32 != 0) return hipErrorInvalidValue;
    const size_t total = (size_t)(C / 32) * 2 * (C / 16) * 64;
    hipLaunchKernelGGL(conv_units2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const _Float16*)hi, (const _Float16*)lo,
                       (uint4*)out, C);
    return hipGetLastError();
}

hipError_t conv_units_launch(const void* hi, const void* lo, void* out, int taps, int N, int Cin, int TN, hipStream_t s) {
    if (N % TN != 0 || Cin % 32 != 0) return hipErrorInvalidValue;
    const size_t total = (size_t)(N / TN) * (Cin / 32) * taps * TN * 8;
    hipLaunchKernelGGL(conv_units_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const _Float16*)hi, (const _Float16*)lo,
                       (uint4*)out, taps, N, Cin, TN);
    return hipGetLastError();
}

hipError_t split_halves_launch(const float* src, void* hi, void* lo, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(split_halves_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, (_Float16*)hi, (_Float16*)lo, n);
    return hipGetLastError();
}

hipError_t enc_first_conv_launch(const float* audio, const float* w, const float* bias, float* out, int B, int T, int C,
                                 size_t out_bs, hipStream_t s) {
    dim3 grid((unsigned)(((size_t)T * C + 255) / 256), B);
    hipLaunchKernelGGL(enc_first_conv_kernel, grid, dim3(256), 0, s, audio, w, bias, out, T, C, out_bs ? out_bs : (size_t)T * C);
    return hipGetLastError();
}
hipError_t zero_tail_launch(float* buf, int B, size_t bs, size_t from, size_t to, hipStream_t s) {
    if (to <= from) return hipSuccess;
    hipLaunchKernelGGL(zero_tail_kernel, dim3((unsigned)((to - from + 255) / 256), B), dim3(256), 0, s, buf, bs, from, to);
    return hipGetLastError();
}
hipError_t conv_pack_strided_launch(const float* w, float* out, int N, int C, int stride, hipStream_t s) {
    const size_t total = (size_t)3 * N * stride * C;
    hipLaunchKernelGGL(conv_pack_strided_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out, N, C, stride);
    return hipGetLastError();
}
hipError_t tile_alpha_launch(const float* alpha, float* out, int C, int reps, hipStream_t s) {
    hipLaunchKernelGGL(tile_alpha_kernel, dim3((C * reps + 255) / 256), dim3(256), 0, s, alpha, out, C, reps);
    return hipGetLastError();
}
hipError_t cb_normalize_launch(const float* cb, float* cbn, float* cbn2, int entries, int dim, hipStream_t s) {
    hipLaunchKernelGGL(cb_normalize_kernel, dim3((entries + 255) / 256), dim3(256), 0, s, cb, cbn, cbn2, entries, dim);
    return hipGetLastError();
}
hipError_t rvq_encode_launch(const RvqArgs& a, hipStream_t s) {
    if (a.dim > 16) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rvq_encode_kernel, dim3(a.F, a.B), dim3(256), (size_t)a.D * sizeof(float), s, a);
    return hipGetLastError();
}
