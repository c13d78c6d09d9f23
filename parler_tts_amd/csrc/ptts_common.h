// Shared device/host definitions for the gfx950 (CDNA4, wave64) Parler-TTS generate() engine.
//
// Memory layouts (all owned by the library, see DESIGN.md "Data layout in HBM"):
//
//  * packed weights  W[N][K] (torch Linear, out x in) are re-tiled once at load into MFMA A-operand
//    order: tile (n/16, k/KSTEP) is 1 KiB, stored lane-major, so one wave64 `global_load_dwordx4`
//    fetches exactly one tile, fully coalesced:
//        bf16: KSTEP=32, lane l element j (0..7)  = W[16*nb + (l&15)][32*ks + 8*(l>>4) + j]   (v_mfma_f32_16x16x32_bf16)
//        f32 : KSTEP=16, lane l element m (0..3)  = W[16*nb + (l&15)][16*ks + 4*m + (l>>4)]   (4 x v_mfma_f32_16x16x4_f32)
//  * activation fragments  X[32 rows][K] of one 32-row m-tile are kept in MFMA B-operand order
//    ("frag-major"), half h = row/16:
//        bf16: lane l element j = X[16*h + (l&15)][32*ks + 8*(l>>4) + j]
//        f32 : lane l element m = X[16*h + (l&15)][16*ks + 4*m + (l>>4)]
//    tile (ks, h) is again 1 KiB lane-major: LDS reads are lane-linear `ds_read_b128` (conflict free).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define PTTS_HEAD_DIM 64
#define PTTS_PAGE_TOKENS 64

// ---- precision policies -------------------------------------------------------------------------
struct PBF16 {
    typedef uint16_t elem_t;                 // storage element
    static constexpr int KSTEP = 32;         // k per 1 KiB tile
    static constexpr int EPL = 8;            // elements per lane per tile (16 bytes)
    static constexpr bool IS_BF16 = true;
};
struct PF32 {
    typedef float elem_t;
    static constexpr int KSTEP = 16;
    static constexpr int EPL = 4;
    static constexpr bool IS_BF16 = false;
};

__host__ __device__ inline float bf16_bits_to_float(uint16_t v) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)v) << 16;
    return c.f;
}

__device__ inline uint16_t float_to_bf16_bits(float f) {
    // plain cast -> v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserving) on gfx950
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

template <typename P> __device__ inline typename P::elem_t to_elem(float f);
template <> __device__ inline uint16_t to_elem<PBF16>(float f) { return float_to_bf16_bits(f); }
template <> __device__ inline float to_elem<PF32>(float f) { return f; }

template <typename P> __host__ __device__ inline float from_elem(typename P::elem_t v);
template <> __host__ __device__ inline float from_elem<PBF16>(uint16_t v) { return bf16_bits_to_float(v); }
template <> __host__ __device__ inline float from_elem<PF32>(float v) { return v; }

// ---- fragment-major offsets (in elements) ----------------------------------------------------------
// offset of activation X[row r (0..31)][k] inside one m-tile's frag buffer
template <typename P> __host__ __device__ inline size_t frag_off(int r, int k);
template <> __host__ __device__ inline size_t frag_off<PBF16>(int r, int k) {
    return ((size_t)((k >> 5) * 2 + (r >> 4)) << 9) + (size_t)((((r & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7));
}
template <> __host__ __device__ inline size_t frag_off<PF32>(int r, int k) {
    return ((size_t)((k >> 4) * 2 + (r >> 4)) << 8) + (size_t)((((r & 15) + 16 * (k & 3)) << 2) + ((k & 15) >> 2));
}
// offset of weight W[n][k] inside the packed matrix (K = in_features)
template <typename P> __host__ __device__ inline size_t wpack_off(int n, int k, int K);
template <> __host__ __device__ inline size_t wpack_off<PBF16>(int n, int k, int K) {
    return ((size_t)(n >> 4) * (size_t)(K >> 5) + (size_t)(k >> 5)) * 512 +
           (size_t)((((n & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7));
}
template <> __host__ __device__ inline size_t wpack_off<PF32>(int n, int k, int K) {
    return ((size_t)(n >> 4) * (size_t)(K >> 4) + (size_t)(k >> 4)) * 256 +
           (size_t)((((n & 15) + 16 * (k & 3)) << 2) + ((k & 15) >> 2));
}

// ---- paged KV pool -------------------------------------------------------------------------------------
// One pool = [L][2 (K,V)][H][n_pages][64 tok][64 dim]: a page id names 64 tokens of ONE utterance in every
// (layer, K/V, head) slab.  One attention launch (one layer) then reads a dense region, and a workgroup
// (one utterance, one head) whose page ids are consecutive streams ONE contiguous run of 8 KiB pages
// (measured: [page][L][2][H] 37.2 us per launch, [L][2][page][H] 34.0 us, this layout see DESIGN.md).
struct KVPool {
    void* base;            // element type = cache dtype
    const int* block_table;  // [B][max_blocks] page ids; NULL = the identity table (utterance b owns pages b*max_blocks ...):
                             // the engine passes NULL while its table IS the identity, which takes the dependent table read
                             // (and, in the decode kernel, the LDS hand-over behind it) out of every attention launch
    int max_blocks;
    int H;
    int L;
    int n_pages;           // pages in the pool
    __host__ __device__ size_t page_elems() const { return (size_t)PTTS_PAGE_TOKENS * PTTS_HEAD_DIM; }   // one page of one (layer, kv, head)
    __host__ __device__ size_t head_elems() const { return (size_t)n_pages * page_elems(); }             // one (layer, kv, head) slab
    __host__ __device__ size_t kv_elems() const { return (size_t)H * head_elems(); }                     // one (layer, kv) slab
    __device__ size_t elem_off(int b, int layer, int kv, int h, int tok) const {
        int page = block_table ? block_table[b * max_blocks + (tok >> 6)] : b * max_blocks + (tok >> 6);
        return ((size_t)((layer * 2 + kv) * H + h) * n_pages + page) * page_elems() + (size_t)(tok & 63) * PTTS_HEAD_DIM;
    }
};

// ---- on-device loop state (one per engine) -----------------------------------------------------------------
struct StepState {
    int pos;           // number of cached self-attention positions (= position id of the token being fed)
    int cur_len;       // columns of raw_ids already filled
    int prompt_len;    // P
    int n_unfinished;  // rows still unfinished after the last select
    int done_len;      // cur_len at which every row was finished (0 = not yet)
    int step_count;    // decode steps executed since prefill
    unsigned int philox_offset;
    int pad_;
};

// ---- loop state read through the scalar cache -----------------------------------------------------------------------
// StepState.pos and the finished-row flags are written by the step's LAST kernels and only read by the following launches.
// Read through a constant-address-space pointer they become scalar loads (s_load_dword: the scalar cache starts cold at every
// launch, so the value is the current one), which neither wait for nor delay the vector-memory stream - a plain `state->pos`
// compiles to a global_load + s_waitcnt vmcnt(0) in front of everything queued behind it.
//
// INVARIANT (every call site repeats it in one line): address space 4 tells LLVM the word is CONSTANT for the kernel, so a
// kernel may read a word this way only if NO launch of that kernel writes it - neither the reading workgroup nor any other
// workgroup of the same dispatch - and the value was written by an EARLIER dispatch on the same stream (the scalar cache is
// invalidated at every dispatch boundary, not inside one).  A fused kernel that both advances the loop state and reads it
// (the reverted merged tail of DESIGN_HISTORY.md) must use plain loads for those words.  The loop state is TWO StepState words
// apart for exactly this reason: state[0] is what a step's kernels read; state[1] holds the counters (pos, cur_len, step_count)
// the last selection left for the next step.  Writers today: select_kernel (state[1], tok_in, unfinished - it reads state[0]),
// advance_kernel and the extra workgroup of embed_codebooks_kernel (state[0] - they read state[1], and so do that launch's
// embedding workgroups), the prefill kernels (both), build_bias_kernel (bias_flag), the host (block_table); readers: the
// attention, gemm-append, embed and select kernels - never a launch that writes the word it reads.  `make EXTRA=-DPTTS_PLAIN_STATE_LOADS` (into a scratch copy of the objects)
// builds the same library with ordinary global loads for an A/B of ids (tests need no change).
typedef const int __attribute__((address_space(4))) ptts_cint;
typedef const float __attribute__((address_space(4))) ptts_cfloat;
#ifdef PTTS_PLAIN_STATE_LOADS
__device__ inline int scalar_load_int(const int* p) { return __builtin_nontemporal_load(p); }
__device__ inline float scalar_load_float(const float* p) { return __builtin_nontemporal_load(p); }
#else
__device__ inline int scalar_load_int(const int* p) { return *reinterpret_cast<ptts_cint*>(reinterpret_cast<uintptr_t>(p)); }
__device__ inline float scalar_load_float(const float* p) { return *reinterpret_cast<ptts_cfloat*>(reinterpret_cast<uintptr_t>(p)); }
#endif

// ---- K-split hand-over stores --------------------------------------------------------------------------------
// MFMA accumulators go to LDS through PLAIN C++ stores, never through an `asm` statement: hipcc pads the gfx950
// "XDL write VGPR -> LDS/VMEM/VALU read" hazard (8 wait states behind v_mfma_f32_16x16x32_bf16, 10 behind
// v_mfma_f32_16x16x4_f32 - what it emits for a bare MFMA -> ds_write pair) only for instructions it models, and an
// inline-asm ds_write is opaque to its hazard recognizer (cdna_hip_programming.md section 5.7 item 2).  Round 1
// issued these stores as `asm("ds_write_b128 ...; s_waitcnt lgkmcnt(0)")` straight behind the last MFMA of the
// K loop; tools/check_mfma_asm_hazard.py found 35 (MFMA, asm store) pairs in 9 instances closer than the table
// allows (8-10 wait states on the shortest path), i.e. a store that can read an accumulator register the matrix
// pipe has not finished retiring: one register, its last-written lanes - the signature DESIGN_HISTORY.md section A records.
// The lint runs in the CPU test suite and fails on any inline-asm reader of an MFMA result.
__device__ inline void lds_store_b128(void* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ inline void lds_store_b32(void* p, float v) { *reinterpret_cast<float*>(p) = v; }

// ---- wave64 all-reduce helpers on DPP (VALU data-parallel primitives: no LDS crossbar round trips) ----
// gfx9 DPP controls: quad_perm 0x00-0xFF, row_ror:n 0x120+n, row_mirror 0x140, row_half_mirror 0x141.
template <int CTRL> __device__ inline float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ inline float lane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// after these every lane of a group holds the group's sum
__device__ inline float sum_quad(float v) { v += dpp_f<0xB1>(v); v += dpp_f<0x4E>(v); return v; }   // 4 lanes
__device__ inline float sum_oct(float v) { v = sum_quad(v); v += dpp_f<0x141>(v); return v; }        // 8 lanes
__device__ inline float sum_row16(float v) { v = sum_oct(v); v += dpp_f<0x140>(v); return v; }       // 16 lanes
__device__ inline float wave_sum(float v) {
    v = sum_row16(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ inline float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}

// ---- diagnostic build only (-DPTTS_STAMPS, `make STAMPS=1` -> libparler_tts_hip_stamps.so; never the product library) ----
// In-kernel clock stamps of wave 0 of every workgroup: where a decode-step launch spends its time (s_memtime: shader cycles)
// and where it sits relative to its neighbours in the stream (s_memrealtime: one 100 MHz counter for the whole device).
// Record of workgroup g of the launch in slot s: stamps[(s * PTTS_STAMP_WGS + g) * 8 + i], i = 0 realtime at entry, 1 .. 6
// memtime at the kernel's named points (1 = entry), 7 realtime at exit.  Values leave only through this buffer.
#define PTTS_STAMP_WGS 1024
#define PTTS_STAMP_SLOTS 256
#ifdef PTTS_STAMPS
struct StampRef { unsigned long long* buf; int slot; };
__device__ inline unsigned long long stamp_cycles() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ inline unsigned long long stamp_realtime() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ inline void stamp_store(const StampRef& r, const unsigned long long (&v)[8]) {
    if (!r.buf || r.slot < 0 || r.slot >= PTTS_STAMP_SLOTS) return;
    const unsigned g = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (threadIdx.x != 0 || g >= PTTS_STAMP_WGS) return;
    unsigned long long* d = r.buf + ((size_t)r.slot * PTTS_STAMP_WGS + g) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = v[i];
}
#endif
