// Host-side interface of fused.hip (internal): QKV projection + self-attention of a decode layer as ONE launch.
#pragma once
#include "attn.h"
#include "gemm.h"

// arrival tags: [layers][m-tiles][heads][32 ints] (12 used: q / k / v x four 16-feature blocks of the head), zeroed at every prefill
inline size_t fused_arrive_ints(int layers, int mtiles, int heads) { return (size_t)layers * mtiles * heads * 32; }

// true when the pair (QKV projection `g`, decode self-attention `a`) of a layer can run as the fused launch: folded-LayerNorm QKV
// instance with an exact-fit tile burst (K / KSTEP = 8 x 4 or 8 x 8), identity page table, no look-ahead QKV, no rotary variant
bool fused_qkv_attn_supported(const GemmArgs& g, const AttnArgs& a, int dtype);
// ev_start / ev_stop as in attn_decode_launch (bound to the dispatch)
// arrive_prod: test hook (PTTS_FUSED_TEST_BREAK=1 at engine creation) - the projection half stores its tags THERE instead of in `arrive`, so the
// attention half never sees them: its bounded, sticky time-out and the error path of the call can be tested
hipError_t fused_qkv_attn_launch(const GemmArgs& g, const AttnArgs& a, int M, int heads, int dtype, int* arrive, int* err,
                                 hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, int* arrive_prod = nullptr);
