// Small per-step kernels: token/prompt embedding + sinusoid positions, token selection with the
// delay-pattern shift and EOS latch, loop-state advance, additive attention masks.
#include "ptts_common.h"
#include "misc.h"

namespace {

// ---- embedding: x[b] = ((((0 + E0[t0]) + E1[t1]) + ...) + E_{K-1}[t_{K-1}]) + pos_table[pos]
// strictly left-to-right like Python's sum() at M:836, then the position row (M:884-886).
template <typename P> struct Vec4;
template <> struct Vec4<PBF16> {
    static __device__ inline void load(const uint16_t* p, float* f) {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        f[0] = __builtin_bit_cast(float, v.x << 16); f[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
        f[2] = __builtin_bit_cast(float, v.y << 16); f[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
    }
};
template <> struct Vec4<PF32> {
    static __device__ inline void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
};

// The loop state is settled LAZILY (round 6): the selection leaves the next step's counters in state[1] and this step's EOS
// latches in `unfinished`; state[0] - what every other kernel of a step reads - is brought up to date from them either by
// advance_kernel (eager steps, and once behind the last replay of a ptts_decode_steps call) or by the one workgroup past the
// utterances of the NEXT step's embedding launch, which runs beside the row gathers instead of as a launch of its own between
// two steps.  Same assignments in both places, and idempotent: settling twice, or with nothing selected in between, changes nothing.
__device__ inline void settle_state(StepState* s, const int* unfinished, int rows) {
    __shared__ int part[4];
    int n = 0;
    for (int r = threadIdx.x; r < rows; r += 256) n += unfinished[r] ? 1 : 0;
    n = (int)wave_sum((float)n);              // rows <= 2^24: exact in fp32
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int unf_total = part[0] + part[1] + part[2] + part[3];
        // scalar_load invariant: neither kernel that settles the state writes state[1] (select_kernel and the prefill kernels do)
        const int pos = scalar_load_int(&s[1].pos), cur = scalar_load_int(&s[1].cur_len), steps = scalar_load_int(&s[1].step_count);
        s[0].pos = pos;
        s[0].cur_len = cur;
        s[0].step_count = steps;
        s[0].n_unfinished = unf_total;
        if (unf_total == 0 && s[0].done_len == 0) s[0].done_len = cur;
    }
}

template <typename P>
__global__ __launch_bounds__(256) void embed_codebooks_kernel(EmbedArgs a) {
    typedef typename P::elem_t elem_t;
    const int b = blockIdx.x;
    if (b == (int)gridDim.x - 1 && a.unfinished) {   // the extra workgroup (uniform branch): the loop state of this step
        settle_state(a.state, a.unfinished, a.n_latch);
        return;
    }
    // the step's position and this utterance's K token ids are uniform per workgroup: scalar-cache reads (no vector round trip,
    // no LDS hand-over, no barrier in front of the row gathers)
    // scalar_load invariant (ptts_common.h): the embed kernel writes neither state[1] nor tok_in (select does, a launch earlier);
    // state[0] is written by this launch's extra workgroup, so the position comes from state[1], which holds the same value
    const int pos = scalar_load_int(&a.state[1].pos);
    int toks[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) toks[k] = k < a.K ? scalar_load_int(a.tok_in + b * a.K + k) : 0;
    const elem_t* tab = reinterpret_cast<const elem_t*>(a.tables);
    const elem_t* ptab = reinterpret_cast<const elem_t*>(a.pos_table) + (size_t)pos * a.D;
    // 4 features per thread (D % 4 == 0); the K row gathers are independent vector loads issued together,
    // the sum order stays strictly left to right per feature
    for (int d = 4 * threadIdx.x; d < a.D; d += 4 * blockDim.x) {
        float e[16][4];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < a.K) Vec4<P>::load(tab + ((size_t)k * a.rows + toks[k]) * a.D + d, e[k]);
        float pe[4];
        Vec4<P>::load(ptab + d, pe);
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < a.K) { s[0] = s[0] + e[k][0]; s[1] = s[1] + e[k][1]; s[2] = s[2] + e[k][2]; s[3] = s[3] + e[k][3]; }
        for (int k = 16; k < a.K; ++k) {
            float t[4];
            Vec4<P>::load(tab + ((size_t)k * a.rows + scalar_load_int(a.tok_in + b * a.K + k)) * a.D + d, t);
            s[0] = s[0] + t[0]; s[1] = s[1] + t[1]; s[2] = s[2] + t[2]; s[3] = s[3] + t[3];
        }
        const float o[4] = {s[0] + pe[0], s[1] + pe[1], s[2] + pe[2], s[3] + pe[3]};
        *reinterpret_cast<float4*>(a.x + (size_t)b * a.D + d) = make_float4(o[0], o[1], o[2], o[3]);
        if (a.xfrag) {
            elem_t* xf = reinterpret_cast<elem_t*>(a.xfrag) + (size_t)(b >> 5) * 32 * a.D;
#pragma unroll
            for (int r = 0; r < 4; ++r) xf[frag_off<P>(b & 31, d + r)] = to_elem<P>(o[r]);
        }
    }
}

// prompt positions: x[b] = embed_prompts[prompt_ids[b][pos]] + pos_table[pos]   (M:2243-2245, M:839-840, M:886)
template <typename P>
__global__ void embed_prompt_kernel(EmbedArgs a) {
    typedef typename P::elem_t elem_t;
    const int b = blockIdx.x;
    const int pos = a.state->pos;
    const elem_t* ptab = reinterpret_cast<const elem_t*>(a.pos_table) + (size_t)pos * a.D;
    elem_t* xf = a.xfrag ? reinterpret_cast<elem_t*>(a.xfrag) + (size_t)(b >> 5) * 32 * a.D : nullptr;
    const float* pre = a.prompt_embeds ? a.prompt_embeds + ((size_t)b * a.P + pos) * a.D : nullptr;
    const elem_t* row = pre ? nullptr
                            : reinterpret_cast<const elem_t*>(a.prompt_table) + (size_t)a.prompt_ids[(size_t)b * a.P + pos] * a.D;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const float v = (pre ? pre[d] : from_elem<P>(row[d])) + from_elem<P>(ptab[d]);
        a.x[(size_t)b * a.D + d] = v;
        if (xf) xf[frag_off<P>(b & 31, d)] = to_elem<P>(v);
    }
}

__global__ void advance_prompt_kernel(StepState* s) { const int p = s[0].pos + 1; s[0].pos = p; s[1].pos = p; }

// loop state of a new call, written on the device (a hipMemcpyAsync from the host's stack needed a stream synchronisation behind
// it - three of them per prefill, each one an idle device and a cold launch queue in front of the next phase)
__global__ void reset_state_kernel(StepState* s, StepState v, int* unfinished, int n_flags) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { s[0] = v; s[1] = v; }
    if (unfinished)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_flags; i += gridDim.x * blockDim.x) unfinished[i] = 1;
}

// batched prefill, prompt rows: one block per (utterance b, chunk position i)
template <typename P>
__global__ void embed_prompt_rows_kernel(EmbedArgs a, int npos, int pos0) {
    typedef typename P::elem_t elem_t;
    const int b = blockIdx.x / npos, i = blockIdx.x - b * npos, pos = pos0 + i;
    const int row = blockIdx.x;
    const elem_t* ptab = reinterpret_cast<const elem_t*>(a.pos_table) + (size_t)pos * a.D;
    elem_t* xf = a.xfrag ? reinterpret_cast<elem_t*>(a.xfrag) + (size_t)(row >> 5) * 32 * a.D : nullptr;
    const float* pre = a.prompt_embeds ? a.prompt_embeds + ((size_t)b * a.P + pos) * a.D : nullptr;
    const elem_t* prow = pre ? nullptr
                             : reinterpret_cast<const elem_t*>(a.prompt_table) + (size_t)a.prompt_ids[(size_t)b * a.P + pos] * a.D;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const float v = (pre ? pre[d] : from_elem<P>(prow[d])) + from_elem<P>(ptab[d]);
        a.x[(size_t)row * a.D + d] = v;
        if (xf) xf[frag_off<P>(row & 31, d)] = to_elem<P>(v);
    }
}

// batched prefill, decoder-prefix rows: the masked ids of column col0 + i (apply_delay_pattern_mask, M:68-74) summed over the
// codebooks strictly left to right (M:836) + the position row
template <typename P>
__global__ void embed_prefix_rows_kernel(EmbedArgs a, int npos, int pos0, int col0, const long long* raw_ids,
                                         const long long* pattern, int rows_src, int L) {
    typedef typename P::elem_t elem_t;
    __shared__ int toks[32];
    const int b = blockIdx.x / npos, i = blockIdx.x - b * npos;
    const int row = blockIdx.x, col = col0 + i;
    if ((int)threadIdx.x < a.K) {
        const int src = (b * a.K + (int)threadIdx.x) % rows_src;      // guidance: the null half repeats the ids (M:2112)
        const long long f = pattern[(size_t)src * L + col];
        toks[threadIdx.x] = (int)(f == -1 ? raw_ids[(size_t)src * L + col] : f);
    }
    __syncthreads();
    const elem_t* tab = reinterpret_cast<const elem_t*>(a.tables);
    const elem_t* ptab = reinterpret_cast<const elem_t*>(a.pos_table) + (size_t)(pos0 + i) * a.D;
    elem_t* xf = a.xfrag ? reinterpret_cast<elem_t*>(a.xfrag) + (size_t)(row >> 5) * 32 * a.D : nullptr;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < a.K; ++k) s = s + from_elem<P>(tab[((size_t)k * a.rows + toks[k]) * a.D + d]);
        const float v = s + from_elem<P>(ptab[d]);
        a.x[(size_t)row * a.D + d] = v;
        if (xf) xf[frag_off<P>(row & 31, d)] = to_elem<P>(v);
    }
}

// ---- Philox4x32-10 counter RNG (one draw per (row, step)) ----
__device__ inline void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ inline float philox_uniform(uint64_t seed, uint32_t row, uint32_t step) {
    uint32_t c0 = row, c1 = step, c2 = 0x9E3779B9u, c3 = 0xBB67AE85u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c0 >> 8) * (1.0f / 16777216.0f);  // [0, 1)
}

__device__ inline uint32_t float_key(float f) {   // order-preserving map to unsigned
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- token selection: one workgroup per (utterance, codebook) row ----
// greedy: first maximal index (torch.argmax).  sample: scores / temperature -> keep scores >= k-th
// largest (ties kept) -> softmax -> one inverse-CDF draw.  Then, as the third-party loop does:
// nxt = tok*unf + pad*(1-unf); raw_ids[:, cur_len] = nxt; unf &= (nxt != eos); and the token fed to
// the next step is the delay pattern's value where it is not -1 (apply_delay_pattern_mask, M:68-74).
__global__ __launch_bounds__(256) void select_kernel(SelectArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sv = reinterpret_cast<float*>(smem);          // [V] scores
    __shared__ float wv[4];
    __shared__ int wi[4];
    __shared__ unsigned hist[256];
    __shared__ unsigned scan[256];
    __shared__ float fscan[256];
    __shared__ unsigned sel_prefix, sel_remaining;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int V = a.vocab;          // valid ids; rows are a.V (= padded stride) apart
    // scalar_load invariant: select_kernel never writes StepState (advance_kernel does, in the launch behind it); the words it DOES
    // write - tok_in, unfinished - it reads with plain loads
    const int cur = scalar_load_int(&a.state->cur_len);      // scalar cache: the logits loads below do not queue behind it
    if (cur >= a.L) return;
    if (row == 0 && tid == 0) {
        // the next step's counters: state[1] is read by no workgroup of this launch, state[0] (read above) is written by none
        a.state[1].pos = scalar_load_int(&a.state->pos) + 1;
        a.state[1].cur_len = cur + 1;
        a.state[1].step_count = scalar_load_int(&a.state->step_count) + 1;
    }
    // what thread 0 needs behind the choice - this row's latch, its delay-pattern value, the forced ids of the tests - does not
    // depend on the scores: requested here, beside the logits, instead of as two more dependent round trips at the kernel's end
    // (the row's own words only: no other workgroup of this launch writes them)
    int unf_in = 0;
    long long forced_in = -1, force_in = -1, fseq_in = -1;
    if (tid == 0) {
        unf_in = a.unfinished[row];
        forced_in = a.pattern[(size_t)row * a.L + cur];
        if (a.force) force_in = a.force[row];
        if (a.force_seq) fseq_in = a.force_seq[(size_t)row * a.L + cur];
    }
    const float* lg = a.logits + (size_t)row * a.V;
    const float* ug = a.guidance > 1.0f ? a.logits + (size_t)(row + a.rows) * a.V : nullptr;
    float gbest = -INFINITY; int gbi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        float s = lg[v];
        if (ug) { float u = ug[v]; s = u + (s - u) * a.guidance; }
        // MinLength / MinNewTokensLength processors run on the raw scores, before the warpers (third-party loop order)
        if (v == a.eos && cur < a.eos_free_until) s = -INFINITY;
        if (a.do_sample && a.temperature != 1.0f) s = s / a.temperature;
        sv[v] = s;
        if (s > gbest || (s == gbest && v < gbi)) { gbest = s; gbi = v; }   // greedy: the lane's candidate straight from the registers
    }
    __syncthreads();

    int choice = 0;
    if (!a.do_sample) {
        float best = gbest; int bi = gbi;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            float ob = __shfl_xor(best, o); int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { wv[w] = best; wi[w] = bi; }
        __syncthreads();
        best = wv[0]; bi = wi[0];
        for (int i = 1; i < 4; ++i) if (wv[i] > best || (wv[i] == best && wi[i] < bi)) { best = wv[i]; bi = wi[i]; }
        choice = bi;
        if (a.probs) for (int v = tid; v < V; v += 256) a.probs[(size_t)row * a.V + v] = sv[v];
    } else {
        // ---- k-th largest by 4-pass radix select over order-preserving keys ----
        float thr = -INFINITY;
        if (a.top_k > 0 && a.top_k < V) {
            if (tid == 0) { sel_prefix = 0; sel_remaining = (unsigned)a.top_k; }
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                hist[tid] = 0;
                __syncthreads();
                const unsigned prefix = sel_prefix;
                const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
                for (int v = tid; v < V; v += 256) {
                    unsigned k = float_key(sv[v]);
                    if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
                }
                __syncthreads();
                // the bin that holds the rem-th largest key: suffix counts S[b] = sum_{b' >= b} hist[b'] by a
                // Hillis-Steele scan over the 256 bins (one per thread), then the unique b with S[b] >= rem > S[b+1]
                // (a serial walk by one thread cost ~10 us per pass: 256 dependent LDS reads)
                const unsigned rem = sel_remaining;
                unsigned sfx = hist[tid];
                for (int off = 1; off < 256; off <<= 1) {
                    scan[tid] = sfx;
                    __syncthreads();
                    if (tid + off < 256) sfx += scan[tid + off];
                    __syncthreads();
                }
                scan[tid] = sfx;
                __syncthreads();
                const unsigned above = tid < 255 ? scan[tid + 1] : 0u;     // keys in strictly higher bins
                if ((sfx >= rem && above < rem) || (tid == 0 && sfx < rem)) {   // (tid 0 fallback: fewer keys than rem)
                    sel_prefix = prefix | ((unsigned)tid << shift);
                    sel_remaining = rem - (sfx >= rem ? above : 0u);
                }
                __syncthreads();
            }
            unsigned key = sel_prefix;   // key of the k-th largest score
            unsigned u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
            thr = __builtin_bit_cast(float, u);
        }
        // ---- masked softmax ----
        float m = -INFINITY;
        for (int v = tid; v < V; v += 256) { float s = sv[v]; if (s >= thr) m = fmaxf(m, s); }
        m = wave_max(m);
        if (lane == 0) wv[w] = m;
        __syncthreads();
        m = fmaxf(fmaxf(wv[0], wv[1]), fmaxf(wv[2], wv[3]));
        __syncthreads();
        float sum = 0.f;
        for (int v = tid; v < V; v += 256) {
            float s = sv[v];
            float e = (s >= thr) ? expf(s - m) : 0.f;
            sv[v] = e; sum += e;
        }
        sum = wave_sum(sum);
        if (lane == 0) wv[w] = sum;
        __syncthreads();
        sum = (wv[0] + wv[1]) + (wv[2] + wv[3]);
        for (int v = tid; v < V; v += 256) {
            float p = sv[v] / sum;
            sv[v] = p;
            if (a.probs) a.probs[(size_t)row * a.V + v] = p;
        }
        __syncthreads();
        // ---- top-p (TopPLogitsWarper, applied after top-k): in ascending order of probability, drop the
        //      tokens whose cumulative probability is <= 1 - top_p; the most probable token always stays ----
        if (a.top_p < 1.0f) {
            float* keep = sv + a.V;                     // second LDS row
            float pmax = 0.f;
            for (int v = tid; v < V; v += 256) pmax = fmaxf(pmax, sv[v]);
            pmax = wave_max(pmax);
            if (lane == 0) wv[w] = pmax;
            __syncthreads();
            pmax = fmaxf(fmaxf(wv[0], wv[1]), fmaxf(wv[2], wv[3]));
            __syncthreads();
            // The rule keeps v iff (mass of all smaller probabilities) + (mass of the equal ones with index <= v) exceeds
            // 1 - top_p, i.e. everything above a threshold value tau, nothing below it, and the equal ones by rank.
            // tau and the mass below it come from a 4-pass radix descent over the IEEE bits of the (non-negative)
            // probabilities: thread b sums bin b in index order, bins are prefix-scanned in a fixed pattern - deterministic,
            // O(4V) per thread instead of the O(V^2 / 256) pairwise form (460 us per step at V = 1088).
            const float target = 1.0f - a.top_p;
            __shared__ unsigned tp_prefix;
            __shared__ float tp_below;
            __shared__ int tp_bin[2];
            // compact list of the indices with non-zero probability (ascending): after top-k only k of the V entries
            // survive, and every loop below runs over the survivors only
            int* nz = reinterpret_cast<int*>(sv + 2 * a.V);
            int nnz;
            {
                const int CH = (V + 255) / 256;
                const int v0 = tid * CH, v1 = min(V, v0 + CH);
                unsigned cnt = 0;
                for (int v = v0; v < v1; ++v) cnt += sv[v] > 0.f ? 1u : 0u;
                unsigned inc = cnt;
                for (int off = 1; off < 256; off <<= 1) {
                    scan[tid] = inc;
                    __syncthreads();
                    if (tid >= off) inc += scan[tid - off];
                    __syncthreads();
                }
                scan[tid] = inc;
                int pos = (int)(inc - cnt);
                for (int v = v0; v < v1; ++v) if (sv[v] > 0.f) nz[pos++] = v;
                __syncthreads();
                nnz = (int)scan[255];
            }
            if (tid == 0) { tp_prefix = 0u; tp_below = 0.f; }
            __syncthreads();
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
                const unsigned prefix = tp_prefix;
                const float below = tp_below;
                float mass = 0.f;
                for (int i = 0; i < nnz; ++i) {
                    const float pu = sv[nz[i]];
                    const unsigned ku = __builtin_bit_cast(unsigned, pu);
                    if ((ku & pmask) == prefix && ((ku >> shift) & 255u) == (unsigned)tid) mass += pu;
                }
                float inc = mass;                        // inclusive ascending scan over the bins
                for (int off = 1; off < 256; off <<= 1) {
                    fscan[tid] = inc;
                    __syncthreads();
                    if (tid >= off) inc += fscan[tid - off];
                    __syncthreads();
                }
                if (tid == 0) { tp_bin[0] = 256; tp_bin[1] = -1; }
                __syncthreads();
                if (below + inc > target) atomicMin(&tp_bin[0], tid);     // first bin whose inclusive mass crosses
                if (mass > 0.f) atomicMax(&tp_bin[1], tid);               // fallback: the top non-empty bin
                __syncthreads();
                const int bsel = tp_bin[0] < 256 ? tp_bin[0] : tp_bin[1];
                if (tid == bsel) {
                    tp_prefix = prefix | ((unsigned)tid << shift);
                    tp_below = below + (inc - mass);
                }
                __syncthreads();
            }
            const unsigned tau = tp_prefix;
            const float mless = tp_below;
            for (int v = tid; v < V; v += 256) {
                const float pv = sv[v];
                float k = 0.f;
                if (pv > 0.f) {
                    const unsigned kv = __builtin_bit_cast(unsigned, pv);
                    if (kv > tau) k = 1.f;
                    else if (kv == tau) {
                        float c = mless;                 // equal values: by rank (index order), exactly as the pairwise rule
                        for (int i = 0; i < nnz && nz[i] <= v; ++i) if (__builtin_bit_cast(unsigned, sv[nz[i]]) == tau) c += pv;
                        k = c > target ? 1.f : 0.f;
                    }
                    if (pv == pmax) k = 1.f;
                }
                keep[v] = k;
            }
            __syncthreads();
            float s2 = 0.f;
            for (int v = tid; v < V; v += 256) { const float pk = sv[v] * keep[v]; sv[v] = pk; s2 += pk; }
            s2 = wave_sum(s2);
            if (lane == 0) wv[w] = s2;
            __syncthreads();
            s2 = (wv[0] + wv[1]) + (wv[2] + wv[3]);
            for (int v = tid; v < V; v += 256) {
                const float pk = sv[v] / s2;
                sv[v] = pk;
                if (a.probs) a.probs[(size_t)row * a.V + v] = pk;
            }
            __syncthreads();
        }
        // ---- inverse CDF in index order: thread t owns the contiguous block [t*CH, (t+1)*CH); block sums are prefix-
        //      scanned over the 256 threads, then each thread walks its own block for the first crossing of u.
        //      (One thread walking all V entries cost tens of microseconds; the summation order differs from a plain
        //      left-to-right cumsum only in the last bits, which moves a draw only when u sits within ~1e-7 of an edge.)
        {
            const float u = philox_uniform(a.seed, (uint32_t)row, (uint32_t)a.state->step_count);
            const int CH = (V + 255) / 256;
            const int v0 = tid * CH, v1 = min(V, v0 + CH);
            float bsum = 0.f;
            int blast = -1;
            for (int v = v0; v < v1; ++v) { const float p = sv[v]; if (p > 0.f) { bsum += p; blast = v; } }
            float inc = bsum;                       // inclusive scan of the block sums
            for (int off = 1; off < 256; off <<= 1) {
                fscan[tid] = inc;
                __syncthreads();
                if (tid >= off) inc += fscan[tid - off];
                __syncthreads();
            }
            int pick = 0x7fffffff;
            float c = inc - bsum;                   // mass before this block
            for (int v = v0; v < v1; ++v) { const float p = sv[v]; if (p > 0.f) { c += p; if (pick == 0x7fffffff && c > u) pick = v; } }
            // first crossing over all blocks = smallest index; fallback = last index with non-zero probability
            int lastnz = blast;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { pick = min(pick, __shfl_xor(pick, o)); lastnz = max(lastnz, __shfl_xor(lastnz, o)); }
            __shared__ int wlast[4];
            if (lane == 0) { wi[w] = pick; wlast[w] = lastnz; }
            __syncthreads();
            pick = min(min(wi[0], wi[1]), min(wi[2], wi[3]));
            lastnz = max(max(wlast[0], wlast[1]), max(wlast[2], wlast[3]));
            __syncthreads();
            if (tid == 0) wi[0] = pick == 0x7fffffff ? lastnz : pick;
        }
        __syncthreads();
        choice = wi[0];
    }

    if (tid == 0) {
        long long tok = choice;
        if (a.own) a.own[(size_t)row * a.L + cur] = tok;
        if (force_in >= 0) tok = force_in;      // teacher forcing (tests)
        if (fseq_in >= 0) tok = fseq_in;
        const int unf = unf_in;
        long long nxt = unf ? tok : (long long)a.pad;
        a.raw_ids[(size_t)row * a.L + cur] = nxt;
        const int unf_new = unf && (nxt != (long long)a.eos);
        a.unfinished[row] = unf_new;
        const long long forced = forced_in;
        const int feed = (int)(forced == -1 ? nxt : forced);
        a.tok_in[row] = feed;
        if (a.guidance > 1.0f) a.tok_in[row + a.rows] = feed;   // both halves see the same ids (M:2112)
        // the loop state is settled from these latches by a LATER launch (settle_state): a "last block advances" ticket
        // here cost 2 x rows serialised device-scope atomics per step (~8 of this kernel's 13 us at 288 rows)
    }
}

// one workgroup: counts the rows still unfinished after the selection and settles the loop state from the counters the
// selection left in state[1] (unfinished_sequences.max() == 0 / MaxLengthCriteria of the third-party loop)
__global__ __launch_bounds__(256) void advance_kernel(StepState* s, const int* unfinished, int rows, int L) {
    settle_state(s, unfinished, rows);
}

// feed tokens for the first decode step: the masked last column of the start ids (M:2107, M:2120-2130)
__global__ void init_tokens_kernel(const long long* raw_ids, const long long* pattern, int* tok_in, int rows, int rows_eff,
                                   int L, int col) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_eff) return;
    int src = r % rows;
    long long f = pattern[(size_t)src * L + col];
    tok_in[r] = (int)(f == -1 ? raw_ids[(size_t)src * L + col] : f);
}

// additive masks: 0 where attended, finfo.min where not (masked_fill semantics of the HF helpers, M:870-879)
__global__ void build_bias_kernel(const int* mask, float* bias, int B, int n_mask, int ld_bias, int* any_masked) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * ld_bias) return;
    int b = i / ld_bias, s = i % ld_bias;
    float v = 0.f;
    if (s < n_mask && mask[b * n_mask + s] == 0) {
        v = -3.4028234663852886e38f;
        if (any_masked) *any_masked = 1;       // every writer stores the same value: no atomic needed (zeroed before the launch)
    }
    bias[i] = v;
}

// the same rule for a segment of key positions [offset, offset + n) (decoder_attention_mask over the prefix columns)
__global__ void build_bias_segment_kernel(const int* mask, float* bias, int B, int n, int ld_bias, int offset) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n) return;
    int b = i / n, s = i % n;
    bias[(size_t)b * ld_bias + offset + s] = mask[i] == 0 ? -3.4028234663852886e38f : 0.f;
}

// encoder states pass through unprojected when the widths match: enc_h = enc * attention_mask (M:2016-2017)
__global__ void rowmask_mul_kernel(const float* x, const int* rowmask, float* y, int rows, int D) {
    const int r = blockIdx.x;
    const float m = (float)rowmask[r];
    for (int d = threadIdx.x; d < D; d += blockDim.x) y[(size_t)r * D + d] = x[(size_t)r * D + d] * m;
}

__global__ void t5_embed_kernel(const long long* ids, const void* table, float* x, int D, int is_bf16) {
    const int r = blockIdx.x;
    const long long tok = ids[r];
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float v = is_bf16 ? bf16_bits_to_float(reinterpret_cast<const uint16_t*>(table)[(size_t)tok * D + d])
                          : reinterpret_cast<const float*>(table)[(size_t)tok * D + d];
        x[(size_t)r * D + d] = v;
    }
}

// final T5 RMSNorm (row-major in/out)
__global__ void rmsnorm_rows_kernel(const float* x, const float* w, float* y, int D, float eps) {
    const int r = blockIdx.x;
    __shared__ float part[4];
    float q = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) { float v = x[(size_t)r * D + d]; q += v * v; }
    q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = q;
    __syncthreads();
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += part[i];
    float rstd = 1.0f / sqrtf(tot / (float)D + eps);
    for (int d = threadIdx.x; d < D; d += blockDim.x) y[(size_t)r * D + d] = w[d] * (x[(size_t)r * D + d] * rstd);
}


// ---- decoder-op variants (extensions; see misc.h) ----
// one wave per (row, head): lane i < 32 owns the pair (i, i + 32).  Products and sums are kept as separate roundings
// (no fma contraction), the order of the fp32 PyTorch statement in oracle/cpu_ref.py: a * cos - b * sin, b * cos + a * sin.
template <typename P>
__global__ __launch_bounds__(64) void rope_kernel(RopeArgs a) {
    typedef typename P::elem_t elem_t;
    const int h = blockIdx.x, row = blockIdx.y, i = threadIdx.x;
    if (i >= PTTS_HEAD_DIM / 2) return;
    int b, pos;
    if (a.npos > 0) { b = row / a.npos; pos = a.pos0 + (row - b * a.npos); }
    else { b = row; pos = a.state->pos; }
    const float c = a.cos_t[(size_t)pos * (PTTS_HEAD_DIM / 2) + i], sn = a.sin_t[(size_t)pos * (PTTS_HEAD_DIM / 2) + i];
    float* q = a.qk + (size_t)row * a.ld + h * PTTS_HEAD_DIM;
    const float qa = q[i], qb = q[i + 32];
    q[i] = __fsub_rn(__fmul_rn(qa, c), __fmul_rn(qb, sn));
    q[i + 32] = __fadd_rn(__fmul_rn(qb, c), __fmul_rn(qa, sn));
    const float* k = q + a.D;
    const float ka = k[i], kb = k[i + 32];
    elem_t* dst = reinterpret_cast<elem_t*>(a.pool.base) + a.pool.elem_off(b, a.layer, 0, h, pos);
    dst[i] = to_elem<P>(__fsub_rn(__fmul_rn(ka, c), __fmul_rn(kb, sn)));
    dst[i + 32] = to_elem<P>(__fadd_rn(__fmul_rn(kb, c), __fmul_rn(ka, sn)));
}

template <typename P>
__global__ __launch_bounds__(256) void swiglu_frag_kernel(const float* __restrict__ raw, typename P::elem_t* __restrict__ of, int rows, int F) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * F) return;
    const int r = (int)(idx / F), n = (int)(idx - (size_t)r * F);
    const float g = raw[(size_t)r * 2 * F + n], u = raw[(size_t)r * 2 * F + F + n];
    const float y = __fmul_rn(g / (1.0f + expf(-g)), u);          // F.silu(g) * u
    of[(size_t)(r >> 5) * 32 * F + frag_off<P>(r & 31, n)] = to_elem<P>(y);
}

}  // namespace

hipError_t embed_codebooks_launch(const EmbedArgs& a, int B, int dtype, hipStream_t s) {
    const int extra = a.unfinished ? 1 : 0;     // one more workgroup settles the loop state
    if (dtype == PTTS_BF16) hipLaunchKernelGGL(embed_codebooks_kernel<PBF16>, dim3(B + extra), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(embed_codebooks_kernel<PF32>, dim3(B + extra), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t embed_prompt_launch(const EmbedArgs& a, int B, int dtype, hipStream_t s) {
    if (dtype == PTTS_BF16) hipLaunchKernelGGL(embed_prompt_kernel<PBF16>, dim3(B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(embed_prompt_kernel<PF32>, dim3(B), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t embed_prompt_rows_launch(const EmbedArgs& a, int B, int npos, int pos0, int dtype, hipStream_t s) {
    if (dtype == PTTS_BF16) hipLaunchKernelGGL(embed_prompt_rows_kernel<PBF16>, dim3(B * npos), dim3(256), 0, s, a, npos, pos0);
    else hipLaunchKernelGGL(embed_prompt_rows_kernel<PF32>, dim3(B * npos), dim3(256), 0, s, a, npos, pos0);
    return hipGetLastError();
}
hipError_t embed_prefix_rows_launch(const EmbedArgs& a, int B, int npos, int pos0, int col0, const long long* raw_ids,
                                    const long long* pattern, int rows_src, int L, int dtype, hipStream_t s) {
    if (dtype == PTTS_BF16)
        hipLaunchKernelGGL(embed_prefix_rows_kernel<PBF16>, dim3(B * npos), dim3(256), 0, s, a, npos, pos0, col0, raw_ids, pattern, rows_src, L);
    else
        hipLaunchKernelGGL(embed_prefix_rows_kernel<PF32>, dim3(B * npos), dim3(256), 0, s, a, npos, pos0, col0, raw_ids, pattern, rows_src, L);
    return hipGetLastError();
}
hipError_t reset_state_launch(StepState* st, const StepState& v, int* unfinished, int n_flags, hipStream_t s) {
    hipLaunchKernelGGL(reset_state_kernel, dim3(1), dim3(256), 0, s, st, v, unfinished, n_flags);
    return hipGetLastError();
}
hipError_t advance_prompt_launch(StepState* st, hipStream_t s) {
    hipLaunchKernelGGL(advance_prompt_kernel, dim3(1), dim3(1), 0, s, st);
    return hipGetLastError();
}
hipError_t select_launch(const SelectArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(select_kernel, dim3(a.rows), dim3(256), (size_t)3 * a.V * sizeof(float), s, a);   // scores | keep | compact list
    return hipGetLastError();
}
hipError_t advance_launch(StepState* st, const int* unfinished, int rows, int L, hipStream_t s) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(256), 0, s, st, unfinished, rows, L);
    return hipGetLastError();
}
hipError_t init_tokens_launch(const long long* raw_ids, const long long* pattern, int* tok_in, int rows, int rows_eff,
                              int L, int col, hipStream_t s) {
    hipLaunchKernelGGL(init_tokens_kernel, dim3((rows_eff + 255) / 256), dim3(256), 0, s, raw_ids, pattern, tok_in, rows,
                       rows_eff, L, col);
    return hipGetLastError();
}
hipError_t build_bias_segment_launch(const int* mask, float* bias, int B, int n, int ld_bias, int offset, hipStream_t s) {
    if (B * n <= 0) return hipSuccess;
    hipLaunchKernelGGL(build_bias_segment_kernel, dim3((B * n + 255) / 256), dim3(256), 0, s, mask, bias, B, n, ld_bias, offset);
    return hipGetLastError();
}

hipError_t build_bias_launch(const int* mask, float* bias, int B, int n_mask, int ld_bias, hipStream_t s, int* any_masked) {
    int n = B * ld_bias;
    if (any_masked) {
        hipError_t e = hipMemsetAsync(any_masked, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(build_bias_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mask, bias, B, n_mask, ld_bias, any_masked);
    return hipGetLastError();
}
hipError_t rowmask_mul_launch(const float* x, const int* rowmask, float* y, int rows, int D, hipStream_t s) {
    hipLaunchKernelGGL(rowmask_mul_kernel, dim3(rows), dim3(256), 0, s, x, rowmask, y, rows, D);
    return hipGetLastError();
}
hipError_t t5_embed_launch(const long long* ids, const void* table, float* x, int rows, int D, int dtype, hipStream_t s) {
    hipLaunchKernelGGL(t5_embed_kernel, dim3(rows), dim3(256), 0, s, ids, table, x, D, dtype == PTTS_BF16 ? 1 : 0);
    return hipGetLastError();
}
hipError_t rmsnorm_rows_launch(const float* x, const float* w, float* y, int rows, int D, float eps, hipStream_t s) {
    hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(rows), dim3(256), 0, s, x, w, y, D, eps);
    return hipGetLastError();
}

hipError_t rope_launch(const RopeArgs& a, int dtype, hipStream_t s) {
    if (a.rows <= 0 || a.H <= 0 || a.ld < 2 * a.D || !a.cos_t || !a.sin_t) return hipErrorInvalidValue;
    dim3 grid(a.H, a.rows);
    if (dtype == PTTS_BF16) hipLaunchKernelGGL(rope_kernel<PBF16>, grid, dim3(64), 0, s, a);
    else hipLaunchKernelGGL(rope_kernel<PF32>, grid, dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t swiglu_frag_launch(const float* raw, void* ofrag, int rows, int F, int dtype, hipStream_t s) {
    const size_t total = (size_t)rows * F;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == PTTS_BF16) hipLaunchKernelGGL(swiglu_frag_kernel<PBF16>, grid, dim3(256), 0, s, raw, (uint16_t*)ofrag, rows, F);
    else hipLaunchKernelGGL(swiglu_frag_kernel<PF32>, grid, dim3(256), 0, s, raw, (float*)ofrag, rows, F);
    return hipGetLastError();
}
