"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU fp32 restatement of the reference's generate() hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``parler_tts_amd``) never does and fails loudly without its HIP library.

Pinning: ``oracle/gen_golden.py`` (container-only) runs the REAL reference code from
``/root/reference`` under four in-process shims and records fixtures under
``tests/golden/``; ``tests/test_oracle_vs_golden.py`` checks every function here against them, so the
oracle's parity is pinned by outputs of the reference itself.  Third-party arithmetic the reference
delegates (T5 encoder: ``transformers``; codec decoder: ``descript-audio-codec``, absent offline) is
pinned against the installed ``transformers`` 5.15.0 ``T5EncoderModel`` / ``DacModel`` with shared weights.

Every function cites the reference lines it follows.  M: = parler_tts/modeling_parler_tts.py,
D: = parler_tts/dac_wrapper/modeling_dac.py, HF: = site-packages/transformers.
Plain stock torch ops on CPU, in the reference's op order, so results are bitwise comparable.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from parler_tts_amd.configuration import ParlerTTSConfig
from parler_tts_amd.delay_pattern import apply_delay_pattern_mask, build_delay_pattern_mask


def _finfo_min(dtype):
    return torch.finfo(dtype).min


# ----------------------------------------------------------------------------------------------
# T5 encoder (third party; HF:models/t5/modeling_t5.py:50-66, 97-108, 144-173, 217-280)
# ----------------------------------------------------------------------------------------------

def t5_rel_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bidirectional bucket of (memory_pos - query_pos), HF t5 ``_relative_position_bucket``."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    ).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


def t5_rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def t5_encode(cfg: ParlerTTSConfig, sd: Dict[str, torch.Tensor], input_ids: torch.Tensor,
              attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """``T5EncoderModel(input_ids, attention_mask).last_hidden_state`` (called at M:2229)."""
    t5 = cfg.text_encoder
    B, T = input_ids.shape
    H, dk = t5.num_heads, t5.d_kv
    x = F.embedding(input_ids, sd["text_encoder.shared.weight"])
    if attention_mask is None:
        attention_mask = torch.ones(B, T, dtype=torch.long)
    ext = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * _finfo_min(x.dtype)
    pos = torch.arange(T)
    bucket = t5_rel_bucket(pos[None, :] - pos[:, None], t5.relative_attention_num_buckets,
                           t5.relative_attention_max_distance)
    bias = F.embedding(bucket, sd["text_encoder.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"])
    bias = bias.permute(2, 0, 1).unsqueeze(0)  # [1,H,T,T]
    for i in range(t5.num_layers):
        p = f"text_encoder.encoder.block.{i}."
        h = t5_rms_norm(x, sd[p + "layer.0.layer_norm.weight"], t5.layer_norm_epsilon)
        q = F.linear(h, sd[p + "layer.0.SelfAttention.q.weight"]).view(B, T, H, dk).transpose(1, 2)
        k = F.linear(h, sd[p + "layer.0.SelfAttention.k.weight"]).view(B, T, H, dk).transpose(1, 2)
        v = F.linear(h, sd[p + "layer.0.SelfAttention.v.weight"]).view(B, T, H, dk).transpose(1, 2)
        s = torch.matmul(q, k.transpose(2, 3)) * 1.0  # T5 does not scale
        s = s + bias
        s = s + ext
        a = F.softmax(s, dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).contiguous().reshape(B, T, H * dk)
        x = x + F.linear(o, sd[p + "layer.0.SelfAttention.o.weight"])
        h = t5_rms_norm(x, sd[p + "layer.1.layer_norm.weight"], t5.layer_norm_epsilon)
        if t5.is_gated_act:
            act = gelu_new if t5.dense_act_fn == "gelu_new" else F.gelu
            g = act(F.linear(h, sd[p + "layer.1.DenseReluDense.wi_0.weight"]))
            u = F.linear(h, sd[p + "layer.1.DenseReluDense.wi_1.weight"])
            ff = g * u
        else:
            ff = F.relu(F.linear(h, sd[p + "layer.1.DenseReluDense.wi.weight"]))
        x = x + F.linear(ff, sd[p + "layer.1.DenseReluDense.wo.weight"])
    return t5_rms_norm(x, sd["text_encoder.encoder.final_layer_norm.weight"], t5.layer_norm_epsilon)


# ----------------------------------------------------------------------------------------------
# decoder
# ----------------------------------------------------------------------------------------------

class KVCache:
    """Per layer (self_k, self_v, cross_k, cross_v), each [B,H,len,hd] (type described at M:611-621).

    ``mode='cat'`` grows the self K/V by torch.cat every step exactly like M:303-304;
    ``mode='static'`` preallocates and writes in place (same values, no re-copy)."""

    def __init__(self, mode: str = "cat", max_len: int = 0):
        self.mode = mode
        self.max_len = max_len
        self.self_k: List[torch.Tensor] = []
        self.self_v: List[torch.Tensor] = []
        self.cross_k: List[torch.Tensor] = []
        self.cross_v: List[torch.Tensor] = []
        self.length = 0  # cached self positions


def _shape(t: torch.Tensor, B: int, H: int, hd: int) -> torch.Tensor:
    return t.view(B, -1, H, hd).transpose(1, 2).contiguous()  # M:261-262


def _attention(q, k, v, mask, B, H, hd):
    """bmm / +mask / softmax / bmm, M:320-378.  q already scaled.  q:[B,H,tq,hd]."""
    tq = q.shape[2]
    qf = q.reshape(B * H, tq, hd)
    kf = k.reshape(B * H, -1, hd)
    vf = v.reshape(B * H, -1, hd)
    w = torch.bmm(qf, kf.transpose(1, 2))
    if mask is not None:
        w = (w.view(B, H, tq, -1) + mask).view(B * H, tq, -1)
    w = F.softmax(w, dim=-1)
    o = torch.bmm(w, vf)
    return o.view(B, H, tq, hd).transpose(1, 2).reshape(B, tq, H * hd)


def causal_4d_mask(mask2d: Optional[torch.Tensor], B: int, tq: int, past: int, dtype) -> Optional[torch.Tensor]:
    """HF ``_prepare_4d_causal_attention_mask`` (imported at M:31, used M:870-872).

    No 2-D mask: causal triangle only, and ``None`` for a single query.  With a 2-D mask the padded
    keys are masked_fill'ed with finfo.min on top of the triangle (never summed)."""
    S = past + tq
    mn = _finfo_min(dtype)
    causal = None
    if tq > 1:
        causal = torch.full((tq, tq), mn, dtype=dtype)
        causal = torch.triu(causal, diagonal=1)
        if past > 0:
            causal = torch.cat([torch.zeros(tq, past, dtype=dtype), causal], dim=-1)
        causal = causal[None, None].expand(B, 1, tq, S)
    if mask2d is None:
        return causal
    pad = (1.0 - mask2d[:, None, None, :].to(dtype)).expand(B, 1, tq, S)
    pad = pad.masked_fill(pad.bool(), mn)
    if causal is None:
        return pad
    return causal.masked_fill(pad.bool(), mn)


def decoder_forward(
    cfg: ParlerTTSConfig,
    sd: Dict[str, torch.Tensor],
    input_ids: torch.Tensor,                 # [B*K, q] (delay mask already applied)
    encoder_hidden_states: torch.Tensor,     # [B,T,D] (projected + masked, M:2010-2017)
    encoder_attention_mask: Optional[torch.Tensor],  # [B,T] or None
    prompt_hidden_states: Optional[torch.Tensor],    # [B,P,D], first call only
    prompt_attention_mask: Optional[torch.Tensor],   # [B,P] or None
    cache: KVCache,
    return_hidden: bool = False,
    decoder_attention_mask: Optional[torch.Tensor] = None,   # [B, past_generated + q_in] or None
):
    """ParlerTTSDecoder.forward + 9 LM heads: M:794-978, M:1096-1185.  Returns logits [B*K, q, V].

    The `norm_type` / `position_embedding` / `mlp_type` branches restate the decoder-op VARIANTS of configuration.py
    (RMSNorm, rotary positions on the self-attention, SwiGLU).  They follow no line of the reference - its decoder has none of
    them - so everything tested through them is "unpinned by the reference": this plain fp32
    PyTorch statement is the only checker that can exist for them."""
    dec = cfg.decoder
    rms = getattr(dec, "norm_type", "layer_norm") == "rms_norm"
    rope = getattr(dec, "position_embedding", "sinusoidal") == "rope"
    swiglu = getattr(dec, "mlp_type", "gelu") == "swiglu"

    def norm(t, wkey, bkey):
        if rms:
            return t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-5) * sd[wkey]
        return F.layer_norm(t, (t.shape[-1],), sd[wkey], sd[bkey])

    D, H, K, L = dec.hidden_size, dec.num_attention_heads, dec.num_codebooks, dec.num_hidden_layers
    hd = D // H
    pre = "decoder.model.decoder."
    ids = input_ids.reshape(-1, K, input_ids.shape[-1])          # M:823
    B, _, q_in = ids.shape
    past = cache.length                                           # M:833
    x = sum([F.embedding(ids[:, k], sd[f"{pre}embed_tokens.{k}.weight"]) for k in range(K)])  # M:836
    if prompt_hidden_states is not None:
        x = torch.cat([prompt_hidden_states, x], dim=1)          # M:839-840
    tq = x.shape[1]
    dtype = x.dtype

    mask2d = None
    if prompt_attention_mask is not None and decoder_attention_mask is not None:     # M:845-846
        mask2d = torch.cat([prompt_attention_mask, decoder_attention_mask], dim=1)
    elif decoder_attention_mask is not None:                     # no prompt mask: the decoder mask alone (M:870-872)
        mask2d = decoder_attention_mask
    elif prompt_attention_mask is not None:                      # M:847-867
        if past == 0:
            mask2d = torch.cat([prompt_attention_mask, torch.ones(B, q_in, dtype=prompt_attention_mask.dtype)], 1)
        else:
            gen_len = past - prompt_attention_mask.shape[1] + 1
            mask2d = torch.cat([prompt_attention_mask, torch.ones(B, gen_len, dtype=prompt_attention_mask.dtype)], 1)
    self_mask = causal_4d_mask(mask2d, B, tq, past, dtype)       # M:870-872
    cross_mask = None
    if encoder_attention_mask is not None and encoder_hidden_states is not None:                       # M:875-879
        inv = 1.0 - encoder_attention_mask[:, None, None, :].to(dtype).expand(B, 1, tq, -1)
        cross_mask = inv.masked_fill(inv.bool(), _finfo_min(dtype))

    if not rope:
        pos = sd[f"{pre}embed_positions.weights"].index_select(0, torch.arange(tq) + past)  # M:219-223
        x = x + pos                                                   # M:886
    scaling = hd ** -0.5
    if rope:      # variant: angles of the positions past .. past + tq - 1 (prompt positions count, as M:882-883 counts them)
        from parler_tts_amd.weights import rope_tables      # the table recipe is data shared with the engine's loader, not engine code

        cos_t, sin_t = rope_tables(dec, past + tq)
        cos, sin = cos_t[past:past + tq][None, None], sin_t[past:past + tq][None, None]

        def rot(t):       # [B, H, tq, hd]: pairs (i, i + hd / 2)
            a, b = t[..., : hd // 2], t[..., hd // 2:]
            return torch.cat([a * cos - b * sin, b * cos + a * sin], dim=-1)

    first = cache.length == 0 and len(cache.self_k) == 0
    for l in range(L):
        p = f"{pre}layers.{l}."
        # --- self attention, M:445-460 ---
        res = x
        h = norm(x, p + "self_attn_layer_norm.weight", p + "self_attn_layer_norm.bias")
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"]) * scaling      # M:282
        k = _shape(F.linear(h, sd[p + "self_attn.k_proj.weight"]), B, H, hd)
        v = _shape(F.linear(h, sd[p + "self_attn.v_proj.weight"]), B, H, hd)
        if rope:
            q = rot(_shape(q, B, H, hd)).transpose(1, 2).reshape(B, tq, D)
            k = rot(k)
        if cache.mode == "cat":
            if past > 0:
                k = torch.cat([cache.self_k[l], k], dim=2)               # M:303-304
                v = torch.cat([cache.self_v[l], v], dim=2)
            if first:
                cache.self_k.append(k); cache.self_v.append(v)
            else:
                cache.self_k[l] = k; cache.self_v[l] = v
        else:
            if first:
                cache.self_k.append(torch.zeros(B, H, cache.max_len, hd, dtype=dtype))
                cache.self_v.append(torch.zeros(B, H, cache.max_len, hd, dtype=dtype))
            cache.self_k[l][:, :, past:past + tq] = k
            cache.self_v[l][:, :, past:past + tq] = v
            k = cache.self_k[l][:, :, : past + tq]
            v = cache.self_v[l][:, :, : past + tq]
        a = _attention(_shape(q, B, H, hd), k, v, self_mask, B, H, hd)
        x = res + F.linear(a, sd[p + "self_attn.out_proj.weight"])
        # --- cross attention, M:462-483 (the whole block is skipped without encoder states, M:465) ---
        if encoder_hidden_states is not None:
            res = x
            h = norm(x, p + "encoder_attn_layer_norm.weight", p + "encoder_attn_layer_norm.bias")
            q = F.linear(h, sd[p + "encoder_attn.q_proj.weight"]) * scaling
            if first:                                                        # M:295-298
                cache.cross_k.append(_shape(F.linear(encoder_hidden_states, sd[p + "encoder_attn.k_proj.weight"]), B, H, hd))
                cache.cross_v.append(_shape(F.linear(encoder_hidden_states, sd[p + "encoder_attn.v_proj.weight"]), B, H, hd))
            a = _attention(_shape(q, B, H, hd), cache.cross_k[l], cache.cross_v[l], cross_mask, B, H, hd)
            x = res + F.linear(a, sd[p + "encoder_attn.out_proj.weight"])
        # --- MLP, M:485-492 ---
        res = x
        h = norm(x, p + "final_layer_norm.weight", p + "final_layer_norm.bias")
        if swiglu:
            h = F.silu(F.linear(h, sd[p + "fc1.weight"])) * F.linear(h, sd[p + "fc1_up.weight"])
        else:
            h = F.gelu(F.linear(h, sd[p + "fc1.weight"]))
        x = res + F.linear(h, sd[p + "fc2.weight"])
    cache.length = past + tq
    x = norm(x, f"{pre}layer_norm.weight", f"{pre}layer_norm.bias")  # M:959
    logits = torch.stack([F.linear(x, sd[f"decoder.lm_heads.{k}.weight"]) for k in range(K)], dim=1)  # M:1143
    logits = logits.reshape(-1, *logits.shape[2:])                      # M:1172
    if return_hidden:
        return logits, x
    return logits


def encoder_glue(cfg: ParlerTTSConfig, sd, enc_last_hidden: torch.Tensor,
                 attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """enc_to_dec_proj + mask multiply, M:2010-2017 (the reference redoes this every step)."""
    h = enc_last_hidden
    if cfg.text_encoder.d_model != cfg.decoder.hidden_size:
        h = F.linear(h, sd["enc_to_dec_proj.weight"], sd["enc_to_dec_proj.bias"])
    if attention_mask is not None:
        h = h * attention_mask[..., None]
    return h


# ----------------------------------------------------------------------------------------------
# token selection (third party: transformers 4.3x GenerationMixin.greedy_search / .sample)
# ----------------------------------------------------------------------------------------------

def warp_logits(logits: torch.Tensor, temperature: float, top_k: int, top_p: float = 1.0) -> torch.Tensor:
    """TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper in the order `_get_logits_warper` builds
    them: scores/tau; everything below the k-th largest value becomes -inf (ties at the threshold are kept);
    then, in ascending order of probability, tokens whose cumulative probability is <= 1 - top_p are removed
    (the most probable token always stays)."""
    s = logits
    if temperature is not None and temperature != 1.0:
        s = s / temperature
    if top_k is not None and top_k > 0:
        kk = min(top_k, s.shape[-1])
        thr = torch.topk(s, kk)[0][..., -1, None]
        s = s.masked_fill(s < thr, -float("inf"))
    if top_p is not None and 0.0 < top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(s, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[..., -1:] = False
        s = s.masked_fill(remove.scatter(-1, sorted_idx, remove), -float("inf"))
    return s


def cfg_combine(logits: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    """ClassifierFreeGuidanceLogitsProcessor (appended at M:2556-2558): rows [0:n) conditional,
    [n:2n) unconditional -> uncond + (cond - uncond) * scale."""
    n = logits.shape[0] // 2
    cond, uncond = logits[:n], logits[n:]
    return uncond + (cond - uncond) * guidance_scale


@torch.no_grad()
def generate_ids(
    cfg: ParlerTTSConfig,
    sd: Dict[str, torch.Tensor],
    encoder_states: torch.Tensor,            # T5 last_hidden_state [B,T,d_enc]
    attention_mask: Optional[torch.Tensor],  # [B,T]
    prompt_input_ids: Optional[torch.Tensor],  # [B,P]
    prompt_attention_mask: Optional[torch.Tensor],
    max_length: int,
    do_sample: bool = False,
    temperature: float = 1.0,
    top_k: int = 50,
    top_p: float = 1.0,
    guidance_scale: Optional[float] = None,
    generator: Optional[torch.Generator] = None,
    kv_mode: str = "cat",
    record_logits: int = 0,
    forced_tokens: Optional[Dict[int, torch.Tensor]] = None,
    step_times: Optional[list] = None,
    logits_processor: Optional[list] = None,
    decoder_input_ids: Optional[torch.Tensor] = None,
    decoder_attention_mask: Optional[torch.Tensor] = None,
    prompt_hidden_states: Optional[torch.Tensor] = None,
    decoder_only: bool = False,
    num_return_sequences: int = 1,
    record_steps=None,
) -> Dict[str, torch.Tensor]:
    """The hot loop: M:2530-2537 (delay mask) + prepare_inputs_for_generation M:2081-2148 +
    forward M:1927-2079 + the third-party greedy/sample step.

    Returns raw loop ids [B*K, L_out] (pre-mask samples, as the reference keeps them), per-step
    logits for the first ``record_logits`` steps, and the top-2 margin of every greedy argmax.
    ``forced_tokens[step]`` overrides the sampled tokens of that step (teacher forcing in tests).
    """
    import time

    dec = cfg.decoder
    K = dec.num_codebooks
    pad, eos, bos = dec.pad_token_id, dec.eos_token_id, dec.bos_token_id
    use_cfg = guidance_scale is not None and guidance_scale > 1
    if decoder_only:     # M:1218-1232: the driver doubles ids and prompt only; encoder states come with 2B rows, or not at all
        B = prompt_hidden_states.shape[0] if prompt_hidden_states is not None else encoder_states.shape[0] // (2 if use_cfg else 1)
    else:
        B = encoder_states.shape[0]

    enc = encoder_states
    amask = attention_mask
    if use_cfg and not decoder_only:  # M:2231-2237
        enc = torch.cat([enc, torch.zeros_like(enc)], dim=0)
        if amask is not None:
            amask = torch.cat([amask, torch.zeros_like(amask)], dim=0)
    prompt_h = F.embedding(prompt_input_ids, sd["embed_prompts.weight"]) if prompt_input_ids is not None else None
    if prompt_hidden_states is not None:          # pre-embedded prompt (M:2476; the decoder-only driver's only form)
        prompt_h = prompt_hidden_states
    pmask = prompt_attention_mask
    if use_cfg and decoder_only:  # M:1218-1232: zeroed prompt under an all-zero prompt mask for the null half
        if prompt_h is not None:
            prompt_h = torch.cat([prompt_h, torch.zeros_like(prompt_h)], dim=0)
        if pmask is not None:
            pmask = torch.cat([pmask, torch.zeros_like(pmask)], dim=0)
    elif use_cfg:  # M:2109-2118
        if prompt_h is not None:
            prompt_h = prompt_h.repeat(2, 1, 1)
        if pmask is not None:
            pmask = pmask.repeat(2, 1)
    # decoder-only driver (ParlerTTSForCausalLM.generate, M:1281-1552): encoder_hidden_states go to the decoder as given
    enc_h = enc if decoder_only else encoder_glue(cfg, sd, enc, amask)  # constant across steps

    start_tok = cfg.decoder_start_token_id if cfg.decoder_start_token_id is not None else bos
    ids = torch.full((B * K, 1), start_tok, dtype=torch.long)                    # M:2174-2181
    if decoder_input_ids is not None:                                            # M:2163-2185: user prefix, start column ensured
        prepend = bool((decoder_input_ids[..., 0] != start_tok).all())
        ids = decoder_input_ids if not prepend else torch.cat([ids, decoder_input_ids], -1)
        if prepend and decoder_attention_mask is not None:                       # M:2186-2193
            decoder_attention_mask = torch.cat([torch.ones_like(decoder_attention_mask)[:, :1], decoder_attention_mask], -1)
    dmask = decoder_attention_mask
    ids, pattern = build_delay_pattern_mask(ids, bos, pad, max_length, K)       # M:2530-2535
    if num_return_sequences > 1:
        # M:2600-2606 -> third-party `_expand_inputs_for_generation`: repeat_interleave(n, dim=0) of the id rows [B*K, s] and
        # of every tensor kwarg - encoder states, masks, prompt states and the stashed delay pattern [B*K, L].  The decoder
        # then reads row (b*K + k)*n + j as codebook ((b*K + k)*n + j) % K of utterance ((b*K + k)*n + j) // K (M:823).
        n = num_return_sequences
        rep = lambda t: None if t is None else t.repeat_interleave(n, dim=0)
        if use_cfg:     # both guidance halves were concatenated before the expansion (M:2231-2237 precede M:2600)
            h = enc_h.shape[0] // 2
            enc_h = torch.cat([rep(enc_h[:h]), rep(enc_h[h:])])
            amask = None if amask is None else torch.cat([rep(amask[:h]), rep(amask[h:])])
            prompt_h = None if prompt_h is None else torch.cat([rep(prompt_h[:h]), rep(prompt_h[h:])])
            pmask = None if pmask is None else torch.cat([rep(pmask[:h]), rep(pmask[h:])])
        else:
            enc_h, amask, prompt_h, pmask = rep(enc_h), rep(amask), rep(prompt_h), rep(pmask)
        ids, pattern, dmask = rep(ids), rep(pattern), rep(dmask)
    cache = KVCache(kv_mode, max_len=(0 if prompt_h is None else prompt_h.shape[1]) + max_length)
    unfinished = torch.ones(ids.shape[0], dtype=torch.long)
    step_logits, margins = [], []
    step = 0
    while True:
        t0 = time.perf_counter()
        masked = apply_delay_pattern_mask(ids, pattern)                          # M:2107
        model_in = masked if cache.length == 0 else masked[:, -1:]              # M:2120-2130
        if use_cfg:
            model_in = model_in.repeat(2, 1)
        dm = None if dmask is None else (dmask.repeat(2, 1) if use_cfg else dmask)     # M:2112-2113
        logits = decoder_forward(cfg, sd, model_in, enc_h, amask,
                                 prompt_h if cache.length == 0 else None, pmask, cache, decoder_attention_mask=dm)[:, -1, :]
        if dmask is not None:      # third-party _update_model_kwargs_for_generation: one column of ones per new token
            dmask = torch.cat([dmask, dmask.new_ones((dmask.shape[0], 1))], dim=-1)
        if use_cfg:
            logits = cfg_combine(logits, guidance_scale)
        for proc in (logits_processor or []):      # user processors follow the guidance processor (M:2556-2567), before the warpers
            logits = proc(ids, logits)
        if step < record_logits or (record_steps is not None and step in record_steps):
            step_logits.append(logits.clone())
        if do_sample:
            probs = F.softmax(warp_logits(logits, temperature, top_k, top_p), dim=-1)
            nxt = torch.multinomial(probs, 1, generator=generator).squeeze(1)
        else:
            top2 = torch.topk(logits, 2, dim=-1)[0]
            margins.append(top2[:, 0] - top2[:, 1])
            nxt = torch.argmax(logits, dim=-1)
        if forced_tokens is not None and step in forced_tokens:
            nxt = forced_tokens[step].clone()
        nxt = nxt * unfinished + pad * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != eos).long()
        step += 1
        if step_times is not None:
            step_times.append(time.perf_counter() - t0)
        if int(unfinished.max()) == 0 or ids.shape[-1] >= max_length:
            break
    out = {"raw_ids": ids, "pattern": pattern}
    if step_logits:
        out["step_logits"] = torch.stack(step_logits)
    if margins:
        out["margins"] = torch.stack(margins)
    return out


def undelay(cfg: ParlerTTSConfig, raw_ids: torch.Tensor, pattern: torch.Tensor, num_return_sequences: int = 1) -> torch.Tensor:
    """M:2633-2648: apply the stashed mask, rebuild the mask at the ACTUAL length (from the start ids of every row, a
    regular pattern also under num_return_sequences), keep free positions, reshape to [1, B, K, -1] with the ORIGINAL
    batch size (M:2645: n * F columns per utterance under num_return_sequences = n)."""
    dec = cfg.decoder
    K = dec.num_codebooks
    out = apply_delay_pattern_mask(raw_ids, pattern)
    start = torch.full((raw_ids.shape[0], 1), dec.bos_token_id, dtype=torch.long)
    _, m = build_delay_pattern_mask(start, dec.bos_token_id, dec.pad_token_id, out.shape[1], K)
    keep = (m != dec.bos_token_id) & (m != dec.pad_token_id)
    return out[keep].reshape(raw_ids.shape[0] // (K * num_return_sequences), K, -1)[None, ...]


# ----------------------------------------------------------------------------------------------
# codec decoder (third party descript DAC; structure per HF:models/dac/modeling_dac.py:86-100,
# 175-210, 236-266, 347-372, 407-442)
# ----------------------------------------------------------------------------------------------

def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def dac_from_codes(cfg: ParlerTTSConfig, sd, codes: torch.Tensor) -> torch.Tensor:
    """quantizer.from_codes (D:130): sum_k out_proj_k(codebook_k[codes[:,k]]) -> [B, latent, F]."""
    p = "audio_encoder.model.quantizer.quantizers."
    z = 0.0
    for k in range(codes.shape[1]):
        e = F.embedding(codes[:, k, :], sd[f"{p}{k}.codebook.weight"]).transpose(1, 2)
        z = z + F.conv1d(e, sd[f"{p}{k}.out_proj.weight"], sd[f"{p}{k}.out_proj.bias"])
    return z


def dac_decoder(cfg: ParlerTTSConfig, sd, z: torch.Tensor) -> torch.Tensor:
    """DAC.decode (D:131): conv7 -> R x [snake, convT, 3 x residual unit] -> snake -> conv7 -> tanh."""
    dc = cfg.audio_encoder
    p = "audio_encoder.model.decoder.model."
    x = F.conv1d(z, sd[p + "0.weight"], sd[p + "0.bias"], padding=3)
    for n, stride in enumerate(dc.upsampling_ratios, start=1):
        b = f"{p}{n}.block."
        x = snake(x, sd[b + "0.alpha"])
        x = F.conv_transpose1d(x, sd[b + "1.weight"], sd[b + "1.bias"], stride=stride,
                               padding=math.ceil(stride / 2))
        for r, dil in zip((2, 3, 4), (1, 3, 9)):
            u = f"{b}{r}.block."
            y = F.conv1d(snake(x, sd[u + "0.alpha"]), sd[u + "1.weight"], sd[u + "1.bias"],
                         dilation=dil, padding=((7 - 1) * dil) // 2)
            y = F.conv1d(snake(y, sd[u + "2.alpha"]), sd[u + "3.weight"], sd[u + "3.bias"])
            crop = (x.shape[-1] - y.shape[-1]) // 2
            if crop > 0:
                x = x[..., crop:-crop]
            x = x + y
    R = len(dc.upsampling_ratios)
    x = snake(x, sd[f"{p}{R + 1}.alpha"])
    x = F.conv1d(x, sd[f"{p}{R + 2}.weight"], sd[f"{p}{R + 2}.bias"], padding=3)
    return torch.tanh(x)


def dac_encoder(cfg: ParlerTTSConfig, sd, audio: torch.Tensor) -> torch.Tensor:
    """descript DAC encoder conv stack (third party; structure per HF:models/dac/modeling_dac.py DacEncoder /
    DacEncoderBlock / DacResidualUnit): audio [B, 1, T] -> latent [B, latent, T // hop] (each strided conv floors)."""
    a = cfg.audio_encoder
    p = "audio_encoder.model.encoder.block."
    h = F.conv1d(audio, sd[p + "0.weight"], sd[p + "0.bias"], padding=3)
    strides = list(reversed(a.upsampling_ratios))
    for n, stride in enumerate(strides, start=1):
        b = f"{p}{n}.block."
        for r, dil in zip((0, 1, 2), (1, 3, 9)):
            u = f"{b}{r}.block."
            y = F.conv1d(snake(h, sd[u + "0.alpha"]), sd[u + "1.weight"], sd[u + "1.bias"], dilation=dil, padding=3 * dil)
            y = F.conv1d(snake(y, sd[u + "2.alpha"]), sd[u + "3.weight"], sd[u + "3.bias"])
            h = h + y
        h = F.conv1d(snake(h, sd[b + "3.alpha"]), sd[b + "4.weight"], sd[b + "4.bias"], stride=stride, padding=math.ceil(stride / 2))
    R = len(strides)
    return F.conv1d(snake(h, sd[f"{p}{R + 1}.alpha"]), sd[f"{p}{R + 2}.weight"], sd[f"{p}{R + 2}.bias"], padding=1)


def dac_quantize(cfg: ParlerTTSConfig, sd, z: torch.Tensor, margins: Optional[list] = None) -> torch.Tensor:
    """Residual vector quantisation, encode direction (HF DacResidualVectorQuantizer.forward in eval mode +
    DacVectorQuantize.decode_latents): per codebook project to codebook_dim, L2-normalise encodings and codebook,
    nearest entry, subtract its out-projection from the residual.  latent [B, latent, F] -> codes [B, K, F]."""
    a = cfg.audio_encoder
    p = "audio_encoder.model.quantizer.quantizers."
    residual = z
    codes = []
    for i in range(a.num_codebooks):
        ze = F.conv1d(residual, sd[f"{p}{i}.in_proj.weight"], sd[f"{p}{i}.in_proj.bias"])          # [B, 8, F]
        Bz, Dz, Fz = ze.shape
        enc = F.normalize(ze.permute(0, 2, 1).reshape(Bz * Fz, Dz))
        cb = F.normalize(sd[f"{p}{i}.codebook.weight"])
        dist = -(enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cb.t()) + cb.pow(2).sum(1, keepdim=True).t()
        idx = dist.max(1)[1].reshape(Bz, Fz)
        if margins is not None:      # top-2 gap of every nearest-entry decision (fixture bookkeeping)
            t2 = torch.topk(dist, 2, dim=1)[0]
            margins.append(float((t2[:, 0] - t2[:, 1]).min()))
        codes.append(idx)
        zq = F.embedding(idx, sd[f"{p}{i}.codebook.weight"]).transpose(1, 2)                         # [B, 8, F]
        zq = F.conv1d(zq, sd[f"{p}{i}.out_proj.weight"], sd[f"{p}{i}.out_proj.bias"])
        residual = residual - zq
    return torch.stack(codes, dim=1)


def dac_encode(cfg: ParlerTTSConfig, sd, input_values: torch.Tensor) -> torch.Tensor:
    """DACModel.encode (D:25-96) with chunking disabled.  D:56 right-pads to a multiple of the hop length
    (descript DAC.preprocess), but the single chunk is then cut as ``audio_data[:, :, 0:input_length]`` (D:64-66, D:85-86:
    chunk_length = stride = input_length), so the encoder sees the audio at its ORIGINAL length and descript's
    ``DAC.encode`` (encoder + quantizer, no padding of its own) yields floor(T / hop) frames.
    input_values [B, 1, T] -> audio_codes [1, B, K, T // hop]."""
    _, channels, T = input_values.shape
    if channels < 1 or channels > 2:
        raise ValueError(f"Number of audio channels must be 1 or 2, but got {channels}")       # D:53-54
    return dac_quantize(cfg, sd, dac_encoder(cfg, sd, input_values.float()))[None, ...]


def dac_decode(cfg: ParlerTTSConfig, sd, audio_codes: torch.Tensor) -> torch.Tensor:
    """DACModel.decode, D:98-134.  audio_codes [1,B,K,F] -> audio_values [B,1,F*hop]."""
    if len(audio_codes) != 1:
        raise ValueError(f"Expected one frame, got {len(audio_codes)}")
    return dac_decoder(cfg, sd, dac_from_codes(cfg, sd, audio_codes.squeeze(0)))


@torch.no_grad()
def codes_to_waveform(cfg: ParlerTTSConfig, sd, output_ids: torch.Tensor) -> torch.Tensor:
    """M:2650-2680: batched decode, or the per-sample path when any special id survived."""
    dec = cfg.decoder
    B = output_ids.shape[1]
    special = [dec.bos_token_id, dec.pad_token_id, dec.eos_token_id]
    sequential = any(bool((output_ids == s).any()) for s in special)
    if not sequential:
        return dac_decode(cfg, sd, output_ids).squeeze(1)
    vals = []
    for b in range(B):
        sample = output_ids[:, b]
        ok = (sample >= cfg.audio_encoder.codebook_size).sum(dim=(0, 1)) == 0
        if int(ok.sum()) > 0:
            wav = dac_decode(cfg, sd, sample[:, :, ok][None, ...])
            vals.append(wav.transpose(0, 2))
        else:
            vals.append(torch.zeros(1, 1, 1))
    return torch.nn.utils.rnn.pad_sequence(vals, batch_first=True, padding_value=0).squeeze(-1).squeeze(-1)


@torch.no_grad()
def generate(cfg: ParlerTTSConfig, sd, input_ids, attention_mask=None, prompt_input_ids=None,
             prompt_attention_mask=None, max_length=None, **gen) -> torch.Tensor:
    """End-to-end M:2342-2686 for the text-conditioned path: description ids -> waveform [B, samples]."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)  # M:2462-2465 (pad == eos disables HF's inference)
    enc = t5_encode(cfg, sd, input_ids, attention_mask)
    r = generate_ids(cfg, sd, enc, attention_mask, prompt_input_ids, prompt_attention_mask, max_length, **gen)
    return codes_to_waveform(cfg, sd, undelay(cfg, r["raw_ids"], r["pattern"]))
