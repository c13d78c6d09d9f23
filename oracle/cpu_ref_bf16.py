"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU restatement of the decoder in the engine's BF16 MODE.

BASELINE configs[1] runs "bf16": weights, embedding tables, KV caches and the activations that feed a matrix
product are bf16, every sum / softmax / LayerNorm statistic / residual is fp32.  ``oracle/cpu_ref.py`` is the fp32
restatement that the reference's own outputs pin bit for bit; this module is THE SAME decoder (same reference lines:
M:264-382 attention, M:415-502 layer, M:794-978 decoder, M:1096-1185 heads, M:2081-2148 step inputs) with a
round-to-nearest-even bf16 rounding inserted at exactly the points where the HIP engine rounds
(parler_tts_amd/csrc), so that the bf16 engine can be held to a TIGHT tolerance: what is left
between the two is the order of fp32 additions (and the rare bf16 rounding it flips), not bf16 noise.

Rounding points (R = fp32 -> bf16 -> fp32):
  * every projection weight, embedding / position table:  R(W)                                   (load time)
  * LayerNorm-following projections (QKV, cross-q, fc1, LM heads) are "folded" (gemm.hip): W' = R(gamma * R(W)),
    c1 = sum_k W', c2 = sum_k beta * R(W);  y = rstd * (R(x) . W'^T - mean * c1) + c2  with the row statistics
    taken over R(x):  mean = sum R(x) / K,  var = max(sum R(x)^2 / K - mean^2, 0),  rstd = 1 / sqrt(var + eps)
  * self / cross K and V are cached as R(k), R(v); q stays fp32 (scaled by hd^-0.5 as M:282)
  * attention output, GELU output: R(.) before the next projection; residual stream x: fp32, projected from R(x)
  * look-ahead cross-attention query (engine.hip run_layers; active when the description fits one 64-token page):
    raw = R(x0) . W'q^T + R(a) . R(W'q . R(Wo))^T   (x0 = residual BEFORE the self-attention out-projection, a = the
    self-attention output), normalised with the statistics of R(x1), x1 = x0 + R(a) . R(Wo)^T
  * enc_to_dec_proj: (R(enc) . R(Wp)^T + b) * mask in fp32; cross K/V from R(of that)

What "tight" can mean: two bf16 pipelines with the SAME rounding points but a different order of fp32 additions do not
stay 1e-6 apart.  A rounding turns an fp32-level difference eps into a whole-ulp flip with probability eps / ulp, i.e.
into an RMS difference of sqrt(eps * ulp) >> eps, and after a few rounding stages the two pipelines differ by a fraction
of a bf16 ulp in every activation (measured: 2 layers 2e-3 on the logits, 24 layers 1.5e-2, against 3e-2 .. 8e-2 for
bf16 vs fp32).  ``acc64=True`` evaluates the same restatement with every matrix product and statistic accumulated in
fp64 (a second, maximally different summation order); tests bound |engine - restatement| by a small multiple of
|restatement(fp32 sums) - restatement(fp64 sums)|, the restatement's own re-association spread.

Pinning: with ``exact=True`` every R is the identity and the LayerNorm is ATen's two-pass form; that mode is
checked against ``cpu_ref.decoder_forward`` (which the reference pins bitwise) in tests/test_oracle_vs_golden.py, so the
only un-pinned content of this file is where the roundings sit - and those are read off the kernels, cited below.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from parler_tts_amd.configuration import ParlerTTSConfig
from parler_tts_amd.delay_pattern import apply_delay_pattern_mask, build_delay_pattern_mask

FMIN = torch.finfo(torch.float32).min


def R(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round to nearest even, v_cvt_pk_bf16_f32) -> fp32"""
    return t.to(torch.bfloat16).to(torch.float32)


def lin(x: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """x . W^T accumulated in W's dtype (fp32, or fp64 in the acc64 variant), result fp32"""
    return F.linear(x.to(W.dtype), W).float()


class Folded:
    """gemm.hip fold_ln_kernel: W' = R(gamma * R(W)), c1[n] = sum_k W'[n][k], c2[n] = sum_k beta[k] * R(W)[n][k]"""

    def __init__(self, W: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, rnd, wdt=torch.float32):
        Wb = rnd(W)
        Wf = rnd(gamma[None, :] * Wb)
        self.c1 = Wf.to(wdt).sum(1).float()
        self.c2 = (beta[None, :] * Wb).to(wdt).sum(1).float()
        self.Wf = Wf.to(wdt)

    def __call__(self, xb: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
        """xb = R(x) rows [M, K] -> LayerNorm(x) . W^T through the folded form (gemm.hip epilogue, FOLD)"""
        K = xb.shape[-1]
        acc = lin(xb, self.Wf)
        xs = xb.to(self.Wf.dtype)
        mean = (xs.sum(-1, keepdim=True) / K).float()
        var = ((xs * xs).sum(-1, keepdim=True) / K).float()
        var = (var - mean * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(var + eps)
        return rstd * (acc - mean * self.c1) + self.c2


class EngineBf16Decoder:
    """Position-at-a-time decoder with a static KV cache, the way the engine runs it (prompt positions, prefix
    columns and decode tokens all pass through the same layer stack one cached position at a time; causal attention
    over a cache IS incremental decode, M:794-978 with past_key_values)."""

    def __init__(self, cfg: ParlerTTSConfig, sd: Dict[str, torch.Tensor], exact: bool = False, lookahead: Optional[bool] = None,
                 acc64: bool = False):
        self.cfg, self.exact = cfg, exact
        self.rnd = (lambda t: t) if exact else R
        self.wdt = wdt = torch.float64 if acc64 else torch.float32
        rnd = self.rnd
        dec = cfg.decoder
        self.D, self.H, self.K, self.L = dec.hidden_size, dec.num_attention_heads, dec.num_codebooks, dec.num_hidden_layers
        self.hd = self.D // self.H
        self.lookahead = (not exact) if lookahead is None else lookahead
        self.trace = None            # debugging: a list collects the per-layer intermediates of every step_hidden call
        pre = "decoder.model.decoder."
        f32 = lambda k: sd[k].float()
        self.emb = [rnd(f32(f"{pre}embed_tokens.{k}.weight")) for k in range(self.K)]
        self.pos = rnd(f32(f"{pre}embed_positions.weights"))
        self.prompt_table = rnd(f32("embed_prompts.weight")) if "embed_prompts.weight" in sd else None
        self.layers = []
        for l in range(self.L):
            p = f"{pre}layers.{l}."
            w = {}
            qkv = torch.cat([f32(p + "self_attn.q_proj.weight"), f32(p + "self_attn.k_proj.weight"), f32(p + "self_attn.v_proj.weight")])
            w["ln1"] = (f32(p + "self_attn_layer_norm.weight"), f32(p + "self_attn_layer_norm.bias"))
            w["ln2"] = (f32(p + "encoder_attn_layer_norm.weight"), f32(p + "encoder_attn_layer_norm.bias"))
            w["ln3"] = (f32(p + "final_layer_norm.weight"), f32(p + "final_layer_norm.bias"))
            w["o"] = rnd(f32(p + "self_attn.out_proj.weight")).to(wdt)
            w["co"] = rnd(f32(p + "encoder_attn.out_proj.weight")).to(wdt)
            w["ckv"] = rnd(torch.cat([f32(p + "encoder_attn.k_proj.weight"), f32(p + "encoder_attn.v_proj.weight")])).to(wdt)
            w["fc2"] = rnd(f32(p + "fc2.weight")).to(wdt)
            if exact:
                w["qkv"], w["cq"], w["fc1"] = qkv.to(wdt), f32(p + "encoder_attn.q_proj.weight").to(wdt), f32(p + "fc1.weight").to(wdt)
            else:
                w["qkv_f"] = Folded(qkv, *w["ln1"], rnd, wdt)
                w["cq_f"] = Folded(f32(p + "encoder_attn.q_proj.weight"), *w["ln2"], rnd, wdt)
                w["fc1_f"] = Folded(f32(p + "fc1.weight"), *w["ln3"], rnd, wdt)
                # gemm.hip lookahead_build_kernel: R( sum_k W'q[n][k] * R(Wo)[k][j] ), an fp32 fma chain over k
                w["look"] = rnd((w["cq_f"].Wf @ w["o"]).float()).to(wdt)
            self.layers.append(w)
        self.lnf = (f32(f"{pre}layer_norm.weight"), f32(f"{pre}layer_norm.bias"))
        heads = torch.cat([f32(f"decoder.lm_heads.{k}.weight") for k in range(self.K)])
        self.V = sd["decoder.lm_heads.0.weight"].shape[0]
        if exact:
            self.heads = heads.to(wdt)
        else:
            self.heads_f = Folded(heads, *self.lnf, rnd, wdt)
        self.has_proj = cfg.text_encoder.d_model != dec.hidden_size and "enc_to_dec_proj.weight" in sd
        if self.has_proj:
            self.proj_w, self.proj_b = rnd(f32("enc_to_dec_proj.weight")).to(wdt), f32("enc_to_dec_proj.bias")

    # ---- prefill: encoder glue + cross K/V (engine.hip ptts_prefill; M:2010-2017, M:295-298) ----
    def start(self, enc_states: torch.Tensor, attention_mask: torch.Tensor, max_positions: int, project: bool = True):
        rnd = self.rnd
        B, T, _ = enc_states.shape
        self.B, self.T = B, T
        if project and self.has_proj:      # PRO_ROWMAJOR rounds the rows, EPI_ROWMAJOR adds the bias and applies the row mask
            h = (lin(rnd(enc_states.float()), self.proj_w) + self.proj_b) * attention_mask[..., None].float()
        elif project:
            h = enc_states.float() * attention_mask[..., None].float()
        else:
            h = enc_states.float()
        hb = rnd(h)
        self.ck, self.cv = [], []
        for w in self.layers:
            kv = lin(hb, w["ckv"])                                                 # [B, T, 2D]
            k, v = kv[..., : self.D], kv[..., self.D:]
            self.ck.append(rnd(k).view(B, T, self.H, self.hd).transpose(1, 2).contiguous().to(self.wdt))
            self.cv.append(rnd(v).view(B, T, self.H, self.hd).transpose(1, 2).contiguous().to(self.wdt))
        self.cross_bias = torch.where(attention_mask.bool(), 0.0, FMIN)[:, None, None, :].float()      # M:875-879
        self.sk = [torch.zeros(B, self.H, max_positions, self.hd, dtype=self.wdt) for _ in self.layers]     # bf16 VALUES,
        self.sv = [torch.zeros(B, self.H, max_positions, self.hd, dtype=self.wdt) for _ in self.layers]     # held in wdt
        self.self_bias = torch.zeros(B, max_positions)
        self.n = 0                                                                 # cached self positions

    def mask_positions(self, start: int, mask2d: torch.Tensor):
        """additive key bias for cache positions [start, start + n): 0 attended / finfo.min not (M:870-872)"""
        n = mask2d.shape[1]
        self.self_bias[:, start:start + n] = torch.where(mask2d.bool(), 0.0, FMIN)

    def _attend(self, q, k, v, bias):
        """q [B,H,1,hd] fp32 (scaled), k / v [B,H,S,hd] (bf16 values); attn.hip: scores + bias, exact fp32 softmax, P.V"""
        wdt = self.wdt
        s = torch.matmul(q.to(wdt), k.transpose(2, 3)).float()
        if bias is not None:
            s = s + bias
        p = F.softmax(s, dim=-1)
        return torch.matmul(p.to(wdt), v).float().transpose(1, 2).reshape(q.shape[0], self.D)

    def _ln(self, x, gb):
        return F.layer_norm(x, (self.D,), gb[0], gb[1])

    def step_hidden(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, D] fp32 (embedding + position, already summed) at cache position self.n -> final residual [B, D]"""
        rnd, B, H, hd, D = self.rnd, self.B, self.H, self.hd, self.D
        scale = hd ** -0.5
        n = self.n
        sb = self.self_bias[:, None, None, : n + 1]
        for l, w in enumerate(self.layers):
            x0b = rnd(x)
            y = lin(self._ln(x, w["ln1"]), w["qkv"]) if self.exact else w["qkv_f"](x0b)
            q = (y[:, :D] * scale).view(B, 1, H, hd).transpose(1, 2)
            self.sk[l][:, :, n] = rnd(y[:, D:2 * D]).view(B, H, hd).to(self.wdt)
            self.sv[l][:, :, n] = rnd(y[:, 2 * D:]).view(B, H, hd).to(self.wdt)
            a = rnd(self._attend(q, self.sk[l][:, :, : n + 1], self.sv[l][:, :, : n + 1], sb))
            x1 = x + lin(a, w["o"])
            if self.exact:
                cq = lin(self._ln(x1, w["ln2"]), w["cq"])
            elif self.lookahead and self.T <= 64:
                f = w["cq_f"]
                raw = lin(x0b, f.Wf) + lin(a, w["look"])
                x1b = rnd(x1).to(self.wdt)
                mean = (x1b.sum(-1, keepdim=True) / D).float()
                var = (((x1b * x1b).sum(-1, keepdim=True) / D).float() - mean * mean).clamp_min(0.0)
                cq = (1.0 / torch.sqrt(var + 1e-5)) * (raw - mean * f.c1) + f.c2
            else:
                cq = w["cq_f"](rnd(x1))
            q = (cq * scale).view(B, 1, H, hd).transpose(1, 2)
            c = rnd(self._attend(q, self.ck[l], self.cv[l], self.cross_bias))
            x2 = x1 + lin(c, w["co"])
            h = lin(self._ln(x2, w["ln3"]), w["fc1"]) if self.exact else w["fc1_f"](rnd(x2))
            x = x2 + lin(rnd(F.gelu(h)), w["fc2"])
            if self.trace is not None:
                self.trace.append(dict(layer=l, pos=n, q=y[:, :D] * scale, a=a, x1=x1, cq=cq * scale, c=c, x2=x2, x3=x))
        self.n = n + 1
        return x

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """final LayerNorm + K heads -> [B*K, V] (row b*K + k, M:1143 / M:1172)"""
        y = lin(self._ln(x, self.lnf), self.heads) if self.exact else self.heads_f(self.rnd(x))
        return y.view(self.B * self.K, self.V)

    def embed_tokens(self, tok: torch.Tensor) -> torch.Tensor:
        """tok [B, K] -> sum over codebooks left to right (M:836) + position row (M:886)"""
        x = 0
        for k in range(self.K):
            x = x + self.emb[k][tok[:, k]]
        return x + self.pos[self.n]

    def embed_prompt(self, ids: Optional[torch.Tensor], embeds: Optional[torch.Tensor], p: int) -> torch.Tensor:
        row = embeds[:, p].float() if embeds is not None else self.prompt_table[ids[:, p]]
        return row + self.pos[self.n]


_decoders: Dict = {}


def decoder_for(cfg, sd, exact=False, lookahead=None, acc64=False) -> EngineBf16Decoder:
    """folded / rounded weight sets are expensive to build (24 layers at mini_v0.1 widths: ~10 s): keep the last few, keyed
    by the identity of the state dict"""
    key = (id(sd), exact, lookahead, acc64)
    if key not in _decoders:
        while len(_decoders) >= 2:
            _decoders.pop(next(iter(_decoders)))
        _decoders[key] = (sd, EngineBf16Decoder(cfg, sd, exact=exact, lookahead=lookahead, acc64=acc64))
    return _decoders[key][1]


@torch.no_grad()
def generate_ids(cfg: ParlerTTSConfig, sd, encoder_states, attention_mask, prompt_input_ids, prompt_attention_mask,
                 max_length: int, exact: bool = False, forced: Optional[torch.Tensor] = None, record=None,
                 guidance_scale: Optional[float] = None, lookahead: Optional[bool] = None, decoder: Optional[EngineBf16Decoder] = None,
                 acc64: bool = False):
    """Greedy loop of cpu_ref.generate_ids on the engine-numerics decoder.  ``forced`` [B*K, L]: teacher forcing along
    a given raw-id matrix (column j+1 overrides the choice of step j); ``record``: None (no logits kept), an int n (first
    n steps) or a set of step indices.  Returns raw_ids, pattern, step_logits [n, B*K, V], logit_steps, own (the
    loop's own argmax of every step, before forcing)."""
    dec = cfg.decoder
    K, pad, eos, bos = dec.num_codebooks, dec.pad_token_id, dec.eos_token_id, dec.bos_token_id
    B = encoder_states.shape[0]
    use_cfg = guidance_scale is not None and guidance_scale > 1
    enc, amask = encoder_states, attention_mask
    pids, pmask = prompt_input_ids, prompt_attention_mask
    if use_cfg:                                                                   # M:2231-2237, M:2109-2118
        enc = torch.cat([enc, torch.zeros_like(enc)])
        amask = torch.cat([amask, torch.zeros_like(amask)])
        pids = None if pids is None else pids.repeat(2, 1)
        pmask = None if pmask is None else pmask.repeat(2, 1)
    P = 0 if pids is None else pids.shape[1]
    m = decoder or decoder_for(cfg, sd, exact=exact, lookahead=lookahead, acc64=acc64)
    m.start(enc, amask, P + max_length)
    if pmask is not None:
        m.mask_positions(0, pmask)
    for p in range(P):
        m.step_hidden(m.embed_prompt(pids, None, p))
    start_tok = cfg.decoder_start_token_id if cfg.decoder_start_token_id is not None else bos
    ids = torch.full((B * K, 1), start_tok, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, bos, pad, max_length, K)
    unfinished = torch.ones(B * K, dtype=torch.long)
    step_logits, steps, own, step = [], [], [], 0
    while True:
        tok = apply_delay_pattern_mask(ids, pattern)[:, -1].view(B, K)            # M:2107, M:2120-2130
        if use_cfg:
            tok = tok.repeat(2, 1)
        lg = m.logits(m.step_hidden(m.embed_tokens(tok)))
        if use_cfg:
            n = B * K
            lg = lg[n:] + (lg[:n] - lg[n:]) * guidance_scale
        if record is not None and ((step in record) if isinstance(record, (set, frozenset)) else step < record):
            step_logits.append(lg.clone())
            steps.append(step)
        nxt = torch.argmax(lg, dim=-1)
        own.append(nxt.clone())
        if forced is not None:
            nxt = forced[:, ids.shape[1]].clone()
        nxt = nxt * unfinished + pad * (1 - unfinished)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != eos).long()
        step += 1
        if int(unfinished.max()) == 0 or ids.shape[-1] >= max_length:
            break
    out = {"raw_ids": ids, "pattern": pattern, "own": torch.stack(own)}
    if step_logits:
        out["step_logits"] = torch.stack(step_logits)
        out["logit_steps"] = torch.tensor(steps)
    return out
