"""State-dict contract of the engine: key layout, synthetic weights, checkpoint loading.

The key layout is the reference's (dumped from the live reference classes):

  text_encoder.*          upstream T5 encoder names (reference instantiates it at M:1598)
  embed_prompts.weight    [vocab, D]                                   (M:1642)
  enc_to_dec_proj.*       Linear(d_enc -> D) with bias                 (M:1635-1639)
  decoder.model.decoder.* embed_tokens.{k}, embed_positions.weights, layers.{l}.*, layer_norm (M:760-781)
  decoder.lm_heads.{k}.weight [V, D]                                   (M:1069-1071)
  audio_encoder.model.*   descript DAC state dict (weight-norm pairs are folded at load)

No pretrained checkpoint exists offline, so tests and the benchmark use :func:`synthetic_state_dict`:
a seeded, platform-independent (numpy PCG64) random initialisation of the named architecture with
the reference's initialiser scale (normal(0, 0.02), M:517-526).
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Iterable, Optional

import numpy as np
import torch

from .configuration import DACConfig, ParlerTTSConfig, ParlerTTSDecoderConfig, T5EncoderConfig


# ----------------------------------------------------------------------------------------------
# sinusoidal position table (reference M:199-213): [cos | sin] halves, NOT interleaved
# ----------------------------------------------------------------------------------------------

def sinusoid_table(num_positions: int, dim: int) -> torch.Tensor:
    half = dim // 2
    scale = math.log(10000.0) / (half - 1)
    inv = np.exp(np.arange(half, dtype=np.float64) * -scale).astype(np.float32).astype(np.float64)
    ang = (np.arange(num_positions, dtype=np.float64)[:, None] * inv[None, :]).astype(np.float32).astype(np.float64)
    tab = np.concatenate([np.cos(ang), np.sin(ang)], axis=1)
    if dim % 2 == 1:
        tab = np.concatenate([tab, np.zeros((num_positions, 1))], axis=1)
    return torch.from_numpy(tab.astype(np.float32))


def rope_tables(cfg: ParlerTTSDecoderConfig, num_positions: int = 0):
    """(cos, sin), each [positions, head_dim / 2] fp32, of the rotary-position VARIANT (configuration.py: an extension with no
    counterpart in the reference): angle(pos, i) = pos * rope_theta^(-2 i / head_dim), every step in fp32.  The engine and the
    fp32 PyTorch statement in oracle/cpu_ref.py both read these tables, so they rotate by bit-identical angles."""
    hd = cfg.hidden_size // cfg.num_attention_heads
    n = num_positions or cfg.max_position_embeddings
    inv = torch.pow(torch.tensor(float(cfg.rope_theta), dtype=torch.float32), -torch.arange(0, hd, 2, dtype=torch.float32) / hd)
    ang = torch.arange(n, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cos(ang), torch.sin(ang)


# ----------------------------------------------------------------------------------------------
# key enumeration
# ----------------------------------------------------------------------------------------------

def decoder_keys(cfg: ParlerTTSDecoderConfig) -> Dict[str, tuple]:
    D, F, V, K = cfg.hidden_size, cfg.ffn_dim, cfg.vocab_size, cfg.num_codebooks
    p = "decoder.model.decoder."
    keys: Dict[str, tuple] = {}
    for k in range(K):
        keys[f"{p}embed_tokens.{k}.weight"] = (V + 1, D)
    keys[f"{p}embed_positions.weights"] = (cfg.max_position_embeddings, D)
    for l in range(cfg.num_hidden_layers):
        q = f"{p}layers.{l}."
        for attn in ("self_attn", "encoder_attn"):
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                keys[f"{q}{attn}.{proj}.weight"] = (D, D)
        for ln in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            keys[f"{q}{ln}.weight"] = (D,)
            keys[f"{q}{ln}.bias"] = (D,)
        keys[f"{q}fc1.weight"] = (F, D)
        keys[f"{q}fc2.weight"] = (D, F)
        if getattr(cfg, "mlp_type", "gelu") == "swiglu":      # extension (configuration.py): the "up" matrix of the gated MLP
            keys[f"{q}fc1_up.weight"] = (F, D)
    keys[f"{p}layer_norm.weight"] = (D,)
    keys[f"{p}layer_norm.bias"] = (D,)
    for k in range(K):
        keys[f"decoder.lm_heads.{k}.weight"] = (V, D)
    return keys


def t5_keys(cfg: T5EncoderConfig) -> Dict[str, tuple]:
    d, inner, ff = cfg.d_model, cfg.num_heads * cfg.d_kv, cfg.d_ff
    keys: Dict[str, tuple] = {"text_encoder.shared.weight": (cfg.vocab_size, d)}
    for i in range(cfg.num_layers):
        b = f"text_encoder.encoder.block.{i}."
        for n in ("q", "k", "v"):
            keys[f"{b}layer.0.SelfAttention.{n}.weight"] = (inner, d)
        keys[f"{b}layer.0.SelfAttention.o.weight"] = (d, inner)
        if i == 0:
            keys[f"{b}layer.0.SelfAttention.relative_attention_bias.weight"] = (
                cfg.relative_attention_num_buckets, cfg.num_heads)
        keys[f"{b}layer.0.layer_norm.weight"] = (d,)
        if cfg.is_gated_act:
            keys[f"{b}layer.1.DenseReluDense.wi_0.weight"] = (ff, d)
            keys[f"{b}layer.1.DenseReluDense.wi_1.weight"] = (ff, d)
        else:
            keys[f"{b}layer.1.DenseReluDense.wi.weight"] = (ff, d)
        keys[f"{b}layer.1.DenseReluDense.wo.weight"] = (d, ff)
        keys[f"{b}layer.1.layer_norm.weight"] = (d,)
    keys["text_encoder.encoder.final_layer_norm.weight"] = (d,)
    return keys


def dac_decoder_keys(cfg: DACConfig) -> Dict[str, tuple]:
    """Folded (plain ``weight``) form of the descript DAC decode-side parameters.

    descript layout: ``quantizer.quantizers.{i}.{codebook,out_proj}``, ``decoder.model.{n}`` with
    n=0 first conv, n=1..R decoder blocks (``block.0`` snake, ``block.1`` transposed conv,
    ``block.2..4`` residual units each ``block.{0,2}`` snake / ``block.{1,3}`` conv), n=R+1 snake,
    n=R+2 last conv.  (Package is absent offline; names restated from its public source.)
    """
    p = "audio_encoder.model."
    keys: Dict[str, tuple] = {}
    for i in range(cfg.num_codebooks):
        keys[f"{p}quantizer.quantizers.{i}.codebook.weight"] = (cfg.codebook_size, cfg.codebook_dim)
        keys[f"{p}quantizer.quantizers.{i}.out_proj.weight"] = (cfg.latent_dim, cfg.codebook_dim, 1)
        keys[f"{p}quantizer.quantizers.{i}.out_proj.bias"] = (cfg.latent_dim,)
    C = cfg.decoder_hidden_size
    keys[f"{p}decoder.model.0.weight"] = (C, cfg.latent_dim, 7)
    keys[f"{p}decoder.model.0.bias"] = (C,)
    cin = C
    for n, stride in enumerate(cfg.upsampling_ratios, start=1):
        cout = cin // 2
        b = f"{p}decoder.model.{n}.block."
        keys[f"{b}0.alpha"] = (1, cin, 1)
        keys[f"{b}1.weight"] = (cin, cout, 2 * stride)  # ConvTranspose1d: [in, out, k]
        keys[f"{b}1.bias"] = (cout,)
        for r in (2, 3, 4):
            keys[f"{b}{r}.block.0.alpha"] = (1, cout, 1)
            keys[f"{b}{r}.block.1.weight"] = (cout, cout, 7)
            keys[f"{b}{r}.block.1.bias"] = (cout,)
            keys[f"{b}{r}.block.2.alpha"] = (1, cout, 1)
            keys[f"{b}{r}.block.3.weight"] = (cout, cout, 1)
            keys[f"{b}{r}.block.3.bias"] = (cout,)
        cin = cout
    R = len(cfg.upsampling_ratios)
    keys[f"{p}decoder.model.{R + 1}.alpha"] = (1, cin, 1)
    keys[f"{p}decoder.model.{R + 2}.weight"] = (1, cin, 7)
    keys[f"{p}decoder.model.{R + 2}.bias"] = (1,)
    return keys


def dac_encoder_keys(cfg: DACConfig) -> Dict[str, tuple]:
    """Folded (plain ``weight``) form of the descript DAC ENCODE-side parameters (audio prompts, D:25-96).

    descript layout: ``encoder.block.0`` first conv (1 -> d), ``encoder.block.{n}`` n=1..R encoder blocks
    (``block.{0,1,2}`` residual units each ``block.{0,2}`` snake / ``block.{1,3}`` conv, ``block.3`` snake,
    ``block.4`` strided conv d -> 2d with kernel 2s), ``encoder.block.{R+1}`` snake, ``encoder.block.{R+2}`` last conv
    (k=3, -> latent); ``quantizer.quantizers.{i}.in_proj`` (latent -> codebook_dim).  Strides are the upsampling ratios
    reversed.  Kept out of :func:`all_keys` on purpose: synthetic values are keyed by enumeration index."""
    p = "audio_encoder.model."
    keys: Dict[str, tuple] = {}
    d = cfg.encoder_hidden_size
    keys[f"{p}encoder.block.0.weight"] = (d, 1, 7)
    keys[f"{p}encoder.block.0.bias"] = (d,)
    strides = list(reversed(cfg.upsampling_ratios))
    for n, stride in enumerate(strides, start=1):
        b = f"{p}encoder.block.{n}.block."
        for r in (0, 1, 2):
            keys[f"{b}{r}.block.0.alpha"] = (1, d, 1)
            keys[f"{b}{r}.block.1.weight"] = (d, d, 7)
            keys[f"{b}{r}.block.1.bias"] = (d,)
            keys[f"{b}{r}.block.2.alpha"] = (1, d, 1)
            keys[f"{b}{r}.block.3.weight"] = (d, d, 1)
            keys[f"{b}{r}.block.3.bias"] = (d,)
        keys[f"{b}3.alpha"] = (1, d, 1)
        keys[f"{b}4.weight"] = (2 * d, d, 2 * stride)
        keys[f"{b}4.bias"] = (2 * d,)
        d *= 2
    R = len(strides)
    keys[f"{p}encoder.block.{R + 1}.alpha"] = (1, d, 1)
    keys[f"{p}encoder.block.{R + 2}.weight"] = (cfg.latent_dim, d, 3)
    keys[f"{p}encoder.block.{R + 2}.bias"] = (cfg.latent_dim,)
    for i in range(cfg.num_codebooks):
        keys[f"{p}quantizer.quantizers.{i}.in_proj.weight"] = (cfg.codebook_dim, cfg.latent_dim, 1)
        keys[f"{p}quantizer.quantizers.{i}.in_proj.bias"] = (cfg.codebook_dim,)
    return keys


def synthetic_dac_encoder_state_dict(cfg: DACConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded encode-side codec weights (own PCG64 streams, offset 1_000_000, so the decode-side values of
    :func:`synthetic_state_dict` are untouched).  Fan-in scaled so that activations stay O(1) through the stack."""
    sd: Dict[str, torch.Tensor] = {}
    for i, (name, shape) in enumerate(dac_encoder_keys(cfg).items()):
        rng = np.random.Generator(np.random.PCG64([seed, 1_000_000 + i]))
        a = rng.standard_normal(int(np.prod(shape)), dtype=np.float32).reshape(shape)
        if name.endswith(".alpha"):
            a = a * np.float32(0.1) + np.float32(1.0)
        elif name.endswith(".bias"):
            a = a * np.float32(0.02)
        elif ".in_proj." in name:
            a = a * np.float32(shape[1] ** -0.5)
        elif name.count(".block.") == 3:
            a = a * np.float32(0.3 * (shape[1] * shape[2]) ** -0.5)       # residual units
        else:
            a = a * np.float32(1.0 * (shape[1] * shape[2]) ** -0.5)       # first / strided / last conv
        sd[name] = torch.from_numpy(a)
    return sd


def all_keys(cfg: ParlerTTSConfig) -> Dict[str, tuple]:
    keys = dict(t5_keys(cfg.text_encoder))
    keys["embed_prompts.weight"] = (cfg.vocab_size, cfg.decoder.hidden_size)
    if cfg.text_encoder.d_model != cfg.decoder.hidden_size:
        keys["enc_to_dec_proj.weight"] = (cfg.decoder.hidden_size, cfg.text_encoder.d_model)
        keys["enc_to_dec_proj.bias"] = (cfg.decoder.hidden_size,)
    keys.update(decoder_keys(cfg.decoder))
    keys.update(dac_decoder_keys(cfg.audio_encoder))
    return keys


# ----------------------------------------------------------------------------------------------
# synthetic weights
# ----------------------------------------------------------------------------------------------

def synthetic_state_dict(
    cfg: ParlerTTSConfig,
    seed: int = 0,
    never_eos: bool = True,
    ln_jitter: float = 0.0,
    parts: Iterable[str] = ("text_encoder", "decoder", "audio_encoder"),
) -> Dict[str, torch.Tensor]:
    """Seeded random-init fp32 state dict (CPU tensors) in the reference key layout.

    * decoder / embed_prompts / enc_to_dec_proj: normal(0, initializer_factor) (M:517-526);
      LayerNorm 1/0 (+ ``ln_jitter`` * normal to exercise the affine path in tests)
    * text encoder: upstream T5 initialiser scales (factor 1.0)
    * codec decoder: fan-in scaled normals, snake alpha ~ 1
    * ``never_eos``: LM-head rows >= codebook_size are zeroed, so the heads can only emit valid codec
      ids, never EOS: every run lasts exactly ``max_length - 1`` steps (SURVEY.md section 8a note (v))
    Each tensor draws from its own PCG64 stream keyed by (seed, key index) so that a subset
    (``parts``) reproduces the same values as the full dict.
    """
    parts = set(parts)
    keys = all_keys(cfg)
    dec = cfg.decoder
    std = dec.initializer_factor
    sd: Dict[str, torch.Tensor] = {}

    def draw(idx: int, shape, scale: float, mean: float = 0.0) -> torch.Tensor:
        rng = np.random.Generator(np.random.PCG64([seed, idx]))
        a = rng.standard_normal(int(np.prod(shape)), dtype=np.float32)
        if scale != 1.0:
            a *= np.float32(scale)
        if mean != 0.0:
            a += np.float32(mean)
        return torch.from_numpy(a.reshape(shape))

    t5 = cfg.text_encoder
    for idx, (name, shape) in enumerate(keys.items()):
        top = name.split(".")[0]
        part = "decoder" if top in ("decoder", "embed_prompts", "enc_to_dec_proj") else top
        if part not in parts:
            continue
        if name.endswith("embed_positions.weights"):
            sd[name] = sinusoid_table(*shape)
        elif top == "text_encoder":
            if name.endswith("layer_norm.weight"):
                sd[name] = draw(idx, shape, ln_jitter, 1.0) if ln_jitter else torch.ones(shape)
            elif name.endswith("shared.weight"):
                sd[name] = draw(idx, shape, 1.0)
            elif ".SelfAttention.q." in name:
                sd[name] = draw(idx, shape, (t5.d_model * t5.d_kv) ** -0.5)
            elif ".SelfAttention.k." in name or ".SelfAttention.v." in name:
                sd[name] = draw(idx, shape, t5.d_model ** -0.5)
            elif ".SelfAttention.o." in name:
                sd[name] = draw(idx, shape, (t5.num_heads * t5.d_kv) ** -0.5)
            elif "relative_attention_bias" in name:
                sd[name] = draw(idx, shape, t5.d_model ** -0.5)
            elif ".wo." in name:
                sd[name] = draw(idx, shape, t5.d_ff ** -0.5)
            else:  # wi, wi_0, wi_1
                sd[name] = draw(idx, shape, t5.d_model ** -0.5)
        elif top == "audio_encoder":
            # gains keep the random conv stack from saturating the final tanh (pre-tanh rms ~0.5),
            # so that waveform parity checks are sensitive
            if name.endswith(".alpha"):
                sd[name] = draw(idx, shape, 0.1, 1.0)
            elif name.endswith("codebook.weight"):
                sd[name] = draw(idx, shape, 1.0)
            elif name.endswith(".bias"):
                sd[name] = draw(idx, shape, 0.02)
            elif "quantizer" in name:
                sd[name] = draw(idx, shape, 0.5 * shape[1] ** -0.5)
            elif ".block.1.weight" in name and name.count(".block.") == 1:
                # transposed conv [in, out, k]: each output sees in * k / stride = 2 * in taps
                sd[name] = draw(idx, shape, 1.0 * (shape[0] * 2) ** -0.5)
            elif name.count(".block.") == 2:
                sd[name] = draw(idx, shape, 0.3 * (shape[1] * shape[2]) ** -0.5)   # residual units
            else:
                sd[name] = draw(idx, shape, 0.75 * (shape[1] * shape[2]) ** -0.5)   # first / last conv
        else:
            if "layer_norm" in name:
                if name.endswith(".weight"):
                    sd[name] = draw(idx, shape, ln_jitter, 1.0) if ln_jitter else torch.ones(shape)
                else:
                    sd[name] = draw(idx, shape, ln_jitter) if ln_jitter else torch.zeros(shape)
            else:
                sd[name] = draw(idx, shape, std)
                if never_eos and name.startswith("decoder.lm_heads."):
                    sd[name][cfg.audio_encoder.codebook_size:] = 0.0
    return sd


# ----------------------------------------------------------------------------------------------
# checkpoint loading
# ----------------------------------------------------------------------------------------------

def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``weight = g * v / ||v||`` with the norm over all dims but 0 (torch weight_norm dim=0).

    For Conv1d dim 0 is the output channel; for ConvTranspose1d it is the *input* channel - which is
    what descript's ``WNConvTranspose1d`` does as well, so one rule serves both.  Accepts both the
    legacy (``weight_g``/``weight_v``) and the parametrize (``parametrizations.weight.original0/1``) names.
    """
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(".weight_v") or k.endswith(".parametrizations.weight.original1"):
            if k.endswith(".weight_v"):
                base, gk = k[: -len(".weight_v")], k[: -len(".weight_v")] + ".weight_g"
            else:
                base = k[: -len(".parametrizations.weight.original1")]
                gk = base + ".parametrizations.weight.original0"
            g = sd[gk].float()
            vf = v.float()
            norm = vf.reshape(vf.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (vf.dim() - 1))
            out[base + ".weight"] = g * vf / norm
        elif k.endswith(".weight_g") or k.endswith(".parametrizations.weight.original0"):
            continue
        else:
            out[k] = v
    return out


def load_state_dict_from_dir(path: str) -> Dict[str, torch.Tensor]:
    """Read every ``*.safetensors`` shard (or ``pytorch_model.bin``) under a local checkpoint dir."""
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    sd: Dict[str, torch.Tensor] = {}
    if files:
        from safetensors.torch import load_file

        for f in files:
            sd.update(load_file(os.path.join(path, f)))
    elif os.path.exists(os.path.join(path, "pytorch_model.bin")):
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    else:
        raise FileNotFoundError(f"no *.safetensors or pytorch_model.bin under {path}")
    return sd


def normalize_state_dict(sd: Dict[str, torch.Tensor], cfg: ParlerTTSConfig) -> Dict[str, torch.Tensor]:
    """Bring a checkpoint's tensors to the engine's contract: fold weight-norm, add what HF omits
    (tied ``encoder.embed_tokens``; a missing sinusoid table), check shapes, report missing keys."""
    sd = fold_weight_norm(sd)
    if "text_encoder.shared.weight" not in sd and "text_encoder.encoder.embed_tokens.weight" in sd:
        sd["text_encoder.shared.weight"] = sd["text_encoder.encoder.embed_tokens.weight"]
    pos = "decoder.model.decoder.embed_positions.weights"
    if pos not in sd:
        sd[pos] = sinusoid_table(cfg.decoder.max_position_embeddings, cfg.decoder.hidden_size)
    want = all_keys(cfg)
    missing = [k for k in want if k not in sd]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, first: {missing[:5]}")
    for k, shape in want.items():
        if tuple(sd[k].shape) != tuple(shape):
            raise ValueError(f"{k}: expected shape {tuple(shape)}, found {tuple(sd[k].shape)}")
    out = {k: sd[k] for k in want}
    # codec ENCODE side (audio prompts): optional as a block - a real checkpoint carries all of it
    enc = dac_encoder_keys(cfg.audio_encoder)
    if all(k in sd for k in enc):
        for k, shape in enc.items():
            if tuple(sd[k].shape) != tuple(shape):
                raise ValueError(f"{k}: expected shape {tuple(shape)}, found {tuple(sd[k].shape)}")
            out[k] = sd[k]
    return out


def save_pretrained(path: str, cfg: ParlerTTSConfig, sd: Dict[str, torch.Tensor], generation_config=None) -> None:
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    cfg.save_pretrained(path)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(path, "model.safetensors"))
    if generation_config is not None:
        with open(os.path.join(path, "generation_config.json"), "w") as f:
            json.dump(generation_config.to_dict(), f, indent=2, sort_keys=True)
