"""ORACLE - TEST INFRASTRUCTURE ONLY, and CONTAINER-ONLY: imports the real reference from /root/reference.

The reference cannot be imported as-is under transformers 5.15; these four
in-process shims make its own ``forward`` / ``prepare_inputs_for_generation`` / delay-pattern functions
run unchanged on CPU:

1. stub modules ``dac`` / ``dac.model`` (the descript package is absent; D:7)
2. ``AutoConfig.register`` / ``AutoModel.register`` forced to ``exist_ok=True`` (M:51-52, __init__:15-16)
3. ``cross_attention_hidden_size=None`` handed to the decoder config (M:1583 reads it)
4. ``tie_weights`` accepting the kwargs transformers 5.x passes (M:1670)

Nothing here travels to the GPU box as reference code: ``gen_golden.py`` uses it to write data
fixtures (``tests/golden/*.npz``) and never copies reference source.  /root/reference does not exist
on the GPU box, so this module must only be imported by ``gen_golden.py`` and by container-side tests
that skip when the path is missing.
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import torch
import torch.nn as nn

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "parler_tts"))


_ref = None


def load_reference():
    """Import /root/reference/parler_tts under the shims; returns the module."""
    global _ref
    if _ref is not None:
        return _ref
    if not reference_available():
        raise RuntimeError("reference tree not present (this only works in the build container)")
    warnings.filterwarnings("ignore")
    # shim 1
    dac = types.ModuleType("dac")
    dac_model = types.ModuleType("dac.model")

    class DAC(nn.Module):
        def __init__(self, n_codebooks=9, latent_dim=1024, codebook_size=1024, **kw):
            super().__init__()

    dac_model.DAC = DAC
    dac.model = dac_model
    sys.modules.setdefault("dac", dac)
    sys.modules.setdefault("dac.model", dac_model)
    # shim 2
    from transformers import AutoConfig, AutoModel

    orig_c = AutoConfig.register
    orig_m = AutoModel.register.__func__
    AutoConfig.register = staticmethod(lambda model_type, config, exist_ok=False: orig_c(model_type, config, exist_ok=True))
    AutoModel.register = classmethod(lambda cls, config_class, model_class, exist_ok=False: orig_m(cls, config_class, model_class, exist_ok=True))
    # the repository root holds an alias package of the same name (parler_tts/__init__.py -> parler_tts_amd): evict it and put the
    # reference tree in front, so that `parler_tts` below IS /root/reference/parler_tts
    for name in [m for m in sys.modules if m == "parler_tts" or m.startswith("parler_tts.")]:
        if getattr(sys.modules.get("parler_tts"), "__ptts_alias__", False):
            del sys.modules[name]
    if REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    import parler_tts  # noqa: the real reference

    if not os.path.realpath(parler_tts.__file__).startswith(REFERENCE_ROOT + os.sep):
        raise RuntimeError(f"`parler_tts` resolved to {parler_tts.__file__}, not to the reference tree")

    # shim 4
    parler_tts.ParlerTTSForConditionalGeneration.tie_weights = lambda self, *a, **k: None
    _ref = parler_tts
    return _ref


def build_reference_model(cfg, state_dict):
    """Instantiate the reference's ParlerTTSForConditionalGeneration for an engine config and load an
    engine state dict (text encoder + decoder parts; the codec is a stub)."""
    ref = load_reference()
    from transformers import T5Config

    t5 = cfg.text_encoder
    t5_cfg = T5Config(
        vocab_size=t5.vocab_size, d_model=t5.d_model, d_kv=t5.d_kv, d_ff=t5.d_ff, num_layers=t5.num_layers,
        num_heads=t5.num_heads, relative_attention_num_buckets=t5.relative_attention_num_buckets,
        relative_attention_max_distance=t5.relative_attention_max_distance,
        layer_norm_epsilon=t5.layer_norm_epsilon, feed_forward_proj=t5.feed_forward_proj,
        dropout_rate=0.1, is_encoder_decoder=False, use_cache=False,
    )
    d = cfg.decoder
    dec_cfg = ref.ParlerTTSDecoderConfig(
        vocab_size=d.vocab_size, max_position_embeddings=d.max_position_embeddings,
        num_hidden_layers=d.num_hidden_layers, ffn_dim=d.ffn_dim, num_attention_heads=d.num_attention_heads,
        layerdrop=d.layerdrop, use_cache=True, activation_function=d.activation_function,
        hidden_size=d.hidden_size, dropout=d.dropout, attention_dropout=d.attention_dropout,
        activation_dropout=d.activation_dropout, pad_token_id=d.pad_token_id, eos_token_id=d.eos_token_id,
        bos_token_id=d.bos_token_id, num_codebooks=d.num_codebooks,
        cross_attention_hidden_size=None,  # shim 3
    )
    a = cfg.audio_encoder
    dac_cfg = ref.DACConfig(num_codebooks=a.num_codebooks, codebook_size=a.codebook_size, latent_dim=a.latent_dim,
                            frame_rate=a.frame_rate, sampling_rate=a.sampling_rate)
    full = ref.ParlerTTSConfig.from_sub_models_config(t5_cfg, dac_cfg, dec_cfg, vocab_size=cfg.vocab_size)
    model = ref.ParlerTTSForConditionalGeneration(full).eval()
    sd = {k: v for k, v in state_dict.items() if not k.startswith("audio_encoder.")}
    if "text_encoder.shared.weight" in sd:
        sd["text_encoder.encoder.embed_tokens.weight"] = sd["text_encoder.shared.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.startswith("audio_encoder.")]
    if missing or unexpected:
        raise RuntimeError(f"state-dict mismatch vs the reference classes: missing={missing[:5]} unexpected={unexpected[:5]}")
    return model


@torch.no_grad()
def reference_generate_ids(model, input_ids, attention_mask, prompt_input_ids, prompt_attention_mask,
                           max_length, do_sample=False, temperature=1.0, top_k=50, guidance_scale=None,
                           generator=None, record_logits=0, forced_tokens=None, decoder_input_ids=None,
                           decoder_attention_mask=None, num_return_sequences=1, encoder_states=None):
    """Drive the REAL reference ``forward`` (M:1927) + ``prepare_inputs_for_generation`` (M:2081) +
    ``build/apply_delay_pattern_mask`` with the third-party greedy/sample step restated in a few lines
    (transformers 4.3x ``greedy_search``/``sample`` no longer exist).
    Then the un-delay of M:2633-2648 using the reference's own functions.
    ``record_logits``: an int n (the first n steps) or a set of step indices (long runs keep a sample).
    ``encoder_states``: text-encoder output handed in (as ``generate(encoder_outputs=...)`` allows, M:2226) instead of running
    the model's own T5 - used to put a bfloat16 decoder behind the SAME encoder states an fp32 run saw.
    The model may have been cast with ``.to(torch.bfloat16)``: logits are recorded as float32 copies of what it returns."""
    from transformers.modeling_outputs import BaseModelOutput

    dec = model.decoder.config
    K = dec.num_codebooks
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    if encoder_states is None:
        enc = model.text_encoder(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
    else:
        enc = encoder_states.clone()
    enc_t5 = enc.clone()
    use_cfg = guidance_scale is not None and guidance_scale > 1
    amask = attention_mask
    if use_cfg:  # M:2231-2237
        enc = torch.concatenate([enc, torch.zeros_like(enc)], dim=0)
        amask = torch.concatenate([amask, torch.zeros_like(amask)], dim=0)
    kw = dict(encoder_outputs=BaseModelOutput(last_hidden_state=enc), attention_mask=amask, use_cache=True,
              guidance_scale=guidance_scale)
    if prompt_input_ids is not None:
        kw["prompt_hidden_states"] = model.embed_prompts(prompt_input_ids)  # M:2243-2245
    if prompt_attention_mask is not None:
        kw["prompt_attention_mask"] = prompt_attention_mask
    ids = torch.full((B * K, 1), dec.bos_token_id, dtype=torch.long)
    if decoder_input_ids is not None:
        # the reference's own _prepare_decoder_input_ids_for_generation (M:2150-2195) prepends the start column (and a
        # ones column to decoder_attention_mask); _get_decoder_start_token_id is gone from transformers 5.x, so the
        # instance gets the one-line 4.3x body
        mk = {"decoder_input_ids": decoder_input_ids}
        if decoder_attention_mask is not None:
            mk["decoder_attention_mask"] = decoder_attention_mask
        model._get_decoder_start_token_id = lambda a=None, b=None: a if a is not None else dec.bos_token_id
        ids, mk = model._prepare_decoder_input_ids_for_generation(B, "input_ids", mk, dec.bos_token_id, dec.bos_token_id, ids.device)
        if "decoder_attention_mask" in mk:
            kw["decoder_attention_mask"] = mk["decoder_attention_mask"]
    ids, pattern = model.decoder.build_delay_pattern_mask(ids, dec.bos_token_id, dec.pad_token_id, max_length)
    if num_return_sequences > 1:
        # M:2600-2606: the third-party helper the reference calls (a static method, still present in transformers 5.15)
        # repeat_interleaves the id rows and every tensor kwarg, the stashed delay pattern included
        from transformers import GenerationMixin

        ids, kw = GenerationMixin._expand_inputs_for_generation(
            expand_size=num_return_sequences, is_encoder_decoder=True, input_ids=ids,
            decoder_delay_pattern_mask=pattern, **kw)
        pattern = kw.pop("decoder_delay_pattern_mask")
    start_ids = ids[:, :1].clone() if decoder_input_ids is None else None
    unf = torch.ones(ids.shape[0], dtype=torch.long)
    past = None
    step_logits, margins, owns, step = [], [], [], 0
    while True:
        mi = model.prepare_inputs_for_generation(ids, past_key_values=past, decoder_delay_pattern_mask=pattern, **kw)
        out = model(**mi, return_dict=True)
        past = out.past_key_values
        logits = out.logits[:, -1, :].float()        # transformers >= 4.4x selects on an fp32 copy of the last row
        if use_cfg:
            n = logits.shape[0] // 2
            logits = logits[n:] + (logits[:n] - logits[n:]) * guidance_scale
        if (step in record_logits) if isinstance(record_logits, (set, frozenset)) else (step < record_logits):
            step_logits.append(logits.clone())
        if do_sample:
            s = logits / temperature if temperature != 1.0 else logits
            if top_k:
                thr = torch.topk(s, min(top_k, s.shape[-1]))[0][..., -1, None]
                s = s.masked_fill(s < thr, -float("inf"))
            nxt = torch.multinomial(torch.softmax(s, dim=-1), 1, generator=generator).squeeze(1)
        else:
            t2 = torch.topk(logits, 2, dim=-1)[0]
            margins.append(t2[:, 0] - t2[:, 1])
            nxt = torch.argmax(logits, dim=-1)
            owns.append(nxt.clone())
        if forced_tokens is not None and step in forced_tokens:
            nxt = forced_tokens[step].clone()
        nxt = nxt * unf + dec.pad_token_id * (1 - unf)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        if "decoder_attention_mask" in kw:   # third-party _update_model_kwargs_for_generation (4.3x): a column of ones per token
            dm = kw["decoder_attention_mask"]
            kw["decoder_attention_mask"] = torch.cat([dm, dm.new_ones((dm.shape[0], 1))], dim=-1)
        unf = unf * (nxt != dec.eos_token_id).long()
        step += 1
        if int(unf.max()) == 0 or ids.shape[-1] >= max_length:
            break
    # M:2633-2648
    out_ids = model.decoder.apply_delay_pattern_mask(ids, pattern)
    start = torch.full((ids.shape[0], 1), dec.bos_token_id, dtype=torch.long)     # M:2637 passes the (expanded) start ids
    _, m = model.decoder.build_delay_pattern_mask(start, dec.bos_token_id, dec.pad_token_id, out_ids.shape[1])
    keep = (m != dec.bos_token_id) & (m != dec.pad_token_id)
    codes = out_ids[keep].reshape(B, K, -1)                                         # M:2645: the ORIGINAL batch size
    res = {"raw_ids": ids, "pattern": pattern, "codes": codes, "enc_t5": enc_t5}
    if step_logits:
        res["step_logits"] = torch.stack(step_logits)
    if margins:
        res["margins"] = torch.stack(margins)
        res["own"] = torch.stack(owns)        # the model's own argmax of every step, before any forcing
    return res


@torch.no_grad()
def reference_causal_generate_ids(causal, encoder_hidden_states, encoder_attention_mask, prompt_hidden_states,
                                  prompt_attention_mask, max_length, record_logits=0, guidance_scale=None):
    """The decoder-only driver (M:1281-1552) on the REAL ``ParlerTTSForCausalLM``: its ``forward`` (M:1096-1185) and
    ``prepare_inputs_for_generation`` (M:1187-1247) with the third-party greedy step restated as in
    ``reference_generate_ids``; un-delay per M:1532-1546."""
    dec = causal.config
    K = dec.num_codebooks
    use_cfg = guidance_scale is not None and guidance_scale > 1
    # under guidance the driver doubles only the decoder ids and the prompt (M:1218-1232): the caller passes encoder states
    # with 2 x batch rows; without encoder states the layers skip their cross-attention block (M:465)
    B = prompt_hidden_states.shape[0] if prompt_hidden_states is not None else encoder_hidden_states.shape[0] // (2 if use_cfg else 1)
    kw = dict(encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=encoder_attention_mask,
              prompt_hidden_states=prompt_hidden_states, prompt_attention_mask=prompt_attention_mask, use_cache=True,
              guidance_scale=guidance_scale)
    ids = torch.full((B * K, 1), dec.bos_token_id, dtype=torch.long)
    ids, pattern = causal.build_delay_pattern_mask(ids, dec.bos_token_id, dec.pad_token_id, max_length)
    unf = torch.ones(B * K, dtype=torch.long)
    past, step_logits, margins, step = None, [], [], 0
    while True:
        mi = causal.prepare_inputs_for_generation(ids, past_key_values=past, delay_pattern_mask=pattern, **kw)
        out = causal(**mi, return_dict=True)
        past = out.past_key_values
        logits = out.logits[:, -1, :]
        if use_cfg:          # ClassifierFreeGuidanceLogitsProcessor, appended at M:1475-1477
            n = logits.shape[0] // 2
            logits = logits[n:] + (logits[:n] - logits[n:]) * guidance_scale
        if step < record_logits:
            step_logits.append(logits.clone())
        t2 = torch.topk(logits, 2, dim=-1)[0]
        margins.append(t2[:, 0] - t2[:, 1])
        nxt = torch.argmax(logits, dim=-1)
        nxt = nxt * unf + dec.pad_token_id * (1 - unf)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unf = unf * (nxt != dec.eos_token_id).long()
        step += 1
        if int(unf.max()) == 0 or ids.shape[-1] >= max_length:
            break
    out_ids = causal.apply_delay_pattern_mask(ids, pattern)
    start = torch.full((B * K, 1), dec.bos_token_id, dtype=torch.long)
    _, m = causal.build_delay_pattern_mask(start, dec.bos_token_id, dec.pad_token_id, out_ids.shape[1])
    keep = (m != dec.bos_token_id) & (m != dec.pad_token_id)
    return {"raw_ids": ids, "pattern": pattern, "codes": out_ids[keep].reshape(B, K, -1),
            "step_logits": torch.stack(step_logits), "margins": torch.stack(margins)}


def build_hf_dac(cfg, state_dict):
    """installed transformers DacModel (structural stand-in for descript DAC) with engine weights."""
    from transformers import DacConfig, DacModel

    a = cfg.audio_encoder
    hf_cfg = DacConfig(n_codebooks=a.num_codebooks, codebook_size=a.codebook_size, codebook_dim=a.codebook_dim,
                       hidden_size=a.latent_dim, decoder_hidden_size=a.decoder_hidden_size,
                       upsampling_ratios=list(a.upsampling_ratios), downsampling_ratios=list(reversed(a.upsampling_ratios)),
                       encoder_hidden_size=getattr(a, "encoder_hidden_size", max(a.decoder_hidden_size // 2 ** len(a.upsampling_ratios), 8)))
    m = DacModel(hf_cfg).eval()
    p = "audio_encoder.model."
    own = m.state_dict()
    new = {}
    for i in range(a.num_codebooks):
        new[f"quantizer.quantizers.{i}.codebook.weight"] = state_dict[f"{p}quantizer.quantizers.{i}.codebook.weight"]
        new[f"quantizer.quantizers.{i}.out_proj.weight"] = state_dict[f"{p}quantizer.quantizers.{i}.out_proj.weight"]
        new[f"quantizer.quantizers.{i}.out_proj.bias"] = state_dict[f"{p}quantizer.quantizers.{i}.out_proj.bias"]
    new["decoder.conv1.weight"] = state_dict[p + "decoder.model.0.weight"]
    new["decoder.conv1.bias"] = state_dict[p + "decoder.model.0.bias"]
    R = len(a.upsampling_ratios)
    for n in range(1, R + 1):
        s, d = f"{p}decoder.model.{n}.block.", f"decoder.block.{n - 1}."
        new[d + "snake1.alpha"] = state_dict[s + "0.alpha"]
        new[d + "conv_t1.weight"] = state_dict[s + "1.weight"]
        new[d + "conv_t1.bias"] = state_dict[s + "1.bias"]
        for r in (2, 3, 4):
            u, e = f"{s}{r}.block.", f"{d}res_unit{r - 1}."
            new[e + "snake1.alpha"] = state_dict[u + "0.alpha"]
            new[e + "conv1.weight"] = state_dict[u + "1.weight"]
            new[e + "conv1.bias"] = state_dict[u + "1.bias"]
            new[e + "snake2.alpha"] = state_dict[u + "2.alpha"]
            new[e + "conv2.weight"] = state_dict[u + "3.weight"]
            new[e + "conv2.bias"] = state_dict[u + "3.bias"]
    if p + "encoder.block.0.weight" in state_dict:      # encode side (audio prompts)
        new["encoder.conv1.weight"] = state_dict[p + "encoder.block.0.weight"]
        new["encoder.conv1.bias"] = state_dict[p + "encoder.block.0.bias"]
        for n in range(1, R + 1):
            s_, d_ = f"{p}encoder.block.{n}.block.", f"encoder.block.{n - 1}."
            for r in (0, 1, 2):
                u, e = f"{s_}{r}.block.", f"{d_}res_unit{r + 1}."
                new[e + "snake1.alpha"] = state_dict[u + "0.alpha"]
                new[e + "conv1.weight"] = state_dict[u + "1.weight"]
                new[e + "conv1.bias"] = state_dict[u + "1.bias"]
                new[e + "snake2.alpha"] = state_dict[u + "2.alpha"]
                new[e + "conv2.weight"] = state_dict[u + "3.weight"]
                new[e + "conv2.bias"] = state_dict[u + "3.bias"]
            new[d_ + "snake1.alpha"] = state_dict[s_ + "3.alpha"]
            new[d_ + "conv1.weight"] = state_dict[s_ + "4.weight"]
            new[d_ + "conv1.bias"] = state_dict[s_ + "4.bias"]
        new["encoder.snake1.alpha"] = state_dict[f"{p}encoder.block.{R + 1}.alpha"]
        new["encoder.conv2.weight"] = state_dict[f"{p}encoder.block.{R + 2}.weight"]
        new["encoder.conv2.bias"] = state_dict[f"{p}encoder.block.{R + 2}.bias"]
        for i in range(a.num_codebooks):
            new[f"quantizer.quantizers.{i}.in_proj.weight"] = state_dict[f"{p}quantizer.quantizers.{i}.in_proj.weight"]
            new[f"quantizer.quantizers.{i}.in_proj.bias"] = state_dict[f"{p}quantizer.quantizers.{i}.in_proj.bias"]
    new["decoder.snake1.alpha"] = state_dict[f"{p}decoder.model.{R + 1}.alpha"]
    new["decoder.conv2.weight"] = state_dict[f"{p}decoder.model.{R + 2}.weight"]
    new["decoder.conv2.bias"] = state_dict[f"{p}decoder.model.{R + 2}.bias"]
    for k, v in new.items():
        if k not in own:
            raise KeyError(f"HF DacModel has no parameter {k}")
        if tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f"{k}: HF shape {tuple(own[k].shape)} vs engine {tuple(v.shape)}")
    m.load_state_dict(new, strict=False)
    return m
