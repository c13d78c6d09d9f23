"""Python API surface of the reference, driving the MI355X HIP engine.

``ParlerTTSForConditionalGeneration.generate()`` keeps the reference's signature and semantics
(/root/reference/parler_tts/modeling_parler_tts.py:2342-2686): description ids -> T5 encoder (once) ->
autoregressive multi-codebook decoder under the delay pattern -> DAC decode -> waveform.  The token
loop (decoder forward + selection + EOS latch, i.e. M:2081-2148, M:1927-2079 and the third-party
greedy/sample step) runs entirely inside ``libparler_tts_hip.so``; PyTorch only owns device buffers
and the current stream.  Nothing here falls back to CPU or to the test oracle.
"""
from __future__ import annotations

import copy
import ctypes as C
import json
import math
import os
from typing import Any, Dict, Optional

import torch

from . import _hip
from .configuration import DACConfig, GenerationConfig, ParlerTTSConfig, ParlerTTSDecoderConfig
from .delay_pattern import apply_delay_pattern_mask, build_delay_pattern_mask
from .weights import load_state_dict_from_dir, normalize_state_dict, synthetic_state_dict

_ACT = {"gelu": 1, "gelu_new": 2, "relu": 3}


class ModelOutput(dict):
    """Minimal attribute-dict standing in for HF ModelOutput (only what generate() returns)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def t5_relative_buckets(T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bucket of every offset (key - query) in [-(T-1), T-1], bidirectional T5 rule
    (HF models/t5/modeling_t5.py ``_relative_position_bucket``).  Host-side integer table."""
    rel = torch.arange(-(T - 1), T)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


class HipEngine:
    """Owns one ``ptts_engine`` handle (weights packed in HBM, KV pools, workspaces)."""

    def __init__(self, cfg: ParlerTTSConfig, state_dict: Dict[str, torch.Tensor], dtype: str, device: torch.device,
                 max_batch: int, max_desc_len: int, max_prompt_len: int, max_length: int, with_t5: bool = True,
                 codec_only: bool = False):
        self.lib = _hip.load_library()
        if device.type != "cuda":
            raise RuntimeError("the HIP engine needs a ROCm device (torch device type 'cuda')")
        self.device = device
        self.dtype = dtype
        d, t5 = cfg.decoder, cfg.text_encoder
        self.codec_only = codec_only
        self.has_codec = any(k.startswith("audio_encoder.model.decoder.") for k in state_dict)
        self.has_codec_encoder = False
        if codec_only or not any(k.startswith("text_encoder.") for k in state_dict):
            with_t5 = False          # no text-encoder tensors: callers pass `encoder_outputs`
        c = _hip.PttsConfig()
        c.abi_version = _hip.ABI_VERSION
        c.dtype = _hip.DTYPE_BF16 if dtype == "bf16" else _hip.DTYPE_F32
        c.hidden, c.layers, c.heads, c.ffn = d.hidden_size, d.num_hidden_layers, d.num_attention_heads, d.ffn_dim
        c.vocab, c.codebooks, c.max_positions = d.vocab_size, d.num_codebooks, d.max_position_embeddings
        c.pad_token_id, c.eos_token_id, c.bos_token_id = d.pad_token_id, d.eos_token_id, d.bos_token_id
        c.prompt_vocab, c.enc_hidden = cfg.vocab_size, t5.d_model
        c.t5_layers = t5.num_layers if with_t5 else 0
        c.t5_heads, c.t5_dkv, c.t5_dff, c.t5_vocab = t5.num_heads, t5.d_kv, t5.d_ff, t5.vocab_size
        c.t5_gated = 1 if t5.is_gated_act else 0
        c.t5_act = _ACT[t5.dense_act_fn]
        c.t5_eps = t5.layer_norm_epsilon
        c.max_batch, c.max_desc_len, c.max_prompt_len, c.max_length = max_batch, max_desc_len, max_prompt_len, max_length
        c.dec_variant = 0 if codec_only else int(getattr(d, "variant_flags", 0))     # RMSNorm / RoPE / SwiGLU extensions
        if codec_only:   # a placeholder decoder: only ptts_dac_decode will be called on this handle
            c.hidden, c.layers, c.heads, c.ffn, c.vocab, c.max_positions = 64, 0, 1, 32, 16, 64
            c.prompt_vocab, c.enc_hidden, c.max_batch, c.max_desc_len, c.max_prompt_len, c.max_length = 16, 32, 1, 1, 0, 2
        if self.has_codec:
            a = cfg.audio_encoder
            c.dac_codebook_size, c.dac_codebook_dim, c.dac_latent = a.codebook_size, a.codebook_dim, a.latent_dim
            c.dac_hidden, c.dac_n_up = a.decoder_hidden_size, len(a.upsampling_ratios)
            for i, r in enumerate(a.upsampling_ratios):
                c.dac_up[i] = int(r)
            # encode side (audio prompts) only when its tensors came with the checkpoint
            self.has_codec_encoder = "audio_encoder.model.encoder.block.0.weight" in state_dict
            c.dac_enc_hidden = int(getattr(a, "encoder_hidden_size", 64)) if self.has_codec_encoder else 0
        self.limits = (max_batch, max_desc_len, max_prompt_len, max_length)
        self.cfg = cfg
        self.handle = C.c_void_p()
        with torch.cuda.device(device):
            _hip.check(self.lib.ptts_create(C.byref(c), C.byref(self.handle)))
            self._upload(state_dict, with_t5)
        self.rel_bias_weight = None
        if with_t5:
            self.rel_bias_weight = state_dict[
                "text_encoder.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float()

    def _upload(self, sd: Dict[str, torch.Tensor], with_t5: bool) -> None:
        n = 0
        staged = []   # the fp32 staging copies must outlive the asynchronous pack kernels that read them
        for name, t in sd.items():
            if name.startswith("audio_encoder.") and not self.has_codec:
                continue
            if self.codec_only and not name.startswith("audio_encoder."):
                continue
            if name.startswith("text_encoder.") and not with_t5:
                continue
            g = t.to(device=self.device, dtype=torch.float32).contiguous()
            shape = (C.c_int64 * g.dim())(*g.shape)
            _hip.check(self.lib.ptts_load_tensor(self.handle, name.encode(), _ptr(g), shape, g.dim(), _stream()))
            staged.append(g)
            n += g.numel()
            if n > 200_000_000:  # bound the transient fp32 staging copies
                torch.cuda.synchronize(self.device)
                staged.clear()
                n = 0
        if not self.codec_only and getattr(self.cfg.decoder, "position_embedding", "sinusoidal") == "rope":
            from .weights import rope_tables

            for nm, t in zip(("decoder.rope.cos", "decoder.rope.sin"), rope_tables(self.cfg.decoder)):
                g = t.to(device=self.device, dtype=torch.float32).contiguous()
                shape = (C.c_int64 * 2)(*g.shape)
                _hip.check(self.lib.ptts_load_tensor(self.handle, nm.encode(), _ptr(g), shape, 2, _stream()))
                staged.append(g)
        torch.cuda.synchronize(self.device)
        staged.clear()
        if not self.codec_only:
            _hip.check(self.lib.ptts_weights_ready(self.handle))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.ptts_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thin wrappers ----
    def t5_encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        t5 = self.cfg.text_encoder
        B, T = input_ids.shape
        # bucketed relative-position bias [H][2T-1]: integer table on the host, gathered once per description length and kept on the
        # device (it was rebuilt and copied host -> device at every call: a synchronous copy in front of the time to first token)
        cache = self.__dict__.setdefault("_rel_bias_tables", {})
        table = cache.get(T)
        if table is None:
            buckets = t5_relative_buckets(T, t5.relative_attention_num_buckets, t5.relative_attention_max_distance)
            table = cache[T] = self.rel_bias_weight[buckets].t().contiguous().to(self.device)
            if len(cache) > 64:
                cache.pop(next(iter(cache)))
        ids = input_ids.to(self.device, torch.int64).contiguous()
        mask = attention_mask.to(self.device, torch.int32).contiguous()
        out = torch.empty(B, T, t5.d_model, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):       # streams, lazy allocations and launches follow the CURRENT device
            _hip.check(self.lib.ptts_t5_encode(self.handle, B, T, _ptr(ids), _ptr(mask), _ptr(table), _ptr(out), _stream()))
            if _stream().value:
                # asynchronous on the caller's stream: the temporaries stay referenced until the next call has queued its own work
                # behind this one (same stream: by then the kernels that read them are done or ordered in front)
                self._t5_keep = (ids, mask, table)
                ids.record_stream(torch.cuda.current_stream()); mask.record_stream(torch.cuda.current_stream())
            else:
                torch.cuda.current_stream().synchronize()  # NULL stream: the engine call was host-synchronous anyway
        return out

    def dac_decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, K, F] int64 -> waveform [B, F * hop] fp32 (ptts_dac_decode)."""
        B, K, F = codes.shape
        hop = self.cfg.audio_encoder.hop_length
        codes = codes.to(self.device, torch.int64).contiguous()
        # ids the LM heads can emit but the codebooks do not hold (codebook_size .. vocab_size-1, SURVEY 8a note (v)): the
        # reference fails in the codebook embedding lookup with an IndexError; the kernel must never see them
        if codes.numel() and (int(codes.max()) >= self.cfg.audio_encoder.codebook_size or int(codes.min()) < 0):
            raise IndexError("index out of range in self: codec ids must be in [0, codebook_size)")
        out = torch.empty(B, F * hop, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.ptts_dac_decode(self.handle, B, F, _ptr(codes), _ptr(out), _stream()))
            torch.cuda.current_stream().synchronize()   # `codes` may be a temporary
        return out

    def dac_encode(self, audio: torch.Tensor) -> torch.Tensor:
        """audio [B, T] fp32 -> codes [B, K, T // hop] int64 (ptts_dac_encode)."""
        if not self.has_codec_encoder:
            raise RuntimeError("this checkpoint holds no codec encoder weights (audio_encoder.model.encoder.*)")
        B, T = audio.shape
        a = self.cfg.audio_encoder
        if T < a.hop_length:
            raise ValueError(f"audio prompt of {T} samples is shorter than one codec frame ({a.hop_length})")
        audio = audio.to(self.device, torch.float32).contiguous()
        codes = torch.empty(B, a.num_codebooks, T // a.hop_length, device=self.device, dtype=torch.int64)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.ptts_dac_encode(self.handle, B, T, _ptr(audio), _ptr(codes), _stream()))
            torch.cuda.current_stream().synchronize()   # `audio` may be a temporary
        return codes

    def prefill(self, B, B_eff, enc_states, attention_mask, prompt_ids, prompt_mask, pattern, raw_ids, start_len,
                prompt_embeds=None, dec_mask=None):
        T = enc_states.shape[1]
        P = prompt_embeds.shape[1] if prompt_embeds is not None else (0 if prompt_ids is None else prompt_ids.shape[1])
        L = pattern.shape[1]
        self._keep = (enc_states, attention_mask, prompt_ids, prompt_mask, pattern, raw_ids, prompt_embeds, dec_mask)  # raw pointers
        if dec_mask is not None:           # decoder_attention_mask over the prefix columns (M:845-846)
            _hip.check(self.lib.ptts_set_decoder_mask(self.handle, _ptr(dec_mask)))
        if prompt_embeds is not None:      # pre-embedded prompts override the id lookup for this prefill (M:2476)
            _hip.check(self.lib.ptts_set_prompt_embeds(self.handle, _ptr(prompt_embeds)))
        _hip.check(self.lib.ptts_prefill(self.handle, B, B_eff, T, P, L, _ptr(enc_states), _ptr(attention_mask),
                                         _ptr(prompt_ids), _ptr(prompt_mask), _ptr(pattern), _ptr(raw_ids), start_len,
                                         _stream()))

    def decode_steps(self, n: int, sampling: _hip.PttsSampling, profile_every: int = 0) -> None:
        _hip.check(self.lib.ptts_decode_steps(self.handle, n, C.byref(sampling), profile_every, _stream()))

    def state(self) -> _hip.PttsState:
        st = _hip.PttsState()
        _hip.check(self.lib.ptts_get_state(self.handle, C.byref(st), _stream()))
        return st

    def state_snapshot(self, slot: int) -> None:
        """Enqueue a copy of the loop state into pinned host slot `slot` behind the work already on the stream (no wait)."""
        _hip.check(self.lib.ptts_state_snapshot(self.handle, slot, _stream()))

    def state_poll(self, slot: int) -> _hip.PttsState:
        """The snapshot of `slot`: waits for that copy only, later work on the stream keeps running."""
        st = _hip.PttsState()
        _hip.check(self.lib.ptts_state_poll(self.handle, slot, C.byref(st)))
        return st

    def last_logits(self, rows: int, probs: bool = False) -> torch.Tensor:
        """Copy of the last step's logits [rows][vocab] (device -> new torch tensor)."""
        lp, pp = C.c_void_p(), C.c_void_p()
        _hip.check(self.lib.ptts_last_logits(self.handle, C.byref(lp), C.byref(pp)))
        V = self.cfg.decoder.vocab_size
        Vp = (V + 15) // 16 * 16
        out = torch.empty(rows, Vp, device=self.device, dtype=torch.float32)
        torch.cuda.current_stream().synchronize()
        src = pp if probs else lp
        rc = _hipMemcpyDtoD(out.data_ptr(), src.value, rows * Vp * 4)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
        return out[:, :V].contiguous()

    def forward_step(self) -> None:
        _hip.check(self.lib.ptts_forward_step(self.handle, _stream()))

    def select_step(self, sampling: _hip.PttsSampling) -> None:
        _hip.check(self.lib.ptts_select_step(self.handle, C.byref(sampling), _stream()))

    def set_logits(self, scores: torch.Tensor) -> None:
        """Overwrite the first rows of the engine's logits buffer with `scores` [rows][vocab] (processed scores)."""
        lp, pp = C.c_void_p(), C.c_void_p()
        _hip.check(self.lib.ptts_last_logits(self.handle, C.byref(lp), C.byref(pp)))
        V = self.cfg.decoder.vocab_size
        Vp = (V + 15) // 16 * 16
        buf = torch.full((scores.shape[0], Vp), float("-inf"), device=self.device, dtype=torch.float32)
        buf[:, :V] = scores.to(self.device, torch.float32)
        torch.cuda.current_stream().synchronize()
        rc = _hipMemcpyDtoD(lp.value, buf.data_ptr(), buf.numel() * 4)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")

    def set_forced_tokens(self, tokens: Optional[torch.Tensor]) -> None:
        if tokens is None:
            _hip.check(self.lib.ptts_set_forced_tokens(self.handle, None, 0))
            return
        t = tokens.to("cpu", torch.int64).contiguous()
        arr = (C.c_int64 * t.numel())(*t.tolist())
        _hip.check(self.lib.ptts_set_forced_tokens(self.handle, arr, t.numel()))

    def set_forced_sequence(self, forced: Optional[torch.Tensor], own: Optional[torch.Tensor] = None) -> None:
        """Device-side teacher forcing (ptts_set_forced_sequence): `forced` [B*K, L] int64 on the engine's device
        (values < 0 leave a column alone), `own` receives the engine's own choices.  The caller keeps both alive."""
        self._forced_keep = (forced, own)
        _hip.check(self.lib.ptts_set_forced_sequence(self.handle, _ptr(forced), _ptr(own)))

    def set_block_table(self, table: torch.Tensor) -> None:
        t = table.to("cpu", torch.int32).contiguous()
        arr = (C.c_int32 * t.numel())(*t.flatten().tolist())
        _hip.check(self.lib.ptts_set_block_table(self.handle, arr, t.shape[0], t.shape[1]))

    def pool_info(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int64()
        _hip.check(self.lib.ptts_pool_info(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def profile_read(self, reset: bool = True) -> _hip.PttsProfile:
        p = _hip.PttsProfile()
        _hip.check(self.lib.ptts_profile_read(self.handle, C.byref(p), 1 if reset else 0))
        return p


_hiprt = None


_side_streams: Dict[int, "torch.cuda.ExternalStream"] = {}


def _own_stream(device: torch.device) -> "torch.cuda.ExternalStream":
    """ONE non-blocking HIP stream PER DEVICE AND PROCESS, created directly and handed to torch as an ExternalStream
    (`torch.cuda.Stream()` draws from torch's pool, which creates 32 streams per priority level on first use; the engine needs
    one queue, not 64).  Shared by every model of the process on that device and never destroyed: a raw stream per model was
    leaked at `model.to(device)` / model deletion, and destroying one under torch's caching allocator - which keys
    its free blocks by stream - is not safe.  generate() is host-driven and not thread-safe per engine, so models on one device
    share the queue without losing anything.
    Note for SHARED cards (several processes on one GPU - not the deployment, one process per GPU is): a process that keeps its
    queue full through this stream shares badly - the 2-rank shared-GPU rehearsal of the benchmark takes 14.9-18.0 s per pass
    against 6.2 s on the host-synchronous NULL-stream path, which drains the queue every few steps.  `PTTS_NO_SIDE_STREAM=1`
    selects that path (bench.py sets it under PTTS_SHARE_GPU)."""
    global _hiprt
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx in _side_streams:
        return _side_streams[idx]
    if _hiprt is None:
        _hiprt = C.CDLL("libamdhip64.so")
        _hiprt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hiprt.hipMemcpy.restype = C.c_int
    _hiprt.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    _hiprt.hipStreamCreateWithFlags.restype = C.c_int
    h = C.c_void_p()
    with torch.cuda.device(device):
        rc = _hiprt.hipStreamCreateWithFlags(C.byref(h), 1)        # hipStreamNonBlocking
    if rc != 0 or not h.value:
        raise RuntimeError(f"hipStreamCreateWithFlags failed: {rc}")
    _side_streams[idx] = torch.cuda.ExternalStream(h.value, device=device)
    return _side_streams[idx]


def _hipMemcpyDtoD(dst: int, src: int, nbytes: int) -> int:
    global _hiprt
    if _hiprt is None:
        _hiprt = C.CDLL("libamdhip64.so")
        _hiprt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hiprt.hipMemcpy.restype = C.c_int
    return _hiprt.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), nbytes, 3)  # hipMemcpyDeviceToDevice


def _warp_scores(scores: torch.Tensor, gen) -> torch.Tensor:
    """What the third-party `sample` loop stores as a step's scores: the processed scores after its warpers, in their order -
    TemperatureLogitsWarper, TopKLogitsWarper (threshold = the k-th largest value, ties kept), TopPLogitsWarper (ascending
    cumulative probability <= 1 - top_p removed, at least one token kept).  Host-side copy of what select_kernel applies on the
    device; only used to REPORT scores (`output_scores=True`), never to choose a token."""
    s = scores.float()
    if gen.temperature is not None and float(gen.temperature) != 1.0:
        s = s / float(gen.temperature)
    if gen.top_k is not None and int(gen.top_k) > 0:
        k = min(int(gen.top_k), s.shape[-1])
        thr = torch.topk(s, k)[0][..., -1, None]
        s = s.masked_fill(s < thr, -float("inf"))
    if gen.top_p is not None and float(gen.top_p) < 1.0:
        srt, idx = torch.sort(s, descending=False)
        cum = srt.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - float(gen.top_p))
        remove[..., -1:] = False
        s = s.masked_fill(remove.scatter(1, idx, remove), -float("inf"))
    return s


class DACModel:
    """Codec wrapper with the reference's ``decode`` contract (dac_wrapper/modeling_dac.py:98-134):
    ``audio_codes [1, B, K, F]`` -> ``audio_values [B, 1, F * hop]``.  The arithmetic (quantizer.from_codes +
    the DAC decoder conv stack) runs in ``ptts_dac_decode`` (csrc/conv.hip: implicit-GEMM MFMA kernels with
    fused Snake / bias / residual / tanh).  ``encode`` (D:25-96, audio prompts) runs the encoder conv stack and the
    residual vector quantiser in ``ptts_dac_encode``."""

    config_class = DACConfig

    def __init__(self, config, state_dict: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0", engine_provider=None):
        # `config`: DACConfig, or a full ParlerTTSConfig when built stand-alone from a state dict
        self.device = torch.device(device)
        self._provider = engine_provider
        self._own = None
        if isinstance(config, ParlerTTSConfig):
            self.config, full = config.audio_encoder, config
        else:
            self.config, full = config, None
        if engine_provider is None:
            if state_dict is None:
                raise RuntimeError("DACModel needs a state dict (audio_encoder.model.* tensors) or an engine")
            if full is None:
                from .configuration import ParlerTTSDecoderConfig, T5EncoderConfig

                full = ParlerTTSConfig.from_sub_models_config(
                    T5EncoderConfig(), self.config, ParlerTTSDecoderConfig(num_codebooks=self.config.num_codebooks))
            sd = {k: v for k, v in state_dict.items() if k.startswith("audio_encoder.")}
            if not sd:
                raise RuntimeError("state dict holds no audio_encoder.model.* tensors")
            self._own = HipEngine(full, sd, "fp32", self.device, 1, 1, 0, 2, codec_only=True)

    def _engine(self) -> HipEngine:
        if self._own is not None:
            return self._own
        eng = self._provider()          # the owning model's CURRENT engine (rebuilt after model.to(...)): never a closed handle
        self.device = eng.device
        return eng

    @torch.no_grad()
    def decode(self, audio_codes, audio_scales=None, padding_mask=None, return_dict=None):
        if len(audio_codes) != 1:
            raise ValueError(f"Expected one frame, got {len(audio_codes)}")
        codes = audio_codes.squeeze(0)
        if codes.dim() != 3:
            raise ValueError("audio_codes must be [1, batch, codebooks, frames]")
        wav = self._engine().dac_decode(codes)
        return ModelOutput(audio_values=wav[:, None, :])

    @torch.no_grad()
    def encode(self, input_values, padding_mask=None, bandwidth=None, return_dict=None, n_quantizers=None, sample_rate=None):
        """D:25-96: ``input_values [B, channels, T]`` -> ``audio_codes [1, B, K, T // hop]`` (+ ``audio_scales [None]``).
        One chunk (the reference hard-codes chunk_length = None, D:62-66); the encoder sees the audio at its original
        length (D:85-86 cut the preprocess padding off again).  ``padding_mask`` / ``bandwidth`` are unused as there."""
        if input_values.dim() != 3:
            raise ValueError("input_values must be [batch, channels, samples]")
        _, channels, _ = input_values.shape
        if channels < 1 or channels > 2:
            raise ValueError(f"Number of audio channels must be 1 or 2, but got {channels}")      # D:53-54
        if channels != 1:
            # the reference's check (D:53-54) admits two channels and the descript encoder's first layer, Conv1d(1, d, 7), then
            # raises torch's RuntimeError: the same exception type and message shape here
            d0 = getattr(self.config, "encoder_hidden_size", 64)
            raise RuntimeError(f"Given groups=1, weight of size [{d0}, 1, 7], expected input{list(input_values.shape)} to have 1 "
                               f"channels, but got {channels} channels instead")
        if n_quantizers is not None and n_quantizers != self.config.num_codebooks:
            raise NotImplementedError("n_quantizers < num_codebooks is not on the generate() path")
        codes = self._engine().dac_encode(input_values[:, 0])
        out = ModelOutput(audio_codes=codes[None, ...], audio_scales=[None])
        if return_dict is False:
            return (out.audio_codes, out.audio_scales)
        return out

    def forward(self, tensor):
        raise ValueError("`DACModel.forward` not implemented yet")   # same message as the reference (D:136-137)


class ParlerTTSForConditionalGeneration:
    """Drop-in for the reference class of the same name (M:1560-2686), inference only."""

    config_class = ParlerTTSConfig
    main_input_name = "input_ids"

    def __init__(self, config: ParlerTTSConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 dtype: str = "bf16", device="cuda:0", generation_config: Optional[GenerationConfig] = None):
        _hip.load_library()  # fail now, loudly, if the HIP library is missing
        if dtype not in ("bf16", "fp32"):
            raise ValueError("dtype must be 'bf16' or 'fp32'")
        self.config = config
        self.dtype = dtype
        self.device = torch.device(device)
        self._sd = state_dict
        self._engine: Optional[HipEngine] = None
        self._dac: Optional[DACModel] = None
        d = config.decoder
        self.generation_config = generation_config or GenerationConfig(
            max_length=int(30 * config.audio_encoder.frame_rate), do_sample=True, guidance_scale=1,
            decoder_start_token_id=config.decoder_start_token_id if config.decoder_start_token_id is not None else d.bos_token_id,
            pad_token_id=d.pad_token_id, eos_token_id=d.eos_token_id, bos_token_id=d.bos_token_id)
        if self.generation_config.bos_token_id is None:
            self.generation_config.bos_token_id = d.bos_token_id  # M:2532 reads it; the recipe leaves it unset
        self.profile_every = 0
        # generate() without a streamer: decode steps queued between two reads of the device's loop state (EOS criterion)
        self.eos_poll_steps = 16
        self.phase_times = None   # bench.py: {'t5_prefill_ms','decode_ms','post_ms'} accumulated with syncs at the seams

    # ---- construction ----
    @classmethod
    def from_pretrained(cls, path: str, dtype: str = "bf16", device="cuda:0", **kw):
        """Local checkpoint directory in the reference layout (config.json + safetensors), M:1701-1720."""
        if not os.path.isdir(path):
            raise OSError(f"{path} is not a local directory (hub downloads are not available here)")
        td = kw.pop("torch_dtype", None)          # the transformers spelling: torch.bfloat16 / torch.float32 (or their names)
        cfg = ParlerTTSConfig.from_pretrained(path)
        names = {torch.bfloat16: "bf16", torch.float32: "fp32", "bfloat16": "bf16", "float32": "fp32", "bf16": "bf16", "fp32": "fp32"}
        if isinstance(td, str) and td == "auto":
            # transformers: "auto" = the dtype recorded in the checkpoint's config.json, else the loader's default
            extra = getattr(cfg, "_extra", {})
            td = extra.get("torch_dtype", extra.get("dtype"))        # transformers 4.x / 5.x spelling of the same field
            td = td if td in names else None
        if td is not None:
            if td not in names:
                raise NotImplementedError(f"torch_dtype={td}: the engine computes in bfloat16 (default) or float32; half precision "
                                          "(float16) is not an engine mode")
            dtype = names[td]
        sd = normalize_state_dict(load_state_dict_from_dir(path), cfg)
        gen = None
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            with open(gpath) as f:
                g = json.load(f)
            g.pop("model_type", None)
            gen = GenerationConfig(**g)
        return cls(cfg, sd, dtype=dtype, device=device, generation_config=gen)

    def save_pretrained(self, path: str) -> None:
        """Checkpoint directory in the reference layout (config.json, generation_config.json, model.safetensors with the reference's
        parameter names): what `from_pretrained` reads back, the counterpart of the inherited `PreTrainedModel.save_pretrained`."""
        from .weights import save_pretrained

        if self._sd is None:
            raise RuntimeError("no weights to save")
        save_pretrained(path, self.config, self._sd, self.generation_config)

    @classmethod
    def from_synthetic(cls, config: ParlerTTSConfig, seed: int = 0, dtype: str = "bf16", device="cuda:0", **sd_kw):
        return cls(config, synthetic_state_dict(config, seed=seed, **sd_kw), dtype=dtype, device=device)

    def to(self, *args, **kwargs):
        """`nn.Module.to` as inference users call it: a device, a dtype (`torch.bfloat16` = the bf16 engine mode, `torch.float32` = the
        bit-exact fp32 mode; half precision is not an engine mode), or both.  A change drops the engine: it is rebuilt (weights
        re-tiled in the new dtype / on the new device) at the next call."""
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        if dtype is not None:
            names = {torch.bfloat16: "bf16", torch.float32: "fp32"}
            if dtype not in names:
                raise NotImplementedError(f"{dtype}: the engine computes in bfloat16 or float32")
            if names[dtype] != self.dtype:
                self._drop_engine()
                self.dtype = names[dtype]
        if device is not None:
            dev = torch.device(device)
            if dev.type != "cuda":
                raise RuntimeError("this engine only runs on an MI355X (device type 'cuda' under ROCm)")
            if dev != self.device:
                self._drop_engine()
                self.device = dev
        return self

    def _drop_engine(self) -> None:
        """Release the HIP engine NOW (KV pool, codec workspaces, graphs) instead of at garbage collection; it is rebuilt at the
        next call.  A codec wrapper handed out earlier (`model.audio_encoder`) stays valid: it asks the model for its engine at
        every call and is re-bound to the model's device here."""
        if self._engine is not None:
            self._engine.close()
        self._engine = None
        self._dac = None

    def bfloat16(self):
        return self.to(torch.bfloat16)

    def float(self):
        return self.to(torch.float32)

    def half(self):
        return self.to(torch.float16)          # raises: not an engine mode

    def eval(self):
        return self

    @property
    def audio_encoder(self) -> DACModel:
        if self._dac is None:
            if not any(k.startswith("audio_encoder.model.decoder.") for k in (self._sd or {})):
                raise RuntimeError("this model was built without codec weights (audio_encoder.model.*)")
            self._dac = DACModel(self.config.audio_encoder, device=self.device, engine_provider=self.engine)
        return self._dac

    def engine(self, B_eff: int = 1, T: int = 1, P: int = 0, L: int = 2) -> HipEngine:
        """Engine sized for at least these limits (re-created when a call exceeds the current pools)."""
        if self._engine is not None:
            mb, mt, mp, ml = self._engine.limits
            if B_eff <= mb and T <= mt and P <= mp and L <= ml:
                return self._engine
            self._engine.close()
            B_eff, T, P, L = max(B_eff, mb), max(T, mt), max(P, mp), max(L, ml)
        if self._sd is None:
            raise RuntimeError("no weights: construct with a state dict, from_pretrained() or from_synthetic()")
        self._engine = HipEngine(self.config, self._sd, self.dtype, self.device, B_eff, T, P, L,
                                 **getattr(self, "_engine_kw", {}))
        return self._engine

    # ---- generate ----
    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, generation_config: Optional[GenerationConfig] = None,
                 logits_processor=None, stopping_criteria=None, synced_gpus: Optional[bool] = None, streamer=None,
                 **kwargs):
        # `synced_gpus` (M:2349, handed to the third-party loop at M:2590): under ZeRO-3 it keeps every rank stepping until all
        # have finished, because the WEIGHTS are sharded; here every rank holds all weights (distributed.py), a rank's extra
        # steps could only produce pad tokens, so the flag is accepted and changes nothing.
        # every stream handle, lazy allocation and launch below follows the CURRENT device: make it the model's
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream(self.device)
            if cur.cuda_stream != 0 or os.environ.get("PTTS_LEGACY_POLL") or os.environ.get("PTTS_NO_SIDE_STREAM"):
                return self._generate(inputs, generation_config, logits_processor, stopping_criteria, synced_gpus, streamer,
                                      **kwargs)
            # The legacy NULL stream cannot be captured into a hipGraph and every engine call on it is host-synchronous
            # (csrc/engine.hip: enter_stream / leave_stream).  The call runs on a stream of the model's own instead: it waits
            # for what the caller has queued on the NULL stream, and the NULL stream waits for it at the end, so the caller
            # sees the ordering of the stream it called from.
            if getattr(self, "_side_stream", None) is None or self._side_stream.device != self.device:
                self._side_stream = _own_stream(self.device)
            side = self._side_stream
            side.wait_stream(cur)
            try:
                with torch.cuda.stream(side):
                    return self._generate(inputs, generation_config, logits_processor, stopping_criteria, synced_gpus,
                                          streamer, **kwargs)
            finally:
                cur.wait_stream(side)

    def _generate(self, inputs, generation_config, logits_processor, stopping_criteria, synced_gpus, streamer, **kwargs):
        cfgm, dec = self.config, self.config.decoder
        K = dec.num_codebooks
        # 1. generation config (M:2416-2422)
        gen = copy.deepcopy(generation_config if generation_config is not None else self.generation_config)
        user_max_length = kwargs.get("max_length")
        model_kwargs = gen.update(**kwargs)
        gen.validate()
        logits_processor = list(logits_processor) if logits_processor else []
        stopping_criteria = list(stopping_criteria) if stopping_criteria else []
        allowed = {"input_ids", "attention_mask", "prompt_input_ids", "prompt_attention_mask", "encoder_outputs",
                   "prompt_hidden_states", "decoder_input_ids", "return_raw_ids", "input_values", "padding_mask",
                   "decoder_attention_mask", "_decoder_only"}
        for k in model_kwargs:
            if k not in allowed:
                raise ValueError(f"The following `model_kwargs` are not used by the model: ['{k}']")
        if gen.pad_token_id is None and gen.eos_token_id is not None:   # M:2432-2442
            gen.pad_token_id = gen.eos_token_id
        # 3. inputs (M:2449-2452)
        input_ids = inputs if inputs is not None else model_kwargs.get("input_ids")
        enc_out = model_kwargs.get("encoder_outputs")
        # decoder-only driver (ParlerTTSForCausalLM.generate, M:1281-1552): {"batch": B} - encoder states are optional there
        # (no cross-attention block without them, M:465) and arrive ALREADY doubled under guidance (M:1218-1232 doubles only
        # the decoder ids and the prompt; encoder_hidden_states go to the decoder as the caller passed them)
        dec_only = model_kwargs.get("_decoder_only")
        if input_ids is None and enc_out is None and dec_only is None:
            raise ValueError("generate() needs description `input_ids` (or `encoder_outputs`)")
        dev = self.device
        enc = None
        if dec_only is not None and enc_out is None:
            B, T = int(dec_only["batch"]), 0
        elif enc_out is not None:
            enc = enc_out[0] if isinstance(enc_out, (tuple, list)) else getattr(enc_out, "last_hidden_state", enc_out)
            enc = enc.to(dev, torch.float32)
            B, T = enc.shape[:2]
            if dec_only is not None:
                B = int(dec_only["batch"])
        else:
            input_ids = input_ids.to(dev)
            B, T = input_ids.shape
        gs0 = gen.guidance_scale
        enc_rows = B * (2 if (dec_only is not None and gs0 is not None and gs0 > 1) else 1)      # rows the caller must have passed
        if enc is not None and enc.shape[0] != enc_rows:
            raise ValueError(f"encoder hidden states must have {enc_rows} rows (batch {B}"
                             + (", doubled by the caller under guidance_scale > 1: the decoder-only driver doubles only the "
                                "decoder ids and the prompt, M:1218-1232" if enc_rows != B else "") + f"), got {enc.shape[0]}")
        attention_mask = model_kwargs.get("attention_mask")
        if attention_mask is None:    # M:2462-2465 (pad == eos, so HF does not infer a mask from pad tokens)
            attention_mask = torch.ones(enc_rows, T, dtype=torch.long, device=dev)
        attention_mask = attention_mask.to(dev)
        prompt_ids = model_kwargs.get("prompt_input_ids")
        prompt_mask = model_kwargs.get("prompt_attention_mask")
        # pre-embedded prompts win over ids: the reference embeds ids only when no hidden states came in (M:2476-2480)
        prompt_embeds = model_kwargs.get("prompt_hidden_states")
        if prompt_embeds is not None:
            if prompt_embeds.dim() != 3 or prompt_embeds.shape[0] != B or prompt_embeds.shape[2] != dec.hidden_size:
                raise ValueError(f"prompt_hidden_states must be [batch {B}, prompt_len, {dec.hidden_size}], got {tuple(prompt_embeds.shape)}")
            prompt_ids = None
            P = prompt_embeds.shape[1]
        else:
            P = 0 if prompt_ids is None else prompt_ids.shape[1]
        use_cfg = gen.guidance_scale is not None and gen.guidance_scale > 1
        B_eff = 2 * B if use_cfg else B
        # 5./6. start ids and max_length (M:2490-2527)
        start_tok = gen.decoder_start_token_id if gen.decoder_start_token_id is not None else gen.bos_token_id
        ids = torch.full((B * K, 1), start_tok, dtype=torch.long, device=dev)
        dec_in = model_kwargs.get("decoder_input_ids")
        if dec_in is None and model_kwargs.get("input_values") is not None:
            # audio prompt (M:2483-2487 -> M:2247-2292): codec ids of the waveform become the decoder prefix
            if self._engine is None:   # size the engine for the whole call before the codec runs on it
                n_pre = model_kwargs["input_values"].shape[-1] // cfgm.audio_encoder.hop_length + 1
                L_est = gen.max_new_tokens + n_pre if gen.max_new_tokens is not None else int(gen.max_length)
                if P + L_est <= dec.max_position_embeddings:
                    self.engine(B_eff, T, P, L_est)
            enc_codes = self.audio_encoder.encode(model_kwargs["input_values"].to(dev)).audio_codes
            frames, bsz, codebooks, seq_len = enc_codes.shape
            if frames != 1:
                raise ValueError(f"Expected 1 frame in the audio code outputs, got {frames} frames. Ensure chunking is "
                                 "disabled by setting `chunk_length=None` in the audio encoder.")
            dec_in = enc_codes[0].reshape(bsz * K, seq_len)
        if dec_in is not None:       # user prefix [B*K, s]; the start column is prepended when it is missing (M:2184-2185)
            dec_in = dec_in.to(dev, torch.long)
            if dec_in.dim() != 2 or dec_in.shape[0] != B * K:
                raise ValueError(f"decoder_input_ids must be [batch * num_codebooks = {B * K}, prefix_len], got {tuple(dec_in.shape)}")
            prepend = bool((dec_in[..., 0] != start_tok).all())
            ids = dec_in if not prepend else torch.cat([ids, dec_in], dim=-1)
        dec_mask = model_kwargs.get("decoder_attention_mask")
        if dec_mask is not None:
            # [B, prefix_len] over the prefix columns; a ones column joins a prepended start column (M:2186-2193).  The
            # decoder concatenates it behind prompt_attention_mask (M:845-846); with a prompt but no prompt mask the
            # reference's 4-D mask helper fails on the length mismatch (M:870-872), so that combination is rejected here.
            dec_mask = dec_mask.to(dev)
            if dec_in is not None and prepend:
                dec_mask = torch.cat([torch.ones_like(dec_mask)[:, :1], dec_mask], dim=-1)
            if dec_mask.dim() != 2 or dec_mask.shape[0] != B or dec_mask.shape[1] != ids.shape[1]:
                raise ValueError(f"decoder_attention_mask must be [batch {B}, prefix_len {ids.shape[1]}], got {tuple(dec_mask.shape)}")
            if P > 0 and prompt_mask is None:
                raise ValueError("decoder_attention_mask with a prompt needs prompt_attention_mask as well: the reference "
                                 "concatenates the two (M:845-846) and cannot build its causal mask from the shorter one")
        if gen.max_new_tokens is not None:
            gen.max_length = gen.max_new_tokens + ids.shape[-1]
        if gen.min_length is not None and gen.min_length > gen.max_length:
            raise ValueError(f"Unfeasible length constraints: the minimum length ({gen.min_length}) is larger than"
                             f" the maximum length ({gen.max_length})")
        L = int(gen.max_length)
        if P + L > dec.max_position_embeddings:
            raise ValueError(f"prompt ({P}) + max_length ({L}) exceeds max_position_embeddings ({dec.max_position_embeddings})")
        # 7. mode (M:2544-2553, 2575-2579, 2622-2626)
        greedy = gen.num_beams == 1 and gen.num_beam_groups == 1 and gen.do_sample is False
        sample = gen.num_beams == 1 and gen.num_beam_groups == 1 and gen.do_sample is True
        if not (greedy or sample):
            raise ValueError("Got incompatible mode for generation, should be one of greedy or sampling. "
                             "Ensure that beam search is de-activated by setting `num_beams=1` and `num_beam_groups=1`.")
        if greedy and gen.num_return_sequences > 1:
            raise ValueError(f"num_return_sequences has to be 1 when doing greedy search, but is {gen.num_return_sequences}.")
        nrs = int(gen.num_return_sequences) if sample else 1
        # EOS suppression of the default processors (third-party `_get_logits_processor`, called at M:2561-2567)
        eos_free_until = 0
        if gen.eos_token_id is not None:
            if gen.min_length is not None and gen.min_length > 0:
                eos_free_until = int(gen.min_length)                                   # MinLengthLogitsProcessor
            if gen.min_new_tokens is not None and gen.min_new_tokens > 0:              # MinNewTokensLengthLogitsProcessor
                eos_free_until = max(eos_free_until, ids.shape[-1] + int(gen.min_new_tokens))
        if eos_free_until and use_cfg:
            # the reference runs these processors BEFORE its guidance processor (user / guidance processors are merged
            # behind the default ones): eos is -inf in both halves and uncond + (cond - uncond) * scale makes it NaN,
            # which argmax then picks and multinomial rejects.  Not reproduced.
            raise NotImplementedError("min_length / min_new_tokens together with guidance_scale > 1 yields NaN scores in "
                                      "the reference (-inf - -inf in the guidance combine)")

        # id ranges: the reference fails with an IndexError in its embedding lookups; the gather kernels must never be handed
        # an id outside their tables
        def _in_range(name, t, hi):
            if t is not None and t.numel() and (int(t.min()) < 0 or int(t.max()) >= hi):
                raise IndexError(f"index out of range in self: `{name}` must be in [0, {hi})")

        if tuple(attention_mask.shape) != (enc_rows, T):
            raise ValueError(f"attention_mask must be [batch {enc_rows}, description length {T}], got {tuple(attention_mask.shape)}")
        if prompt_ids is not None and (prompt_ids.dim() != 2 or prompt_ids.shape[0] != B):
            raise ValueError(f"prompt_input_ids must be [batch {B}, prompt_len], got {tuple(prompt_ids.shape)}")
        if prompt_mask is not None and tuple(prompt_mask.shape) != (B, P):
            raise ValueError(f"prompt_attention_mask must be [batch {B}, prompt length {P}], got {tuple(prompt_mask.shape)}")
        if enc_out is not None and enc is not None and enc.shape[-1] != cfgm.text_encoder.d_model:
            raise ValueError(f"encoder_outputs must be [batch, T, {cfgm.text_encoder.d_model}], got {tuple(enc.shape)}")
        if enc_out is None:
            _in_range("input_ids", input_ids, cfgm.text_encoder.vocab_size)
        _in_range("prompt_input_ids", prompt_ids, cfgm.vocab_size)
        _in_range("decoder_input_ids", ids, dec.vocab_size + 1)       # embed_tokens has vocab_size + 1 rows (M:769-773)
        eng = self.engine(B_eff * nrs, max(T, 1), P, L)
        import time as _time

        def _tick():
            if self.phase_times is None:
                return 0.0
            torch.cuda.synchronize(dev)
            return _time.perf_counter()

        t_start = _tick()
        if enc_out is None and dec_only is None:
            enc = eng.t5_encode(input_ids, attention_mask)                                   # M:2467-2474
        if enc is None:                        # no encoder states: a placeholder row set the engine never reads (T = 0)
            enc = torch.zeros(B_eff, 0, cfgm.text_encoder.d_model, device=dev)
        # classifier-free guidance: zeroed 'null' half (M:2231-2237), prompts repeated (M:2109-2118)
        amask32 = attention_mask.to(torch.int32)
        if use_cfg and dec_only is None:
            enc = torch.cat([enc, torch.zeros_like(enc)], dim=0)
            amask32 = torch.cat([amask32, torch.zeros_like(amask32)], dim=0)
        pids = None if prompt_ids is None else prompt_ids.to(dev, torch.int64)
        pmask32 = None if prompt_mask is None else prompt_mask.to(dev, torch.int32)
        pemb = None if prompt_embeds is None else prompt_embeds.to(dev, torch.float32)
        if use_cfg and dec_only is not None:
            # decoder-only driver (M:1218-1232): the null half gets a ZEROED prompt and an all-zero prompt mask
            pemb = None if pemb is None else torch.cat([pemb, torch.zeros_like(pemb)], dim=0)
            pmask32 = None if pmask32 is None else torch.cat([pmask32, torch.zeros_like(pmask32)], dim=0)
        elif use_cfg:
            pids = None if pids is None else pids.repeat(2, 1)
            pmask32 = None if pmask32 is None else pmask32.repeat(2, 1)
            pemb = None if pemb is None else pemb.repeat(2, 1, 1)                            # M:2115-2116
        # delay pattern (M:2530-2537): built for the B utterances, BEFORE any num_return_sequences expansion
        ids, pattern = build_delay_pattern_mask(ids, gen.bos_token_id, gen.pad_token_id, L, K)
        if streamer is not None:
            streamer.put(ids.cpu())                                                          # M:2540-2541
        dmask32 = None
        if dec_mask is not None:
            dmask32 = dec_mask.to(torch.int32)
            dmask32 = (dmask32.repeat(2, 1) if use_cfg else dmask32)                         # M:2112-2113
        B_gen = B
        if nrs > 1:
            # `_expand_inputs_for_generation` (third party, called at M:2600-2606): repeat_interleave along dim 0 of the id
            # rows [B*K, s] AND of every tensor kwarg, the stashed delay pattern [B*K, L] included.  Row (b*K + k)*n + j is
            # then read by the decoder's reshape(-1, K, len) (M:823) as codebook ((b*K + k)*n + j) % K of utterance
            # ((b*K + k)*n + j) // K, under the delay pattern of codebook k: reproduced as is (the engine takes any pattern).
            rep = lambda t: None if t is None else t.repeat_interleave(nrs, dim=0)
            enc, amask32, pids, pmask32, pemb, dmask32 = rep(enc), rep(amask32), rep(pids), rep(pmask32), rep(pemb), rep(dmask32)
            ids, pattern = rep(ids), rep(pattern)
            B_gen, B_eff = B * nrs, B_eff * nrs
        pemb = None if pemb is None else pemb.contiguous()
        dmask32 = None if dmask32 is None else dmask32.contiguous()
        enc = enc.contiguous()
        amask32 = amask32.contiguous()
        pids = None if pids is None else pids.contiguous()
        pmask32 = None if pmask32 is None else pmask32.contiguous()
        pattern = pattern.contiguous()
        start_len = ids.shape[1]
        raw_ids = torch.full((B_gen * K, L), gen.pad_token_id, dtype=torch.long, device=dev)
        raw_ids[:, :start_len] = ids
        sp = _hip.PttsSampling()
        sp.do_sample = 1 if sample else 0
        sp.temperature = float(gen.temperature if gen.temperature is not None else 1.0)
        sp.top_k = int(gen.top_k or 0)
        sp.guidance_scale = float(gen.guidance_scale) if use_cfg else 1.0
        sp.top_p = float(gen.top_p) if (gen.top_p is not None and 0.0 < gen.top_p < 1.0) else 1.0
        if gen.seed is not None:
            seed = int(gen.seed)
        elif sample:
            # like torch.multinomial in the reference, an unseeded call consumes the global CPU generator: two calls draw
            # different streams, torch.manual_seed(s) reproduces a run
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        else:
            seed = 0
        sp.seed = seed & 0xFFFFFFFFFFFFFFFF
        self.last_seed = sp.seed
        sp.eos_token_id = -1 if gen.eos_token_id is None else int(gen.eos_token_id)
        sp.pad_token_id = -1 if gen.pad_token_id is None else int(gen.pad_token_id)
        if gen.eos_token_id is None:
            sp.eos_token_id = -2          # no EOS criterion: nothing latches (the third-party loop skips the check)
        sp.eos_free_until = eos_free_until
        eng.prefill(B_gen, B_eff, enc, amask32, pids, pmask32, pattern, raw_ids, start_len, prompt_embeds=pemb, dec_mask=dmask32)
        t_dec = _tick()
        # ---- the hot loop: state lives on the device, the host only polls for "all rows finished" ----
        remaining = L - start_len
        L_out = L
        # `output_scores=True` with `return_dict_in_generate=True` (third-party loop, reached from M:2582 / M:2608): the processed
        # scores of every step come back as a tuple of [batch * codebooks, vocab] tensors - after the guidance combine, the
        # default / user processors and, under sampling, the warpers.  They exist on the host only in the step-wise form below.
        collect_scores = bool(gen.output_scores and gen.return_dict_in_generate)
        step_scores = []
        if logits_processor or stopping_criteria or collect_scores:
            # custom processors / criteria are host callables (M:2561-2573): one step at a time, the scores pass through
            # them between the forward half and the selection half, in the order of the third-party loop
            # (processors on [B*K, vocab] scores after the guidance combine, then the warpers inside the select kernel)
            V = dec.vocab_size
            n = B_gen * K
            cur = start_len                       # columns filled so far: known on the host (one column per step)
            while remaining > 0:
                eng.forward_step()
                scores = eng.last_logits(B_eff * K)          # synchronises the stream; the only copy out per step
                if use_cfg:
                    scores = scores[n:] + (scores[:n] - scores[n:]) * float(gen.guidance_scale)      # M:2556-2558
                seen = raw_ids[:, :cur]
                if collect_scores and eos_free_until and cur < eos_free_until:
                    # MinLength / MinNewTokensLength sit in FRONT of the user's processors in the merged list; the select kernel
                    # masks EOS again below the bound (idempotent), here the reported scores get what those processors write
                    scores = scores.clone()
                    scores[:, gen.eos_token_id] = -float("inf")
                for proc in logits_processor:
                    scores = proc(seen, scores)
                if collect_scores:
                    step_scores.append(_warp_scores(scores, gen) if sample else scores.clone())
                eng.set_logits(torch.cat([scores, scores], dim=0) if use_cfg else scores)
                eng.select_step(sp)
                remaining -= 1
                cur += 1
                st = eng.state()                               # one read-back per step: the EOS criterion is a host decision here
                if streamer is not None:
                    streamer.put(raw_ids[:, cur - 1].cpu())
                if st.done_len:
                    L_out = st.done_len
                    break
                if stopping_criteria:
                    seen = raw_ids[:, :cur]
                    stop = [c(seen, scores) for c in stopping_criteria]
                    if any(bool(torch.as_tensor(x).all()) for x in stop):
                        L_out = cur
                        break
            remaining = 0
        # a streamer is fed one column per decoding step (M:2540-2541 + the third-party loop), but the host only polls the
        # device every `poll_steps` steps (streamer attribute, default 8): one state read-back and one copy per poll
        # Without a streamer the loop state is read through PIPELINED snapshots (ptts_state_snapshot / ptts_state_poll): chunk
        # i + 1 is queued before chunk i's snapshot is awaited, so the device never runs dry and an EOS-terminated call runs at
        # most 2 * eos_poll_steps - 1 steps past its last token (rounds 1-4 read the state with a blocking copy every 128 steps:
        # up to 127 extra steps, ~64 on average).  The third-party loop tests `unfinished_sequences.max() == 0` after every step.
        legacy_poll = bool(os.environ.get("PTTS_LEGACY_POLL"))      # A/B switch: rounds 1-4's blocking read every 128 steps
        poll_eos = streamer is None and gen.eos_token_id is not None and not legacy_poll
        chunk = max(1, int(getattr(streamer, "poll_steps", 8))) if streamer is not None else \
            (max(1, int(os.environ.get("PTTS_EOS_POLL_STEPS", self.eos_poll_steps))) if poll_eos else 128)
        fed = start_len                                   # columns already handed to the streamer
        pending, slot = None, 0                           # snapshot slot not polled yet / next slot to use
        while remaining > 0:
            n = min(chunk, remaining)
            eng.decode_steps(n, sp, self.profile_every)
            remaining -= n
            if poll_eos:
                eng.state_snapshot(slot)
                if pending is not None:
                    st = eng.state_poll(pending)
                    if st.done_len:
                        L_out = st.done_len
                        eng.state_poll(slot)              # retire the snapshot just taken
                        pending = None
                        break
                pending, slot = slot, (slot + 1) % _hip.STATE_SLOTS
                continue
            if streamer is not None:
                st = eng.state()
                upto = st.done_len or st.cur_len
                if upto > fed:
                    cols = raw_ids[:, fed:upto].cpu()
                    for c in range(cols.shape[1]):
                        streamer.put(cols[:, c])
                    fed = upto
                if st.done_len:
                    L_out = st.done_len
                    break
            elif legacy_poll and gen.eos_token_id is not None:
                st = eng.state()
                if st.done_len:
                    L_out = st.done_len
                    break
        if pending is not None:                           # the run's last chunk
            st = eng.state_poll(pending)
            if st.done_len:
                L_out = st.done_len
        if streamer is not None:
            streamer.end()
        # ONE blocking read of the loop state behind the last step, on every path (it is also the synchronisation this point
        # needs): it is what turns a failed in-grid hand-off of the fused launch into a PttsError BEFORE any id is used.  Without
        # it a call with no streamer and no EOS criterion (or under PTTS_LEGACY_POLL) never looked at the error word and would
        # have returned ids computed from a stale query.
        st_end = eng.state()
        if st_end.done_len and st_end.done_len < L_out:
            L_out = st_end.done_len
        t_post = _tick()
        if self.phase_times is not None:
            self.phase_times["t5_prefill_ms"] += (t_dec - t_start) * 1e3
            self.phase_times["decode_ms"] += (t_post - t_dec) * 1e3
        self.last_raw_ids = raw_ids[:, :L_out]
        # ---- un-delay (M:2633-2648): integer bookkeeping on the device tensors ----
        out_ids = apply_delay_pattern_mask(raw_ids[:, :L_out], pattern)
        # M:2637 rebuilds the mask from the (expanded) start ids: a REGULAR pattern over B_gen * K rows, then M:2645 reshapes
        # with the ORIGINAL batch size - under num_return_sequences = n every utterance ends up with n * F columns
        start = torch.full((B_gen * K, 1), start_tok, dtype=torch.long, device=dev)
        _, m = build_delay_pattern_mask(start, gen.bos_token_id, gen.pad_token_id, L_out, K)
        keep = (m != gen.bos_token_id) & (m != gen.pad_token_id)
        output_ids = out_ids[keep].reshape(B, K, -1)[None, ...]
        self.last_codes = output_ids
        if model_kwargs.get("return_raw_ids"):
            if self.phase_times is not None:
                self.phase_times["post_ms"] += (_tick() - t_post) * 1e3
            return output_ids
        # ---- DAC decode (M:2650-2680) ----
        special = (gen.bos_token_id, gen.pad_token_id, gen.eos_token_id)
        sequential = any(bool((output_ids == s).any()) for s in special if s is not None)
        if not sequential:
            values = self.audio_encoder.decode(output_ids, audio_scales=[None] * B).audio_values.squeeze(1)
        else:
            vals = []
            for b in range(B):
                sample_b = output_ids[:, b]
                ok = (sample_b >= cfgm.audio_encoder.codebook_size).sum(dim=(0, 1)) == 0
                if int(ok.sum()) > 0:
                    wav = self.audio_encoder.decode(sample_b[:, :, ok][None, ...], [None]).audio_values
                    vals.append(wav.transpose(0, 2))
                else:
                    vals.append(torch.zeros(1, 1, 1, device=dev))
            values = torch.nn.utils.rnn.pad_sequence(vals, batch_first=True, padding_value=0).squeeze(-1).squeeze(-1)
        if self.phase_times is not None:
            self.phase_times["post_ms"] += (_tick() - t_post) * 1e3
        if gen.return_dict_in_generate:
            # M:2682-2684: the third-party loop's own output object with `.sequences` overwritten by the waveform.  That object
            # is a Greedy/SampleEncoderDecoderOutput (transformers 4.3x): its other fields are None unless `output_scores` /
            # `output_attentions` / `output_hidden_states` were asked for; `scores` is served (above), attentions / hidden states never leave the kernels
            return ModelOutput(sequences=values, scores=tuple(step_scores) if collect_scores else None, encoder_attentions=None,
                               encoder_hidden_states=None, decoder_attentions=None, cross_attentions=None, decoder_hidden_states=None)
        return values


class ParlerTTSForCausalLM:
    """Decoder-only driver with the reference's ``generate()`` contract (M:1281-1552): codec ids in, codec ids
    ``[batch, num_codebooks, frames]`` out, conditioned on caller-supplied ``encoder_hidden_states`` (already in the decoder
    width: the composite's ``enc_to_dec_proj`` is not part of this class) and optional ``prompt_hidden_states``.

    Same engine and kernels as the composite: the decoder state dict (``model.decoder.*``, ``lm_heads.*``; a composite
    checkpoint's ``decoder.``-prefixed keys are accepted too) is loaded into a ``ptts_engine`` whose encoder width equals
    the decoder width, so ``ptts_prefill`` skips the projection.  The reference multiplies nothing into
    ``encoder_hidden_states`` here (M:1096-1185), the composite path multiplies by the mask (M:2016-2017); masked keys
    carry weight exactly 0 in cross-attention either way, so the ids are the same.

    ``guidance_scale > 1``: as M:1218-1232, only the decoder ids and the prompt are doubled here (the null half gets a zeroed
    prompt under an all-zero prompt mask); ``encoder_hidden_states`` / ``encoder_attention_mask`` must arrive with 2 x batch
    rows.  Without ``encoder_hidden_states`` the layers run without their cross-attention block (M:465)."""

    config_class = ParlerTTSDecoderConfig

    def __init__(self, config: ParlerTTSDecoderConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 dtype: str = "bf16", device="cuda:0", generation_config: Optional[GenerationConfig] = None):
        from .configuration import T5EncoderConfig

        _hip.load_library()
        if state_dict is None:
            raise RuntimeError("no weights: pass the decoder state dict (model.decoder.*, lm_heads.*)")
        self.config = config
        self.num_codebooks = config.num_codebooks
        D = config.hidden_size
        sd = {}
        for k, v in state_dict.items():
            if k.startswith("decoder."):
                sd[k] = v
            elif k.startswith(("model.decoder.", "lm_heads.")):
                sd["decoder." + k] = v
        pos = "decoder.model.decoder.embed_positions.weights"
        if pos not in sd:
            from .weights import sinusoid_table

            sd[pos] = sinusoid_table(config.max_position_embeddings, D)
        sd["embed_prompts.weight"] = torch.zeros(1, D)      # prompts arrive pre-embedded; the table is never indexed
        wrap = ParlerTTSConfig.from_sub_models_config(T5EncoderConfig(d_model=D), DACConfig(num_codebooks=config.num_codebooks),
                                                      config, vocab_size=1)
        self._inner = ParlerTTSForConditionalGeneration(
            wrap, sd, dtype=dtype, device=device,
            generation_config=generation_config or GenerationConfig(
                max_length=20, do_sample=False, guidance_scale=None, decoder_start_token_id=config.bos_token_id,
                pad_token_id=config.pad_token_id, eos_token_id=config.eos_token_id, bos_token_id=config.bos_token_id))
        self._inner._engine_kw = dict(with_t5=False)
        self.generation_config = self._inner.generation_config
        self.device = self._inner.device

    def to(self, *args, **kwargs):
        self._inner.to(*args, **kwargs)
        self.device = self._inner.device
        return self

    def eval(self):
        return self

    def build_delay_pattern_mask(self, input_ids, bos_token_id, pad_token_id, max_length=None):
        max_length = max_length if max_length is not None else self.generation_config.max_length      # M:1265-1266
        return build_delay_pattern_mask(input_ids, bos_token_id, pad_token_id, max_length, self.num_codebooks)

    @staticmethod
    def apply_delay_pattern_mask(input_ids, decoder_pad_token_mask):
        return apply_delay_pattern_mask(input_ids, decoder_pad_token_mask)

    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, generation_config: Optional[GenerationConfig] = None,
                 logits_processor=None, stopping_criteria=None, synced_gpus: Optional[bool] = None, streamer=None,
                 **kwargs):
        """M:1281-1552.  ``inputs`` / ``input_ids``: decoder ids ``[batch * num_codebooks, s]`` (default: one BOS column,
        which then needs ``batch_size`` from ``encoder_hidden_states``); ``attention_mask``: ``[batch, s]`` over those
        columns; ``encoder_hidden_states`` ``[batch, T, hidden]`` + ``encoder_attention_mask``; ``prompt_hidden_states``
        ``[batch, P, hidden]`` + ``prompt_attention_mask``.  Returns the un-delayed ids ``[batch, num_codebooks, frames]``
        (M:1532-1546)."""
        kwargs = dict(kwargs)
        enc = kwargs.pop("encoder_hidden_states", None)
        if enc is not None and (enc.dim() != 3 or enc.shape[-1] != self.config.hidden_size):
            raise ValueError(f"encoder_hidden_states must be [batch, T, {self.config.hidden_size}], got {tuple(enc.shape)}")
        ids = inputs if inputs is not None else kwargs.pop("input_ids", None)
        kwargs.pop("input_ids", None)
        prompt_h = kwargs.get("prompt_hidden_states")
        if ids is not None:
            batch = ids.shape[0] // self.num_codebooks
        elif prompt_h is not None:
            batch = prompt_h.shape[0]
        elif enc is not None:
            gs = kwargs.get("guidance_scale", (generation_config or self.generation_config).guidance_scale)
            batch = enc.shape[0] // (2 if (gs is not None and gs > 1) else 1)
        else:
            batch = 1                                   # M:1441-1446: one BOS column
        inner = dict(attention_mask=kwargs.pop("encoder_attention_mask", None), return_raw_ids=True, _decoder_only={"batch": batch})
        if enc is not None:
            inner["encoder_outputs"] = (enc,)
        if ids is not None:
            inner["decoder_input_ids"] = ids
        dmask = kwargs.pop("attention_mask", None)
        if dmask is not None:
            inner["decoder_attention_mask"] = dmask
        out = self._inner.generate(generation_config=generation_config, logits_processor=logits_processor,
                                   stopping_criteria=stopping_criteria, synced_gpus=synced_gpus, streamer=streamer,
                                   **inner, **kwargs)
        self.last_raw_ids = self._inner.last_raw_ids
        codes = out[0]                                   # [B, K, F]
        if (generation_config or self.generation_config).return_dict_in_generate or kwargs.get("return_dict_in_generate"):
            return ModelOutput(sequences=codes)
        return codes
