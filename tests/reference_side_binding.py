# parler_tts/hip_backend.py  (reference side) - the binding a maintainer of danlyth/parler-tts would add.
# ctypes over include/parler_tts_hip.h ONLY: nothing of parler_tts_amd is imported here.  `model` is the reference's own
# ParlerTTSForConditionalGeneration (anything with .state_dict(), .config, .decoder.config, .text_encoder.config does).
# INTEGRATION.md section B quotes this file byte for byte (tests/test_abi.py); tests/test_gpu_reference_binding.py runs it.
import ctypes as C
import os

import torch

_lib = C.CDLL(os.environ.get("PTTS_HIP_LIB", "libparler_tts_hip.so"))          # include/parler_tts_hip.h
_lib.ptts_last_error.restype = C.c_char_p
_I32 = C.c_int32


class PttsConfig(C.Structure):                 # struct ptts_config
    _fields_ = [(n, _I32) for n in (
        "abi_version dtype hidden layers heads ffn vocab codebooks max_positions pad_token_id eos_token_id bos_token_id "
        "prompt_vocab enc_hidden t5_layers t5_heads t5_dkv t5_dff t5_vocab t5_gated t5_act").split()] + \
        [("t5_eps", C.c_float)] + [(n, _I32) for n in "max_batch max_desc_len max_prompt_len max_length "
        "dac_codebook_size dac_codebook_dim dac_latent dac_hidden dac_n_up".split()] + \
        [("dac_up", _I32 * 8), ("dac_enc_hidden", _I32), ("dec_variant", _I32)]


class PttsSampling(C.Structure):               # struct ptts_sampling
    _fields_ = [("do_sample", _I32), ("temperature", C.c_float), ("top_k", _I32), ("guidance_scale", C.c_float),
                ("seed", C.c_uint64), ("top_p", C.c_float), ("eos_token_id", _I32), ("pad_token_id", _I32),
                ("eos_free_until", _I32)]


class PttsState(C.Structure):                  # struct ptts_state
    _fields_ = [(n, _I32) for n in "pos cur_len prompt_len n_unfinished done_len step_count".split()]


def _ck(rc):
    if rc:
        raise RuntimeError(f"parler_tts_hip error {rc}: {_lib.ptts_last_error().decode()}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _fold_weight_norm(sd):
    """descript's WNConv1d / WNConvTranspose1d keep (weight_g, weight_v); the engine takes weight = g * v / ||v|| (norm over
    all dims but 0, torch weight_norm dim=0), under either spelling of the pair."""
    out = {}
    for k, v in sd.items():
        for vs, gs in ((".weight_v", ".weight_g"), (".parametrizations.weight.original1", ".parametrizations.weight.original0")):
            if k.endswith(vs):
                g, vf = sd[k[: -len(vs)] + gs].float(), v.float()
                norm = vf.reshape(vf.shape[0], -1).norm(dim=1).reshape([-1] + [1] * (vf.dim() - 1))
                out[k[: -len(vs)] + ".weight"] = g * vf / norm
                break
            if k.endswith(gs):
                break
        else:
            out[k] = v
    return out


class HipDecoder:
    """Owns one engine handle.  T5 stays in transformers on this side (t5_layers = 0): `run` is handed the text encoder's
    last_hidden_state UNPROJECTED - the engine applies enc_to_dec_proj and the mask multiply once (M:2010-2017)."""

    def __init__(self, model, max_batch, max_desc_len, max_prompt_len, max_length, bf16=True, device="cuda:0"):
        d, t5 = model.decoder.config, model.text_encoder.config
        sd = _fold_weight_norm({k: v for k, v in model.state_dict().items() if not k.startswith("text_encoder.")})
        cfg = PttsConfig()
        cfg.abi_version = _lib.ptts_abi_version()       # never a literal: the library refuses any other value
        cfg.dtype = int(bf16)
        cfg.hidden, cfg.layers, cfg.heads, cfg.ffn = d.hidden_size, d.num_hidden_layers, d.num_attention_heads, d.ffn_dim
        cfg.vocab, cfg.codebooks, cfg.max_positions = d.vocab_size, d.num_codebooks, d.max_position_embeddings
        cfg.pad_token_id, cfg.eos_token_id, cfg.bos_token_id = d.pad_token_id, d.eos_token_id, d.bos_token_id
        cfg.prompt_vocab, cfg.enc_hidden = model.config.vocab_size, t5.d_model
        cfg.t5_layers, cfg.t5_heads, cfg.t5_dkv, cfg.t5_dff, cfg.t5_vocab = 0, t5.num_heads, t5.d_kv, t5.d_ff, t5.vocab_size
        cfg.t5_gated, cfg.t5_act, cfg.t5_eps = 1, 2, t5.layer_norm_epsilon
        cfg.max_batch, cfg.max_desc_len, cfg.max_prompt_len, cfg.max_length = max_batch, max_desc_len, max_prompt_len, max_length
        # codec geometry from the tensors themselves (DACConfig, dac_wrapper/configuration_dac.py:8-25, holds only
        # codebook_size / latent_dim / num_codebooks; the rest are the descript package's constructor defaults)
        dec0 = sd.get("audio_encoder.model.decoder.model.0.weight")
        self.hop = 0
        if dec0 is not None:
            book = sd["audio_encoder.model.quantizer.quantizers.0.codebook.weight"]
            cfg.dac_codebook_size, cfg.dac_codebook_dim = book.shape
            cfg.dac_hidden, cfg.dac_latent = dec0.shape[0], dec0.shape[1]
            n = 0
            while f"audio_encoder.model.decoder.model.{n + 1}.block.1.weight" in sd:      # ConvTranspose1d [Cin, Cout, 2 * stride]
                cfg.dac_up[n] = sd[f"audio_encoder.model.decoder.model.{n + 1}.block.1.weight"].shape[2] // 2
                n += 1
            cfg.dac_n_up = n
            self.hop = 1
            for i in range(n):
                self.hop *= cfg.dac_up[i]
            enc0 = sd.get("audio_encoder.model.encoder.block.0.weight")
            cfg.dac_enc_hidden = 0 if enc0 is None else enc0.shape[0]
        self.codebooks, self.pad, self.device = d.num_codebooks, d.pad_token_id, torch.device(device)
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _ck(_lib.ptts_create(C.byref(cfg), C.byref(self.h)))
            keep = []                                       # the staging copies must outlive the asynchronous pack kernels
            for name, p in sd.items():                      # the reference's parameter names are the ABI's tensor names
                t = p.detach().to(self.device, torch.float32).contiguous()
                keep.append(t)
                shape = (C.c_int64 * t.dim())(*t.shape)
                _ck(_lib.ptts_load_tensor(self.h, name.encode(), _p(t), shape, t.dim(), None))
            torch.cuda.synchronize(self.device)
            _ck(_lib.ptts_weights_ready(self.h))

    def close(self):
        if self.h:
            _lib.ptts_destroy(self.h)
            self.h = C.c_void_p()

    def run(self, encoder_hidden, attention_mask, prompt_input_ids, prompt_attention_mask, delay_pattern_mask, input_ids,
            max_length, do_sample=False, temperature=1.0, top_k=0, top_p=1.0, guidance_scale=None, seed=0,
            eos_token_id=-1, pad_token_id=-1, min_length=0, min_new_tokens=0):
        """replaces M:2582-2620: returns the raw output ids [B*K, L_out] exactly as greedy_search / sample would.
        Under guidance the caller passes what M:2231-2237 built: B_eff = 2 B rows, the second half zeroed."""
        dev = self.device
        BK, L = delay_pattern_mask.shape               # BK = B * num_codebooks
        assert L == max_length and BK % self.codebooks == 0
        B = BK // self.codebooks
        B_eff = encoder_hidden.shape[0]
        start = input_ids.shape[1]
        raw = torch.full((BK, L), pad_token_id if pad_token_id >= 0 else self.pad, dtype=torch.long, device=dev)
        raw[:, :start] = input_ids.to(dev)
        enc = encoder_hidden.to(dev, torch.float32).contiguous()
        am = attention_mask.to(dev, torch.int32).contiguous()
        pids = prompt_input_ids.to(dev, torch.long).contiguous()
        pm = None if prompt_attention_mask is None else prompt_attention_mask.to(dev, torch.int32).contiguous()
        pat = delay_pattern_mask.to(dev, torch.long).contiguous()
        # MinLength / MinNewTokensLength processors (the `_get_logits_processor` call at M:2561-2567) as one bound
        free_until = max(min_length or 0, start + (min_new_tokens or 0)) if (min_length or min_new_tokens) else 0
        sp = PttsSampling(int(do_sample), temperature, top_k or 0, guidance_scale or 1.0, seed, top_p or 1.0,
                          eos_token_id, pad_token_id, free_until)
        st = PttsState()
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)                # stream NULL below = host-synchronous calls on the engine's own stream
            _ck(_lib.ptts_prefill(self.h, B, B_eff, enc.shape[1], pids.shape[1], L, _p(enc), _p(am), _p(pids), _p(pm),
                                  _p(pat), _p(raw), start, None))
            _ck(_lib.ptts_decode_steps(self.h, L - start, C.byref(sp), 0, None))
            # the loop state MUST be read before the ids are used: it carries the end of the run (every row at EOS) and any
            # error of the in-grid hand-off (include/parler_tts_hip.h, ptts_get_state)
            _ck(_lib.ptts_get_state(self.h, C.byref(st), None))
        return raw[:, : (st.done_len or st.cur_len)]    # the un-delay continues unchanged at M:2633

    def dac_decode(self, audio_codes):
        """replaces self.audio_encoder.decode(...) at M:2660 / M:2671: audio_codes [1, B, K, F] -> [B, 1, F * hop]"""
        codes = audio_codes.squeeze(0).to(self.device, torch.long).contiguous()
        B, K, F = codes.shape
        out = torch.empty(B, F * self.hop, device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _ck(_lib.ptts_dac_decode(self.h, B, F, _p(codes), _p(out), None))
        return out[:, None, :]

    def dac_encode(self, input_values):
        """replaces encoder.encode(...) at M:2276 (audio prompts): input_values [B, 1, T] -> audio_codes [1, B, K, T // hop]"""
        x = input_values[:, 0].to(self.device, torch.float32).contiguous()
        B, T = x.shape
        codes = torch.empty(B, self.codebooks, T // self.hop, dtype=torch.long, device=self.device)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            _ck(_lib.ptts_dac_encode(self.h, B, T, _p(x), _p(codes), None))
        return codes[None]
