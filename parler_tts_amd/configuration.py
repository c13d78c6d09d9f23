"""Configuration objects for the MI355X Parler-TTS generate() engine.

Field names and defaults follow the reference so that a checkpoint's ``config.json``
loads unchanged:

* ``ParlerTTSDecoderConfig``  - /root/reference/parler_tts/configuration_parler_tts.py:82-127
* ``ParlerTTSConfig``         - /root/reference/parler_tts/configuration_parler_tts.py:193-238
* ``DACConfig``               - /root/reference/parler_tts/dac_wrapper/configuration_dac.py:8-25
* ``T5EncoderConfig``         - the subset of the upstream T5 config the text-encoder path reads
                                (third party; the reference instantiates it at modeling_parler_tts.py:1598)

These are plain Python objects (no ``transformers`` dependency): the product path only needs
``to_dict``/``from_dict`` and ``from_pretrained(local_dir)`` reading an HF style ``config.json``.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, Optional


class _ConfigBase:
    model_type = ""

    def to_dict(self) -> Dict[str, Any]:
        out = {}
        for k, v in self.__dict__.items():
            out[k] = v.to_dict() if isinstance(v, _ConfigBase) else copy.deepcopy(v)
        out["model_type"] = self.model_type
        return out

    @classmethod
    def from_dict(cls, d: Dict[str, Any]):
        d = dict(d)
        d.pop("model_type", None)
        return cls(**d)

    @classmethod
    def from_pretrained(cls, path: str):
        cfg_file = path if path.endswith(".json") else os.path.join(path, "config.json")
        with open(cfg_file, "r") as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)

    def __repr__(self):
        return f"{type(self).__name__} {json.dumps(self.to_dict(), indent=2, sort_keys=True)}"

    def __eq__(self, other):
        return type(self) is type(other) and self.to_dict() == other.to_dict()


class ParlerTTSDecoderConfig(_ConfigBase):
    """Decoder (MusicGen-style) hyper-parameters; defaults = reference C:82-103."""

    model_type = "parler_tts_decoder"

    def __init__(
        self,
        vocab_size: int = 2049,
        max_position_embeddings: int = 2048,
        num_hidden_layers: int = 24,
        ffn_dim: int = 4096,
        num_attention_heads: int = 16,
        layerdrop: float = 0.0,
        use_cache: bool = True,
        activation_function: str = "gelu",
        hidden_size: int = 1024,
        dropout: float = 0.1,
        attention_dropout: float = 0.0,
        activation_dropout: float = 0.0,
        initializer_factor: float = 0.02,
        scale_embedding: bool = False,
        num_codebooks: int = 4,
        pad_token_id: int = 2048,
        bos_token_id: int = 2049,
        eos_token_id: int = 2048,
        tie_word_embeddings: bool = False,
        **kwargs,
    ):
        self.vocab_size = vocab_size
        self.max_position_embeddings = max_position_embeddings
        self.hidden_size = hidden_size
        self.ffn_dim = ffn_dim
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.dropout = dropout
        self.attention_dropout = attention_dropout
        self.activation_dropout = activation_dropout
        self.activation_function = activation_function
        self.initializer_factor = initializer_factor
        self.layerdrop = layerdrop
        self.use_cache = use_cache
        self.scale_embedding = scale_embedding
        self.num_codebooks = num_codebooks
        self.pad_token_id = pad_token_id
        self.bos_token_id = bos_token_id
        self.eos_token_id = eos_token_id
        self.tie_word_embeddings = tie_word_embeddings
        # audio_channels is read at M:2042; mono unless a checkpoint says otherwise
        self.audio_channels = kwargs.pop("audio_channels", 1)
        self.cross_attention_hidden_size = kwargs.pop("cross_attention_hidden_size", None)
        # Decoder-op variants for random-init "scaled decoder" studies (BASELINE configs[4]).  They are
        # EXTENSIONS: the reference's decoder has LayerNorm with bias, additive sinusoidal positions and a 2-matrix GELU MLP
        # (M:402-413, M:181-223, M:488-490) and no switch for anything else, so these have no oracle in the reference
        # ("unpinned"); the defaults are the reference's ops.
        #   norm_type          "layer_norm" | "rms_norm"  (x * rsqrt(mean(x^2) + 1e-5) * weight; the bias tensors are ignored)
        #   position_embedding "sinusoidal" | "rope"      (no additive table; q, k of the SELF-attention rotated per head,
        #                                                  pairs (i, i + head_dim / 2), angle pos * rope_theta^(-2 i / head_dim))
        #   mlp_type           "gelu" | "swiglu"          (fc2(silu(fc1(x)) * fc1_up(x)); extra tensor layers.{l}.fc1_up.weight)
        self.norm_type = kwargs.pop("norm_type", "layer_norm")
        self.position_embedding = kwargs.pop("position_embedding", "sinusoidal")
        self.rope_theta = float(kwargs.pop("rope_theta", 10000.0))
        self.mlp_type = kwargs.pop("mlp_type", "gelu")
        if self.norm_type not in ("layer_norm", "rms_norm") or self.position_embedding not in ("sinusoidal", "rope") or \
                self.mlp_type not in ("gelu", "swiglu"):
            raise ValueError("norm_type / position_embedding / mlp_type: unknown variant")
        self._extra = {k: v for k, v in kwargs.items() if not k.startswith("_")}

    def to_dict(self):
        d = super().to_dict()
        d.update(d.pop("_extra", {}))
        return d

    @property
    def variant_flags(self) -> int:
        """bit 0 RMSNorm, bit 1 RoPE, bit 2 SwiGLU (ptts_config.dec_variant)"""
        return (1 if self.norm_type == "rms_norm" else 0) | (2 if self.position_embedding == "rope" else 0) | \
            (4 if self.mlp_type == "swiglu" else 0)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


class T5EncoderConfig(_ConfigBase):
    """The T5 fields the encoder-only path needs (upstream names). Defaults = flan-t5-base."""

    model_type = "t5"

    def __init__(
        self,
        vocab_size: int = 32128,
        d_model: int = 768,
        d_kv: int = 64,
        d_ff: int = 2048,
        num_layers: int = 12,
        num_heads: int = 12,
        relative_attention_num_buckets: int = 32,
        relative_attention_max_distance: int = 128,
        layer_norm_epsilon: float = 1e-6,
        feed_forward_proj: str = "gated-gelu",
        **kwargs,
    ):
        self.vocab_size = vocab_size
        self.d_model = d_model
        self.d_kv = d_kv
        self.d_ff = d_ff
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.relative_attention_num_buckets = relative_attention_num_buckets
        self.relative_attention_max_distance = relative_attention_max_distance
        self.layer_norm_epsilon = layer_norm_epsilon
        self.feed_forward_proj = feed_forward_proj
        self._extra = {k: v for k, v in kwargs.items() if not k.startswith("_")}

    def to_dict(self):
        d = super().to_dict()
        d.update(d.pop("_extra", {}))
        return d

    @property
    def hidden_size(self) -> int:
        return self.d_model

    @property
    def is_gated_act(self) -> bool:
        return self.feed_forward_proj.startswith("gated-")

    @property
    def dense_act_fn(self) -> str:
        act = self.feed_forward_proj.split("-")[-1]
        # upstream maps "gated-gelu" to the tanh approximation ("gelu_new")
        return "gelu_new" if self.feed_forward_proj == "gated-gelu" else act


class DACConfig(_ConfigBase):
    """Codec description, reference dac_wrapper/configuration_dac.py:8-25.

    ``decoder_hidden_size`` / ``upsampling_ratios`` / ``codebook_dim`` are the descript-audio-codec
    package defaults the reference relies on implicitly (D:19-23 passes only three arguments).
    """

    model_type = "dac"

    def __init__(
        self,
        num_codebooks: int = 9,
        model_bitrate: int = 8,
        codebook_size: int = 1024,
        latent_dim: int = 1024,
        frame_rate: int = 86,
        sampling_rate: int = 44100,
        decoder_hidden_size: int = 1536,
        upsampling_ratios=(8, 8, 4, 2),
        codebook_dim: int = 8,
        encoder_hidden_size: int = 64,
        **kwargs,
    ):
        self.encoder_hidden_size = encoder_hidden_size    # descript `encoder_dim` default; strides = reversed(upsampling_ratios)
        self.codebook_size = codebook_size
        self.model_bitrate = model_bitrate
        self.latent_dim = latent_dim
        self.num_codebooks = num_codebooks
        self.frame_rate = frame_rate
        self.sampling_rate = sampling_rate
        self.decoder_hidden_size = decoder_hidden_size
        self.upsampling_ratios = list(upsampling_ratios)
        self.codebook_dim = codebook_dim
        self._extra = {k: v for k, v in kwargs.items() if not k.startswith("_")}

    def to_dict(self):
        d = super().to_dict()
        d.update(d.pop("_extra", {}))
        return d

    @property
    def hop_length(self) -> int:
        h = 1
        for r in self.upsampling_ratios:
            h *= r
        return h


class GenerationConfig(_ConfigBase):
    """The generation parameters ``generate()`` reads (M:2416-2421 merges kwargs into a copy).

    Defaults follow ``helpers/model_init_scripts/init_model_600M.py:55-62`` plus the 4.3x-era HF
    defaults (top_k=50, top_p=1.0, temperature=1.0, num_beams=1).
    """

    model_type = "generation"

    def __init__(self, **kwargs):
        self.max_length = kwargs.pop("max_length", 2580)
        self.max_new_tokens = kwargs.pop("max_new_tokens", None)
        self.min_length = kwargs.pop("min_length", 0)
        self.min_new_tokens = kwargs.pop("min_new_tokens", None)
        self.do_sample = kwargs.pop("do_sample", True)
        self.temperature = kwargs.pop("temperature", 1.0)
        self.top_k = kwargs.pop("top_k", 50)
        self.top_p = kwargs.pop("top_p", 1.0)
        self.num_beams = kwargs.pop("num_beams", 1)
        self.num_beam_groups = kwargs.pop("num_beam_groups", 1)
        self.num_return_sequences = kwargs.pop("num_return_sequences", 1)
        self.guidance_scale = kwargs.pop("guidance_scale", 1)
        self.pad_token_id = kwargs.pop("pad_token_id", None)
        self.bos_token_id = kwargs.pop("bos_token_id", None)
        self.eos_token_id = kwargs.pop("eos_token_id", None)
        self.decoder_start_token_id = kwargs.pop("decoder_start_token_id", None)
        self.use_cache = kwargs.pop("use_cache", True)
        self.return_dict_in_generate = kwargs.pop("return_dict_in_generate", False)
        self.output_scores = kwargs.pop("output_scores", False)
        self.seed = kwargs.pop("seed", None)  # Philox seed of the on-device sampler (extension)
        self._extra = {k: v for k, v in kwargs.items() if not k.startswith("_")}

    def to_dict(self):
        d = super().to_dict()
        d.update(d.pop("_extra", {}))
        return d

    _FIELDS = (
        "max_length max_new_tokens min_length min_new_tokens do_sample temperature top_k top_p num_beams num_beam_groups "
        "num_return_sequences guidance_scale pad_token_id bos_token_id eos_token_id decoder_start_token_id "
        "use_cache return_dict_in_generate output_scores seed"
    ).split()

    def update(self, **kwargs) -> Dict[str, Any]:
        """Absorb the kwargs that are generation parameters; return the rest (= model kwargs)."""
        rest = {}
        for k, v in kwargs.items():
            if k in self._FIELDS:
                setattr(self, k, v)
            else:
                rest[k] = v
        return rest

    def validate(self) -> None:
        if self.max_length is not None and self.max_length <= 0:
            raise ValueError("`max_length` must be positive")
        if self.do_sample and self.temperature is not None and self.temperature <= 0:
            raise ValueError("`temperature` must be strictly positive when sampling")
        if self.top_k is not None and self.top_k < 0:
            raise ValueError("`top_k` must be >= 0")
        # fields of a generation_config.json this engine does not act on: bookkeeping keys are ignored, a generation
        # parameter left at its neutral HF default is ignored, anything else would silently change the output -> refuse
        for k, v in self._extra.items():
            if k in self._INERT or k.startswith("output_"):
                continue
            if k in self._NEUTRAL and (v is None or v == self._NEUTRAL[k]):
                continue
            raise NotImplementedError(f"generation_config field `{k}` = {v!r} is not served by this engine "
                                      "(it would change the generated tokens in the reference)")

    _INERT = {"transformers_version", "_from_model_config", "_commit_hash", "_original_object_hash", "return_legacy_cache",
              "cache_implementation", "cache_config", "use_cache", "max_time", "generation_kwargs", "is_assistant",
              "return_dict_in_generate", "low_memory", "watermarking_config", "token_healing", "compile_config",
              "disable_compile", "trust_remote_code", "stop_strings"}
    _NEUTRAL = {"repetition_penalty": 1.0, "encoder_repetition_penalty": 1.0, "no_repeat_ngram_size": 0,
                "encoder_no_repeat_ngram_size": 0, "bad_words_ids": None, "force_words_ids": None, "typical_p": 1.0,
                "epsilon_cutoff": 0.0, "eta_cutoff": 0.0, "min_p": None, "diversity_penalty": 0.0, "length_penalty": 1.0,
                "early_stopping": False, "penalty_alpha": None, "renormalize_logits": False, "constraints": None,
                "forced_bos_token_id": None, "forced_eos_token_id": None, "remove_invalid_values": False,
                "exponential_decay_length_penalty": None, "suppress_tokens": None, "begin_suppress_tokens": None,
                "forced_decoder_ids": None, "sequence_bias": None, "num_assistant_tokens": None, "dola_layers": None,
                "prompt_lookup_num_tokens": None, "temperature_last_layer": None, "top_h": None}


class ParlerTTSConfig(_ConfigBase):
    """Composite config: text encoder + audio codec + decoder (reference C:193-238)."""

    model_type = "parler_tts"
    is_composition = True

    def __init__(self, vocab_size: int = 1024, **kwargs):
        if "text_encoder" not in kwargs or "audio_encoder" not in kwargs or "decoder" not in kwargs:
            raise ValueError("Config has to be initialized with text_encoder, audio_encoder and decoder config")

        def _sub(v, cls):
            return v if isinstance(v, cls) else cls.from_dict(v.to_dict() if hasattr(v, "to_dict") else v)

        self.vocab_size = vocab_size
        self.text_encoder = _sub(kwargs.pop("text_encoder"), T5EncoderConfig)
        self.audio_encoder = _sub(kwargs.pop("audio_encoder"), DACConfig)
        self.decoder = _sub(kwargs.pop("decoder"), ParlerTTSDecoderConfig)
        self.is_encoder_decoder = True
        self.pad_token_id = kwargs.pop("pad_token_id", None)
        self.decoder_start_token_id = kwargs.pop("decoder_start_token_id", None)
        self._extra = {k: v for k, v in kwargs.items() if not k.startswith("_")}

    def to_dict(self):
        d = super().to_dict()
        d.update(d.pop("_extra", {}))
        return d

    @classmethod
    def from_sub_models_config(cls, text_encoder_config, audio_encoder_config, decoder_config, **kwargs):
        """Same entry point as the reference (C:212-233)."""
        return cls(
            text_encoder=text_encoder_config,
            audio_encoder=audio_encoder_config,
            decoder=decoder_config,
            **kwargs,
        )

    @property
    def sampling_rate(self) -> int:
        # reference C:235-238
        return self.audio_encoder.sampling_rate


# --------------------------------------------------------------------------------------
# Named architectures used by tests / bench (no pretrained weights exist offline).
# --------------------------------------------------------------------------------------

def mini_v01_config() -> ParlerTTSConfig:
    """parler_tts_mini_v0.1 as helpers/model_init_scripts/init_model_600M.py:25-65 builds it."""
    dec = ParlerTTSDecoderConfig(
        vocab_size=1024 + 64,
        max_position_embeddings=4096,
        num_hidden_layers=24,
        ffn_dim=4096,
        num_attention_heads=16,
        layerdrop=0.0,
        use_cache=True,
        activation_function="gelu",
        hidden_size=1024,
        dropout=0.1,
        attention_dropout=0.0,
        activation_dropout=0.0,
        pad_token_id=1024,
        eos_token_id=1024,
        bos_token_id=1025,
        num_codebooks=9,
    )
    return ParlerTTSConfig.from_sub_models_config(
        T5EncoderConfig(), DACConfig(), dec, vocab_size=32128, pad_token_id=1024, decoder_start_token_id=1025
    )


def scaled_config() -> ParlerTTSConfig:
    """BASELINE.json configs[4]: random-init decoder d_model=1536, 24 layers, 24 heads, ffn 6144."""
    cfg = mini_v01_config()
    cfg.decoder.hidden_size = 1536
    cfg.decoder.num_attention_heads = 24
    cfg.decoder.ffn_dim = 6144
    return cfg


def tiny_config(num_codebooks: int = 4, hidden: int = 128, layers: int = 2, heads: int = 2, ffn: int = 256,
                codebook_size: int = 64, enc_d: int = 64, max_positions: int = 256, **decoder_variants) -> ParlerTTSConfig:
    """A toy architecture with the same structure, for second-scale CPU tests.

    hidden must be a multiple of 64 (head_dim is 64 as in every released Parler-TTS decoder).
    """
    dec = ParlerTTSDecoderConfig(
        vocab_size=codebook_size + 64,
        max_position_embeddings=max_positions,
        num_hidden_layers=layers,
        ffn_dim=ffn,
        num_attention_heads=heads,
        hidden_size=hidden,
        pad_token_id=codebook_size,
        eos_token_id=codebook_size,
        bos_token_id=codebook_size + 1,
        num_codebooks=num_codebooks,
        **decoder_variants,       # norm_type / position_embedding / rope_theta / mlp_type (extensions, ParlerTTSDecoderConfig)
    )
    t5 = T5EncoderConfig(vocab_size=512, d_model=enc_d, d_kv=32, d_ff=128, num_layers=2, num_heads=2)
    dac = DACConfig(num_codebooks=num_codebooks, codebook_size=codebook_size, latent_dim=64,
                    decoder_hidden_size=256, upsampling_ratios=(4, 2, 2, 2), encoder_hidden_size=16, codebook_dim=8)
    return ParlerTTSConfig.from_sub_models_config(
        t5, dac, dec, vocab_size=512, pad_token_id=codebook_size, decoder_start_token_id=codebook_size + 1
    )
