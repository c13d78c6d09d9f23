"""The checkpoint boundary pinned to the REFERENCE's writer.  `from_pretrained` is the reference user's
entry point (parler_tts/modeling_parler_tts.py:1701-1720); what it must read is what the reference's own classes emit:

* `tests/golden/ref_config_{mini,tiny}.json` - the text of the real `ParlerTTSConfig.to_json_string(use_diff=False)`
  (configuration_parler_tts.py:193-238 + the transformers base class: every bookkeeping key it adds is in there);
* `tests/golden/ref_state_manifest_{mini,tiny}.json` - {parameter name: shape} of the real
  `ParlerTTSForConditionalGeneration.state_dict()`.

Both were written in the build container by `oracle/gen_golden.py --only-ref-config` from the imported reference (data: names,
shapes, hyper-parameters).  No GPU, no reference tree needed here."""
import json
import os

import pytest
import torch

from parler_tts_amd.configuration import ParlerTTSConfig, mini_v01_config, tiny_config
from parler_tts_amd.weights import all_keys, normalize_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARCH = {"mini": mini_v01_config, "tiny": tiny_config}
# the reference's composite state dict lists the T5 input embedding twice (`shared` and the tied `encoder.embed_tokens`)
TIED = "text_encoder.encoder.embed_tokens.weight"


def _own_fields(cfg):
    """the hyper-parameters this package's config classes define themselves (everything else a file carries sits in `_extra`)"""
    return {k: v for k, v in cfg.__dict__.items() if k != "_extra" and not hasattr(v, "to_dict")}


@pytest.mark.parametrize("tag", ["mini", "tiny"])
def test_config_json_written_by_the_reference_classes_loads_with_every_field(tag, tmp_path):
    path = os.path.join(GOLDEN, f"ref_config_{tag}.json")
    raw = json.load(open(path))
    assert raw["model_type"] == "parler_tts" and "transformers_version" in raw          # the reference's writer, not ours
    cfg = ParlerTTSConfig.from_pretrained(path)
    want = ARCH[tag]()
    for part in ("decoder", "text_encoder", "audio_encoder"):
        got_p, want_p = _own_fields(getattr(cfg, part)), _own_fields(getattr(want, part))
        default_p = _own_fields(type(getattr(cfg, part))())
        assert set(got_p) == set(want_p), part
        for k, v in want_p.items():
            if k in raw[part]:                   # the field is in the reference's file: it must arrive unchanged
                assert got_p[k] == raw[part][k] == v or (isinstance(v, (list, tuple)) and list(got_p[k]) == list(v)), (part, k, got_p[k], v)
            else:                                # not in the reference's file (codec internals it leaves to descript's package
                assert got_p[k] == default_p[k], (part, k)     # defaults, D:19-23; derived T5 fields): this package's class default
    assert cfg.vocab_size == raw["vocab_size"] == want.vocab_size and cfg.is_encoder_decoder is True
    # nothing the reference wrote is dropped: bookkeeping keys survive a save / load round trip through this package
    cfg.save_pretrained(str(tmp_path))
    again = json.load(open(tmp_path / "config.json"))
    for k in ("transformers_version", "architectures", "model_type"):
        assert again[k] == raw[k]
    assert ParlerTTSConfig.from_pretrained(str(tmp_path)) == cfg
    # the directory form (config.json inside a checkpoint directory), as from_pretrained(path) reads it
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "config.json").write_text(open(path).read())
    assert ParlerTTSConfig.from_pretrained(str(d)) == cfg


@pytest.mark.parametrize("tag", ["mini", "tiny"])
def test_state_dict_manifest_of_the_reference_class_is_the_key_layout_the_loader_expects(tag):
    man = json.load(open(os.path.join(GOLDEN, f"ref_state_manifest_{tag}.json")))["tensors"]
    cfg = ParlerTTSConfig.from_pretrained(os.path.join(GOLDEN, f"ref_config_{tag}.json"))
    mine = {k: list(v) for k, v in all_keys(cfg).items() if not k.startswith("audio_encoder.")}
    assert set(man) - set(mine) == {TIED} and not (set(mine) - set(man))
    assert man[TIED] == man["text_encoder.shared.weight"]
    bad = [(k, mine[k], man[k]) for k in mine if mine[k] != man[k]]
    assert not bad, bad[:5]
    dec = cfg.decoder
    # decoder incl. heads, embeddings and the sinusoid table = 427.1 M parameters at mini_v0.1
    if tag == "mini":
        n_dec = sum(int(torch.Size(v).numel()) for k, v in man.items() if k.startswith("decoder."))
        assert round(n_dec / 1e6, 1) == 427.1
        assert man["decoder.model.decoder.embed_positions.weights"] == [dec.max_position_embeddings, dec.hidden_size]
        assert man[f"decoder.lm_heads.{dec.num_codebooks - 1}.weight"] == [dec.vocab_size, dec.hidden_size]
        assert man["decoder.model.decoder.embed_tokens.0.weight"] == [dec.vocab_size + 1, dec.hidden_size]


def test_a_state_dict_with_exactly_the_reference_names_passes_the_loader():
    """tiny architecture: tensors of the manifest's names and shapes (+ the codec's, which the stubbed reference class does not
    hold) go through `normalize_state_dict` - the tied embedding is accepted, nothing is reported missing, nothing is renamed."""
    man = json.load(open(os.path.join(GOLDEN, "ref_state_manifest_tiny.json")))["tensors"]
    cfg = ParlerTTSConfig.from_pretrained(os.path.join(GOLDEN, "ref_config_tiny.json"))
    # the reference's file does not carry the codec internals the build needs for the tiny codec: take them from the named architecture
    cfg.audio_encoder = tiny_config().audio_encoder
    sd = {k: torch.zeros(*v) for k, v in man.items()}
    sd.update({k: torch.zeros(*v) for k, v in all_keys(cfg).items() if k.startswith("audio_encoder.")})
    del sd["text_encoder.shared.weight"]                 # a safetensors file keeps ONE of two tied tensors: either must do
    out = normalize_state_dict(dict(sd), cfg)
    assert set(all_keys(cfg)) <= set(out) and out["text_encoder.shared.weight"].shape == torch.Size(man[TIED])
    sd2 = dict(sd)
    del sd2["decoder.model.decoder.layers.0.fc1.weight"]
    with pytest.raises(KeyError):
        normalize_state_dict(sd2, cfg)


def test_reference_package_name_resolves_to_the_engine():
    """`from parler_tts import ParlerTTSForConditionalGeneration` (the reference README's first line of code, README.md:36) must
    run unchanged against this tree.  In a subprocess: the container-side oracle tests import the REAL reference under the same
    module name (oracle/ref_shim.py evicts the alias first)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import parler_tts, parler_tts_amd; from parler_tts import ParlerTTSForConditionalGeneration, ParlerTTSForCausalLM, "
            "ParlerTTSConfig, ParlerTTSDecoderConfig, DACConfig, DACModel, build_delay_pattern_mask, apply_delay_pattern_mask; "
            "assert ParlerTTSForConditionalGeneration is parler_tts_amd.ParlerTTSForConditionalGeneration; "
            "assert ParlerTTSConfig is parler_tts_amd.ParlerTTSConfig and parler_tts.__version__ == parler_tts_amd.__version__; "
            "print('alias ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300,
                       env={**os.environ, "PYTHONPATH": root})
    assert r.returncode == 0 and "alias ok" in r.stdout, r.stderr[-1500:]
