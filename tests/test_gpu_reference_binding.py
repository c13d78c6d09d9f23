"""The reference-side binding of INTEGRATION.md section B, EXECUTED: tests/reference_side_binding.py (ctypes over
include/parler_tts_hip.h, nothing of parler_tts_amd imported) drives fixtures the real reference wrote
(modeling_parler_tts.py:2582-2620 is the call site it replaces) and must return the reference's raw ids."""
import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.helpers import case_inputs, config_for, load_case, state_dict_for

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "parler_tts_amd", "csrc", "libparler_tts_hip.so")


@pytest.fixture(scope="module")
def binding():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on an MI355X box")
    old = os.environ.get("PTTS_HIP_LIB")
    os.environ["PTTS_HIP_LIB"] = old or LIB         # what LD_LIBRARY_PATH would be on the reference side
    try:
        sys.modules.pop("tests.reference_side_binding", None)
        mod = importlib.import_module("tests.reference_side_binding")
    finally:
        if old is None:
            del os.environ["PTTS_HIP_LIB"]
    src = open(mod.__file__).read()
    assert "parler_tts_amd" not in src.replace("nothing of parler_tts_amd is imported", "")
    return mod


def _reference_shaped_model(cfg, sd):
    """what the binding touches of the reference's ParlerTTSForConditionalGeneration: state_dict() under the reference's
    parameter names (tests/golden/ref_state_manifest_*.json pins them) and the three config objects"""
    return SimpleNamespace(state_dict=lambda: sd, config=cfg, decoder=SimpleNamespace(config=cfg.decoder),
                           text_encoder=SimpleNamespace(config=cfg.text_encoder))


def _split_weight_norm(sd):
    """codec conv weights as descript stores them: (weight_g, weight_v) pairs the binding has to fold"""
    out = {}
    for k, v in sd.items():
        if k.startswith("audio_encoder.model.") and k.endswith(".weight") and v.dim() == 3:
            out[k[:-7] + ".weight_g"] = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
            out[k[:-7] + ".weight_v"] = v.clone()
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("name", ["tiny_greedy", "tiny_greedy_ragged", "tiny_cfg"])
def test_reference_side_binding_returns_the_reference_ids(binding, name):
    f, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    L = int(f["max_length"])
    gs = meta["gen"].get("guidance_scale")
    enc = torch.from_numpy(f["enc_t5"])                 # T5EncoderModel's last_hidden_state, as M:2229 leaves it
    am, pids, pm = amask, prompt, pmask
    if gs is not None and gs > 1:                       # M:2231-2237 / M:2476-2480: the null half
        enc = torch.cat([enc, torch.zeros_like(enc)])
        am = torch.cat([am, torch.zeros_like(am)])
        pids = pids.repeat(2, 1)
        pm = None if pm is None else pm.repeat(2, 1)
    dec = binding.HipDecoder(_reference_shaped_model(cfg, sd), enc.shape[0], desc.shape[1], prompt.shape[1], L, bf16=False)
    try:
        gold = torch.from_numpy(f["raw_ids"])
        raw = dec.run(enc, am, pids, pm, torch.from_numpy(f["pattern"]), gold[:, :1], L, guidance_scale=gs).cpu()
        # a second call on the same handle (the reference calls generate() repeatedly on one model)
        raw2 = dec.run(enc, am, pids, pm, torch.from_numpy(f["pattern"]), gold[:, :1], L, guidance_scale=gs).cpu()
    finally:
        dec.close()
    assert raw.shape == gold.shape
    free = f["pattern"] == -1                           # columns the delay pattern leaves to the model
    free[:, 0] = False
    assert np.array_equal(raw.numpy()[free], f["raw_ids"][free])
    assert torch.equal(raw, raw2)


def test_reference_side_binding_codec_both_directions(binding):
    from parler_tts_amd.weights import synthetic_dac_encoder_state_dict, synthetic_state_dict

    cfg = config_for("dac_tiny")

    def decoder_for(seed):
        sd = synthetic_state_dict(cfg, seed=seed, parts=("decoder", "audio_encoder"))
        sd.update(synthetic_dac_encoder_state_dict(cfg.audio_encoder, seed=seed))
        return binding.HipDecoder(_reference_shaped_model(cfg, _split_weight_norm(sd)), 2, 4, 4, 8, bf16=False)

    f, _ = load_case("dac_tiny")
    dec = decoder_for(int(f["seed"]))
    try:
        assert dec.hop == cfg.audio_encoder.hop_length
        wav = dec.dac_decode(torch.from_numpy(f["codes"])[None]).cpu().numpy()
    finally:
        dec.close()
    assert wav.shape == f["wav"].shape
    assert float(np.sqrt(((wav - f["wav"]) ** 2).mean())) < 1e-5
    fe, _ = load_case("dac_enc_tiny")
    dec = decoder_for(int(fe["seed"]))
    try:
        codes = dec.dac_encode(torch.from_numpy(fe["audio"])).cpu().numpy()
    finally:
        dec.close()
    assert np.array_equal(codes, fe["codes"])


def test_reference_side_binding_refuses_a_stale_abi_literal(binding):
    """what INTEGRATION.md's round-5 snippet did: a literal ABI version from an older header is PTTS_E_INVALID, loudly"""
    import ctypes as C

    cfg = binding.PttsConfig()
    cfg.abi_version = binding._lib.ptts_abi_version() - 1
    h = C.c_void_p()
    assert binding._lib.ptts_create(C.byref(cfg), C.byref(h)) == -1
    assert b"ABI version" in binding._lib.ptts_last_error()
