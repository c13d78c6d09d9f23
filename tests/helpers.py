"""Shared helpers for the parity tests (fixture loading, synthetic models)."""
import json
import os

import numpy as np
import torch

from parler_tts_amd.configuration import mini_v01_config, tiny_config
from parler_tts_amd.weights import synthetic_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_sd_cache = {}


def config_for(name, meta=None):
    if name.startswith("mini") or name == "dac_mini":
        return mini_v01_config()
    return tiny_config(**((meta or {}).get("cfg_kw") or {}))


def load_case(name):
    f = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(f["meta"])) if "meta" in f else {}
    return f, meta


def state_dict_for(name, meta, parts=("text_encoder", "decoder")):
    cfg = config_for(name, meta)
    key = (name.split("_")[0], json.dumps(meta["sd_kw"], sort_keys=True), json.dumps(meta.get("cfg_kw"), sort_keys=True), tuple(parts))
    if key not in _sd_cache:
        _sd_cache.clear()  # mini dicts are GBs: keep one at a time
        _sd_cache[key] = synthetic_state_dict(cfg, parts=parts, **meta["sd_kw"])
    return cfg, _sd_cache[key]


def case_inputs(f):
    t = lambda k: torch.from_numpy(f[k])
    pmask = t("pmask") if int(f["has_pmask"]) else None
    return t("desc"), t("amask"), t("prompt"), pmask


def forced_tokens(f):
    if "forced_steps" not in f:
        return None
    return {int(s): torch.from_numpy(v) for s, v in zip(f["forced_steps"], f["forced_vals"])}


def free_positions(pattern, steps):
    """[steps, rows] bool: the token produced at step j (column j+1) is kept by the delay pattern"""
    return (pattern[:, 1:1 + steps] == -1).T


def bf16_bits_to_float(bits):
    """int16 array holding the upper halves of float32 words (how the *_refbf16 fixtures store bfloat16 logits) -> float32"""
    return (bits.astype(np.int32) << 16).view(np.float32)
