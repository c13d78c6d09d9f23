#!/usr/bin/env python3
"""A/B on one box: decode loop driven from torch's legacy NULL stream (engine redirects to its own stream, fenced by
events) vs from a non-blocking torch stream handed straight to the engine."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

cfg = mini_v01_config()
B, T, P, L = 32, 40, 16, 2580
model = ParlerTTSForConditionalGeneration.from_synthetic(cfg, seed=0, dtype="bf16", never_eos=True)
g = torch.Generator().manual_seed(1234)
desc = torch.randint(2, 32100, (B, T), generator=g)
prompt = torch.randint(2, 32100, (B, P), generator=g)
kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)


def run(tag):
    for i in range(2):
        model.phase_times = {"t5_prefill_ms": 0.0, "decode_ms": 0.0, "post_ms": 0.0}
        model.generate(**kw)
        print(f"{tag} run {i}: decode {model.phase_times['decode_ms']:.0f} ms", flush=True)


run("NULL stream " + os.environ.get("TAG", ""))
