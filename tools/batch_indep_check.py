#!/usr/bin/env python3
"""Are greedy codes (a) repeatable run to run, (b) independent of how the batch is split over engines?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
from parler_tts_amd.weights import synthetic_state_dict

L = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
B, T, P = 32, 40, 16
g = torch.Generator().manual_seed(1234)
desc = torch.randint(2, 32100, (B, T), generator=g)
prompt = torch.randint(2, 32100, (B, P), generator=g)
m = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")


def run(lo, hi):
    return m.generate(input_ids=desc[lo:hi], prompt_input_ids=prompt[lo:hi], do_sample=False, max_length=L, return_raw_ids=True)[0]


a = run(0, 32)
b = run(0, 32)
print("repeatable:", bool(torch.equal(a, b)))
for n in (16, 8, 17):
    parts = torch.cat([run(lo, min(lo + n, 32)) for lo in range(0, 32, n)], 0)
    neq = (parts != a)
    first = [int(neq[i].any(0).nonzero()[0]) if neq[i].any() else -1 for i in range(32)]
    print(f"split into batches of {n}: equal {bool(torch.equal(parts, a))}; first differing frame per utterance: {first}")
