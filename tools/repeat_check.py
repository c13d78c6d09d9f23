#!/usr/bin/env python3
"""Repeatability under GPU sharing: run N identical greedy passes and compare each with the first.
   python3 tools/repeat_check.py [max_length] [passes] [tag]   (start two of these at once to share the GPU)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
from parler_tts_amd.weights import synthetic_state_dict

L = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tag = sys.argv[3] if len(sys.argv) > 3 else ""
dtype = sys.argv[4] if len(sys.argv) > 4 else "bf16"
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
B, T, P = 32, 40, 16
g = torch.Generator().manual_seed(1234)
desc = torch.randint(2, 32100, (B, T), generator=g)
prompt = torch.randint(2, 32100, (B, P), generator=g)
m = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
ref = None
for i in range(N):
    t0 = time.perf_counter()
    c = m.generate(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ref is None:
        ref = c
    neq = c != ref
    bad = [int(neq[b].any(0).nonzero()[0]) for b in range(B) if neq[b].any()]
    if i == 0 or bad or i == N - 1:
        print(f"{tag} pass {i}: {dt * 1e3:.0f} ms, sum {int(c.sum())}, equal to pass 0: {not bad}" + (f" first differing frames {bad[:8]}" if bad else ""), flush=True)
