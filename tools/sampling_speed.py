#!/usr/bin/env python3
"""Decode-loop time of greedy vs sampling (top_k=50, temperature=0.7, BASELINE configs[3]) at batch 32."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
from parler_tts_amd.weights import synthetic_state_dict

L = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
m = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
g = torch.Generator().manual_seed(1234)
desc = torch.randint(2, 32100, (32, 40), generator=g)
prompt = torch.randint(2, 32100, (32, 16), generator=g)
for name, kw in (("greedy", dict(do_sample=False)), ("sample top_k=50 T=0.7", dict(do_sample=True, top_k=50, temperature=0.7)),
                 ("sample top_k=50 top_p=0.9", dict(do_sample=True, top_k=50, top_p=0.9, temperature=1.0)),
                 ("greedy + guidance_scale 3 (64 rows)", dict(do_sample=False, guidance_scale=3.0))):
    for rep in range(2):
        m.phase_times = {"t5_prefill_ms": 0.0, "decode_ms": 0.0, "post_ms": 0.0}
        m.generate(input_ids=desc, prompt_input_ids=prompt, max_length=L, return_raw_ids=True, **kw)
    print(f"{name}: decode {m.phase_times['decode_ms']:.1f} ms = {m.phase_times['decode_ms'] / (L - 1) * 1e3:.1f} us/step", flush=True)
