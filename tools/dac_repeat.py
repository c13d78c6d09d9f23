#!/usr/bin/env python3
"""Codec decode repeated under GPU sharing: bitwise repeatable?  (start two at once)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import DACModel
from parler_tts_amd.weights import synthetic_state_dict

tag = sys.argv[1] if len(sys.argv) > 1 else ""
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=1, parts=("audio_encoder",))
dac = DACModel(cfg, sd, device="cuda:0")
codes = torch.randint(0, 1024, (1, 4, 9, 600), generator=torch.Generator().manual_seed(3))
ref = None
bad = 0
for i in range(N):
    w = dac.decode(codes, None).audio_values
    torch.cuda.synchronize()
    if ref is None:
        ref = w.clone()
    elif not torch.equal(w, ref):
        bad += 1
        d = (w != ref)
        if bad <= 4:
            print(f"{tag} pass {i}: {int(d.sum())} samples differ, maxdiff {(w - ref).abs().max().item():.3e}, first at {d.nonzero()[0].tolist()}", flush=True)
print(f"{tag}: {bad} of {N - 1} repeats differ", flush=True)
