#!/usr/bin/env python3
"""Stand-alone codec decode at benchmark size for kernel-level profiling:
   rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/dac_profile.py [B] [F]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import DACModel
from parler_tts_amd.weights import synthetic_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2571
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=1, parts=("audio_encoder",))
dac = DACModel(cfg, sd, device="cuda:0")
torch.manual_seed(0)
codes = torch.randint(0, 1024, (1, B, 9, F))
dac.decode(codes, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
wav = dac.decode(codes, None).audio_values
torch.cuda.synchronize()
dt = time.perf_counter() - t0
import hashlib
print("sha1 of the waveform bytes:", hashlib.sha1(wav.cpu().numpy().tobytes()).hexdigest()[:16])
print(f"B={B} F={F}: {dt * 1e3:.1f} ms, {B * F * 1.61e9 / dt / 1e12:.1f} TFLOP/s (1.61 GFLOP per frame)")
if os.environ.get("PTTS_SAVE_WAV"):
    import numpy as np
    np.save(os.environ["PTTS_SAVE_WAV"], wav.cpu().numpy())
