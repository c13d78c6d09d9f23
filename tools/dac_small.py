#!/usr/bin/env python3
"""Small codec decode (mini codec, B x F frames) for kernel bring-up: prints a checksum, optionally saves the waveform.
   python3 tools/dac_small.py [B] [F] [out.npy]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import DACModel
from parler_tts_amd.weights import synthetic_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=1, parts=("audio_encoder",))
dac = DACModel(cfg, sd, device="cuda:0")
codes = torch.randint(0, 1024, (1, B, 9, F), generator=torch.Generator().manual_seed(0))
wav = dac.decode(codes, None).audio_values
torch.cuda.synchronize()
w = wav.cpu().numpy()
print("finite:", bool(np.isfinite(w).all()), "absmax", float(np.abs(w).max()), "sha1", hashlib.sha1(w.tobytes()).hexdigest()[:16], flush=True)
if len(sys.argv) > 3:
    np.save(sys.argv[3], w)
