#!/usr/bin/env python3
"""Codec ENCODE (audio prompts, ptts_dac_encode) at mini_v0.1 size: time per call and two size-independent checks
(an utterance alone == the same utterance in a batch; encode -> decode -> encode keeps the frame count).
   python3 tools/dac_encode_speed.py [B] [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import DACModel
from parler_tts_amd.weights import synthetic_dac_encoder_state_dict, synthetic_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=1, parts=("audio_encoder",))
sd.update(synthetic_dac_encoder_state_dict(cfg.audio_encoder, seed=1))
dac = DACModel(cfg, sd, device="cuda:0")
T = int(secs * cfg.audio_encoder.sampling_rate) + 123          # not a multiple of the hop on purpose
g = torch.Generator().manual_seed(3)
audio = ((torch.rand(B, 1, T, generator=g) * 2 - 1) * 0.5).cuda()
codes = dac.encode(audio).audio_codes
torch.cuda.synchronize()
t0 = time.perf_counter()
codes = dac.encode(audio).audio_codes
torch.cuda.synchronize()
dt = time.perf_counter() - t0
F = T // cfg.audio_encoder.hop_length
assert codes.shape == (1, B, 9, F), codes.shape
assert int(codes.min()) >= 0 and int(codes.max()) < 1024
one = dac.encode(audio[B // 2: B // 2 + 1]).audio_codes
same = bool(torch.equal(one[0, 0], codes[0, B // 2]))
wav = dac.decode(codes, [None]).audio_values
again = dac.encode(wav).audio_codes
# encoder MACs per input sample (mini: d = 64, strides 2,4,8,8): residual units (7 + 1 taps) * C^2 per sample at that rate
d, rate, macs = 64, 1.0, 7 * 64
for s in (2, 4, 8, 8):
    macs += rate * 3 * (7 + 1) * d * d
    macs += rate / s * (2 * s) * d * (2 * d)
    d, rate = 2 * d, rate / s
macs += rate * 3 * d * 1024
print(f"B={B} T={T} ({secs:.0f} s clips, {F} frames): {dt * 1e3:.1f} ms per call, {B * T * 2 * macs / dt / 1e12:.1f} TFLOP/s fp32-equivalent "
      f"({2 * macs / 1e6:.2f} MFLOP per sample), alone==in-batch: {same}, re-encode frames {again.shape[-1]} == {F}")
assert same and again.shape[-1] == F
