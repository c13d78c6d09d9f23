#!/usr/bin/env python3
"""Codec decode at random (batch, frames): the round-6 kernel variants against the kernels they replaced, bit for bit.
   python3 tools/fuzz_codec.py [cases] [seed]
Per case: one decode with the library's defaults (packed hand-over + convt_dma_kernel, 64 x 96 wave tiles, four-wave fused units at
C = 96) and one with PTTS_DAC_NO_PRESPLIT=1 PTTS_DAC_WAVE_TILE=32x192 PTTS_DAC_FUSED96_WAVES=8 (the round-5 dispatch) - the waveforms
must be EQUAL; a random clip decoded alone must equal its rows in the batch.  Frame counts are drawn to straddle the row-tile edges
of every block (256-row tiles at strides 8, 64, 256, 512 per frame) and the ring kernel's minimum launch length."""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import DACModel
from parler_tts_amd.weights import synthetic_state_dict

OLD = {"PTTS_DAC_NO_PRESPLIT": "1", "PTTS_DAC_WAVE_TILE": "32x192", "PTTS_DAC_FUSED96_WAVES": "8"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 71
    rng = random.Random(seed)
    cfg = mini_v01_config()
    sd = synthetic_state_dict(cfg, seed=seed, parts=("audio_encoder",))
    g = torch.Generator().manual_seed(seed)
    bad = 0
    for case in range(n):
        B = rng.choice([1, 1, 2, 3, 5])
        kind = rng.random()
        if kind < 0.3:
            F = rng.randint(1, 40)
        elif kind < 0.7:
            F = 32 * rng.randint(1, 40) + rng.choice([-1, 0, 1, 2])       # 256-row tile edges at stride 8: every 32 frames
        else:
            F = rng.randint(41, 1500)
        F = max(1, F)
        codes = torch.randint(0, 1024, (B, 9, F), generator=g)
        out = {}
        for old in (False, True):
            for k, v in OLD.items():
                if old:
                    os.environ[k] = v
                else:
                    os.environ.pop(k, None)
            dac = DACModel(cfg, sd, device="cuda:0")
            out[old] = dac.decode(codes[None], [None]).audio_values.cpu()
            if not old and B > 1:
                r = rng.randrange(B)
                one = dac.decode(codes[None, r:r + 1], [None]).audio_values.cpu()
                if not torch.equal(one, out[old][r:r + 1]):
                    print(f"BAD  case {case}: B {B} F {F}: clip {r} alone differs from the clip in the batch")
                    bad += 1
            del dac
            torch.cuda.empty_cache()
        for k in OLD:
            os.environ.pop(k, None)
        same = torch.equal(out[False], out[True])
        amp = float(out[False].abs().max())
        sat = float((out[False].abs() > 0.999).float().mean())      # a saturated tanh would hide differences
        med = float(out[False].abs().median())
        print(f"{'ok  ' if same and amp > 0.01 and sat < 0.5 else 'BAD '} case {case}: B {B} F {F}: |wave| max {amp:.3f} median {med:.3f} saturated {sat:.3f}, "
              f"equal to the round-5 dispatch {same}")
        bad += 1 if sat >= 0.5 else 0
        bad += 0 if same and amp > 0.01 else 1
    print(f"{n - bad if bad <= n else 0} of {n} cases passed" if not bad else f"FAILED: {bad} bad checks in {n} cases")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
