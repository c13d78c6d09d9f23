#!/usr/bin/env python3
"""T5 encoder + prefill of the benchmark shape alone (mini_v0.1, bf16, B = 32, T = 40, P = 16), a few short generate() calls: run it
under `rocprofv3 --kernel-trace --stats` for the per-kernel time of the one-off phases, or plainly for the host-timed phase.

    python3 tools/prefill_profile.py [passes] [dtype]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from parler_tts_amd.configuration import mini_v01_config  # noqa: E402
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration  # noqa: E402
from parler_tts_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
    cfg = mini_v01_config()
    sd = synthetic_state_dict(cfg, seed=1234, never_eos=True, parts=("text_encoder", "decoder"))
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
    g = torch.Generator().manual_seed(1234)
    B, T, P = 32, 40, 16
    desc = torch.randint(2, 32100, (B, T), generator=g).cuda()
    prompt = torch.randint(2, 32100, (B, P), generator=g).cuda()
    amask = torch.ones(B, T, dtype=torch.long, device="cuda")
    kw = dict(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=12, return_raw_ids=True)
    model.generate(**kw)            # folds, look-ahead weights, workspaces
    model.generate(**kw)
    torch.cuda.synchronize()
    model.phase_times = {"t5_prefill_ms": 0.0, "decode_ms": 0.0, "post_ms": 0.0}
    t0 = time.perf_counter()
    for _ in range(passes):
        model.generate(**kw)
    torch.cuda.synchronize()
    print(f"t5_prefill_ms per pass (host, synchronised at both ends): {model.phase_times['t5_prefill_ms'] / passes:.3f}  "
          f"(whole short call {(time.perf_counter() - t0) / passes * 1e3:.3f} ms)")


if __name__ == "__main__":
    main()
