#!/usr/bin/env python3
"""Do independent utterance groups overlap on one GPU?  G engines (batch 32/G each), each driven from its own host
thread on its own non-blocking stream, vs one engine with batch 32.   python3 tools/two_engines.py [max_length] [G ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
from parler_tts_amd.weights import synthetic_state_dict

L = int(sys.argv[1]) if len(sys.argv) > 1 else 2580
groups = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
B, T, P = 32, 40, 16
g = torch.Generator().manual_seed(1234)
desc = torch.randint(2, 32100, (B, T), generator=g)
prompt = torch.randint(2, 32100, (B, P), generator=g)

ref_codes = None
for G in groups:
    models = [ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0") for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    n = B // G
    outs = [None] * G

    def work(i):
        with torch.cuda.stream(streams[i]):
            outs[i] = models[i].generate(input_ids=desc[i * n:(i + 1) * n], prompt_input_ids=prompt[i * n:(i + 1) * n],
                                         do_sample=False, max_length=L, return_raw_ids=True)
            streams[i].synchronize()

    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=work, args=(i,)) for i in range(G)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    codes = torch.cat([o[0] for o in outs], dim=0)
    if ref_codes is None:
        ref_codes = codes
    same = bool(torch.equal(codes, ref_codes))
    if not same:
        neq = codes != ref_codes
        print("   first differing frame per utterance:", [int(neq[i].any(0).nonzero()[0]) if neq[i].any() else -1 for i in range(B)])
    print(f"G={G} engines x batch {n}: {dt * 1e3:.0f} ms per pass (T5+prefill+decode, no DAC), "
          f"{B * 9 * (L - 1) / dt:.0f} tok/s, codes equal to G={groups[0]}: {same}", flush=True)
    del models
    torch.cuda.empty_cache()
