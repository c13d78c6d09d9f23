#!/usr/bin/env python3
"""Feasibility probe: can the codec decode run on a CU-masked side stream while the (latency-bound) decode loop runs,
and what does each side lose?   python3 tools/overlap_probe.py [n_cus ...]"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]


def masked_stream(n_cus, pattern):
    words = (C.c_uint32 * 8)()
    if pattern == "low":
        bits = range(n_cus)
    else:   # spread: every (256/n)-th CU
        bits = range(0, 256, 256 // n_cus)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return s


cfg = mini_v01_config()
B, T, P, L = 32, 40, 16, 2580
model = ParlerTTSForConditionalGeneration.from_synthetic(cfg, seed=0, dtype="bf16", never_eos=True)
g = torch.Generator().manual_seed(1234)
desc = torch.randint(2, 32100, (B, T), generator=g)
prompt = torch.randint(2, 32100, (B, P), generator=g)
kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)
main = torch.cuda.Stream()      # non-blocking: the legacy NULL stream would fence against the masked stream
torch.cuda.set_stream(main)
model.phase_times = {"t5_prefill_ms": 0.0, "decode_ms": 0.0, "post_ms": 0.0}
model.generate(**kw)
model.phase_times = {"t5_prefill_ms": 0.0, "decode_ms": 0.0, "post_ms": 0.0}
model.generate(**kw)
print(f"decode loop alone: {model.phase_times['decode_ms']:.0f} ms", flush=True)
eng = model.engine(B, T, P, L)
codes = torch.randint(0, 1024, (16, 9, 2571), device="cuda:0")
out = torch.empty(16, 2571 * 512, device="cuda:0")
torch.cuda.synchronize()


def dac_pass(stream):
    t0 = time.perf_counter()
    for _ in range(2):   # 32 utterances = 2 x 16
        rc = eng.lib.ptts_dac_decode(eng.handle, 16, 2571, C.c_void_p(codes.data_ptr()), C.c_void_p(out.data_ptr()), stream)
        assert rc == 0, rc
    hip.hipStreamSynchronize(stream)
    return (time.perf_counter() - t0) * 1e3


for spec in sys.argv[1:] or ["256:low", "64:low", "32:low", "16:low"]:
    n, pat = spec.split(":")
    s = masked_stream(int(n), pat)
    dac_pass(s)
    alone = dac_pass(s)
    res = {}

    def side():
        time.sleep(0.3)            # let the decode loop get going
        res["dac"] = dac_pass(s)

    th = threading.Thread(target=side)
    model.phase_times = {"t5_prefill_ms": 0.0, "decode_ms": 0.0, "post_ms": 0.0}
    t0 = time.perf_counter()
    th.start()
    model.generate(**kw)
    th.join()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(f"mask {spec}: DAC(32 utt) alone {alone:.0f} ms | concurrent: decode {model.phase_times['decode_ms']:.0f} ms, "
          f"DAC {res['dac']:.0f} ms, wall {wall:.0f} ms", flush=True)
