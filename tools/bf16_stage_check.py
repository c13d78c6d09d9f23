"""GPU-box debugging aid: bf16 engine vs the engine-numerics restatement (oracle/cpu_ref_bf16.py), kernel by kernel
(ptts_debug_stop_after) on prompt position 0 of a fixture.  Prints the max |diff| of the residual stream after every
layer and of q / attention fragments inside the first layers.   python tools/bf16_stage_check.py [case] [layers_detail]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import cpu_ref_bf16
from parler_tts_amd import _hip
from parler_tts_amd.delay_pattern import build_delay_pattern_mask
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration, _hipMemcpyDtoD
from tests.helpers import case_inputs, load_case, state_dict_for

name = sys.argv[1] if len(sys.argv) > 1 else "mini_greedy"
detail = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f, meta = load_case(name)
cfg, sd = state_dict_for(name, meta)
model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16")
desc, amask, prompt, pmask = case_inputs(f)
B, T = desc.shape
P, L, K = prompt.shape[1], int(f["max_length"]), cfg.decoder.num_codebooks
D, NL = cfg.decoder.hidden_size, cfg.decoder.num_hidden_layers
eng = model.engine(B, T, P, L)
lib = eng.lib


def buf(nm, numel, dt=torch.float32):
    p, n = C.c_void_p(), C.c_int64()
    _hip.check(lib.ptts_debug_buffer(eng.handle, nm.encode(), C.byref(p), C.byref(n)))
    out = torch.empty(numel, dtype=dt, device="cuda")
    torch.cuda.synchronize()
    assert _hipMemcpyDtoD(out.data_ptr(), p.value, numel * out.element_size()) == 0
    return out.cpu()


def defrag(flat, rows, Kf):
    r = torch.arange(rows)[:, None]
    k = torch.arange(Kf)[None, :]
    off = (((k >> 5) * 2 + (r >> 4)) << 9) + ((((r & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7))
    return flat[off].float()


def run(stop):
    dev = "cuda:0"
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.to(dev).contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    _hip.check(lib.ptts_debug_stop_after(eng.handle, stop))
    eng.prefill(B, B, torch.from_numpy(f["enc_t5"]).to(dev).contiguous(), amask.to(dev, torch.int32).contiguous(),
                prompt.to(dev).contiguous(), None if pmask is None else pmask.to(dev, torch.int32).contiguous(), pattern, raw, 1)
    torch.cuda.synchronize()


m = cpu_ref_bf16.EngineBf16Decoder(cfg, sd)
m.start(torch.from_numpy(f["enc_t5"]), amask, P + L)
if pmask is not None:
    m.mask_positions(0, pmask)
m.trace = []
x0 = m.embed_prompt(prompt, None, 0)
m.step_hidden(x0)


def cmp(tag, got, want):
    d = (got - want).abs()
    print(f"{tag:26s} max|diff| {float(d.max()):.3e}  mean {float(d.mean()):.3e}   ref absmax {float(want.abs().max()):.3e}", flush=True)


look = os.environ.get("PTTS_LOOKAHEAD", "1") != "0" and T <= 64
per = 7 if look else 8
run(1)        # the debug gate counts the prompt position itself as launch 0
cmp("x after embed", buf("x", 32 * D).reshape(32, D)[:B], x0)
for l in range(NL):
    t = m.trace[l]
    base = 1 + per * l
    if l < detail:
        run(base + 1); cmp(f"L{l} q", buf("q", 32 * D).reshape(32, D)[:B], t["q"])
        run(base + 2); cmp(f"L{l} attn frag", defrag(buf("attn_frag", 32 * D, torch.bfloat16), 32, D)[:B], t["a"])
        run(base + 3); cmp(f"L{l} x1", buf("x", 32 * D).reshape(32, D)[:B], t["x1"])
        if not look:
            run(base + 4); cmp(f"L{l} cross q", buf("q", 32 * D).reshape(32, D)[:B], t["cq"])
        run(base + per - 3); cmp(f"L{l} cross frag", defrag(buf("attn_frag", 32 * D, torch.bfloat16), 32, D)[:B], t["c"])
        run(base + per - 2); cmp(f"L{l} x2", buf("x", 32 * D).reshape(32, D)[:B], t["x2"])
    run(base + per); cmp(f"L{l} x out", buf("x", 32 * D).reshape(32, D)[:B], t["x3"])
