#!/usr/bin/env python3
"""A/B of the look-ahead cross-q path (PTTS_LOOKAHEAD=1) against the default path: same mini_v0.1 model, batch 32,
same teacher-forced tokens; per-step logits must agree to bf16 tolerance.   python3 tools/lookahead_check.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd import _hip
from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.delay_pattern import build_delay_pattern_mask
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
from parler_tts_amd.weights import synthetic_state_dict

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
T, P, L = 40, 16, steps + 2
K = cfg.decoder.num_codebooks
g = torch.Generator().manual_seed(7)
desc = torch.randint(2, 32100, (B, T), generator=g)
prompt = torch.randint(2, 32100, (B, P), generator=g)
forced = torch.randint(0, 1024, (steps, B * K), generator=g)


def run(flag):
    os.environ["PTTS_LOOKAHEAD"] = flag
    m = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
    eng = m.engine(B, T, P, L)
    enc = eng.t5_encode(desc.cuda(), torch.ones(B, T, dtype=torch.long, device="cuda"))
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.cuda().contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device="cuda")
    raw[:, :1] = ids.cuda()
    eng.prefill(B, B, enc.contiguous(), torch.ones(B, T, dtype=torch.int32, device="cuda"), prompt.cuda().contiguous(), None,
                pattern, raw, 1)
    sp = _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
    out = []
    for i in range(steps):
        eng.set_forced_tokens(forced[i])
        eng.decode_steps(1, sp)
        out.append(eng.last_logits(B * K).cpu())
    eng.set_forced_tokens(None)
    del m
    torch.cuda.empty_cache()
    return torch.stack(out)


a = run("0")
b = run("1")
d = (a - b).abs()
print(f"logits: max |diff| {d.max().item():.4f}, mean |diff| {d.mean().item():.5f}, logit std {a.std().item():.3f}, "
      f"argmax agreement {(a.argmax(-1) == b.argmax(-1)).float().mean().item():.4f}, nan {bool(torch.isnan(b).any())}")
print("per-step max diff (first 12):", [round(x, 4) for x in d.amax(dim=(1, 2))[:12].tolist()])
