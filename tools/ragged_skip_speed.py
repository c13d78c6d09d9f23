#!/usr/bin/env python3
"""Decode-loop time of the bench workload when half of the batch finishes early (all K codebooks of utterances 16..31 are forced
to EOS at column 64): finished utterances read one key instead of their cache (PTTS_NO_SKIP_FINISHED=1 switches that off).
   python3 tools/ragged_skip_speed.py [finish_column]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from parler_tts_amd import _hip
from parler_tts_amd.configuration import mini_v01_config
from parler_tts_amd.delay_pattern import build_delay_pattern_mask
from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
from parler_tts_amd.weights import synthetic_state_dict

col = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = mini_v01_config()
sd = synthetic_state_dict(cfg, seed=1234, never_eos=True, parts=("text_encoder", "decoder"))
model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
B, T, P, L, K = 32, 40, 16, 2580, cfg.decoder.num_codebooks
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
enc = torch.randn(B, T, cfg.text_encoder.d_model, generator=g).to(dev)
amask = torch.ones(B, T, dtype=torch.int32, device=dev)
prompt = torch.randint(2, cfg.vocab_size, (B, P), generator=g).to(dev)
eng = model.engine(B, T, P, L)
ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
pattern = pattern.to(dev).contiguous()
for rep in range(2):
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    forced = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
    forced[16 * K:, col] = cfg.decoder.eos_token_id
    own = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
    eng.set_forced_sequence(forced, own)
    eng.prefill(B, B, enc, amask, prompt, None, pattern, raw, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.decode_steps(L - 1, _hip.PttsSampling(0, 1.0, 0, 1.0, 0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.state()
    eng.set_forced_sequence(None)
print(f"skip={'off' if os.environ.get('PTTS_NO_SKIP_FINISHED') else 'on'}: decode loop {dt * 1e3:.0f} ms for {L - 1} steps "
      f"({dt / (L - 1) * 1e3:.3f} ms/step), unfinished rows at the end {st.n_unfinished} of {B * K}")
