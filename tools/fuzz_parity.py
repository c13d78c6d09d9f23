#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): fp32 engine against oracle/cpu_ref.py on random tiny architectures and call shapes the
fixtures do not cover - codebooks 1 .. 32, hidden 64 .. 256, 1-3 layers, batch 1 .. 40, descriptions 1 .. 90 tokens, prompts
0 .. 90 tokens (page crossings at 64), max_length from K + 2 up, ragged masks.  Teacher-forced along the oracle's greedy path:
logits within 2e-4 and the argmax equal wherever the oracle's top-2 margin exceeds 1e-3.
    python3 tools/fuzz_parity.py [n_cases] [seed] [generate | bf16 | fused]
With the third argument `fused` (round 5): decoders WIDE enough for the fused QKV + self-attention launch (hidden 768 / 1024 / 1536,
1-2 layers, 1-17 codebooks, batch 1 .. 70 = up to three m-tiles, prompts across KV pages, ragged masks, guidance, live EOS heads):
the public `generate()` free-running under graph replay, with the fused launch and with the two launches of rounds 1-4
(PTTS_NO_FUSED_QKV_ATTN=1) - raw ids and the last step's logits must be BIT-IDENTICAL (no oracle involved: the two forms are the
same arithmetic, only the hand-off differs).
With the third argument `generate`: the public `generate()` call, free-running greedy, LIVE EOS heads, optionally classifier-free
guidance and ragged masks, against `cpu_ref.generate_ids`: the raw ids must agree up to (per utterance) the first step the oracle
itself decides by less than 1e-3; with no such step the codes and the number of columns must be equal.
Test infrastructure (imports oracle/ through tests/test_gpu_parity.py's helper); prints one line per case, exits 1 on a failure."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from parler_tts_amd.configuration import tiny_config  # noqa: E402
from parler_tts_amd.weights import synthetic_state_dict  # noqa: E402
from tests import test_gpu_parity as tg  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.set_num_threads(min(16, os.cpu_count() or 8))
fails = 0
MODE = sys.argv[3] if len(sys.argv) > 3 else "forced"


def generate_case(case):
    from oracle import cpu_ref
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    heads = int(rng.integers(1, 4))
    K = int(rng.choice([1, 2, 4, 5, 9, 12, 17]))
    layers = int(rng.integers(1, 3))
    cb = int(rng.choice([32, 64]))
    B = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
    T = int(rng.choice([1, 3, 9, 33, 65]))
    P = int(rng.choice([1, 2, 5, 16, 63, 65]))
    L = int(K + 2 + rng.integers(0, 30))
    gs = float(rng.choice([0.0, 0.0, 2.5]))
    ragged = bool(rng.integers(0, 2)) and T >= 4 and P >= 2
    tag = f"case {case}: heads {heads} K {K} layers {layers} V {cb} B {B} T {T} P {P} L {L} guidance {gs} ragged {ragged}"
    cfg = tiny_config(num_codebooks=K, hidden=64 * heads, layers=layers, heads=heads, ffn=128, codebook_size=cb, max_positions=256)
    sd = synthetic_state_dict(cfg, seed=int(rng.integers(1, 1000)), never_eos=False, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    g = torch.Generator().manual_seed(int(rng.integers(1, 1000)))
    desc = torch.randint(2, cfg.text_encoder.vocab_size, (B, T), generator=g)
    prompt = torch.randint(2, cfg.vocab_size, (B, P), generator=g)
    amask = torch.ones(B, T, dtype=torch.long)
    pmask = None
    if ragged:
        pmask = torch.ones(B, P, dtype=torch.long)
        for b in range(1, B):
            amask[b, T - (7 * b) % (T // 2):] = 0
            pmask[b, : (5 * b) % (P // 2 + 1)] = 0
    kw = dict(guidance_scale=gs) if gs > 1 else {}
    okw, extra = {}, ""
    # call shapes beside the plain one: a decoder prefix (M:2150-2195), with a decoder_attention_mask over it when the call carries a
    # prompt mask (M:845-846), or the prompt handed in pre-embedded (prompt_hidden_states, M:2476)
    variant = int(rng.integers(0, 4))
    if variant == 1 and L > K + 8:
        s_pre = int(rng.integers(1, 6))
        prefix = torch.randint(0, cfg.decoder.pad_token_id, (B * K, s_pre), generator=g)
        okw["decoder_input_ids"] = prefix
        extra = f" prefix {s_pre}"
        if ragged and s_pre >= 2:
            dmask = torch.ones(B, s_pre, dtype=torch.long)
            dmask[B - 1, : s_pre // 2] = 0
            prefix.reshape(B, K, s_pre)[B - 1, :, : s_pre // 2] = cfg.decoder.pad_token_id
            okw["decoder_attention_mask"] = dmask
            extra += " masked"
    elif variant == 2:
        okw["prompt_hidden_states"] = torch.nn.functional.embedding(prompt, sd["embed_prompts.weight"])
        extra = " pre-embedded prompt"
    tag += extra
    enc = cpu_ref.t5_encode(cfg, sd, desc, amask)
    ref = cpu_ref.generate_ids(cfg, sd, enc, amask, None if "prompt_hidden_states" in okw else prompt, pmask, L, **kw, **okw)
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    gk = dict(input_ids=desc, attention_mask=amask, do_sample=False, max_length=L, return_raw_ids=True, **kw, **okw)
    if "prompt_hidden_states" not in okw:
        gk["prompt_input_ids"] = prompt
    if pmask is not None:
        gk["prompt_attention_mask"] = pmask
    model.generate(**gk)
    got = model.last_raw_ids.cpu()
    want, margins = ref["raw_ids"], ref["margins"]            # [rows, L_out], [steps, rows]
    n = min(got.shape[1], want.shape[1])
    benign, bad = 0, None
    for b in range(B):
        rows = slice(b * K, (b + 1) * K)
        diff = (got[rows, :n] != want[rows, :n]).any(0).nonzero()
        if len(diff) == 0:
            continue
        c = int(diff[0])
        # the decision that filled column c: step c - c0 (c0 = columns present before the first step: 1, or 1 + the prefix);
        # benign if the oracle's own gap on a differing row is tiny
        c0 = want.shape[1] - margins.shape[0]
        if c < c0:
            bad = (b, c, float("inf"))
            continue
        m = float(margins[c - c0][rows][(got[rows, c] != want[rows, c])].min())
        if m < 1e-3:
            benign += 1
        else:
            bad = (b, c, m)
    # Verdict.  No utterance may differ on a clear margin.  Without a benign divergence the two id matrices must be
    # EQUAL, shape included.  With one, the other utterances were already compared over the common columns above (no entry in
    # `diverged`), and a length difference must be explainable by the diverged utterance alone: a run ends when EVERY row has
    # finished, so an utterance with identical ids in both runs had finished by the shorter run's last column and the longer
    # run may only hold pad for it beyond that column.
    diverged = set()
    for b in range(B):
        if bool((got[b * K:(b + 1) * K, :n] != want[b * K:(b + 1) * K, :n]).any()):
            diverged.add(b)
    tail_ok = True
    longer = got if got.shape[1] > want.shape[1] else want
    if longer.shape[1] > n:
        for b in range(B):
            if b not in diverged and bool((longer[b * K:(b + 1) * K, n:] != cfg.decoder.pad_token_id).any()):
                tail_ok = False
                bad = bad or (b, n, float("nan"))
    ok = bad is None and tail_ok and (benign > 0 or (got.shape == want.shape and bool(torch.equal(got, want))))
    print(f"{'ok  ' if ok else 'FAIL'} {tag}: columns {tuple(got.shape)} vs {tuple(want.shape)}, utterances that left on a sub-1e-3 margin {benign}"
          + ("" if bad is None else f", utterance {bad[0]} differs at column {bad[1]} where the oracle's margin is {bad[2]:.2e}"), flush=True)
    del model
    torch.cuda.empty_cache()
    return ok


def fused_case(case):
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    D = int(rng.choice([768, 1024, 1024, 1536]))
    K = int(rng.choice([1, 4, 9, 9, 17]))
    layers = int(rng.integers(1, 3))
    B = int(rng.choice([1, 2, 7, 16, 19, 32, 33, 40, 70]))
    T = int(rng.choice([1, 5, 12, 40, 63]))
    P = int(rng.choice([0, 1, 16, 63, 65, 130]))
    L = int(K + 4 + rng.integers(4, 60))
    gs = float(rng.choice([0.0, 0.0, 3.0])) if B <= 35 else 0.0
    dtype = "bf16" if rng.integers(0, 4) else "fp32"
    ragged = bool(rng.integers(0, 2)) and T >= 4 and P >= 2
    never_eos = bool(rng.integers(0, 2))
    tag = f"case {case} [{dtype}]: D {D} K {K} layers {layers} B {B} T {T} P {P} L {L} guidance {gs} ragged {ragged} never_eos {never_eos}"
    cfg = mini_v01_config()
    dec = cfg.decoder
    dec.hidden_size, dec.num_attention_heads, dec.ffn_dim, dec.num_hidden_layers, dec.num_codebooks = D, D // 64, 2 * D, layers, K
    cfg.audio_encoder.num_codebooks = K
    sd = synthetic_state_dict(cfg, seed=int(rng.integers(1, 1000)), never_eos=never_eos, ln_jitter=0.1, parts=("decoder",))
    g = torch.Generator().manual_seed(int(rng.integers(1, 1000)))
    enc = torch.randn(B, T, cfg.text_encoder.d_model, generator=g) * 0.3
    prompt = torch.randint(2, 32000, (B, P), generator=g) if P else None
    amask = torch.ones(B, T, dtype=torch.long)
    pmask = None
    if ragged:
        pmask = torch.ones(B, P, dtype=torch.long)
        for b in range(1, B):
            amask[b, T - (7 * b) % (T // 2):] = 0
            pmask[b, : (5 * b) % (P // 2 + 1)] = 0
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, do_sample=False, max_length=L, return_raw_ids=True)
    if prompt is not None:
        kw["prompt_input_ids"] = prompt
    if pmask is not None:
        kw["prompt_attention_mask"] = pmask
    if gs > 1:
        kw["guidance_scale"] = gs
    outs = []
    env = {"PTTS_LN_FOLD": "1"} if dtype == "fp32" else {}
    for fused in (True, False):
        os.environ.update(env)
        if not fused:
            os.environ["PTTS_NO_FUSED_QKV_ATTN"] = "1"
        try:
            model = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
            model.generate(**kw)
            eng = model.engine()
            rows = (2 * B if gs > 1 else B) * K
            prof = eng.profile_read(reset=True)
            outs.append((model.last_raw_ids.cpu().clone(), eng.last_logits(rows).cpu().clone()))
            del model
            torch.cuda.empty_cache()
        finally:
            os.environ.pop("PTTS_NO_FUSED_QKV_ATTN", None)
            for k in env:
                os.environ.pop(k, None)
    ok = outs[0][0].shape == outs[1][0].shape and bool(torch.equal(outs[0][0], outs[1][0])) and bool(torch.equal(outs[0][1], outs[1][1]))
    print(f"{'ok  ' if ok else 'FAIL'} {tag}: columns {tuple(outs[0][0].shape)} vs {tuple(outs[1][0].shape)}, ids and last logits bit-identical {ok}", flush=True)
    return ok


for case in range(n_cases if MODE == "fused" else 0):
    try:
        fails += 0 if fused_case(case) else 1
    except Exception as e:
        fails += 1
        print(f"EXC  case {case}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()
if MODE == "fused":
    print(f"{n_cases - fails} of {n_cases} cases passed")
    sys.exit(1 if fails else 0)
for case in range(n_cases if MODE == "generate" else 0):
    try:
        fails += 0 if generate_case(case) else 1
    except Exception as e:
        fails += 1
        print(f"EXC  case {case}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()
for case in range(0 if MODE == "generate" else n_cases):
    heads = int(rng.integers(1, 5))
    K = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 16, 17, 24, 32]))
    layers = int(rng.integers(1, 4))
    ffn = int(rng.choice([64, 128, 256, 320]))
    cb = int(rng.choice([32, 64, 128]))
    B = int(rng.choice([1, 2, 3, 5, 8, 16, 17, 31, 32, 33, 40]))
    T = int(rng.choice([1, 2, 7, 16, 33, 63, 64, 65, 90]))
    P = int(rng.choice([1, 2, 5, 16, 31, 62, 63, 64, 65, 90]))
    L = int(K + 2 + rng.integers(0, 40))
    steps = min(L - 1, 24)
    ragged = bool(rng.integers(0, 2)) and T >= 4 and P >= 2
    dtype = "bf16" if MODE == "bf16" else "fp32"
    tag = f"case {case} [{dtype}]: heads {heads} K {K} layers {layers} ffn {ffn} V {cb} B {B} T {T} P {P} L {L} steps {steps} ragged {ragged}"
    try:
        cfg = tiny_config(num_codebooks=K, hidden=64 * heads, layers=layers, heads=heads, ffn=ffn, codebook_size=cb, max_positions=256)
        sd = synthetic_state_dict(cfg, seed=int(rng.integers(1, 1000)), never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
        got, want, ref_ids, got_raw = tg._oracle_vs_engine(cfg, sd, dtype, B=B, T=T, P=P, L=L, seed=int(rng.integers(1, 1000)), steps=steps,
                                                           ragged=ragged, oracle="bf16" if dtype == "bf16" else "fp32")
        err = float((got - want).abs().max())
        # fp32: 2e-4 / clear margins 1e-3.  bf16 (third argument `bf16`): the engine-numerics restatement oracle/cpu_ref_bf16.py within
        # max(4e-3, 2 x its own re-association spread), argmax equal where its margin exceeds twice that (tests/test_gpu_parity.py)
        tol = tg._bf16_bound(cfg, tg._oracle_vs_engine.spread) if dtype == "bf16" else 2e-4
        top2 = torch.topk(want, 2, dim=-1)[0]
        safe = (top2[..., 0] - top2[..., 1]) > (2 * tol if dtype == "bf16" else 1e-3)
        same = bool(torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe]))
        ok = err < tol and same and bool(torch.isfinite(got).all())
        print(f"{'ok  ' if ok else 'FAIL'} {tag}: max |dlogit| {err:.2e}, clear argmaxes equal {same}", flush=True)
        fails += 0 if ok else 1
    except Exception as e:      # a refused shape must be a clean error, never a fault
        fails += 1
        print(f"EXC  {tag}: {type(e).__name__}: {e}", flush=True)
        traceback.print_exc()
print(f"{n_cases - fails} of {n_cases} cases passed")
sys.exit(1 if fails else 0)
