"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI of
parler_tts_amd/csrc/libparler_tts_hip.so; the oracle (oracle/cpu_ref.py, pinned to the real reference by
tests/test_oracle_vs_golden.py) and the committed golden fixtures are only the checkers.

Tolerances
  fp32 mode : token / code ids bit-exact on the fixtures (their top-2 logit margins are >= 6e-4, three
              orders above the fp32 re-association noise measured here); logits |diff| <= 2e-4
  bf16 mode : weights + KV cache in bf16, fp32 accumulate.  Checked against oracle/cpu_ref_bf16.py, the CPU restatement
              with bf16 roundings at the engine's rounding points.
              Two bf16 pipelines with the same rounding points but another fp32 summation order drift apart (every
              rounding turns an fp32-level difference into whole-ulp flips; cpu_ref_bf16.py header), the more the deeper
              and wider the model.  The restatement is therefore evaluated twice, with fp32 and with fp64 accumulation, and
              the engine must lie within max(BF16_TOL, 2x that re-association SPREAD) of it, the argmax equal wherever the
              restatement's margin exceeds twice the bound.  Measured spreads: 4e-4 (tiny, 2 layers), 9e-3 (mini_v0.1
              widths cut to 2 layers), 1.3e-2 (mini_v0.1, 24 layers - where bf16 vs the fp32 reference is 1.7e-2 .. 8e-2).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.helpers import case_inputs, config_for, forced_tokens, free_positions, load_case, state_dict_for

pytestmark = pytest.mark.gpu

BF16_TOL = 4e-3      # one bf16 ulp of a value in [0.5, 1) is 2^-8 = 3.9e-3


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on an MI355X box")


@pytest.fixture(scope="module")
def hip():
    _need_gpu()
    from parler_tts_amd import _hip

    return _hip.load_library()


_models = {}


def model_for(name, meta, dtype):
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    cfg, sd = state_dict_for(name, meta, parts=("text_encoder", "decoder", "audio_encoder")
                             if name.startswith("tiny") else ("text_encoder", "decoder"))
    key = (name.split("_")[0], str(meta["sd_kw"]), str(meta.get("cfg_kw")), dtype)
    if key not in _models:
        _models.clear()
        torch.cuda.empty_cache()
        _models[key] = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
    return cfg, _models[key]


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K,pro,act", [(32, 1024, 1024, 2, 0), (32, 4096, 1024, 2, 1), (32, 1024, 4096, 1, 0),
                                           (7, 256, 128, 3, 2), (70, 768, 768, 1, 0), (64, 2048, 2048, 1, 3)])
def test_gemm_against_torch(hip, dtype, M, N, K, pro, act):
    g = torch.Generator().manual_seed(M * 7 + N + K)
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = torch.randn(M, K, generator=g).cuda()
    gamma = (1 + 0.1 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = (0.1 * torch.randn(N, generator=g)).cuda()
    Y = torch.empty(M, N, device="cuda")
    dt = 1 if dtype == "bf16" else 0
    rc = hip.ptts_test_gemm(dt, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(), beta.data_ptr(),
                            C.c_float(1e-5), act, bias.data_ptr(), Y.data_ptr(), None)
    assert rc == 0, hip.ptts_last_error()
    Xd, Wd = X.double(), W.double()
    if pro == 2:
        Xd = torch.nn.functional.layer_norm(Xd, (K,), gamma.double(), beta.double(), 1e-5)
    elif pro == 3:
        Xd = gamma.double() * (Xd * torch.rsqrt(Xd.pow(2).mean(-1, keepdim=True) + 1e-5))
    if dtype == "bf16":   # the kernel rounds both MFMA operands to bf16, accumulates in fp32
        Xd = Xd.float().bfloat16().double()
        Wd = W.bfloat16().double()
    ref = Xd @ Wd.t() + bias.double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    elif act == 3:
        ref = torch.relu(ref)
    err = (Y.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= (2e-5 if dtype == "fp32" else 2e-3) * max(scale, 1.0), (err, scale)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K,act,mean", [(32, 1024, 1024, 0, 0.0), (32, 3072, 1024, 0, 0.7), (17, 4096, 1024, 1, -1.5),
                                            (40, 1024, 1024, 0, 0.3), (32, 9792, 1024, 0, 2.0), (32, 256, 2048, 0, 0.5)])
def test_folded_layernorm_gemm_against_torch(hip, dtype, M, N, K, act, mean):
    """LayerNorm applied AFTER the MFMA (gamma folded into the weights, statistics from ones.x / x.x^T MFMAs):
    y = rstd * (x.W' - mean * c1) + c2 must equal LayerNorm(x) @ W^T.  Reference M:446/467/487/959 + M:256-259.
    Rows carry a non-zero mean so the E[x^2] - mean^2 form and the mean * c1 cancellation are exercised."""
    g = torch.Generator().manual_seed(M * 13 + N + K)
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = (mean + torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 2)).cuda()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = torch.zeros(N).cuda()
    Y = torch.empty(M, N, device="cuda")
    dt = 1 if dtype == "bf16" else 0
    rc = hip.ptts_test_gemm(dt, M, N, K, W.data_ptr(), X.data_ptr(), 4, gamma.data_ptr(), beta.data_ptr(),
                            C.c_float(1e-5), act, bias.data_ptr(), Y.data_ptr(), None)
    assert rc == 0, hip.ptts_last_error()
    Xd, Wd = X.double(), W.double()
    if dtype == "bf16":   # the MFMA (and therefore the statistics) see the bf16-rounded rows and weights
        Xd = X.bfloat16().double()
        Wd = W.bfloat16().double()
    ref = torch.nn.functional.layer_norm(Xd, (K,), gamma.double(), beta.double(), 1e-5) @ Wd.t()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    err = (Y.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= (5e-5 if dtype == "fp32" else 3e-3) * max(scale, 1.0), (err, scale)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("pro", [4, 5])
@pytest.mark.parametrize("M,N,K", [(2, 384, 128), (4, 128, 128), (2, 128, 512), (4, 1152, 128), (2, 192, 64), (16, 1024, 1024),
                                   (32, 1024, 1024), (32, 1024, 4096), (33, 256, 2048)])
def test_fragment_gemm_partial_bursts(hip, dtype, pro, M, N, K):
    """Fragment-input GEMM (5 = plain rows, 4 = folded LayerNorm) on shapes where a wave owns fewer k-steps than its
    8-slot burst, or none at all: an MFMA of an empty slot must not run (MFMA ignores the EXEC mask; the guards have
    to be scalar branches - tools/check_mfma_guards.py is the ISA-level check of the same property)."""
    g = torch.Generator().manual_seed(M * 31 + N + K + pro)
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = (0.2 + torch.randn(M, K, generator=g)).cuda()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = torch.zeros(N).cuda()
    Y = torch.full((M, N), 7.0, device="cuda")
    rc = hip.ptts_test_gemm(1 if dtype == "bf16" else 0, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(),
                            beta.data_ptr(), C.c_float(1e-5), 0, bias.data_ptr(), Y.data_ptr(), None)
    assert rc == 0, hip.ptts_last_error()
    Xd, Wd = X.double(), W.double()
    if dtype == "bf16":
        Xd, Wd = X.bfloat16().double(), W.bfloat16().double()
    if pro == 4:
        Xd = torch.nn.functional.layer_norm(Xd, (K,), gamma.double(), beta.double(), 1e-5)
    ref = Xd @ Wd.t()
    assert not torch.isnan(Y).any()
    err = (Y.double() - ref).abs().max().item()
    tol = (5e-5 if dtype == "fp32" else (3e-3 if pro == 4 else 2e-5)) * max(ref.abs().max().item(), 1.0)
    assert err <= tol, (err, tol)


@pytest.mark.parametrize("M,N,K,pro,act", [(128, 1024, 1024, 7, 0), (1280, 2304, 768, 7, 0), (300, 9792, 1024, 7, 1), (512, 768, 2048, 7, 0),
                                           (130, 64, 96, 7, 2), (1280, 128, 32, 7, 0), (256, 1072, 256, 7, 0), (7, 256, 128, 7, 0),
                                           (40, 3072, 1024, 8, 0), (512, 3072, 1024, 8, 0),
                                           (200, 4096, 1024, 8, 1), (129, 1024, 1536, 8, 0), (1280, 3072, 1024, 8, 0)])
def test_tile_gemm_against_torch(hip, M, N, K, pro, act):
    """The MFMA tile kernel of the one-off phases (csrc/gemm_tile.hip; T5 encoder M:2229, batched prefill M:794-978, cross
    K / V M:295-298) against torch in float64 on the bf16-rounded operands: plain rows (7) and the folded LayerNorm with its
    matrix-core row statistics (8); ragged last row tile, fewer rows than one tile (one description alone), feature-block counts
    that are no multiple of the tile (67 = 16.75 x 4), odd k-step counts and products shorter than the ring (K = 96, K = 32), rows
    with a non-zero mean."""
    g = torch.Generator().manual_seed(M * 11 + N + K)
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = ((0.4 if pro == 8 else 0.0) + torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g))).cuda()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = (torch.zeros(N) if pro == 8 else 0.1 * torch.randn(N, generator=g)).cuda()
    Y = torch.full((M, N), 7.0, device="cuda")
    rc = hip.ptts_test_gemm(1, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(), beta.data_ptr(), C.c_float(1e-5), act,
                            bias.data_ptr(), Y.data_ptr(), None)
    assert rc == 0, hip.ptts_last_error()
    Xd, Wd = X.bfloat16().double(), W.bfloat16().double()
    if pro == 8:
        ref = torch.nn.functional.layer_norm(Xd, (K,), gamma.double(), beta.double(), 1e-5) @ Wd.t()
    else:
        ref = Xd @ Wd.t() + bias.double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    elif act == 2:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    assert not torch.isnan(Y).any()
    err = (Y.double() - ref).abs().max().item()
    tol = (3e-3 if pro == 8 else 2e-5) * max(ref.abs().max().item(), 1.0)       # 8: gamma is folded into bf16 weights (one more rounding)
    assert err <= tol, (err, tol)
    # the same product on the 32-row decode kernel (prologue 5 / 4): same operands, another fp32 summation order
    Y2 = torch.empty_like(Y)
    rc = hip.ptts_test_gemm(1, M, N, K, W.data_ptr(), X.data_ptr(), 5 if pro == 7 else 4, gamma.data_ptr(), beta.data_ptr(),
                            C.c_float(1e-5), act, bias.data_ptr(), Y2.data_ptr(), None)
    assert rc == 0, hip.ptts_last_error()
    assert (Y - Y2).abs().max().item() <= 2e-5 * max(ref.abs().max().item(), 1.0)


# ------------------------------------------------------------------------------------------------
# decode loop vs golden fixtures of the reference
# ------------------------------------------------------------------------------------------------
def run_engine_case(name, dtype, teacher_force=False, shuffle_pages=False):
    """Drive prefill + per-step decode through the C ABI; returns raw ids, per-step logits."""
    from parler_tts_amd import _hip
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask

    f, meta = load_case(name)
    cfg, model = model_for(name, meta, dtype)
    desc, amask, prompt, pmask = case_inputs(f)
    K = cfg.decoder.num_codebooks
    B = desc.shape[0]
    L = int(f["max_length"])
    gs = meta["gen"].get("guidance_scale")
    use_cfg = gs is not None and gs > 1
    B_eff = 2 * B if use_cfg else B
    eng = model.engine(B_eff, desc.shape[1], prompt.shape[1], L)
    if shuffle_pages:
        n_pages, max_blocks, _ = eng.pool_info()
        perm = torch.randperm(n_pages, generator=torch.Generator().manual_seed(3)).to(torch.int32)
        eng.set_block_table(perm.reshape(-1, max_blocks))
    dev = "cuda:0"
    enc = torch.from_numpy(f["enc_t5"]).to(dev)
    am = amask.to(dev, torch.int32)
    pids = prompt.to(dev)
    pm = None if pmask is None else pmask.to(dev, torch.int32)
    if use_cfg:
        enc = torch.cat([enc, torch.zeros_like(enc)])
        am = torch.cat([am, torch.zeros_like(am)])
        pids = pids.repeat(2, 1)
        pm = None if pm is None else pm.repeat(2, 1)
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.to(dev).contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    eng.prefill(B, B_eff, enc.contiguous(), am.contiguous(), pids.contiguous(), None if pm is None else pm.contiguous(),
                pattern, raw, 1)
    sp = _hip.PttsSampling(0, 1.0, 0, float(gs) if use_cfg else 1.0, 0)
    ft = forced_tokens(f) or {}
    logits = []
    gold = torch.from_numpy(f["raw_ids"])
    for step in range(L - 1):
        if teacher_force:
            eng.set_forced_tokens(gold[:, step + 1])
        elif step in ft:
            eng.set_forced_tokens(ft[step])
        else:
            eng.set_forced_tokens(None)
        eng.decode_steps(1, sp)
        lg = eng.last_logits(B_eff * K)
        if use_cfg:
            n = B * K
            lg = lg[n:] + (lg[:n] - lg[n:]) * gs
        logits.append(lg.cpu())
        st = eng.state()
        if st.done_len:
            break
    eng.set_forced_tokens(None)
    st = eng.state()
    out = raw[:, : (st.done_len or st.cur_len)].cpu()
    if shuffle_pages:
        n_pages, max_blocks, _ = eng.pool_info()
        eng.set_block_table(torch.arange(n_pages, dtype=torch.int32).reshape(-1, max_blocks))
    return f, cfg, out, torch.stack(logits), pattern.cpu()


GREEDY_TINY = ["tiny_greedy", "tiny_greedy_ragged", "tiny_eos_free", "tiny_eos_forced", "tiny_cfg"]


@pytest.mark.parametrize("name", GREEDY_TINY + ["mini_greedy", "mini_greedy_ragged"])
def test_fp32_tokens_bit_exact_and_logits_close(hip, name):
    from oracle import cpu_ref

    f, cfg, raw, logits, pattern = run_engine_case(name, "fp32")
    gold_logits = f["step_logits"]
    n = min(gold_logits.shape[0], logits.shape[0])
    assert np.abs(logits[:n].numpy() - gold_logits[:n]).max() < 2e-4
    assert raw.shape == f["raw_ids"].shape
    free = free_positions(f["pattern"], raw.shape[1] - 1).T
    assert np.array_equal(raw.numpy()[:, 1:][free], f["raw_ids"][:, 1:][free])
    codes = cpu_ref.undelay(cfg, raw, pattern)[0]
    assert np.array_equal(codes.numpy(), f["codes"])


def _bf16_bound(cfg, spread):
    """tolerance of a bf16 engine vs restatement comparison: one bf16 ulp of a logit in [0.5, 1), or 2x the restatement's own
    re-association spread (its fp32-accumulate vs its fp64-accumulate evaluation) where that is larger"""
    return max(BF16_TOL, 2.0 * float(spread))


@pytest.mark.parametrize("name", ["tiny_greedy", "tiny_greedy_ragged", "tiny_cfg", "mini_greedy", "mini_greedy_ragged"])
def test_bf16_teacher_forced_logits_against_the_bf16_restatement(hip, name):
    """bf16 engine vs oracle/cpu_ref_bf16.py (same rounding points, CPU), every step of the fixture: logits within the bound
    of _bf16_bound, argmax equal where the restatement's margin exceeds twice that; equal lengths, ragged masks, guidance."""
    from oracle import cpu_ref_bf16

    f, cfg, raw, logits, pattern = run_engine_case(name, "bf16", teacher_force=True)
    _, meta = load_case(name)
    _, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    kw = dict(forced=torch.from_numpy(f["raw_ids"]), record=int(f["max_length"]), guidance_scale=meta["gen"].get("guidance_scale"))
    want = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]), **kw)["step_logits"]
    alt = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                                    acc64=True, **kw)["step_logits"]
    tol = _bf16_bound(cfg, (alt - want).abs().max())
    n = min(want.shape[0], logits.shape[0])
    diff = (logits[:n] - want[:n]).abs()
    assert diff.max() < tol, (diff.max(), tol)
    t2 = torch.topk(want[:n], 2, dim=-1)[0]
    safe = (t2[..., 0] - t2[..., 1]) > 2 * tol
    assert safe.float().mean() > 0.3
    assert torch.equal(logits[:n].argmax(-1)[safe], want[:n].argmax(-1)[safe])
    gold = f["step_logits"]
    m = min(n, gold.shape[0])
    assert np.abs(logits[:m].numpy() - gold[:m]).max() < 0.08        # bf16 noise vs the fp32 reference


def test_bf16_mini_widths_two_layers_against_the_restatement(hip):
    """mini_v0.1 widths (D 1024, 16 heads, ffn 4096, 9 codebooks of 1088, batch 32: the GEMM, attention and selection
    instances of the benchmark) cut to 2 layers, where two same-rounding bf16 pipelines have drifted least: every rounding
    point of the engine is where the restatement has it - logits within the bound of _bf16_bound (here ~1e-2 on logits of
    magnitude up to ~3, i.e. about one bf16 ulp), argmax equal outside twice that."""
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 2
    sd = synthetic_state_dict(cfg, seed=7, never_eos=True, ln_jitter=0.1, parts=("decoder",))
    for ragged in (False, True):
        got, want, _, _ = _oracle_vs_engine(cfg, sd, "bf16", B=32, T=40, P=16, L=14, seed=8, steps=12, oracle="bf16", ragged=ragged)
        tol = _bf16_bound(cfg, _oracle_vs_engine.spread)
        assert tol < 3e-2 and (got - want).abs().max() < tol, (ragged, (got - want).abs().max(), tol)
        t2 = torch.topk(want, 2, dim=-1)[0]
        safe = (t2[..., 0] - t2[..., 1]) > 2 * tol
        assert safe.float().mean() > 0.3
        assert torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])


def test_paged_kv_is_layout_independent(hip):
    f, cfg, raw0, lg0, _ = run_engine_case("tiny_greedy", "fp32")
    f, cfg, raw1, lg1, _ = run_engine_case("tiny_greedy", "fp32", shuffle_pages=True)
    assert torch.equal(raw0, raw1)
    assert torch.equal(lg0, lg1)


def test_graph_replay_equals_eager_steps(hip):
    """generate() replays the captured hipGraph; run_engine_case steps eagerly via the test hooks."""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    desc, amask, prompt, pmask = case_inputs(f)
    codes = model.generate(encoder_outputs=(torch.from_numpy(f["enc_t5"]),), attention_mask=amask,
                           prompt_input_ids=prompt, do_sample=False, max_length=int(f["max_length"]),
                           return_raw_ids=True)
    assert np.array_equal(codes[0].cpu().numpy(), f["codes"])
    free = free_positions(f["pattern"], int(f["max_length"]) - 1).T
    assert np.array_equal(model.last_raw_ids.cpu().numpy()[:, 1:][free], f["raw_ids"][:, 1:][free])


# ------------------------------------------------------------------------------------------------
# parity at depth: the reference's own long runs, teacher-forced on the device (ptts_set_forced_sequence)
# ------------------------------------------------------------------------------------------------
def run_forced_long(name, dtype, batch_pad=0):
    """Engine teacher-forced along the fixture's raw ids under graph replay; returns the logits at the fixture's sampled
    steps [n, rows, V], the engine's own choice of every step [steps, rows] and the fixture.  batch_pad > 0 appends that
    many random utterances (free-running) so that the fixture rows run inside a larger batch."""
    from parler_tts_amd import _hip
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask

    f, meta = load_case(name)
    cfg, model = model_for(name, meta, dtype)
    desc, amask, prompt, pmask = case_inputs(f)
    enc = torch.from_numpy(f["enc_t5"])
    B0 = desc.shape[0]
    if batch_pad:
        g = torch.Generator().manual_seed(77)
        enc = torch.cat([enc, torch.randn(batch_pad, *enc.shape[1:], generator=g) * enc.std()])
        amask = torch.cat([amask, torch.ones(batch_pad, amask.shape[1], dtype=amask.dtype)])
        prompt = torch.cat([prompt, torch.randint(2, cfg.vocab_size, (batch_pad, prompt.shape[1]), generator=g)])
        if pmask is not None:
            pmask = torch.cat([pmask, torch.ones(batch_pad, pmask.shape[1], dtype=pmask.dtype)])
    K, B, L = cfg.decoder.num_codebooks, enc.shape[0], int(f["max_length"])
    eng = model.engine(B, enc.shape[1], prompt.shape[1], L)
    dev = "cuda:0"
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.to(dev).contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    forced = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
    forced[: B0 * K] = torch.from_numpy(f["raw_ids"]).to(dev)
    own = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
    eng.set_forced_sequence(forced, own)
    try:
        eng.prefill(B, B, enc.to(dev).contiguous(), amask.to(dev, torch.int32).contiguous(), prompt.to(dev).contiguous(),
                    None if pmask is None else pmask.to(dev, torch.int32).contiguous(), pattern, raw, 1)
        sp = _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
        logits, done = [], 0
        for s_ in f["logit_steps"]:
            n = int(s_) + 1 - done
            eng.decode_steps(n, sp)
            done += n
            logits.append(eng.last_logits(B * K)[: B0 * K].cpu())
        eng.decode_steps(L - 1 - done, sp)
        torch.cuda.synchronize()
        assert eng.state().cur_len == L
    finally:
        eng.set_forced_sequence(None)
    assert torch.equal(raw[: B0 * K].cpu(), torch.from_numpy(f["raw_ids"]))          # the forcing took
    return f, cfg, torch.stack(logits), own[: B0 * K, 1:].cpu().numpy().T


@pytest.mark.parametrize("name", ["tinylong_greedy", "minilong_greedy", "minilong2580_greedy"])
def test_fp32_long_horizon_against_the_reference(hip, name, capsys):
    """L = 2580 (tiny shape: 41 KV pages, positions > 2048), L = 512 (mini_v0.1 shape) and L = 2580 at the mini_v0.1 shape
    (`minilong2580_greedy`: 24 layers x 16 heads over all 41 KV pages - BASELINE configs[1]'s exact decoder at its full
    horizon) runs of the REAL reference.  fp32 engine, teacher-forced: logits of the sampled steps (first, EVERY page
    boundary the run crosses, S > 1024, S > 2048, the last 8) within 2e-4, and the engine's own argmax equal to the
    reference's at EVERY step whose top-2 margin is >= 5e-4."""
    f, cfg, logits, own = run_forced_long(name, "fp32")
    d = np.abs(logits.numpy() - f["step_logits"])
    assert d.max() < 2e-4, (d.max(), d.reshape(d.shape[0], -1).max(1))
    L = int(f["max_length"])
    free = free_positions(f["pattern"], L - 1)
    gold = f["raw_ids"][:, 1:].T
    ok = free & (f["margins"] >= FP32_ARGMAX_MARGIN)
    assert ok.mean() > 0.5 * free.mean()
    assert np.array_equal(own[ok], gold[ok])
    last = f["logit_steps"] >= L - 9
    assert last.sum() >= 6 and d[last].max() < 2e-4
    # the kept argmaxes the 5e-4 rule skips: what did the engine choose THERE?  A flip is legitimate only
    # where the reference's own top-2 gap is below twice the largest logit difference measured in this very run (two logits
    # moving towards each other by d.max() each); everything else must agree, and the counts go on the record.
    sub = free & ~ok
    flips = sub & (own != gold)
    with capsys.disabled():
        print(f"\n[fp32 forced] {name}: {int(free.sum())} kept argmaxes, {int(ok.sum())} with a reference margin >= {FP32_ARGMAX_MARGIN:g} "
              f"(all equal), {int(sub.sum())} below it of which the engine's own choice differs at {int(flips.sum())}"
              + (f" (largest reference margin at a differing choice {float(f['margins'][flips].max()):.2e})" if flips.any() else "")
              + f"; max |dlogit| at the sampled steps {d.max():.2e}, smallest kept margin {float(f['margins'][free].min()):.2e}")
    strict = free & (f["margins"] >= 2 * float(d.max()))
    assert np.array_equal(own[strict], gold[strict]), float(f["margins"][strict & (own != gold)].max())


def _bf16_long_reference(name, f):
    """logits / own argmax / margins of the engine-numerics restatement along the fixture's raw ids"""
    from oracle import cpu_ref_bf16

    if name.startswith("mini"):                      # minutes of CPU: written by oracle/gen_golden.py --only-bf16
        g, _ = load_case(name + "_bf16")
        assert np.array_equal(g["logit_steps"], f["logit_steps"])
        spread = float(g["spread"]) if "spread" in g else float(np.abs(g["step_logits"] - g["step_logits_acc64"]).max())
        return g["step_logits"], g["own"], g["margins"], spread
    _, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    dec = cpu_ref_bf16.EngineBf16Decoder(cfg, sd)
    margins, orig = [], dec.logits

    def wrapped(x):
        lg = orig(x)
        t2 = torch.topk(lg, 2, dim=-1)[0]
        margins.append(t2[:, 0] - t2[:, 1])
        return lg

    dec.logits = wrapped
    kw = dict(forced=torch.from_numpy(f["raw_ids"]), record=set(int(x) for x in f["logit_steps"]))
    r = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]), decoder=dec, **kw)
    alt = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]), acc64=True, **kw)
    spread = float((alt["step_logits"] - r["step_logits"]).abs().max())
    return r["step_logits"].numpy(), r["own"].numpy(), torch.stack(margins).numpy(), spread


@pytest.mark.parametrize("name,pad", [("tinylong_greedy", 0), ("minilong_greedy", 0), ("minilong_greedy", 30),
                                      ("minilong2580_greedy", 0), ("minilong2580_greedy", 30)])
def test_bf16_long_horizon_against_the_bf16_restatement(hip, name, pad):
    """The benchmark precision (BASELINE configs[1]) held to the CPU restatement with the engine's rounding points, over the
    same long runs: logits within the bound of _bf16_bound (one bf16 ulp on the 2-layer shape, 2x the restatement's own
    re-association spread at 24 layers) at every sampled step incl. S > 2048, argmax equal wherever the restatement's margin
    exceeds twice the bound.  pad = 30: the two fixture utterances run as rows 0-1 of a batch of 32 (the benchmark's m-tile
    occupancy) and must give the same numbers."""
    f, cfg, logits, own = run_forced_long(name, "bf16", batch_pad=pad)
    want, want_own, margins, spread = _bf16_long_reference(name, f)
    tol = _bf16_bound(cfg, spread)
    d = np.abs(logits.numpy() - want)
    assert d.max() < tol, (d.max(), tol, d.reshape(d.shape[0], -1).max(1))
    L = int(f["max_length"])
    free = free_positions(f["pattern"], L - 1)
    ok = free & (margins >= 2 * tol)
    assert ok.mean() > 0.3 * free.mean()
    assert np.array_equal(own[ok], want_own[ok])
    # bf16 noise proper, vs the fp32 reference (reported in the assertion message only if it is absurd)
    noise = np.abs(logits.numpy() - f["step_logits"]).max()
    assert noise < 0.2, noise



# ------------------------------------------------------------------------------------------------
# the benchmark precision against the REFERENCE's own bfloat16 (oracle/gen_golden.py --only-ref-bf16)
# ------------------------------------------------------------------------------------------------
def _ref_bf16_case(name):
    """(sampled steps, reference fp32 logits, reference bf16 logits, reference bf16 own argmax [steps, rows], summary) of the
    `<name>_refbf16` fixture: the real reference `.to(torch.bfloat16)` on CPU, teacher-forced along the fp32 fixture's ids, its
    decoder fed the fp32 T5 states (cast) exactly as the engine's teacher-forced runs are."""
    from tests.helpers import bf16_bits_to_float

    g, meta = load_case(name + "_refbf16")
    f, _ = load_case(name)
    r32 = g["step_logits_fp32"] if "step_logits_fp32" in g else f["step_logits"]
    return g["logit_steps"], r32, bf16_bits_to_float(g["step_logits_bf16_bits"]), g["own_bf16"].astype(np.int64), meta["summary"]["bf16"]


# The demand: the bf16 engine's error against the fp32 reference must not exceed the REFERENCE's own bf16 error
# (factor 1.0).  The engine keeps fp32 accumulators, an fp32 softmax and an fp32 residual stream where the reference rounds every
# op's output to bf16; measured (round 4, MI355X): max |dlogit| 0.0174-0.0202 against the reference-in-bf16's 0.045-0.049, mean
# 0.0027-0.0029 against 0.0070-0.0073, i.e. 0.36-0.41 of the reference's own error.  The test asserts 0.6: the demand with a
# regression guard that still clears the measurement by a third.
REF_BF16_MAX_FACTOR = 0.6
REF_BF16_MEAN_FACTOR = 0.6


@pytest.mark.parametrize("name", ["mini_greedy", "mini_greedy_ragged", "minilong_greedy", "minilong2580_greedy"])
def test_bf16_engine_is_at_least_as_close_to_the_fp32_reference_as_the_reference_in_bf16(hip, name, capsys):
    """BASELINE configs[1] runs in bf16; the reference computes in the model dtype (softmax M:342, table cast M:213).  So the
    yardstick for the bf16 engine is the reference ITSELF in bfloat16: along the same forced ids, the engine's max and mean
    |logit - fp32 reference logit| must not exceed the bf16 reference's (x the factors above), and its argmax must agree with
    the fp32 ids on at least as many kept positions (24 layers; 23 / 19 / 511 / 2579 steps; equal lengths and ragged masks)."""
    steps, r32, rb, rb_own, summ = _ref_bf16_case(name)
    if name.startswith("minilong"):
        f, cfg, logits, own = run_forced_long(name, "bf16")
        assert np.array_equal(f["logit_steps"], steps)
        e = logits.numpy()
    else:
        f, cfg, raw, logits, pattern = run_engine_case(name, "bf16", teacher_force=True)
        e = logits.numpy()[steps]
        own = logits.argmax(-1).numpy()
    L = int(f["max_length"])
    d_eng, d_ref = np.abs(e - r32), np.abs(rb - r32)
    assert abs(float(d_ref.max()) - summ["max_abs_dlogit"]) < 1e-6          # the fixture's own summary
    free = free_positions(f["pattern"], L - 1)
    gold = f["raw_ids"][:, 1:].T
    agree_eng = float((own == gold)[free].mean())
    agree_ref = float((rb_own == gold)[free].mean())
    with capsys.disabled():
        print(f"\n[ref-bf16] {name}: engine max |dlogit| {d_eng.max():.4f} mean {d_eng.mean():.5f} argmax agreement {agree_eng * 100:.2f} % | "
              f"reference in bf16: max {d_ref.max():.4f} mean {d_ref.mean():.5f} agreement {agree_ref * 100:.2f} % ({int(free.sum())} argmaxes)")
    assert d_eng.max() <= REF_BF16_MAX_FACTOR * d_ref.max(), (float(d_eng.max()), float(d_ref.max()))
    assert d_eng.mean() <= REF_BF16_MEAN_FACTOR * d_ref.mean(), (float(d_eng.mean()), float(d_ref.mean()))
    assert agree_eng >= agree_ref, (agree_eng, agree_ref)
    # where both leave the fp32 choice the cause is the margin, not the engine: every engine disagreement sits on a small margin
    miss = free & (own != gold)
    if miss.any():
        assert float(f["margins"][miss].max()) < 2 * summ["max_abs_dlogit"], float(f["margins"][miss].max())


def _free_running_audit(name, dtype):
    """The engine on a long fixture's inputs with NOTHING forced.  Returns (fixture, per-utterance report, L): for every utterance
    (b, first differing step or None, the reference's own top-2 gap at the codebook rows whose ids first differ, fraction of the
    kept ids equal to the reference's over the whole run)."""
    from parler_tts_amd import _hip
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask

    f, meta = load_case(name)
    cfg, model = model_for(name, meta, dtype)
    desc, amask, prompt, pmask = case_inputs(f)
    enc = torch.from_numpy(f["enc_t5"])
    K, B, L = cfg.decoder.num_codebooks, enc.shape[0], int(f["max_length"])
    eng = model.engine(B, enc.shape[1], prompt.shape[1], L)
    dev = "cuda:0"
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.to(dev).contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    eng.set_forced_sequence(None)
    eng.prefill(B, B, enc.to(dev).contiguous(), amask.to(dev, torch.int32).contiguous(), prompt.to(dev).contiguous(),
                None if pmask is None else pmask.to(dev, torch.int32).contiguous(), pattern, raw, 1)
    eng.decode_steps(L - 1, _hip.PttsSampling(0, 1.0, 0, 1.0, 0))
    torch.cuda.synchronize()
    assert eng.state().cur_len == L
    got, gold = raw.cpu().numpy(), f["raw_ids"]
    free = free_positions(f["pattern"], L - 1).T                      # [rows, L - 1]
    diff = (got[:, 1:] != gold[:, 1:]) & free
    report = []
    for b in range(B):
        rows = slice(b * K, (b + 1) * K)
        cols = np.nonzero(diff[rows].any(0))[0]
        if len(cols) == 0:
            report.append((b, None, None, 1.0))
            continue
        c = int(cols[0])                                              # step c filled column c + 1
        m = float(f["margins"][c][rows][diff[rows][:, c]].max())      # the reference's own top-2 gap where the ids first differ
        report.append((b, c, m, float(1.0 - diff[rows].mean() / max(free[rows].mean(), 1e-9))))
    return f, report, L


def test_bf16_free_running_audit_against_the_fp32_reference_ids(hip, capsys):
    """What the benchmark's precision does to the ids when nothing is forced: the bf16 engine on `minilong2580_greedy`'s inputs
    (mini_v0.1 shape, 2579 steps), free-running.  Per utterance, the first column where one of its codebook rows leaves the fp32
    reference's ids must be a step the reference itself decides by less than 2 x the bf16 bound (the reference-in-bf16's max
    |dlogit|, from the fixture); after that column the utterance follows another greedy path, so the agreement fraction is
    printed, not asserted."""
    name = "minilong2580_greedy"
    bound = _ref_bf16_case(name)[4]["max_abs_dlogit"]
    f, report, L = _free_running_audit(name, "bf16")
    for b, c, m, frac in report:
        assert c is None or m < 2 * bound, (b, c, m, bound)
    with capsys.disabled():
        for b, c, m, frac in report:
            print(f"\n[bf16 free run] utterance {b}: " + ("ids equal to the fp32 reference over the whole run" if c is None else
                  f"first differing step {c} of {L - 1} (reference margin there {m:.2e}, 2 x bf16 bound {2 * bound:.3f}); "
                  f"kept ids equal to the fp32 reference's: {frac * 100:.1f} %"))


FP32_ARGMAX_MARGIN = 5e-4        # below this top-2 gap of the REFERENCE an fp32 re-association may flip the argmax (logits agree to 2e-4)


@pytest.mark.parametrize("name", ["minilong_greedy", "minilong2580_greedy"])
def test_fp32_free_running_audit_against_the_fp32_reference_ids(hip, name, capsys):
    """The bit-exact claim, free-running: the fp32 engine on the REAL reference's long runs at the mini_v0.1
    shape (L = 512 and the benchmark's full horizon L = 2580, 24 layers x 16 heads over all 41 KV pages), nothing forced.  Per
    utterance the ids must EQUAL the reference's up to the first column that the reference itself decides by less than 5e-4
    (`margins` of the fixture: the top-2 gap of every argmax of the reference run); an utterance that leaves the reference's ids
    at a step with a margin >= 5e-4 fails the test.  The first differing step and the agreement over the whole run are printed."""
    f, report, L = _free_running_audit(name, "fp32")
    with capsys.disabled():
        for b, c, m, frac in report:
            print(f"\n[fp32 free run] {name} utterance {b}: " + (f"ids EQUAL to the reference over all {L - 1} steps" if c is None else
                  f"ids equal to the reference up to step {c} of {L - 1}, which the reference decides by {m:.2e} (< {FP32_ARGMAX_MARGIN:g}); "
                  f"kept ids equal over the whole run: {frac * 100:.2f} %"))
    for b, c, m, frac in report:
        assert c is None or m < FP32_ARGMAX_MARGIN, f"utterance {b} leaves the reference's ids at step {c} where the reference's margin is {m:.3e}"


# ------------------------------------------------------------------------------------------------
# text encoder
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,dtype,tol", [("tiny_greedy_ragged", "fp32", 2e-4), ("mini_greedy_ragged", "fp32", 5e-4),
                                            ("mini_greedy", "bf16", 0.15)])
def test_t5_encoder(hip, name, dtype, tol):
    f, meta = load_case(name)
    cfg, model = model_for(name, meta, dtype)
    desc, amask, _, _ = case_inputs(f)
    eng = model.engine(desc.shape[0], desc.shape[1], 8, 32)
    enc = eng.t5_encode(desc, amask).cpu().numpy()
    valid = amask.bool().numpy()
    assert np.abs(enc - f["enc_t5"])[valid].max() < tol


def test_t5_encoder_long_description_up_to_512_tokens(hip):
    """Descriptions longer than the ~300 tokens whose K / V fit the LDS (T5 accepts 512): the attention kernel reads K / V
    from the QKV rows in global memory instead.  T = 400 and T = 512 with right padding, fp32 engine vs the oracle's
    T5EncoderModel restatement; and the global-memory variant forced at a short length must equal the LDS variant."""
    from oracle import cpu_ref
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    _models.clear()
    torch.cuda.empty_cache()
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    g = torch.Generator().manual_seed(2)
    for T in (400, 512):
        desc = torch.randint(2, cfg.text_encoder.vocab_size, (2, T), generator=g)
        amask = torch.ones(2, T, dtype=torch.long)
        amask[1, T - 37:] = 0
        eng = model.engine(2, T, 4, 16)
        got = eng.t5_encode(desc, amask).cpu()
        want = cpu_ref.t5_encode(cfg, sd, desc, amask)
        assert ((got - want).abs() * amask[..., None]).max() < 5e-4, T
    desc = torch.randint(2, cfg.text_encoder.vocab_size, (2, 40), generator=g)
    amask = torch.ones(2, 40, dtype=torch.long)
    eng = model.engine(2, 40, 4, 16)
    a = eng.t5_encode(desc, amask).cpu()
    os.environ["PTTS_T5_GLOBAL_KV"] = "1"
    try:
        b = eng.t5_encode(desc, amask).cpu()
    finally:
        del os.environ["PTTS_T5_GLOBAL_KV"]
    assert (a - b).abs().max() < 1e-6
    del model
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# sampling
# ------------------------------------------------------------------------------------------------
def _fresh_prefill(eng, cfg, f, amask, prompt, L):
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask

    dev = "cuda:0"
    K, B = cfg.decoder.num_codebooks, amask.shape[0]
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.to(dev).contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    eng.prefill(B, B, torch.from_numpy(f["enc_t5"]).to(dev).contiguous(), amask.to(dev, torch.int32).contiguous(),
                prompt.to(dev).contiguous(), None, pattern, raw, 1)
    return raw


def test_sampling_probabilities_and_determinism(hip):
    from oracle import cpu_ref
    from parler_tts_amd import _hip

    f, meta = load_case("tiny_sample")
    cfg, model = model_for("tiny_sample", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    K, B, L = cfg.decoder.num_codebooks, desc.shape[0], int(f["max_length"])
    eng = model.engine(B, desc.shape[1], prompt.shape[1], L)
    gold = torch.from_numpy(f["raw_ids"])

    def run(seed, steps):
        _fresh_prefill(eng, cfg, f, amask, prompt, L)
        sp = _hip.PttsSampling(1, 0.7, 50, 1.0, seed)
        probs = []
        for s in range(steps):
            eng.set_forced_tokens(gold[:, s + 1])     # follow the reference's sampled path
            eng.decode_steps(1, sp)
            probs.append(eng.last_logits(B * K, probs=True).cpu())
        eng.set_forced_tokens(None)
        return torch.stack(probs)

    p = run(7, 8)
    ref = torch.softmax(cpu_ref.warp_logits(torch.from_numpy(f["step_logits"][:8]), 0.7, 50), dim=-1)
    assert (p - ref).abs().max() < 1e-5
    # same survivors as the reference's top-k rule: ties at the k-th value are all kept (the zeroed
    # LM-head rows of the synthetic model tie at logit 0, so more than k entries can survive)
    assert torch.equal(p > 0, ref > 0) and ((p > 0).sum(-1) >= 50).all()
    assert torch.equal(p, run(7, 8))

    # free-running draws: same seed -> same ids, different seed -> different ids
    def free(seed):
        model.generate(encoder_outputs=(torch.from_numpy(f["enc_t5"]),), attention_mask=amask,
                       prompt_input_ids=prompt, do_sample=True, temperature=0.7, top_k=50, max_length=L,
                       seed=seed, return_raw_ids=True)
        return model.last_raw_ids.clone()

    a, b, c = free(11), free(11), free(12)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_sampler_frequencies_chi_square(hip):
    """Inverse-CDF draws follow the warped distribution: chi-square over repeated draws of ONE fixed
    distribution (row 0, first step) under different Philox seeds."""
    from parler_tts_amd import _hip

    f, meta = load_case("tiny_sample")
    cfg, model = model_for("tiny_sample", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    K, B, L = cfg.decoder.num_codebooks, desc.shape[0], 10
    eng = model.engine(B, desc.shape[1], prompt.shape[1], 24)
    counts = torch.zeros(cfg.decoder.vocab_size)
    probs = None
    n_draws = 400
    for seed in range(n_draws):
        raw = _fresh_prefill(eng, cfg, f, amask, prompt, L)
        eng.decode_steps(1, _hip.PttsSampling(1, 1.0, 20, 1.0, seed))
        if probs is None:
            probs = eng.last_logits(B * K, probs=True).cpu()[0]
        torch.cuda.synchronize()
        counts[int(raw[0, 1])] += 1
    support = probs > 0
    assert int(support.sum()) == 20 and counts[~support].sum() == 0
    exp = probs[support] * n_draws
    chi2 = float(((counts[support] - exp) ** 2 / exp).sum())
    assert chi2 < 50.0, chi2    # 19 dof: p(chi2 > 50) ~ 1e-4


# ------------------------------------------------------------------------------------------------
# end to end: description ids -> waveform, against the oracle's generate()
# ------------------------------------------------------------------------------------------------
def test_generate_waveform_matches_oracle(hip):
    from oracle import cpu_ref

    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    _, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    desc, amask, prompt, _ = case_inputs(f)
    wav = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False,
                         max_length=int(f["max_length"])).cpu()
    assert np.array_equal(model.last_codes[0].cpu().numpy(), f["codes"])
    ref = cpu_ref.generate(cfg, sd, desc, amask, prompt, None, int(f["max_length"]))
    assert wav.shape == ref.shape
    rms = float((wav - ref).pow(2).mean().sqrt())
    assert rms < 1e-4, rms


def test_generate_api_errors(hip):
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, num_beams=4, do_sample=False)
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, do_sample=False, num_return_sequences=2)
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, min_length=50, max_length=20)
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, bogus_kwarg=1)
    with pytest.raises(ValueError):
        model.audio_encoder.decode(torch.zeros(2, 1, 4, 3, dtype=torch.long), [None])


def test_generate_rejects_ids_outside_the_tables(hip):
    """ids past an embedding table: IndexError as from the reference's nn.Embedding, never an out-of-bounds gather"""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    kw = dict(attention_mask=amask, do_sample=False, max_length=12)
    with pytest.raises(IndexError):
        model.generate(input_ids=desc + cfg.text_encoder.vocab_size, prompt_input_ids=prompt, **kw)
    with pytest.raises(IndexError):
        model.generate(input_ids=desc, prompt_input_ids=prompt + cfg.vocab_size, **kw)
    K = cfg.decoder.num_codebooks
    with pytest.raises(IndexError):
        model.generate(input_ids=desc, prompt_input_ids=prompt,
                       decoder_input_ids=torch.full((desc.shape[0] * K, 2), cfg.decoder.vocab_size + 1), **kw)
    # shapes the raw pointers behind the C ABI cannot check
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, prompt_input_ids=prompt, attention_mask=amask[:, :-1], do_sample=False, max_length=12)
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, prompt_input_ids=prompt, prompt_attention_mask=torch.ones(desc.shape[0], 1), **kw)
    with pytest.raises(ValueError):
        model.generate(input_ids=desc, prompt_input_ids=prompt[:-1], **kw)


def test_generate_eos_paths(hip):
    """EOS rows: the per-sample (sequential) codec path and the reference's frame filter."""
    from oracle import cpu_ref

    f, meta = load_case("tiny_eos_free")
    cfg, model = model_for("tiny_eos_free", meta, "fp32")
    _, sd = state_dict_for("tiny_eos_free", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    desc, amask, prompt, _ = case_inputs(f)
    L = int(f["max_length"])
    wav = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False,
                         max_length=L).cpu()
    assert np.array_equal(model.last_codes[0].cpu().numpy(), f["codes"])
    ref = cpu_ref.generate(cfg, sd, desc, amask, prompt, None, L)
    assert wav.shape == ref.shape and float((wav - ref).pow(2).mean().sqrt()) < 1e-4


def test_finished_utterances_stop_streaming_their_cache(hip):
    """Extension of SURVEY 8f.2: once all K codebook rows of an utterance have latched EOS its self-attention reads one key instead
    of the cache (every later token of it is pad whatever the logits say; the reference keeps computing such rows).  Ragged
    finish by teacher forcing: utterance 1 of 3 gets EOS on every codebook at step 9, utterance 0 on ONE codebook only (it must
    stay alive), utterance 2 runs free.  Raw ids, the loop state and the last logits of the live utterances must equal those of an
    engine created with PTTS_NO_SKIP_FINISHED=1, in fp32 and bf16."""
    from parler_tts_amd import _hip
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask

    f, meta = load_case("tiny_greedy")
    desc, amask, prompt, pmask = case_inputs(f)
    enc = torch.from_numpy(f["enc_t5"])
    for dtype in ("fp32", "bf16"):
        outs = []
        for skip in (True, False):
            if not skip:
                os.environ["PTTS_NO_SKIP_FINISHED"] = "1"
            try:        # the switch is read when the engine is created: a model of its own per setting
                from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
                cfg, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder"))
                model = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
                K, B, L = cfg.decoder.num_codebooks, enc.shape[0], 48
                eng = model.engine(B, enc.shape[1], prompt.shape[1], L)
            finally:
                os.environ.pop("PTTS_NO_SKIP_FINISHED", None)
            dev = "cuda:0"
            ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
            ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
            pattern = pattern.to(dev).contiguous()
            raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
            raw[:, :1] = ids.to(dev)
            forced = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
            forced[K:2 * K, 9] = cfg.decoder.eos_token_id          # utterance 1: every codebook ends at column 9
            forced[0, 9] = cfg.decoder.eos_token_id                # utterance 0: one codebook only
            own = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
            eng.set_forced_sequence(forced, own)
            try:
                eng.prefill(B, B, enc.to(dev).contiguous(), amask.to(dev, torch.int32).contiguous(), prompt.to(dev).contiguous(),
                            None if pmask is None else pmask.to(dev, torch.int32).contiguous(), pattern, raw, 1)
                eng.decode_steps(L - 1, _hip.PttsSampling(0, 1.0, 0, 1.0, 0))
                torch.cuda.synchronize()
                st = eng.state()
                outs.append((raw.cpu().clone(), eng.last_logits(B * K).cpu().clone(), st.cur_len, st.n_unfinished))
            finally:
                eng.set_forced_sequence(None)
        (ra, la, ca, na), (rb, lb, cb, nb) = outs
        B_, K_ = enc.shape[0], ra.shape[0] // enc.shape[0]
        assert torch.equal(ra, rb) and ca == cb and na == nb
        live = [r for r in range(ra.shape[0]) if not (K_ <= r < 2 * K_)]
        assert torch.equal(la[live], lb[live]), dtype                 # live utterances: bit-identical logits
        assert na == ra.shape[0] - K_ - 1                             # K rows of utterance 1 + one row of utterance 0 latched


def test_more_than_16_codebooks_ragged_finish_and_oracle_parity(hip):
    """the finished-utterance test read 16 flags per utterance, so with 17 .. 32 codebooks (ptts_create
    accepts 32) an utterance whose codebooks 0 .. 15 had latched EOS was treated as finished while codebook 16 - the LAST to
    finish under the delay pattern - was still decoding on key 0 alone.  K = 17: (a) the engine against the fp32 oracle (logits
    2e-4, clear argmaxes equal); (b) ragged finish by teacher forcing - utterance 0: all 17 codebooks end at column 9; utterance
    1: codebooks 0 .. 15 end at column 9, codebook 16 runs FREE until column 30; utterance 2 runs free - raw ids, loop state and
    the live rows' last logits bit-identical to an engine created with PTTS_NO_SKIP_FINISHED=1, fp32 and bf16."""
    from parler_tts_amd import _hip
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config(num_codebooks=17)
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    got, want, _, _ = _oracle_vs_engine(cfg, sd, "fp32", B=3, T=7, P=4, L=40, seed=3, steps=39)
    assert (got - want).abs().max() < 2e-4
    top2 = torch.topk(want, 2, dim=-1)[0]
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])

    K, B, T, P, L = 17, 3, 7, 4, 48
    g = torch.Generator().manual_seed(5)
    enc = torch.randn(B, T, cfg.text_encoder.d_model, generator=g)
    amask = torch.ones(B, T, dtype=torch.int32)
    prompt = torch.randint(2, cfg.vocab_size, (B, P), generator=g)
    eos = cfg.decoder.eos_token_id
    for dtype in ("fp32", "bf16"):
        outs = []
        for skip in (True, False):
            if not skip:
                os.environ["PTTS_NO_SKIP_FINISHED"] = "1"
            try:
                model = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
                eng = model.engine(B, T, P, L)
            finally:
                os.environ.pop("PTTS_NO_SKIP_FINISHED", None)
            dev = "cuda:0"
            ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
            ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
            pattern = pattern.to(dev).contiguous()
            raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
            raw[:, :1] = ids.to(dev)
            forced = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
            forced[0:K, 9] = eos                      # utterance 0: every codebook
            forced[K:K + 16, 9] = eos                 # utterance 1: codebooks 0 .. 15 only
            forced[K + 16, 30] = eos                  # ... its codebook 16 much later
            own = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
            eng.set_forced_sequence(forced, own)
            try:
                eng.prefill(B, B, enc.to(dev).contiguous(), amask.to(dev).contiguous(), prompt.to(dev).contiguous(), None, pattern, raw, 1)
                sp = _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
                eng.decode_steps(25, sp)              # cur_len 26: utterance 1 lives on codebook 16 alone
                torch.cuda.synchronize()
                mid = eng.last_logits(B * K).cpu().clone()
                eng.decode_steps(L - 1 - 25, sp)
                torch.cuda.synchronize()
                st = eng.state()
                outs.append((raw.cpu().clone(), mid, eng.last_logits(B * K).cpu().clone(), st.cur_len, st.n_unfinished))
            finally:
                eng.set_forced_sequence(None)
            del model
            torch.cuda.empty_cache()
        (ra, ma, la, ca, na), (rb, mb, lb, cb, nb) = outs
        assert torch.equal(ra, rb) and ca == cb and na == nb, dtype
        assert torch.equal(ma[K + 16], mb[K + 16]), dtype             # the late codebook of utterance 1 while the other 16 are done
        assert torch.equal(ma[2 * K:], mb[2 * K:]) and torch.equal(la[2 * K:], lb[2 * K:]), dtype
        assert na == K                                                 # utterances 0 and 1 latched on every codebook
        # utterance 1's codebook 16 really decoded between the two forcings: its free tokens are not pad
        assert bool((ra[K + 16, 18:30] != cfg.decoder.pad_token_id).all())


# ------------------------------------------------------------------------------------------------
# codec: hand-written implicit-GEMM conv stack vs the fixtures of the structural oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["dac_tiny", "dac_mini"])
def test_dac_decode_kernels(hip, name):
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    f, _ = load_case(name)
    cfg = config_for(name)
    sd = synthetic_state_dict(cfg, seed=int(f["seed"]), parts=("audio_encoder",))
    dac = DACModel(cfg, sd, device="cuda:0")
    codes = torch.from_numpy(f["codes"])
    wav = dac.decode(codes[None], [None]).audio_values.cpu().numpy()
    assert wav.shape == f["wav"].shape
    rms = float(np.sqrt(((wav - f["wav"]) ** 2).mean()))
    assert rms < 1e-5, rms
    assert np.abs(wav - f["wav"]).max() < 1e-4
    with pytest.raises(ValueError):
        dac.decode(torch.zeros(2, 1, codes.shape[1], 3, dtype=torch.long), [None])
    bad = codes.clone()
    bad[0, -1, 0] = cfg.audio_encoder.codebook_size + 2      # an id the LM heads can emit but no codebook holds
    with pytest.raises(IndexError):
        dac.decode(bad[None], [None])


def test_dac_streaming_1x1_conv_equals_the_tiled_kernel(hip):
    """The residual 1x1 convs of the narrow blocks (C = 96 / 192) run on a streaming kernel (activations straight from global
    memory into MFMA B fragments, weights parked in LDS / registers); same products in the same order as the tiled kernel, so
    the waveform must be bit-identical to the one computed with PTTS_DAC_NO_STREAM_1X1=1 (mini codec, 2 x 40 frames)."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    dac = DACModel(cfg, sd, device="cuda:0")
    codes = torch.randint(0, 1024, (1, 2, 9, 40), generator=torch.Generator().manual_seed(1))
    a = dac.decode(codes, None).audio_values.cpu()
    for var in ("PTTS_DAC_NO_STREAM_1X1", "PTTS_DAC_1X1_WAVES8"):
        os.environ[var] = "1"
        try:
            b = dac.decode(codes, None).audio_values.cpu()
        finally:
            del os.environ[var]
        assert torch.equal(a, b), (var, float((a - b).abs().max()))
    # the last layer (96 -> 1 channel) as an fp32 reduction kernel vs the MFMA tile kernel with the fp16 x 3 split
    os.environ["PTTS_DAC_NO_LAST_KERNEL"] = "1"
    try:
        c = dac.decode(codes, None).audio_values.cpu()
    finally:
        del os.environ["PTTS_DAC_NO_LAST_KERNEL"]
    assert float((a - c).abs().max()) < 5e-6 and float(a.abs().max()) > 0.05


def test_dac_ring_kernel_equals_the_tiled_kernel(hip):
    """The wide layers run on conv_kernel_h3r (weights by LDS-DMA through a ring of tap units, counted vmcnt waits, 64 rows per
    wave; 192-wide units for the transposed and 1x1 layers): same products in the same order as conv_kernel_h3, so the waveform
    must be bit-identical with PTTS_DAC_NO_RING=1 - at 40 frames (every workgroup is mostly padding: row masks, clamped tail
    units) and at 700 frames (several 512-row workgroups per layer, halos across their seams), for both unit widths."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    os.environ["PTTS_DAC_RING_MIN_U"] = "1"
    os.environ["PTTS_DAC_NO_FUSE"] = "1"       # the fused residual unit has its own (rounding-level) test below
    try:
        dac = DACModel(cfg, sd, device="cuda:0")
        os.environ["PTTS_DAC_NO_UNIT192"] = "1"
        try:
            dac96 = DACModel(cfg, sd, device="cuda:0")      # the unit width is chosen when the weights are loaded
        finally:
            del os.environ["PTTS_DAC_NO_UNIT192"]
        for shape in ((1, 2, 9, 40), (1, 1, 9, 700)):
            codes = torch.randint(0, 1024, shape, generator=torch.Generator().manual_seed(shape[-1]))
            a = dac.decode(codes, None).audio_values.cpu()
            b = dac96.decode(codes, None).audio_values.cpu()
            os.environ["PTTS_DAC_NO_RING"] = "1"
            try:
                ref = dac.decode(codes, None).audio_values.cpu()
            finally:
                del os.environ["PTTS_DAC_NO_RING"]
            assert float(ref.abs().max()) > 0.05
            assert torch.equal(a, ref), ("192/96-wide units", shape, float((a - ref).abs().max()))
            assert torch.equal(b, ref), ("96-wide units", shape, float((b - ref).abs().max()))
    finally:
        del os.environ["PTTS_DAC_RING_MIN_U"]
        del os.environ["PTTS_DAC_NO_FUSE"]


def test_dac_fused_residual_unit_equals_the_two_kernels(hip):
    """C = 96 / 192 residual units run as ONE launch (7-tap conv tile kept in registers -> bias, Snake, fp16 split -> B operand of
    the 1x1 stage -> + x; at C = 192 the 1x1 weights keep streaming through the ring).  The 1x1 stage sums a k-step's 32 channels
    in the accumulator's channel order, so it agrees with the two-kernel path to rounding, not bit for bit: the waveform has to
    stay within 1e-5 (max) / 1e-6 (rms) of it - a tenth of the fixture bound - at 40 frames (edge masks) and at 700."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    os.environ["PTTS_DAC_RING_MIN_U"] = "1"
    try:
        dac = DACModel(cfg, sd, device="cuda:0")
        for shape in ((1, 2, 9, 40), (1, 1, 9, 700)):
            codes = torch.randint(0, 1024, shape, generator=torch.Generator().manual_seed(shape[-1]))
            a = dac.decode(codes, None).audio_values.cpu()
            os.environ["PTTS_DAC_NO_FUSE"] = "1"
            try:
                ref = dac.decode(codes, None).audio_values.cpu()
            finally:
                del os.environ["PTTS_DAC_NO_FUSE"]
            assert float(ref.abs().max()) > 0.05
            d = (a - ref).double()
            assert float(d.abs().max()) < 1e-5 and float(d.pow(2).mean().sqrt()) < 1e-6, (shape, float(d.abs().max()))
    finally:
        del os.environ["PTTS_DAC_RING_MIN_U"]


@pytest.mark.parametrize("B,F", [(1, 1), (3, 2), (1, 7), (2, 33), (1, 257)])
def test_dac_odd_shapes_on_the_ring_and_fused_kernels(hip, B, F):
    """Edge shapes (one frame, fewer rows than a 16-row tile, lengths that leave partial tiles in every layer): the ring kernel
    at every wide layer (PTTS_DAC_RING_MIN_U=1) and the fused residual units against the register-staged two-kernel path."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    dac = DACModel(cfg, sd, device="cuda:0")
    codes = torch.randint(0, 1024, (1, B, 9, F), generator=torch.Generator().manual_seed(100 * B + F))
    os.environ["PTTS_DAC_RING_MIN_U"] = "1"
    try:
        a = dac.decode(codes, None).audio_values.cpu()
    finally:
        del os.environ["PTTS_DAC_RING_MIN_U"]
    os.environ["PTTS_DAC_NO_RING"] = "1"
    try:
        ref = dac.decode(codes, None).audio_values.cpu()
    finally:
        del os.environ["PTTS_DAC_NO_RING"]
    assert a.shape == ref.shape == (B, 1, F * cfg.audio_encoder.hop_length) and bool(torch.isfinite(a).all())
    d = (a - ref).double()
    assert float(d.abs().max()) < 1e-5 and float(d.pow(2).mean().sqrt()) < 1e-6


def test_dac_ring_kernels_relaunch_bit_identical(hip):
    """The ring kernels order LDS-DMA against the MFMA waves with counted vmcnt waits and raw barriers: a wait that is one
    instruction short shows up as a rare wrong tile, not as a parity failure.  60 relaunches of the whole decoder (ring, 192-wide
    and fused kernels at every layer, 2 x 300 frames) must reproduce the first waveform bit for bit (`tools/dac_repeat.py` is the
    longer form: 300 relaunches of 4 x 600 frames, 0 differences)."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    os.environ["PTTS_DAC_RING_MIN_U"] = "1"
    try:
        dac = DACModel(cfg, sd, device="cuda:0")
        codes = torch.randint(0, 1024, (1, 2, 9, 300), generator=torch.Generator().manual_seed(5))
        ref = dac.decode(codes, None).audio_values.clone()
        bad = sum(0 if torch.equal(dac.decode(codes, None).audio_values, ref) else 1 for _ in range(60))
        assert bad == 0 and float(ref.abs().max()) > 0.05
    finally:
        del os.environ["PTTS_DAC_RING_MIN_U"]


def test_dac_decode_long_clip_chunked_equals_unchunked(hip):
    """time tiles, halos across tile seams and batch chunks: a long clip decoded alone must equal the same
    clip decoded inside a batch (size-independent property used at the BASELINE sizes)."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_tiny")
    sd = synthetic_state_dict(cfg, seed=11, parts=("audio_encoder",))
    dac = DACModel(cfg, sd, device="cuda:0")
    g = torch.Generator().manual_seed(1)
    codes = torch.randint(0, 64, (3, 4, 700), generator=g)
    full = dac.decode(codes[None], [None]).audio_values
    one = dac.decode(codes[1:2][None], [None]).audio_values
    assert torch.equal(full[1:2], one)
    # prefix property: the codec is causal-free but local; far from the cut the prefix decode must agree
    pre = dac.decode(codes[:, :, :400][None], [None]).audio_values
    hop = cfg.audio_encoder.hop_length
    assert torch.allclose(pre[..., : 300 * hop], full[..., : 300 * hop], atol=1e-6)


def test_dac_decode_at_benchmark_length_properties(hip):
    """BASELINE configs[1] codec shape (mini codec, F = 2571 frames = 1.3 M samples per clip; every layer on its production kernel:
    LDS-DMA ring, 192-wide units, fused residual units, streaming 1x1, last-layer reduction).  No oracle finishes this size in
    seconds, so size-independent properties: a clip decoded alone equals the same clip inside a batch (bit for bit: rows are
    independent), the first 2000 frames of the long decode equal the decode of a 2100-frame prefix (locality: the receptive
    field is ~10 frames) and agree with the register-staged two-kernel path (PTTS_DAC_NO_RING=1) to rounding."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    dac = DACModel(cfg, sd, device="cuda:0")
    codes = torch.randint(0, 1024, (3, 9, 2571), generator=torch.Generator().manual_seed(2))
    full = dac.decode(codes[None], [None]).audio_values
    one = dac.decode(codes[1:2][None], [None]).audio_values
    assert full.shape[-1] == 2571 * cfg.audio_encoder.hop_length and torch.equal(full[1:2], one)
    pre = dac.decode(codes[:1, :, :2100][None], [None]).audio_values
    hop = cfg.audio_encoder.hop_length
    assert torch.allclose(pre[..., : 2000 * hop], full[:1, ..., : 2000 * hop], atol=1e-6)
    os.environ["PTTS_DAC_NO_RING"] = "1"
    try:
        ref = dac.decode(codes[:1][None], [None]).audio_values
    finally:
        del os.environ["PTTS_DAC_NO_RING"]
    d = (full[:1] - ref).double()
    assert float(d.abs().max()) < 1e-5 and float(d.pow(2).mean().sqrt()) < 1e-6 and float(ref.abs().max()) > 0.05



def _production_kernel_counts(frames_ge_1024):
    """launches per kernel family of ONE mini-codec decode in the default environment (conv.hip dispatch): conv0 and the
    first transposed conv have F and F + 1 GEMM rows, so below 1024 frames those two stay on the register-staged kernel"""
    want = dict(last=1, fused96=3, fused192=3, s1=6, ring192_t7=6, ring192_t2=3, h3=2, f32=0, stream96=0, stream192=0,
                ring96_t7=0, ring96_t2=0, ring96_t1=0, ring192_t1=0)
    if frames_ge_1024:
        want.update(ring192_t7=7, ring192_t2=4, h3=0)
    return want


@pytest.mark.parametrize("name", ["dac_mini_f300", "dac_mini_f2571"])
def test_dac_production_kernels_against_the_oracle_at_production_sizes(hip, name):
    """The kernels that decode the benchmark's clips - LDS-DMA ring (7-tap and transposed instances), windowless wide 1x1
    (conv_kernel_s1), fused residual units (96 / 192 wide), last-layer reduction - against the structural oracle itself
    (transformers DacModel with the same weights, oracle/gen_golden.py --only-dac-long), at sizes where the DEFAULT dispatch
    selects them: 2 x 300 frames (whole waveform) and 1 x 2571 frames = the benchmark clip (every 64th sample, float64 sums
    and sums of squares over every 4096-sample block, whole windows at the head, across the middle tile seams, around frame
    1000 and at the tail).  No PTTS_DAC_* switch is set; the launch tallies prove which kernels produced the waveform."""
    from parler_tts_amd import _hip
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    assert not [k for k in os.environ if k.startswith("PTTS_DAC_")], "this test is about the default dispatch"
    f, _ = load_case(name)
    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=int(f["seed"]), parts=("audio_encoder",))
    dac = DACModel(cfg, sd, device="cuda:0")
    codes = torch.from_numpy(f["codes"])
    dac.decode(codes[None, :, :, :8], [None])            # weights packed, workspaces allocated
    _hip.dac_kernel_counts(reset=True)
    wav = dac.decode(codes[None], [None]).audio_values.cpu().double().reshape(codes.shape[0], -1)
    counts = _hip.dac_kernel_counts(reset=True)
    assert counts == _production_kernel_counts(codes.shape[-1] >= 1024), counts
    if "wav" in f:
        ref = torch.from_numpy(f["wav"]).double().reshape(codes.shape[0], -1)
        d = wav - ref
        assert wav.shape == ref.shape and float(ref.abs().max()) > 0.05
        assert float(d.pow(2).mean().sqrt()) < 1e-5 and float(d.abs().max()) < 1e-4, (float(d.pow(2).mean().sqrt()), float(d.abs().max()))
        return
    n, stride, block = int(f["n_samples"]), int(f["stride"]), int(f["block"])
    assert wav.shape[1] == n
    d = wav[:, ::stride] - torch.from_numpy(f["wav_stride"]).double()
    assert float(d.pow(2).mean().sqrt()) < 1e-5 and float(d.abs().max()) < 1e-4, (float(d.pow(2).mean().sqrt()), float(d.abs().max()))
    nb = (n + block - 1) // block
    pad = torch.zeros(wav.shape[0], nb * block - n, dtype=wav.dtype)
    sums = torch.cat([wav, pad], 1).reshape(wav.shape[0], nb, block).sum(-1)
    sq = torch.cat([wav * wav, pad], 1).reshape(wav.shape[0], nb, block).sum(-1)
    ref_s, ref_q = torch.from_numpy(f["block_sums"]), torch.from_numpy(f["block_sqsums"])
    # a mean error of 1e-5 per sample over a block, and the matching bound on the energy (2 |x| |dx| summed)
    assert float((sums - ref_s).abs().max()) < block * 1e-5, float((sums - ref_s).abs().max())
    assert float(((sq - ref_q).abs() / (2e-5 * block * ref_q.div(block).sqrt() + 1e-9)).max()) < 1.0
    for i in range(int(f["n_windows"])):
        a, b = (int(x) for x in f[f"win{i}_range"])
        dw = wav[:, a:b] - torch.from_numpy(f[f"win{i}"]).double()
        assert float(dw.pow(2).mean().sqrt()) < 1e-5 and float(dw.abs().max()) < 1e-4, (i, float(dw.abs().max()))


def test_dac_transposed_convs_on_the_dma_window_kernel_equal_the_ring_kernel_bit_for_bit(hip):
    """Round 6: the layer in front of every transposed conv (conv0, the last residual unit of blocks 1-3) applies that conv's
    Snake and the fp16 hi / lo split ONCE per element in its epilogue and stores [hi x 8 | lo x 8] per 8-channel group in place
    (csrc/conv.h: alpha_out / in_packed); the transposed conv then runs on convt_dma_kernel, whose activation window arrives by
    LDS-DMA a chunk ahead instead of through registers, Snake, split, ds_write and two barriers per chunk.  Same Snake, same
    split, same MFMA operands in the same order: the waveform must be BIT-IDENTICAL to the fp32 hand-over + ring kernel
    (PTTS_DAC_NO_PRESPLIT=1) - at the benchmark clip length, in a batch with a ragged last row tile, and at 300 frames where conv0
    and the first transposed conv stay on the register-staged kernel (that pair is not packed) - and a clip alone must equal the
    same clip inside the batch."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=21, parts=("audio_encoder",))
    g = torch.Generator().manual_seed(3)
    for B, F in ((1, 2571), (3, 1100), (2, 300)):
        codes = torch.randint(0, 1024, (B, 9, F), generator=g)
        out = {}
        for off in (False, True):
            if off:
                os.environ["PTTS_DAC_NO_PRESPLIT"] = "1"
            try:
                dac = DACModel(cfg, sd, device="cuda:0")
                out[off] = dac.decode(codes[None], [None]).audio_values.cpu()
                if not off and B > 1:
                    one = dac.decode(codes[None, 1:2], [None]).audio_values.cpu()
                    assert torch.equal(one, out[off][1:2]), (B, F)
                del dac
            finally:
                os.environ.pop("PTTS_DAC_NO_PRESPLIT", None)
            torch.cuda.empty_cache()
        assert float(out[False].abs().max()) > 0.05 and torch.equal(out[False], out[True]), (B, F)


def test_dac_wide_seven_tap_convs_on_64x96_wave_tiles_equal_the_32x192_instance_bit_for_bit(hip):
    """Round 6: the wide residual units' 7-tap convs (C = 768, 384 at mini_v0.1) run the ring kernel with the eight waves of a
    workgroup tiled 4 row groups x 2 column halves (conv_kernel_h3r<192, 64, 7, 4, false, WC = 2>: 64 rows x 96 outputs per
    wave, 20 fragment reads per tap instead of 28) over the SAME workgroup tile, ring units and window as the 32 x 192 instance
    (PTTS_DAC_WAVE_TILE=32x192).  Every output is the same sum in the same order: the waveform must be BIT-IDENTICAL - at the
    benchmark clip length, in a batch with ragged row tiles, at short clips - and a clip alone must equal the clip in the batch."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=29, parts=("audio_encoder",))
    g = torch.Generator().manual_seed(7)
    for B, F in ((1, 2571), (3, 1100), (2, 300), (1, 37)):
        codes = torch.randint(0, 1024, (B, 9, F), generator=g)
        out = {}
        for off in (False, True):
            if off:
                os.environ["PTTS_DAC_WAVE_TILE"] = "32x192"
            try:
                dac = DACModel(cfg, sd, device="cuda:0")
                out[off] = dac.decode(codes[None], [None]).audio_values.cpu()
                if not off and B > 1:
                    one = dac.decode(codes[None, 1:2], [None]).audio_values.cpu()
                    assert torch.equal(one, out[off][1:2]), (B, F)
                del dac
            finally:
                os.environ.pop("PTTS_DAC_WAVE_TILE", None)
            torch.cuda.empty_cache()
        assert float(out[False].abs().max()) > 0.05 and torch.equal(out[False], out[True]), (B, F)


def test_dac_fused_unit_at_96_channels_on_four_wave_workgroups_equals_the_eight_wave_one_bit_for_bit(hip):
    """Round 6: the fused residual units of the last block (C = 96) run as FOUR-wave workgroups over 256 rows with a ring of 3 -
    76 KB of LDS, two workgroups on a CU, so that one's prologue, second stage and epilogue run beside the other's chunk loop
    (conv_kernel_h3r<96, 64, 7, 3, true, 1, NW = 4>) - instead of one eight-wave workgroup over 512 rows
    (PTTS_DAC_FUSED96_WAVES=8).  A wave computes the same 64 x 96 outputs from the same operands in the same order (the second
    stage streams its three k-steps instead of holding them): the waveform must be BIT-IDENTICAL at the benchmark clip length, in a
    batch with ragged row tiles, at short clips; and a clip alone must equal the clip in the batch."""
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=37, parts=("audio_encoder",))
    g = torch.Generator().manual_seed(13)
    for B, F in ((1, 2571), (3, 1100), (2, 300), (1, 37)):
        codes = torch.randint(0, 1024, (B, 9, F), generator=g)
        out = {}
        for off in (False, True):
            if off:
                os.environ["PTTS_DAC_FUSED96_WAVES"] = "8"
            try:
                dac = DACModel(cfg, sd, device="cuda:0")
                out[off] = dac.decode(codes[None], [None]).audio_values.cpu()
                if not off and B > 1:
                    one = dac.decode(codes[None, 1:2], [None]).audio_values.cpu()
                    assert torch.equal(one, out[off][1:2]), (B, F)
                del dac
            finally:
                os.environ.pop("PTTS_DAC_FUSED96_WAVES", None)
            torch.cuda.empty_cache()
        assert float(out[False].abs().max()) > 0.05 and torch.equal(out[False], out[True]), (B, F)


# ------------------------------------------------------------------------------------------------
# codec, encode direction (audio prompts): DAC encoder conv stack + residual vector quantiser
# ------------------------------------------------------------------------------------------------
def _codec_with_encoder(name, seed):
    from parler_tts_amd.modeling import DACModel
    from parler_tts_amd.weights import synthetic_dac_encoder_state_dict, synthetic_state_dict

    cfg = config_for(name)
    sd = synthetic_state_dict(cfg, seed=seed, parts=("audio_encoder",))
    sd.update(synthetic_dac_encoder_state_dict(cfg.audio_encoder, seed=seed))
    return cfg, sd, DACModel(cfg, sd, device="cuda:0")


def _codec_latent(dac, B, F, D):
    """the encoder output of the last ptts_dac_encode call (debug buffer), as [B, D, F]"""
    eng = dac._engine()
    ptr, nbytes = C.c_void_p(), C.c_int64()
    assert eng.lib.ptts_debug_buffer(eng.handle, b"codec_latent", C.byref(ptr), C.byref(nbytes)) == 0
    assert nbytes.value == B * F * D * 4
    out = torch.empty(B, F, D, device="cuda:0", dtype=torch.float32)
    from parler_tts_amd.modeling import _hipMemcpyDtoD

    torch.cuda.synchronize()
    assert _hipMemcpyDtoD(out.data_ptr(), ptr.value, nbytes.value) == 0
    return out.permute(0, 2, 1).cpu()


@pytest.mark.parametrize("name", ["dac_enc_tiny", "dac_enc_mini"])
def test_dac_encode_kernels(hip, name):
    """ptts_dac_encode vs the fixture transformers' DacModel.encode produced (oracle/gen_golden.py): codes bit-exact (every
    nearest-entry decision of the fixture has a top-2 distance gap >= 1e-4; the encoder latent is checked to 2e-5), lengths
    that are no multiple of the hop (every strided conv floors, D:85-86)."""
    f, _ = load_case(name)
    cfg, sd, dac = _codec_with_encoder("dac_mini" if name.endswith("mini") else "dac_tiny", int(f["seed"]))
    audio = torch.from_numpy(f["audio"])
    out = dac.encode(audio)
    codes = out.audio_codes.cpu()
    assert out.audio_scales == [None]
    assert codes.shape == f["codes"].shape and codes.dtype == torch.int64
    B, _, T = audio.shape
    F = T // cfg.audio_encoder.hop_length
    z = _codec_latent(dac, B, F, cfg.audio_encoder.latent_dim).numpy()
    assert np.abs(z - f["latent"]).max() < 2e-5 * max(1.0, float(np.abs(f["latent"]).max())), np.abs(z - f["latent"]).max()
    assert np.array_equal(codes.numpy(), f["codes"])
    # API errors of D:53-54 and the tuple form
    with pytest.raises(ValueError):
        dac.encode(torch.zeros(1, 3, 4096))
    tup = dac.encode(audio, return_dict=False)
    assert torch.equal(tup[0].cpu(), codes) and tup[1] == [None]
    # encode -> decode -> shapes of the round trip (D:98-134 takes what D:25-96 returns)
    wav = dac.decode(out.audio_codes, out.audio_scales).audio_values
    assert wav.shape == (B, 1, F * cfg.audio_encoder.hop_length)


def test_dac_encode_against_the_oracle_at_one_second_of_audio(hip):
    """ptts_dac_encode at 1.2 s of audio (2 x 52813 samples, not a hop multiple; several 512-row workgroups per layer and
    tile seams in every block) vs transformers' DacModel.encode (fixture dac_enc_mini_long): codes bit-exact, latent 2e-5."""
    from parler_tts_amd import _hip

    f, _ = load_case("dac_enc_mini_long")
    cfg, sd, dac = _codec_with_encoder("dac_mini", int(f["seed"]))
    B, T = int(f["B"]), int(f["T"])
    gg = torch.Generator().manual_seed(int(f["seed_in"]))
    audio = (torch.rand(B, 1, T, generator=gg) * 2 - 1) * 0.5
    _hip.dac_kernel_counts(reset=True)
    codes = dac.encode(audio).audio_codes.cpu()
    counts = _hip.dac_kernel_counts(reset=True)
    assert counts["f32"] == 0 and counts["h3"] > 0, counts        # every MFMA layer on the fp16 x 3 fast path
    F = T // cfg.audio_encoder.hop_length
    assert codes.shape == f["codes"].shape == (1, B, cfg.audio_encoder.num_codebooks, F)
    z = _codec_latent(dac, B, F, cfg.audio_encoder.latent_dim).numpy()[:, :, :: int(f["latent_stride"])]
    assert np.abs(z - f["latent"]).max() < 2e-5 * max(1.0, float(np.abs(f["latent"]).max())), np.abs(z - f["latent"]).max()
    assert np.array_equal(codes.numpy(), f["codes"])


def test_dac_encode_batch_and_length_invariants(hip):
    """Size-independent properties at sizes the CPU oracle does not cover: an utterance encoded alone equals the same
    utterance inside a batch (tile seams, batch strides); samples past the last whole frame never reach a code that the
    receptive field cannot see; the fp32-MFMA build of the conv stack (PTTS_DAC_F32) gives the same codes."""
    cfg, sd, dac = _codec_with_encoder("dac_tiny", 13)
    hop = cfg.audio_encoder.hop_length
    g = torch.Generator().manual_seed(5)
    T = hop * 700 + 11
    audio = (torch.rand(3, 1, T, generator=g) * 2 - 1) * 0.5
    full = dac.encode(audio).audio_codes
    assert full.shape == (1, 3, cfg.audio_encoder.num_codebooks, 700)
    one = dac.encode(audio[1:2]).audio_codes
    assert torch.equal(full[:, 1:2], one)
    # the 11 trailing samples lie beyond the last strided window of every stage only if they are dropped by each floor:
    # compare against the oracle on a short clip instead of asserting locality here
    from oracle import cpu_ref

    short = audio[:, :, : hop * 40 + 11]
    ref = cpu_ref.dac_encode(cfg, sd, short)
    got = dac.encode(short).audio_codes.cpu()
    agree = (got == ref).float().mean().item()
    assert agree > 0.995, agree          # random audio: no margin scan here, a near-tie may flip
    old = os.environ.get("PTTS_DAC_F32")
    os.environ["PTTS_DAC_F32"] = "1"
    try:
        _, _, dac32 = _codec_with_encoder("dac_tiny", 13)
        got32 = dac32.encode(short).audio_codes.cpu()
    finally:
        if old is None:
            del os.environ["PTTS_DAC_F32"]
        else:
            os.environ["PTTS_DAC_F32"] = old
    assert (got32 == ref).float().mean().item() > 0.995
    assert (got32 == got).float().mean().item() > 0.995


def test_generate_with_audio_prompt(hip):
    """`input_values` (M:2483-2487 -> M:2247-2292): the waveform is encoded by the codec, its ids become the decoder prefix
    (start column prepended, M:2184-2185), generation continues after it.  fp32 engine vs the oracle, ids bit-exact."""
    from oracle import cpu_ref
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_dac_encoder_state_dict

    f, meta = load_case("tiny_greedy")
    fe, _ = load_case("dac_enc_tiny")
    cfg, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    sd = dict(sd)
    sd.update(synthetic_dac_encoder_state_dict(cfg.audio_encoder, seed=int(fe["seed"])))
    # the fixture's margin scan was done with the codec weights of ITS seed: use them for the whole codec
    from parler_tts_amd.weights import synthetic_state_dict

    sd.update(synthetic_state_dict(cfg, seed=int(fe["seed"]), parts=("audio_encoder",)))
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    desc, amask, prompt, pmask = case_inputs(f)
    B = desc.shape[0]
    K = cfg.decoder.num_codebooks
    rows = [i % 2 for i in range(B)]                   # the fixture holds two clips: repeat them over the batch
    audio = torch.from_numpy(fe["audio"])[rows]
    n_pre = audio.shape[-1] // cfg.audio_encoder.hop_length
    enc = torch.from_numpy(f["enc_t5"])
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, do_sample=False, return_raw_ids=True)
    got = model.generate(input_values=audio, max_new_tokens=12, **kw)
    codes = torch.from_numpy(fe["codes"])[0, rows]
    prefix = codes.reshape(B * K, n_pre)
    L = 12 + n_pre + 1
    ref = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, pmask, L, decoder_input_ids=prefix)
    assert torch.equal(model.last_raw_ids.cpu()[:, n_pre + 1:], ref["raw_ids"][:, n_pre + 1:])
    assert torch.equal(got.cpu(), cpu_ref.undelay(cfg, ref["raw_ids"], ref["pattern"]))
    # explicit decoder_input_ids win over input_values (M:2483)
    again = model.generate(input_values=audio * 0, decoder_input_ids=prefix, max_new_tokens=12, **kw)
    assert torch.equal(again, got)
    # the waveform path: prompt audio + continuation come back as one clip
    wav = model.generate(input_values=audio, max_new_tokens=12, **{k: v for k, v in kw.items() if k != "return_raw_ids"})
    assert wav.dim() == 2 and wav.shape[0] == B and torch.isfinite(wav).all()
    # the codec does not depend on the decoder's storage type: a bf16 engine encodes the prompt to the same ids, and its
    # prefix columns (teacher-forced by the delay pattern) are the fixture's codes
    del model
    torch.cuda.empty_cache()
    m16 = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
    assert torch.equal(m16.audio_encoder.encode(audio).audio_codes.cpu()[0], codes)
    out16 = m16.generate(input_values=audio, max_new_tokens=12, **kw)
    assert torch.equal(out16.cpu()[0][:, :, : n_pre - K + 1], codes[:, :, : n_pre - K + 1])


# ------------------------------------------------------------------------------------------------
# BASELINE-size properties (mini_v0.1, bf16, batch 32, max_length 2580): size-independent invariants; the oracle-pinned
# checks of the same shapes are the long-horizon tests above (the mini fixture as rows 0-1 of a batch of 32)
# ------------------------------------------------------------------------------------------------
def test_baseline_size_invariants_bf16_batch32(hip):
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    _models.clear()
    torch.cuda.empty_cache()
    cfg = mini_v01_config()
    sd = synthetic_state_dict(cfg, seed=1234, never_eos=True, parts=("text_encoder", "decoder"))
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
    B, T, P, L, K = 32, 40, 16, 2580, 9        # BASELINE configs[1] at its full max_length (41 KV pages per utterance)
    g = torch.Generator().manual_seed(1)
    desc = torch.randint(2, 32100, (B, T), generator=g)
    prompt = torch.randint(2, 32100, (B, P), generator=g)
    kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)
    codes = model.generate(**kw)
    raw = model.last_raw_ids.clone()
    # (1) shape and range: every run lasts max_length - 1 steps, F = L - K frames, ids are valid codec ids
    assert codes.shape == (1, B, K, L - K) and raw.shape == (B * K, L)
    assert int(codes.min()) >= 0 and int(codes.max()) < 1024
    # (2) determinism: a second pass (replayed graph, recycled KV pages) is bit-identical
    codes2 = model.generate(**kw)
    assert torch.equal(codes, codes2)
    # (3) batch independence: utterance b alone equals utterance b inside the batch (paged KV rows,
    #     MFMA column b, per-row select are independent).  Only frames produced before the short run's
    #     own PAD tail (columns >= 105 - 9 + 1) starts feeding PAD tokens back in are comparable: 80 frames.
    for b in (0, 17):
        single = model.generate(input_ids=desc[b:b + 1], prompt_input_ids=prompt[b:b + 1], do_sample=False,
                                max_length=105, return_raw_ids=True)
        assert torch.equal(single[0, 0, :, :80], codes[0, b, :, :80])
    # (4) the raw ids honour the loop's pad-out rule: nothing is padded because no row ever emitted EOS
    _, pattern = build_delay_pattern_mask(torch.full((B * K, 1), 1025), 1025, 1024, L, K)
    free = pattern == -1
    assert bool((raw.cpu()[free] < 1024).all())
    del model
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# API surface of generate(): streamer, guidance, ragged masks end to end, checkpoint round trip
# ------------------------------------------------------------------------------------------------
def test_generate_ragged_masks_end_to_end(hip):
    """description mask (right padded) + left-padded prompt mask, T5 on device, vs the reference fixture"""
    f, meta = load_case("tiny_greedy_ragged")
    cfg, model = model_for("tiny_greedy_ragged", meta, "fp32")
    desc, amask, prompt, pmask = case_inputs(f)
    codes = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, prompt_attention_mask=pmask,
                           do_sample=False, max_length=int(f["max_length"]), return_raw_ids=True)
    assert np.array_equal(codes[0].cpu().numpy(), f["codes"])


def test_generate_with_guidance_scale(hip):
    f, meta = load_case("tiny_cfg")
    cfg, model = model_for("tiny_cfg", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    codes = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False,
                           guidance_scale=3.0, max_length=int(f["max_length"]), return_raw_ids=True)
    assert np.array_equal(codes[0].cpu().numpy(), f["codes"])


def test_max_length_below_two_codebook_spans(hip):
    """max_length < 2K - 1: build_delay_pattern_mask returns an all-(-1) mask (M:105-106), so nothing is forced, the un-delay
    keeps every column including the start column, and the special id there sends the codec through the per-sample path
    that drops such frames (M:2654-2680).  fp32 engine vs the oracle: ids bit-exact, waveform close."""
    from oracle import cpu_ref

    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    _, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    desc, amask, prompt, pmask = case_inputs(f)
    K = cfg.decoder.num_codebooks
    L = 2 * K - 2
    enc = torch.from_numpy(f["enc_t5"])
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=L)
    ids = model.generate(return_raw_ids=True, **kw)
    ref = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, pmask, L)
    assert bool((ref["pattern"] == -1).all())
    assert torch.equal(model.last_raw_ids.cpu(), ref["raw_ids"])
    want = cpu_ref.undelay(cfg, ref["raw_ids"], ref["pattern"])
    assert want.shape[-1] == L and torch.equal(ids.cpu(), want)          # every column kept, start column included
    wav = model.generate(**kw).cpu()
    ref_wav = cpu_ref.codes_to_waveform(cfg, sd, want)
    assert wav.shape == ref_wav.shape == (desc.shape[0], (L - 1) * cfg.audio_encoder.hop_length)
    assert float((wav - ref_wav).pow(2).mean().sqrt()) < 1e-4


def test_streamer_protocol(hip):
    """put(start ids) once, then one put per generated column, then end() (M:2540-2541 + the HF loop)"""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)

    class Rec:
        def __init__(self):
            self.items, self.ended = [], False

        def put(self, t):
            assert t.device.type == "cpu"
            self.items.append(t.clone())

        def end(self):
            self.ended = True

    r = Rec()
    L = int(f["max_length"])
    model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=L,
                   streamer=r, return_raw_ids=True)
    assert r.ended and len(r.items) == L          # 1 (start ids) + L-1 steps
    assert r.items[0].shape == (12, 1) and bool((r.items[0] == cfg.decoder.bos_token_id).all())
    streamed = torch.stack(r.items[1:], dim=1)
    assert torch.equal(streamed, model.last_raw_ids.cpu()[:, 1:])


def test_audio_streamer_chunks_add_up_to_the_one_shot_waveform(hip):
    """Streaming with chunked DAC decode: AudioStreamer un-delays the id columns as they arrive and decodes
    windows of play_frames + 2 x context frames; the codec decoder is a stack of finite convolutions, so with a context of
    its receptive field the concatenated chunks equal the waveform generate() returns after the loop, sample for sample.
    Also the polling: with poll_steps = 5 the host reads the device state every 5 steps, the streamer still sees every column
    exactly once."""
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.streamer import AudioStreamer, dac_context_frames
    from parler_tts_amd.weights import synthetic_state_dict

    assert dac_context_frames((8, 8, 4, 2)) == 10           # the 44 kHz DAC: 10 frames of context per side
    cfg = tiny_config(max_positions=512)
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1)
    _models.clear()
    torch.cuda.empty_cache()
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    g = torch.Generator().manual_seed(3)
    desc = torch.randint(2, cfg.text_encoder.vocab_size, (2, 9), generator=g)
    prompt = torch.randint(2, cfg.vocab_size, (2, 5), generator=g)
    L = 160
    kw = dict(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_length=L)
    want = model.generate(**kw).cpu()
    seen = []
    st = AudioStreamer(model, play_frames=20, poll_steps=5, on_audio=lambda c, last: seen.append((c.shape[-1], last)))
    got = model.generate(streamer=st, **kw).cpu()
    assert torch.equal(got, want)
    assert len(st.cols) == L and st.emitted == L - cfg.decoder.num_codebooks
    wav = st.waveform()
    assert wav.shape == want.shape and len(st.chunks) >= 5 and seen[-1][1] and not seen[0][1]
    assert float((wav - want).abs().max()) < 1e-6, float((wav - want).abs().max())
    assert sum(1 for _ in st) == len(st.chunks)             # the queue holds every chunk, then the end marker
    with pytest.raises(ValueError):
        AudioStreamer(model, context_frames=3)
    del model
    torch.cuda.empty_cache()


def test_from_pretrained_roundtrip_with_weight_norm(hip, tmp_path):
    from parler_tts_amd.configuration import GenerationConfig
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import save_pretrained

    from parler_tts_amd.weights import synthetic_dac_encoder_state_dict

    f, meta = load_case("tiny_greedy")
    cfg, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    sd = dict(sd)
    sd.update(synthetic_dac_encoder_state_dict(cfg.audio_encoder, seed=7))      # a full checkpoint carries the codec encoder too
    # write the codec convs the way a descript checkpoint stores them: weight_g / weight_v pairs
    disk = {}
    for k, v in sd.items():
        if k.startswith("audio_encoder.") and k.endswith(".weight") and v.dim() == 3:
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
            disk[k[:-len("weight")] + "weight_g"] = norm.clone()
            disk[k[:-len("weight")] + "weight_v"] = (v * 1.7).clone()      # any positive rescale of v folds back
        else:
            disk[k] = v
    save_pretrained(str(tmp_path), cfg, disk, GenerationConfig(max_length=24, do_sample=False,
                    pad_token_id=cfg.decoder.pad_token_id, eos_token_id=cfg.decoder.eos_token_id,
                    bos_token_id=cfg.decoder.bos_token_id, decoder_start_token_id=cfg.decoder.bos_token_id))
    model = ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path), dtype="fp32")
    assert model.config.sampling_rate == 44100 and model.generation_config.max_length == 24
    desc, amask, prompt, _ = case_inputs(f)
    wav = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt)   # config defaults: greedy, L=24
    assert np.array_equal(model.last_codes[0].cpu().numpy(), f["codes"])
    assert wav.shape == (3, f["codes"].shape[-1] * cfg.audio_encoder.hop_length)
    # the encode side came through the same loader (weight-norm folded): audio prompts work on a loaded checkpoint and agree
    # with the CPU restatement on the un-normed tensors (random audio, no margin scan: a near-tie may flip)
    from oracle import cpu_ref

    g = torch.Generator().manual_seed(9)
    audio = (torch.rand(2, 1, cfg.audio_encoder.hop_length * 30 + 7, generator=g) * 2 - 1) * 0.5
    got = model.audio_encoder.encode(audio).audio_codes.cpu()
    ref = cpu_ref.dac_encode(cfg, sd, audio)
    assert got.shape == ref.shape and (got == ref).float().mean().item() > 0.99
    # model.save_pretrained(): the loaded (normalised) state dict written back and read again gives the same codes
    again = tmp_path / "again"
    model.save_pretrained(str(again))
    model2 = ParlerTTSForConditionalGeneration.from_pretrained(str(again), torch_dtype=torch.float32)     # the transformers spelling
    assert model2.generation_config.max_length == 24
    # torch_dtype="auto" (transformers: the dtype recorded in config.json, else the default): no record here -> the engine default
    assert ParlerTTSForConditionalGeneration.from_pretrained(str(again), torch_dtype="auto").dtype == "bf16"
    with pytest.raises(NotImplementedError):
        ParlerTTSForConditionalGeneration.from_pretrained(str(again), torch_dtype=torch.float16)
    model2.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt)
    assert np.array_equal(model2.last_codes[0].cpu().numpy(), f["codes"])
    del model2
    _models.clear()


# ------------------------------------------------------------------------------------------------
# shapes beyond the fixtures: more than one 32-row m-tile, and BASELINE configs[4] (scaled decoder)
# ------------------------------------------------------------------------------------------------
_last_ref = {}


def _oracle_vs_engine(cfg, sd, dtype, B, T, P, L, seed, steps, ragged=False, check_t5=False, oracle="fp32", sampling=None,
                      reuse_ref=False):
    """teacher-forced along the ORACLE's greedy path; returns (engine logits, oracle logits, oracle ids).
    ragged: right-padded descriptions and LEFT-padded prompts (training script T:916) with their masks.
    oracle = "bf16": the engine-numerics restatement (oracle/cpu_ref_bf16.py) instead of the fp32 one.
    sampling = PttsSampling: the engine runs its sampling warpers (the draws are overridden by the teacher forcing) and the
    returned "logits" are its post-warper probabilities."""
    from oracle import cpu_ref
    from parler_tts_amd import _hip
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    g = torch.Generator().manual_seed(seed)
    desc = torch.randint(2, cfg.text_encoder.vocab_size, (B, T), generator=g)
    prompt = torch.randint(2, cfg.vocab_size, (B, P), generator=g)
    amask = torch.ones(B, T, dtype=torch.long)
    pmask = None
    if ragged:
        pmask = torch.ones(B, P, dtype=torch.long)
        for b in range(1, B):
            amask[b, T - (7 * b) % (T // 2):] = 0
            pmask[b, : (5 * b) % (P // 2 + 1)] = 0
    if "text_encoder.shared.weight" in sd:
        enc = cpu_ref.t5_encode(cfg, sd, desc, amask)
    else:       # decoder-only state dict: random encoder states of the T5 width stand in for the text encoder
        enc = torch.randn(B, T, cfg.text_encoder.d_model, generator=g)
    if reuse_ref:                 # second engine run of the same case (determinism checks): the oracle result is kept
        ref = _last_ref["ref"]
    elif oracle == "bf16":
        from oracle import cpu_ref_bf16

        ref = cpu_ref_bf16.generate_ids(cfg, sd, enc, amask, prompt, pmask, L, record=steps)
        ref["step_logits"] = ref["step_logits"][:steps]
        # the restatement's own re-association spread (see _bf16_bound)
        alt = cpu_ref_bf16.generate_ids(cfg, sd, enc, amask, prompt, pmask, L, record=steps, acc64=True,
                                        forced=ref["raw_ids"])["step_logits"][:steps]
        _oracle_vs_engine.spread = float((alt - ref["step_logits"]).abs().max())
    else:
        ref = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, pmask, L, record_logits=steps)
    _last_ref["ref"] = ref
    _models.clear()
    torch.cuda.empty_cache()
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype=dtype, device="cuda:0")
    eng = model.engine(B, T, P, L)
    K = cfg.decoder.num_codebooks
    dev = "cuda:0"
    ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
    ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
    pattern = pattern.to(dev).contiguous()
    raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
    raw[:, :1] = ids.to(dev)
    if check_t5:       # the HIP T5 encoder at this length, valid rows only (padded rows are never attended)
        enc_gpu = eng.t5_encode(desc.to(dev), amask.to(dev)).cpu()
        assert ((enc_gpu - enc).abs() * amask[..., None]).max() < 2e-4
    eng.prefill(B, B, enc.to(dev).contiguous(), amask.to(dev, torch.int32).contiguous(), prompt.to(dev).contiguous(),
                None if pmask is None else pmask.to(dev, torch.int32).contiguous(), pattern, raw, 1)
    sp = sampling if sampling is not None else _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
    out = []
    for s in range(steps):
        eng.set_forced_tokens(ref["raw_ids"][:, s + 1])
        eng.decode_steps(1, sp)
        out.append(eng.last_logits(B * K, probs=sampling is not None).cpu())
    eng.set_forced_tokens(None)
    got_raw = raw[:, : steps + 1].cpu()
    del model
    _models.clear()
    torch.cuda.empty_cache()
    return torch.stack(out), ref["step_logits"], ref["raw_ids"], got_raw


def test_two_m_tiles_batch_40_fp32(hip):
    """B = 40 > 32: GEMMs run two 32-row m-tiles, attention / select grids grow with the batch"""
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    got, want, ref_ids, got_raw = _oracle_vs_engine(cfg, sd, "fp32", B=40, T=7, P=4, L=16, seed=3, steps=15)
    assert (got - want).abs().max() < 2e-4
    top2 = torch.topk(want, 2, dim=-1)[0]
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3            # only argmaxes the reference itself decides clearly
    assert torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_batch_sizes_1_17_33_reproduce_fixture_rows(hip, dtype):
    """Utterances are independent (no cross-row op anywhere), so any batch built from the fixture's rows must reproduce
    their codes: batch 1 (BASELINE configs[0]), 17 (one partly filled 32-row m-tile, the half-split GEMM grid with an
    almost empty second half) and 33 (a second m-tile with one row).  fp32: bit-exact against the reference's codes;
    bf16: every copy of a row gives the same codes as every other copy."""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, dtype)
    desc, amask, prompt, pmask = case_inputs(f)
    enc = torch.from_numpy(f["enc_t5"])
    want = torch.from_numpy(f["codes"])
    for B in (1, 17, 33):
        rows = [i % desc.shape[0] for i in range(B)]
        got = model.generate(encoder_outputs=(enc[rows],), attention_mask=amask[rows], prompt_input_ids=prompt[rows],
                             do_sample=False, max_length=int(f["max_length"]), return_raw_ids=True)[0].cpu()
        assert got.shape[0] == B
        if dtype == "fp32":
            assert torch.equal(got, want[rows]), B
        else:
            for i in range(B):
                assert torch.equal(got[i], got[rows[i]]), (B, i)


def test_long_description_long_prompt_and_page_crossings_fp32(hip):
    """Lengths the fixtures never reach: a description of 90 tokens (cross-attention over two KV pages on the general
    kernel, T5 over 90 positions), a left-padded prompt of 70 tokens (prefill crosses a page), 60 generated columns (self
    KV positions up to 129: three pages), ragged masks; fp32 engine teacher-forced along the oracle's greedy path."""
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    got, want, ref_ids, got_raw = _oracle_vs_engine(cfg, sd, "fp32", B=3, T=90, P=70, L=61, seed=8, steps=60, ragged=True,
                                                    check_t5=True)
    assert (got - want).abs().max() < 2e-4
    top2 = torch.topk(want, 2, dim=-1)[0]
    safe = (top2[..., 0] - top2[..., 1]) > 1e-3
    assert torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])


VARIANTS = [dict(norm_type="rms_norm"), dict(position_embedding="rope"), dict(mlp_type="swiglu"),
            dict(norm_type="rms_norm", position_embedding="rope", mlp_type="swiglu")]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: "+".join(sorted(v.values())))
def test_decoder_op_variants_unpinned_by_the_reference(hip, dtype, variant):
    """RMSNorm / rotary positions / SwiGLU as decoder variants (configuration.py extensions for the random-init scaled decoder,
    SURVEY.md 8f.4).  UNPINNED BY THE REFERENCE: its decoder has LayerNorm with bias, additive sinusoids and a GELU MLP and no
    switch, so the checker is the plain fp32 PyTorch statement of the same ops in oracle/cpu_ref.decoder_forward - each flag
    alone and all three together, ragged masks, a prompt that crosses into a second KV page (rotation angles of the prefill
    chunk rows AND of the decode steps), teacher-forced: fp32 engine logits within 2e-4 and the argmax equal wherever the
    statement decides clearly; bf16 engine within bf16 noise of it."""
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config(**variant)
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    if "mlp_type" in variant:
        assert "decoder.model.decoder.layers.0.fc1_up.weight" in sd
    got, want, ref_ids, got_raw = _oracle_vs_engine(cfg, sd, dtype, B=3, T=10, P=60, L=21, seed=12, steps=20, ragged=True)
    if dtype == "fp32":
        assert (got - want).abs().max() < 2e-4, (got - want).abs().max()
        top2 = torch.topk(want, 2, dim=-1)[0]
        safe = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert safe.float().mean() > 0.3 and torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])
    else:
        assert (got - want).abs().max() < 0.08, (got - want).abs().max()
        top2 = torch.topk(want, 2, dim=-1)[0]
        safe = (top2[..., 0] - top2[..., 1]) > 0.05
        assert (got.argmax(-1)[safe] == want.argmax(-1)[safe]).float().mean() > 0.9


def test_decoder_op_variants_scaled_config_all_three_bf16(hip):
    """BASELINE configs[4]'s decoder (d_model 1536, 24 heads, ffn 6144; cut to 4 layers for the CPU statement) with all three
    variants on, batch 8, bf16 and fp32 (rows wider than the LDS stage: folded RMSNorm in both), against the fp32 PyTorch
    statement.  Unpinned by the reference, as above."""
    from parler_tts_amd.configuration import scaled_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = scaled_config()
    cfg.decoder.num_hidden_layers = 4
    cfg.decoder.norm_type, cfg.decoder.position_embedding, cfg.decoder.mlp_type = "rms_norm", "rope", "swiglu"
    sd = synthetic_state_dict(cfg, seed=5, never_eos=True, ln_jitter=0.1, parts=("decoder",))
    for dtype, tol in (("fp32", 5e-4), ("bf16", 0.1)):
        got, want, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=8, T=40, P=16, L=9, seed=3, steps=8)
        assert (got - want).abs().max() < tol, (dtype, (got - want).abs().max())


def test_long_description_bf16_lookahead_off_path(hip):
    """bf16 engine with a description longer than one KV page: the look-ahead cross-q is inactive (it rides on the
    single-page cross-attention kernel) and the folded LayerNorm feeds the separate cross-q projection instead."""
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    got, want, _, _ = _oracle_vs_engine(cfg, sd, "bf16", B=3, T=90, P=70, L=41, seed=9, steps=40, ragged=True)
    assert (got - want).abs().max() < 0.08
    top2 = torch.topk(want, 2, dim=-1)[0]
    safe = (top2[..., 0] - top2[..., 1]) > 0.05
    assert (got.argmax(-1)[safe] == want.argmax(-1)[safe]).float().mean() > 0.9


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_scaled_decoder_config_24_layers_batch8(hip, dtype):
    """BASELINE configs[4] on its own workload: d_model 1536, 24 heads, ffn 6144, ALL 24 layers (random init), batch 8 = the
    per-GPU share of batch 64 over 8 GPUs.  fp32 engine vs the fp32 oracle (LayerNorm rows of 1536 > 1024 take the folded
    form, engine.hip ptts_create), bf16 engine vs the engine-numerics restatement; teacher-forced along the oracle's path."""
    from parler_tts_amd.configuration import scaled_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = scaled_config()
    sd = synthetic_state_dict(cfg, seed=5, never_eos=True, parts=("decoder",))
    got, want, ref_ids, got_raw = _oracle_vs_engine(cfg, sd, dtype, B=8, T=20, P=8, L=14, seed=4, steps=5, oracle=dtype)
    tol = 3e-4 if dtype == "fp32" else _bf16_bound(cfg, _oracle_vs_engine.spread)
    assert (got - want).abs().max() < tol, ((got - want).abs().max(), tol)
    top2 = torch.topk(want, 2, dim=-1)[0]
    safe = (top2[..., 0] - top2[..., 1]) > 2 * tol
    assert safe.float().mean() > 0.3
    assert torch.equal(got.argmax(-1)[safe], want.argmax(-1)[safe])


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sampling_mini_batch32_topk50_temperature07(hip, dtype):
    """BASELINE configs[3] on its own workload: parler_tts_mini_v0.1 shapes, batch 32, T = 40, P = 16, do_sample with
    top_k = 50 and temperature = 0.7.  Post-warper probability vectors of the first decode steps (teacher-forced along the
    oracle's path) within 1e-5 (fp32 engine vs the fp32 oracle); bf16 engine vs the engine-numerics restatement in log space
    within what its logit bound allows; identical survivor sets where the k-th largest score is not a near-tie; Philox
    determinism."""
    from oracle import cpu_ref
    from parler_tts_amd import _hip
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    sd = synthetic_state_dict(cfg, seed=1234, never_eos=True, parts=("decoder",))
    sp = _hip.PttsSampling(1, 0.7, 50, 1.0, 99)
    steps = 3
    got, want, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=32, T=40, P=16, L=12, seed=1234, steps=steps, oracle=dtype, sampling=sp)
    ref = torch.softmax(cpu_ref.warp_logits(want, 0.7, 50), dim=-1)
    ltol = 3e-4 if dtype == "fp32" else _bf16_bound(cfg, _oracle_vs_engine.spread)
    assert ((got > 0).sum(-1) >= 50).all() and torch.allclose(got.sum(-1), torch.ones(got.shape[:-1]), atol=1e-5)
    # survivor sets (top-k keeps every score >= the k-th largest): an entry may be kept by one side and dropped by the other
    # only if its score is within the logit tolerance of the threshold
    thr = torch.sort(want, dim=-1, descending=True)[0][..., 49:50]
    mismatch = (got > 0) != (ref > 0)
    assert bool(((want - thr).abs()[mismatch] <= 2 * ltol).all())
    if dtype == "fp32":
        assert mismatch.float().mean() < 1e-4
        assert (got - ref).abs()[~mismatch].max() < 1e-5, (got - ref).abs()[~mismatch].max()
    else:
        # a logit moved by at most ltol moves log p by at most 2 * ltol / temperature (its own score and the normaliser),
        # compared over the common support; SURVEY 8d's flat 1e-3 does not hold at 24 layers, where two same-rounding bf16
        # pipelines already differ by ~1e-2 on the logits
        both = (got > 0) & (ref > 0)
        g = torch.where(both, got, torch.zeros_like(got))
        r = torch.where(both, ref, torch.zeros_like(ref))
        g, r = g / g.sum(-1, keepdim=True), r / r.sum(-1, keepdim=True)
        dlog = (g[both].log() - r[both].log()).abs().max()
        assert dlog < 2 * ltol / 0.7 + 1e-4, (dlog, ltol)
    got2, _, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=32, T=40, P=16, L=12, seed=1234, steps=steps, oracle=dtype, sampling=sp,
                                      reuse_ref=True)
    assert torch.equal(got, got2)


def test_top_p_warper_matches_oracle(hip):
    """temperature -> top-k -> top-p, teacher-forced along the reference's sampled path"""
    from oracle import cpu_ref
    from parler_tts_amd import _hip

    f, meta = load_case("tiny_sample")
    cfg, model = model_for("tiny_sample", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    K, B, L = cfg.decoder.num_codebooks, desc.shape[0], int(f["max_length"])
    eng = model.engine(B, desc.shape[1], prompt.shape[1], L)
    gold = torch.from_numpy(f["raw_ids"])
    _fresh_prefill(eng, cfg, f, amask, prompt, L)
    sp = _hip.PttsSampling(1, 0.7, 40, 1.0, 3, 0.8)
    probs = []
    for s in range(6):
        eng.set_forced_tokens(gold[:, s + 1])
        eng.decode_steps(1, sp)
        probs.append(eng.last_logits(B * K, probs=True).cpu())
    eng.set_forced_tokens(None)
    p = torch.stack(probs)
    ref = torch.softmax(cpu_ref.warp_logits(torch.from_numpy(f["step_logits"][:6]), 0.7, 40, 0.8), dim=-1)
    # The synthetic heads tie 64 logits at exactly 0 and torch.sort orders ties arbitrarily (the kernel uses
    # index order), so WHICH of the tied tokens survive is not defined by the reference; how many survive and
    # with what probabilities is: compare the sorted probability vectors.
    ps, rs = p.sort(-1, descending=True)[0], ref.sort(-1, descending=True)[0]
    assert torch.equal((ps > 0).sum(-1), (rs > 0).sum(-1))
    assert (ps - rs).abs().max() < 1e-5
    untied = torch.from_numpy(f["step_logits"][:6]) != 0          # outside the tie, membership must agree exactly
    assert torch.equal((p > 0)[untied], (ref > 0)[untied])
    assert ((p > 0).sum(-1) < 40).any()          # top-p actually removed something beyond top-k
    assert torch.allclose(p.sum(-1), torch.ones(p.shape[:2]), atol=1e-5)
    # and it is reachable through generate()
    model.generate(encoder_outputs=(torch.from_numpy(f["enc_t5"]),), attention_mask=amask, prompt_input_ids=prompt,
                   do_sample=True, top_k=40, top_p=0.8, temperature=0.7, max_length=L, seed=5, return_raw_ids=True)


def test_decoder_without_enc_to_dec_proj(hip):
    """text-encoder width == decoder width: no enc_to_dec_proj (M:2010-2014 skipped), only the mask multiply"""
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.weights import all_keys, synthetic_state_dict

    cfg = tiny_config(enc_d=128)
    assert "enc_to_dec_proj.weight" not in all_keys(cfg)
    sd = synthetic_state_dict(cfg, seed=13, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    got, want, _, _ = _oracle_vs_engine(cfg, sd, "fp32", B=3, T=7, P=4, L=14, seed=6, steps=10)
    assert (got - want).abs().max() < 2e-4


def _fresh_engine_logits(name, dtype, env):
    """teacher-forced per-step logits from a NEW engine created under `env` (the switches are read in ptts_create)"""
    old = {k: os.environ.get(k) for k in env}
    try:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _models.clear()
        torch.cuda.empty_cache()
        return run_engine_case(name, dtype, teacher_force=True)[3]
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _models.clear()


def test_folded_and_staged_layernorm_agree(hip):
    """same model, same teacher-forced tokens: LayerNorm folded behind the MFMA vs the staged two-pass prologue"""
    a = _fresh_engine_logits("tiny_greedy", "fp32", {"PTTS_LN_FOLD": "1"})
    b = _fresh_engine_logits("tiny_greedy", "fp32", {"PTTS_LN_FOLD": "0"})
    assert (a - b).abs().max().item() < 1e-4
    a = _fresh_engine_logits("tiny_greedy", "bf16", {"PTTS_LN_FOLD": "1"})
    b = _fresh_engine_logits("tiny_greedy", "bf16", {"PTTS_LN_FOLD": "0"})
    assert (a - b).abs().max().item() < 0.08
    assert (a.argmax(-1) == b.argmax(-1)).float().mean().item() > 0.9


def test_specialised_gemm_instances_equal_the_generic_ones_bit_for_bit(hip):
    """The decode projections run on GEMM instances with a compile-time epilogue, one unguarded tile burst per wave and (N = 1024
    residual projections, one m-tile) 8-row quarter workgroups.  Same products, same order: at mini_v0.1 widths, batch 32 and
    batch 19 (a quarter with three rows, an empty one), bf16 and fp32, the teacher-forced logits must be BIT-IDENTICAL to
    the generic instances (PTTS_GEMM_GENERIC=1), to the half-split grid (PTTS_GEMM_NO_QUARTERS=1) and - LM heads on three
    feature blocks per workgroup, bf16 - to the one-block instance (PTTS_GEMM_NO_NB3=1)."""
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 2
    sd = synthetic_state_dict(cfg, seed=7, never_eos=True, ln_jitter=0.1, parts=("decoder",))
    for dtype in ("bf16", "fp32"):
        for B in (32, 19, 8):          # 8: whole 32-row workgroups with a compile-time residual epilogue (M <= 16)
            got, want, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=B, T=40, P=16, L=8, seed=8, steps=6)
            assert (got - want).abs().max() < (0.1 if dtype == "bf16" else 5e-4)
            for var in ("PTTS_GEMM_NO_QUARTERS", "PTTS_GEMM_GENERIC", "PTTS_GEMM_NO_NB3"):
                os.environ[var] = "1"
                try:
                    alt, _, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=B, T=40, P=16, L=8, seed=8, steps=6, reuse_ref=True)
                finally:
                    del os.environ[var]
                assert torch.equal(got, alt), (dtype, B, var, float((got - alt).abs().max()))


def test_fused_qkv_attention_launch_equals_the_two_launches_bit_for_bit(hip):
    """Round 5: the QKV projection and the self-attention of a decode layer run as ONE launch (csrc/fused.hip: the attention
    workgroups are resident behind the projection workgroups of the same grid, stream the cached K rows while the projection
    computes, and wait for their head's q / k / v through arrival tags).  Same products in the same order on both sides, only the
    store / load instructions of the handed-over rows differ: teacher-forced logits must be BIT-IDENTICAL to the two launches of
    rounds 1-4 (PTTS_NO_FUSED_QKV_ATTN=1) - mini_v0.1 widths, bf16 (exact-fit burst of 4 tiles) and fp32 with the folded
    LayerNorm (8 tiles), batch 32, 19 (a ragged m-tile) and 40 (two m-tiles: tags per m-tile), prompts that cross a KV page."""
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 2
    sd = synthetic_state_dict(cfg, seed=7, never_eos=True, ln_jitter=0.1, parts=("decoder",))
    for dtype, env in (("bf16", {}), ("fp32", {"PTTS_LN_FOLD": "1"})):
        os.environ.update(env)
        try:
            for B, P in ((32, 16), (19, 70), (40, 5)):
                got, want, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=B, T=40, P=P, L=10, seed=8, steps=8)
                assert (got - want).abs().max() < (0.1 if dtype == "bf16" else 5e-4)
                for off in (("PTTS_NO_FUSED_QKV_ATTN",),):
                    for k in off:
                        os.environ[k] = "1"
                    try:
                        alt, _, _, _ = _oracle_vs_engine(cfg, sd, dtype, B=B, T=40, P=P, L=10, seed=8, steps=8, reuse_ref=True)
                    finally:
                        for k in off:
                            del os.environ[k]
                    assert torch.equal(got, alt), (dtype, B, off, float((got - alt).abs().max()))
        finally:
            for k in env:
                del os.environ[k]


def test_fused_qkv_attention_free_run_under_graph_replay_and_ragged_finish(hip):
    """The fused launch over a few hundred REPLAYED steps (the arrival tags carry pos + 1 and are never reset inside a call),
    through the public generate(): raw ids equal to the two-launch engine's, column for column, in bf16 at mini_v0.1 widths -
    free-running over four KV pages, then with live EOS heads so that utterances finish at different steps (a finished utterance's
    attention reads key 0 only, its workgroups still wait for their tags), then twice in a row on one engine (a new call restarts
    pos: the tags are cleared at prefill)."""
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 3
    g = torch.Generator().manual_seed(5)
    B, T, P = 7, 12, 9
    enc = torch.randn(B, T, cfg.text_encoder.d_model, generator=g) * 0.3
    prompt = torch.randint(2, 32000, (B, P), generator=g)
    outs = {}
    for never_eos, L in ((True, 270), (False, 120)):
        sd = synthetic_state_dict(cfg, seed=3, never_eos=never_eos, ln_jitter=0.1, parts=("decoder",))
        for fused in (True, False):
            if not fused:
                os.environ["PTTS_NO_FUSED_QKV_ATTN"] = "1"
            try:
                _models.clear()
                torch.cuda.empty_cache()
                model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
                runs = []
                for _ in range(2):
                    model.generate(encoder_outputs=(enc,), prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)
                    runs.append(model.last_raw_ids.cpu().clone())
                assert torch.equal(runs[0], runs[1])
                outs[(never_eos, fused)] = runs[0]
                del model
            finally:
                os.environ.pop("PTTS_NO_FUSED_QKV_ATTN", None)
        a, b = outs[(never_eos, True)], outs[(never_eos, False)]
        assert a.shape == b.shape and torch.equal(a, b), (never_eos, a.shape, b.shape)
    assert outs[(True, True)].shape[1] == 270
    torch.cuda.empty_cache()


def test_deferred_loop_state_equals_the_advance_launch_of_rounds_1_to_5(hip):
    """Round 6: a replayed step no longer ends with a one-workgroup advance launch.  The selection leaves the next step's counters
    in state[1] and its EOS latches in `unfinished`; one extra workgroup of the NEXT step's embedding launch settles state[0]
    from them, and ptts_decode_steps settles it once behind the last replay of a call (csrc/misc.hip settle_state).  Against
    PTTS_EAGER_ADVANCE=1 (every replay ends with advance_kernel, as in rounds 1-5), at mini_v0.1 widths in bf16:
    (a) the loop state read at EVERY C-ABI boundary of a walk in calls of 1, 2, 3, 5, 1, 7 ... steps - replays, eager profiled
        steps and forward / select half steps mixed - with live EOS heads (utterances finish at different steps: n_unfinished
        falls, done_len is latched once), past max_length (replays clamp), and over a second prefill on the same engine;
    (b) raw ids of generate(): greedy without EOS over four KV pages, greedy and sampled with live EOS (the pipelined EOS poll
        decides where the call stops, the Philox draw is keyed by step_count)."""
    from parler_tts_amd import _hip
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 2
    g = torch.Generator().manual_seed(11)
    B, T, P = 5, 12, 9
    K = cfg.decoder.num_codebooks
    enc = torch.randn(B, T, cfg.text_encoder.d_model, generator=g) * 0.3
    prompt = torch.randint(2, 32000, (B, P), generator=g)
    dev = "cuda:0"

    def walk(model, L):
        eng = model.engine(B, T, P, L)
        trace = []
        for call in range(2):
            ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
            ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
            pattern = pattern.to(dev).contiguous()
            raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device=dev)
            raw[:, :1] = ids.to(dev)
            eng.prefill(B, B, enc.to(dev).contiguous(), torch.ones(B, T, dtype=torch.int32, device=dev), prompt.to(dev).contiguous(),
                        None, pattern, raw, 1)
            sp = _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
            st = eng.state()
            trace.append((st.pos, st.cur_len, st.n_unfinished, st.done_len, st.step_count))
            sizes = [1, 2, 3, 5, 1, 7, 11, 4]
            i = 0
            while st.cur_len < L:
                n = sizes[i % len(sizes)]
                if i % 5 == 3:                       # half steps (what user logits processors drive)
                    eng.forward_step()
                    eng.select_step(sp)
                elif i % 7 == 5:                     # a call whose first step is an eager, profiled one
                    eng.decode_steps(n, sp, profile_every=1)
                    eng.profile_read()
                else:
                    eng.decode_steps(n, sp)
                if i % 3 == 0:                       # the pipelined poll's view
                    eng.state_snapshot(0)
                    ps = eng.state_poll(0)
                    trace.append(("snap", ps.pos, ps.cur_len, ps.n_unfinished, ps.done_len, ps.step_count))
                st = eng.state()
                trace.append((st.pos, st.cur_len, st.n_unfinished, st.done_len, st.step_count))
                i += 1
            eng.decode_steps(3, sp)                  # past max_length: nothing moves
            st = eng.state()
            trace.append((st.pos, st.cur_len, st.n_unfinished, st.done_len, st.step_count))
            trace.append(raw.cpu().clone())
        return trace

    outs = {}
    for eager in (False, True):
        if eager:
            os.environ["PTTS_EAGER_ADVANCE"] = "1"
        try:
            for never_eos, L in ((True, 270), (False, 120)):
                _models.clear()
                torch.cuda.empty_cache()
                sd = synthetic_state_dict(cfg, seed=3, never_eos=never_eos, ln_jitter=0.1, parts=("decoder",))
                model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device=dev)
                model.generate(encoder_outputs=(enc,), prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)
                runs = [model.last_raw_ids.cpu().clone()]
                if not never_eos:
                    model.generate(encoder_outputs=(enc,), prompt_input_ids=prompt, do_sample=True, temperature=0.9, top_k=40,
                                   max_length=L, return_raw_ids=True, seed=5)
                    runs.append(model.last_raw_ids.cpu().clone())
                    runs.append(walk(model, 60))
                outs[(eager, never_eos)] = runs
                del model
        finally:
            os.environ.pop("PTTS_EAGER_ADVANCE", None)
    for never_eos in (True, False):
        a, b = outs[(False, never_eos)], outs[(True, never_eos)]
        assert a[0].shape == b[0].shape and torch.equal(a[0], b[0]), never_eos
        if not never_eos:
            assert a[1].shape == b[1].shape and torch.equal(a[1], b[1])
            ta, tb = a[2], b[2]
            assert len(ta) == len(tb)
            for x, y in zip(ta, tb):
                if torch.is_tensor(x):
                    assert torch.equal(x, y)
                else:
                    assert x == y, (x, y)
            states = [x for x in ta if not torch.is_tensor(x) and x[0] != "snap"]
            assert any(s[3] > 0 for s in states) or min(s[2] for s in states) > 0     # done_len latched iff every row finished
            assert states[-1][1] == 60 or states[-1][3] > 0
    assert outs[(False, True)][0].shape[1] == 270
    torch.cuda.empty_cache()


def test_fused_launch_hand_off_failure_is_a_loud_error_not_a_hang(hip):
    """The attention half of the fused launch waits for arrival tags of the projection half with a BOUNDED spin.  If they never come
    (test hook PTTS_FUSED_TEST_BREAK=1: the projection half stores its tags elsewhere) the first workgroups give up after 2 s of
    device wall time, the give-up is sticky - everything behind them stops waiting within ~1 ms - and the call ends in a PttsError
    at the next read of the loop state instead of hanging the GPU or returning ids computed from a stale query.  EVERY way through
    generate() reads that state before it hands out ids: the pipelined EOS poll, a call with NO EOS criterion (which
    polls nothing on the way), a streamer, and the step-wise path of user logits processors."""
    import time

    from parler_tts_amd import _hip
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 2
    sd = synthetic_state_dict(cfg, seed=5, never_eos=True, parts=("decoder",))
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(4, 8, cfg.text_encoder.d_model, generator=g) * 0.3
    prompt = torch.randint(2, 32000, (4, 6), generator=g)
    os.environ["PTTS_FUSED_TEST_BREAK"] = "1"
    try:
        _models.clear()
        torch.cuda.empty_cache()
        model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
        from parler_tts_amd import AudioStreamer  # noqa: F401  (the streamer protocol: put / end)

        class _Sink:
            poll_steps = 4

            def __init__(self):
                self.cols = 0

            def put(self, v):
                self.cols += 1

            def end(self):
                pass

        for kw in (dict(), dict(eos_token_id=None), dict(streamer=_Sink()), dict(logits_processor=[lambda ids, sc: sc])):
            t0 = time.perf_counter()
            with pytest.raises(_hip.PttsError, match="gave up waiting"):
                model.generate(encoder_outputs=(enc,), prompt_input_ids=prompt, do_sample=False, max_length=60, return_raw_ids=True, **kw)
            torch.cuda.synchronize()
            # bounded: seconds, not the hours an unbounded or per-launch time-out would take
            assert time.perf_counter() - t0 < 60.0, (list(kw), time.perf_counter() - t0)
        del model
    finally:
        del os.environ["PTTS_FUSED_TEST_BREAK"]
        _models.clear()
        torch.cuda.empty_cache()
    # the same call on a fresh engine without the hook works
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
    model.generate(encoder_outputs=(enc,), prompt_input_ids=prompt, do_sample=False, max_length=20, return_raw_ids=True)
    assert model.last_raw_ids.shape[1] == 20
    del model
    torch.cuda.empty_cache()


def test_page_attention_kernel_equals_general_kernel(hip):
    """cross-attention over a single page: one-wave kernel vs the general paged kernel (same arithmetic, other order)"""
    for name in ("tiny_greedy_ragged", "tiny_cfg"):      # padded description keys / fully masked null rows
        a = _fresh_engine_logits(name, "fp32", {"PTTS_ATTN_NO_PAGE_KERNEL": None})
        b = _fresh_engine_logits(name, "fp32", {"PTTS_ATTN_NO_PAGE_KERNEL": "1"})
        assert (a - b).abs().max().item() < 2e-5, name


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_batched_prefill_equals_position_at_a_time_prefill(hip, dtype):
    """The batched prefill (all prompt / prefix positions as rows of MFMA GEMMs + causal attention over the cache, the
    reference's single P + 1 position forward M:794-978 / M:870-872) against the round-1 form that fed one cached position
    at a time through the decode kernels (PTTS_PREFILL_SEQ=1), and against itself cut into chunks of 3 positions: same
    teacher-forced logits to fp32 re-association (fp32) / bf16 rounding flips (bf16); ragged masks and guidance included."""
    tol = 2e-5 if dtype == "fp32" else 0.02
    for name in ("tiny_greedy", "tiny_greedy_ragged", "tiny_cfg"):
        a = _fresh_engine_logits(name, dtype, {"PTTS_PREFILL_SEQ": None, "PTTS_PREFILL_CHUNK": None})
        b = _fresh_engine_logits(name, dtype, {"PTTS_PREFILL_SEQ": "1", "PTTS_PREFILL_CHUNK": None})
        c = _fresh_engine_logits(name, dtype, {"PTTS_PREFILL_SEQ": None, "PTTS_PREFILL_CHUNK": "3"})
        assert (a - b).abs().max().item() < tol, (name, (a - b).abs().max().item())
        assert (a - c).abs().max().item() < tol, (name, (a - c).abs().max().item())


def test_one_off_phases_on_the_tile_kernel_against_the_32_row_kernels_bf16(hip):
    """T5 encoder, cross K / V and batched prefill at the benchmark's row counts (32 x 40 = 1280 description rows, 32 x 16 = 512
    prompt rows; ragged masks) on the MFMA tile kernel (default) against the same phases on the 32-row decode kernels
    (PTTS_PREFILL_M128=0): same operands and rounding points, another fp32 summation order -> encoder states within fp32
    rounding amplified by the bf16 roundings between the layers, teacher-forced logits within the bf16 bound of the restatement;
    and the default against the bf16 restatement itself (the yardstick of every bf16 test)."""
    from oracle import cpu_ref
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 2
    cfg.text_encoder.num_layers = 3
    sd = synthetic_state_dict(cfg, seed=7, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    g = torch.Generator().manual_seed(5)
    B, T = 32, 40
    desc = torch.randint(2, cfg.text_encoder.vocab_size, (B, T), generator=g)
    amask = torch.ones(B, T, dtype=torch.long)
    for b in range(1, B):
        amask[b, T - (7 * b) % (T // 2):] = 0
    want = cpu_ref.t5_encode(cfg, sd, desc, amask)
    enc = {}
    for off in (False, True):
        if off:
            os.environ["PTTS_PREFILL_M128"] = "0"
        try:
            _models.clear()
            torch.cuda.empty_cache()
            model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
            enc[off] = model.engine(B, T, 16, 16).t5_encode(desc, amask).cpu()
            del model
        finally:
            os.environ.pop("PTTS_PREFILL_M128", None)
    m = amask[..., None].float()
    assert ((enc[False] - want).abs() * m).max() < 0.15 and ((enc[True] - want).abs() * m).max() < 0.15
    assert ((enc[False] - enc[True]).abs() * m).max() < 0.05, ((enc[False] - enc[True]).abs() * m).max()
    # prefill + cross K / V: teacher-forced logits of the first steps
    sd = synthetic_state_dict(cfg, seed=7, never_eos=True, ln_jitter=0.1, parts=("decoder",))
    for ragged in (False, True):
        got, ref, _, _ = _oracle_vs_engine(cfg, sd, "bf16", B=32, T=40, P=16, L=10, seed=8, steps=8, oracle="bf16", ragged=ragged)
        tol = _bf16_bound(cfg, _oracle_vs_engine.spread)
        os.environ["PTTS_PREFILL_M128"] = "0"
        try:
            alt, _, _, _ = _oracle_vs_engine(cfg, sd, "bf16", B=32, T=40, P=16, L=10, seed=8, steps=8, oracle="bf16", ragged=ragged,
                                             reuse_ref=True)
        finally:
            del os.environ["PTTS_PREFILL_M128"]
        assert (got - ref).abs().max() < tol and (alt - ref).abs().max() < tol, (ragged, (got - ref).abs().max(), tol)
        assert (got - alt).abs().max() < tol, (ragged, (got - alt).abs().max(), tol)
        assert not torch.equal(got, alt)          # the switch does change the path (another summation order)


def test_batched_prefill_long_prompt_and_prefix_fp32(hip):
    """P = 200 prompt positions (two chunks of 128 / 72, four KV pages) and a 70-column decoder prefix in one pass each:
    fp32 engine vs the fp32 oracle, first decode steps teacher-forced along the oracle's path."""
    from oracle import cpu_ref
    from parler_tts_amd.configuration import tiny_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = tiny_config(max_positions=512)
    sd = synthetic_state_dict(cfg, seed=11, never_eos=True, ln_jitter=0.1, parts=("text_encoder", "decoder"))
    got, want, ref_ids, got_raw = _oracle_vs_engine(cfg, sd, "fp32", B=3, T=20, P=200, L=12, seed=9, steps=6, ragged=True)
    assert (got - want).abs().max() < 2e-4
    # decoder prefix: generate() end to end against the oracle loop
    K = cfg.decoder.num_codebooks
    g = torch.Generator().manual_seed(4)
    B, T, P, n_pre, L = 2, 9, 5, 70, 90
    desc = torch.randint(2, cfg.text_encoder.vocab_size, (B, T), generator=g)
    prompt = torch.randint(2, cfg.vocab_size, (B, P), generator=g)
    prefix = torch.randint(0, cfg.decoder.pad_token_id, (B * K, n_pre), generator=g)
    amask = torch.ones(B, T, dtype=torch.long)
    enc = cpu_ref.t5_encode(cfg, sd, desc, amask)
    ref = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, None, L, decoder_input_ids=prefix)
    _models.clear()
    torch.cuda.empty_cache()
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    model.generate(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, decoder_input_ids=prefix,
                   do_sample=False, max_length=L, return_raw_ids=True)
    got = model.last_raw_ids.cpu()
    m = ref["margins"]                                        # step j fills column n_pre + 1 + j
    c0 = ref["raw_ids"].shape[1] - m.shape[0]
    clear = (m >= 5e-4).T & (ref["pattern"][:, c0:] == -1)
    assert clear.float().mean() > 0.3
    assert torch.equal(got[:, c0:][clear], ref["raw_ids"][:, c0:][clear])
    del model
    torch.cuda.empty_cache()


def test_greedy_generation_is_bit_repeatable_bf16_mini(hip):
    """Greedy generation on the mini_v0.1 decoder shapes must be bit-repeatable from pass to pass (single process).
    Round 1 fished for a 1-in-3e4 event with a second GPU process; its cause was an accumulator handed to an inline-asm
    `ds_write` with no MFMA hazard padding (ptts_common.h), which tools/check_mfma_asm_hazard.py now rules out at the
    ISA level in the CPU suite."""
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    _models.clear()
    torch.cuda.empty_cache()
    cfg = mini_v01_config()
    sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
    m = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
    g = torch.Generator().manual_seed(1234)
    desc = torch.randint(2, 32100, (32, 40), generator=g)
    prompt = torch.randint(2, 32100, (32, 16), generator=g)
    ref = None
    for i in range(6):
        c = m.generate(input_ids=desc, prompt_input_ids=prompt, do_sample=False, max_length=160, return_raw_ids=True)[0]
        if ref is None:
            ref = c
        assert torch.equal(c, ref), f"pass {i} differs from pass 0"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("pro,M,N,K", [(4, 32, 1024, 1024), (6, 32, 1024, 1024), (4, 32, 2048, 512), (5, 32, 1024, 4096),
                                       (5, 32, 1024, 1024), (4, 32, 3072, 1024)])
def test_gemm_instances_relaunch_bit_identical(hip, dtype, pro, M, N, K, capfd):
    """Every K-split hand-over instance the decode step uses - folded half-split (4 with N <= 2048: re-enabled this
    round), folded two-half (6 / N > 2048), plain half-split (5) - relaunched 3000 times on one stream, each result
    compared bitwise with the first on the device (ptts_test_gemm hammer mode)."""
    g = torch.Generator().manual_seed(5)
    W = (torch.randn(N, K, generator=g) * 0.05).cuda()
    X = (0.3 + torch.randn(M, K, generator=g)).cuda()
    gamma = (1 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = torch.zeros(N).cuda()
    Y = torch.empty(M, N, device="cuda")
    os.environ["PTTS_TEST_GEMM_ITERS"] = "3000"
    try:
        rc = hip.ptts_test_gemm(1 if dtype == "bf16" else 0, M, N, K, W.data_ptr(), X.data_ptr(), pro, gamma.data_ptr(),
                                beta.data_ptr(), C.c_float(1e-5), 0, bias.data_ptr(), Y.data_ptr(), None)
    finally:
        del os.environ["PTTS_TEST_GEMM_ITERS"]
    assert rc == 0, hip.ptts_last_error()
    err = capfd.readouterr().err
    assert "hammer: 3000 launches, 0 differ" in err, err


def test_prompt_hidden_states_equal_embedded_prompt_ids(hip):
    """`prompt_hidden_states` (pre-embedded prompts, M:839-840) take precedence over `prompt_input_ids` (M:2476) and
    must give exactly what the id lookup gives when they ARE the embedded ids (fp32 mode: the table is not rounded)."""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    _, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    desc, amask, prompt, pmask = case_inputs(f)
    kw = dict(encoder_outputs=(torch.from_numpy(f["enc_t5"]),), attention_mask=amask, do_sample=False,
              max_length=int(f["max_length"]), return_raw_ids=True)
    a = model.generate(prompt_input_ids=prompt, **kw)
    emb = torch.nn.functional.embedding(prompt, sd["embed_prompts.weight"].float())
    b = model.generate(prompt_hidden_states=emb, **kw)
    c = model.generate(prompt_hidden_states=emb, prompt_input_ids=torch.zeros_like(prompt), **kw)   # ids ignored
    assert torch.equal(a, b) and torch.equal(a, c)
    assert np.array_equal(a[0].cpu().numpy(), f["codes"])
    rnd = torch.randn(emb.shape, generator=torch.Generator().manual_seed(5)) * 2.0
    d = model.generate(prompt_hidden_states=rnd, **kw)               # and they are really used
    assert not torch.equal(a, d)
    with pytest.raises(ValueError):
        model.generate(prompt_hidden_states=emb[:, :, :-1], **kw)


def test_custom_logits_processors_and_stopping_criteria(hip):
    """`logits_processor` / `stopping_criteria` lists (M:2561-2573): host callables applied between the forward half and
    the selection half of every step, in the third-party loop's order.  Checked against the oracle loop with the same
    processors (fp32 mode, bit-exact ids), an identity processor, and a criterion that stops the loop early."""
    from oracle import cpu_ref

    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    _, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    desc, amask, prompt, pmask = case_inputs(f)
    L = int(f["max_length"])
    enc = torch.from_numpy(f["enc_t5"])
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=L,
              return_raw_ids=True)
    base = model.generate(**kw)
    same = model.generate(logits_processor=[lambda ids, sc: sc], **kw)
    assert torch.equal(base, same)

    banned = torch.arange(0, cfg.decoder.vocab_size, 3)

    def ban(ids, scores):                      # a NoBadWords-style processor
        scores = scores.clone()
        scores[:, banned.to(scores.device)] = float("-inf")
        return scores

    def ramp(ids, scores):                     # depends on the ids generated so far
        return scores + 0.05 * (ids[:, -1:].to(scores.device).float() % 7) * torch.linspace(-1, 1, scores.shape[1], device=scores.device)

    got = model.generate(logits_processor=[ban, ramp], **kw)
    ref = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, pmask, L, logits_processor=[ban, ramp])
    want = cpu_ref.undelay(cfg, ref["raw_ids"], ref["pattern"])
    assert torch.equal(got.cpu(), want)
    assert not bool(torch.isin(got.cpu(), banned).any())
    assert not torch.equal(got, base)

    calls = []

    def stop_after_8(ids, scores):
        calls.append(ids.shape[1])
        return ids.shape[1] >= 8

    short = model.generate(stopping_criteria=[stop_after_8], **kw)
    assert model.last_raw_ids.shape[1] == 8 and calls[-1] == 8
    assert short.shape[-1] < base.shape[-1] or base.shape[-1] == 0


def test_model_to_dtype_switches_the_engine_mode(hip):
    """`model.to(torch.bfloat16)` / `.float()` as a reference user writes them: the engine is rebuilt in the other mode; back in
    float32 the ids are the reference's again, bit for bit; float16 is refused."""
    f, meta = load_case("tiny_greedy")
    cfg, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    desc, amask, prompt, _ = case_inputs(f)
    kw = dict(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=int(f["max_length"]))
    model.generate(**kw)
    assert np.array_equal(model.last_codes[0].cpu().numpy(), f["codes"])
    assert model.to(torch.bfloat16) is model and model.dtype == "bf16"
    wav16 = model.generate(**kw)
    assert wav16.shape[0] == desc.shape[0] and bool(torch.isfinite(wav16).all())
    assert (model.last_codes[0].cpu().numpy() == f["codes"]).mean() > 0.8          # bf16: close, not bit-exact
    codec = model.audio_encoder                     # a wrapper fetched BEFORE the switch: it must follow the model
    old_engine = model._engine
    model.to("cuda:0", torch.float32)
    assert not old_engine.handle.value                                          # released NOW, not at garbage collection
    wav_c = codec.decode(torch.from_numpy(f["codes"])[None]).audio_values       # re-binds to the rebuilt engine, no freed handle
    model.generate(**kw)
    assert model.dtype == "fp32" and np.array_equal(model.last_codes[0].cpu().numpy(), f["codes"])
    assert codec._engine() is model._engine and wav_c.shape[-1] == f["codes"].shape[-1] * cfg.audio_encoder.hop_length
    with pytest.raises(NotImplementedError):
        model.half()
    del model
    torch.cuda.empty_cache()


def test_output_scores_with_return_dict_in_generate(hip):
    """`generate(output_scores=True, return_dict_in_generate=True)` (third-party loop behind M:2582 / M:2608): one [batch * codebooks,
    vocab] tensor of PROCESSED scores per step.  Greedy: every step's scores equal the reference fixture's logits (2e-4) and their
    argmax are the ids of the run; with `min_length` the EOS column is -inf below the bound.  Sampling: step 0's scores are the
    reference logits after temperature / top-k (the -inf pattern exactly, finite entries 2e-4); later steps follow the engine's own
    Philox draws, so only their shape and top-k count are checked."""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    L, K = int(f["max_length"]), cfg.decoder.num_codebooks
    kw = dict(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, max_length=L, return_dict_in_generate=True)
    out = model.generate(**kw, do_sample=False, output_scores=True)
    assert isinstance(out.scores, tuple) and len(out.scores) == L - 1 and out.sequences.dim() == 2
    got = torch.stack([x.cpu() for x in out.scores])
    assert got.shape == f["step_logits"].shape and np.abs(got.numpy() - f["step_logits"]).max() < 2e-4
    assert np.array_equal(model.last_codes[0].cpu().numpy(), f["codes"])
    free = free_positions(f["pattern"], L - 1)
    assert np.array_equal(got.argmax(-1).numpy()[free], f["raw_ids"][:, 1:].T[free])
    assert model.generate(**kw, do_sample=False).scores is None                              # not asked for: None, the fast path
    eos = cfg.decoder.eos_token_id
    out = model.generate(**kw, do_sample=False, output_scores=True, min_length=8)
    col = torch.stack([x.cpu() for x in out.scores])[:, :, eos]
    assert bool(torch.isinf(col[:7]).all()) and bool(torch.isfinite(col[7:]).all())     # columns 1 .. 7 filled below the bound
    # sampling: the fixture's step-0 logits through the warpers
    g, gmeta = load_case("tiny_sample")
    cfg2, model2 = model_for("tiny_sample", gmeta, "fp32")
    d2, a2, p2, _ = case_inputs(g)
    out = model2.generate(input_ids=d2, attention_mask=a2, prompt_input_ids=p2, max_length=int(g["max_length"]), do_sample=True,
                          temperature=0.7, top_k=50, seed=5, return_dict_in_generate=True, output_scores=True)
    s0 = out.scores[0].cpu()
    ref = torch.from_numpy(g["step_logits"][0]) / 0.7
    thr = torch.topk(ref, 50)[0][..., -1, None]
    keep = ref >= thr
    clear = (ref - thr).abs() > 1e-3                      # an entry within 1e-3 of the threshold may fall on either side
    assert torch.equal(torch.isfinite(s0)[clear], keep[clear])
    both = keep & torch.isfinite(s0)
    assert (s0[both] - ref[both]).abs().max() < 3e-4
    assert all(tuple(x.shape) == tuple(s0.shape) and int(torch.isfinite(x).sum(-1).min()) >= 50 for x in out.scores)


def test_lookahead_cross_q_agrees_with_separate_projection(hip):
    """Look-ahead cross-q (x.Wq' + a.(Wq'.Wo) inside the out-projection launch, LayerNorm applied by the cross-attention
    kernel from per-block statistics) vs the separate folded projection: same model, same teacher-forced tokens."""
    for name in ("tiny_greedy", "tiny_greedy_ragged", "tiny_cfg"):
        a = _fresh_engine_logits(name, "bf16", {"PTTS_LOOKAHEAD": "1"})
        b = _fresh_engine_logits(name, "bf16", {"PTTS_LOOKAHEAD": "0"})
        assert not torch.isnan(a).any()
        assert (a - b).abs().max().item() < 0.05, name
        assert (a.argmax(-1) == b.argmax(-1)).float().mean().item() > 0.95, name
    a = _fresh_engine_logits("tiny_greedy", "fp32", {"PTTS_LOOKAHEAD": "1", "PTTS_LN_FOLD": "1"})
    b = _fresh_engine_logits("tiny_greedy", "fp32", {"PTTS_LOOKAHEAD": "0", "PTTS_LN_FOLD": "1"})
    assert (a - b).abs().max().item() < 1e-4


def test_lookahead_qkv_agrees_with_separate_projection(hip):
    """Look-ahead QKV (layer l+1's x.Wqkv' + g.(Wqkv'.W2) inside layer l's fc2 launch; the self-attention kernel finishes
    the LayerNorm from per-block statistics, scales q and appends k / v) vs the separate folded QKV launch: same model,
    same teacher-forced tokens; ragged masks and guidance rows included; the fp32 engine with the fold on as a tight check."""
    for name in ("tiny_greedy", "tiny_greedy_ragged", "tiny_cfg"):
        a = _fresh_engine_logits(name, "bf16", {"PTTS_LOOKAHEAD2": "1"})
        b = _fresh_engine_logits(name, "bf16", {"PTTS_LOOKAHEAD2": "0"})
        assert not torch.isnan(a).any()
        assert (a - b).abs().max().item() < 0.05, name
        assert (a.argmax(-1) == b.argmax(-1)).float().mean().item() > 0.95, name
    a = _fresh_engine_logits("tiny_greedy", "fp32", {"PTTS_LOOKAHEAD2": "1", "PTTS_LN_FOLD": "1"})
    b = _fresh_engine_logits("tiny_greedy", "fp32", {"PTTS_LOOKAHEAD2": "0", "PTTS_LN_FOLD": "1"})
    assert (a - b).abs().max().item() < 1e-4


def test_decoder_input_ids_prefix(hip):
    """User `decoder_input_ids` (M:2150-2195): the prefix is delay-shifted by build_delay_pattern_mask (M:108-111), passes
    through the cache teacher-forced, and generation continues after it.  fp32 engine vs the oracle loop, bit-exact."""
    from oracle import cpu_ref

    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    _, sd = state_dict_for("tiny_greedy", meta, parts=("text_encoder", "decoder", "audio_encoder"))
    desc, amask, prompt, pmask = case_inputs(f)
    K = cfg.decoder.num_codebooks
    B = desc.shape[0]
    L = int(f["max_length"])
    enc = torch.from_numpy(f["enc_t5"])
    g = torch.Generator().manual_seed(11)
    prefix = torch.randint(0, cfg.decoder.pad_token_id, (B * K, 5), generator=g)      # plain codec ids
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=L,
              return_raw_ids=True)
    got = model.generate(decoder_input_ids=prefix, **kw)
    ref = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, pmask, L, decoder_input_ids=prefix)
    assert torch.equal(model.last_raw_ids.cpu()[:, 6:], ref["raw_ids"][:, 6:])          # sampled columns (prefix = 1 + 5)
    want = cpu_ref.undelay(cfg, ref["raw_ids"], ref["pattern"])
    assert torch.equal(got.cpu(), want)
    # a prefix that already starts with the start token is taken as is
    with_start = torch.cat([torch.full((B * K, 1), cfg.decoder.bos_token_id), prefix], dim=-1)
    again = model.generate(decoder_input_ids=with_start, **kw)
    assert torch.equal(again, got)
    base = model.generate(**kw)
    assert not torch.equal(base, got)
    with pytest.raises(ValueError):
        model.generate(decoder_input_ids=prefix[:-1], **kw)


@pytest.mark.parametrize("name", ["tiny_prefix", "tiny_prefix_masked"])
def test_decoder_prefix_fixtures_from_the_reference(hip, name):
    """Fixtures written by the REAL reference forward with `decoder_input_ids` and, in the masked case, a left-padded
    `decoder_attention_mask` + ragged prompt / description masks (M:845-846, M:2150-2195): fp32 engine ids bit-exact."""
    f, meta = load_case(name)
    cfg, model = model_for(name, meta, "fp32")
    desc, amask, prompt, pmask = case_inputs(f)
    prefix = torch.from_numpy(f["decoder_input_ids"])
    dmask = torch.from_numpy(f["decoder_attention_mask"]) if "decoder_attention_mask" in f else None
    enc = torch.from_numpy(f["enc_t5"])
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, prompt_attention_mask=pmask,
              do_sample=False, max_length=int(f["max_length"]), return_raw_ids=True, decoder_input_ids=prefix)
    if dmask is not None:
        kw["decoder_attention_mask"] = dmask
    got = model.generate(**kw)
    start_len = prefix.shape[1] + 1
    free = torch.from_numpy(f["pattern"][:, start_len:] == -1)
    raw = model.last_raw_ids.cpu()
    assert torch.equal(raw[:, start_len:][free], torch.from_numpy(f["raw_ids"])[:, start_len:][free])
    assert torch.equal(got.cpu()[0], torch.from_numpy(f["codes"]))
    if dmask is not None:
        # the mask matters: without it the padded prefix columns are attended and the ids change
        kw.pop("decoder_attention_mask")
        other = model.generate(**kw)
        assert not torch.equal(other, got)
        # a decoder mask next to a prompt needs the prompt mask too (the reference cannot build its causal mask otherwise)
        with pytest.raises(ValueError):
            model.generate(**{**kw, "decoder_attention_mask": dmask, "prompt_attention_mask": None})
        # under classifier-free guidance the mask is repeated with the rows (M:2112-2113): runs and stays finite
        g = model.generate(**{**kw, "decoder_attention_mask": dmask, "guidance_scale": 2.0})
        assert g.shape == got.shape


def test_decoder_only_driver(hip):
    """ParlerTTSForCausalLM.generate (M:1281-1552) against the fixture the REAL class produced: encoder and prompt hidden
    states handed in directly (ragged masks), decoder-only state dict (model.decoder.*, lm_heads.*), ids bit-exact."""
    from parler_tts_amd.modeling import ParlerTTSForCausalLM

    f, meta = load_case("tiny_causal")
    cfg, sd = state_dict_for("tiny_causal", meta)
    dec_sd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    assert any(k.startswith("lm_heads.") for k in dec_sd) and any(k.startswith("model.decoder.layers.") for k in dec_sd)
    _models.clear()
    model = ParlerTTSForCausalLM(cfg.decoder, dec_sd, dtype="fp32", device="cuda:0")
    t = lambda k: torch.from_numpy(f[k])
    kw = dict(encoder_hidden_states=t("enc_h"), encoder_attention_mask=t("emask"), prompt_hidden_states=t("prompt_h"),
              prompt_attention_mask=t("pmask"), do_sample=False, max_length=int(f["max_length"]))
    codes = model.generate(**kw)
    assert torch.equal(codes.cpu(), t("codes"))
    free = torch.from_numpy(f["pattern"][:, 1:] == -1)
    assert torch.equal(model.last_raw_ids.cpu()[:, 1:][free], t("raw_ids")[:, 1:][free])
    # explicit BOS column == default start; errors of the driver
    K = cfg.decoder.num_codebooks
    again = model.generate(torch.full((3 * K, 1), cfg.decoder.bos_token_id), **kw)
    assert torch.equal(again, codes)
    with pytest.raises(ValueError):           # guidance: the caller must pass encoder states with 2 x batch rows (M:1218-1232)
        model.generate(guidance_scale=3.0, **kw)
    with pytest.raises(ValueError):
        model.generate(num_beams=2, **kw)
    # guidance_scale > 1 and no encoder states at all: fixtures from the real class
    for name in ("tiny_causal_cfg", "tiny_causal_noenc"):
        g, gmeta = load_case(name)
        u = lambda k: torch.from_numpy(g[k]) if k in g else None
        kw2 = dict(prompt_hidden_states=u("prompt_h"), prompt_attention_mask=u("pmask"), do_sample=False, max_length=int(g["max_length"]))
        if u("enc_h") is not None:
            kw2.update(encoder_hidden_states=u("enc_h"), encoder_attention_mask=u("emask"))
        if gmeta["guidance_scale"]:
            kw2["guidance_scale"] = gmeta["guidance_scale"]
        codes2 = model.generate(**kw2)
        assert torch.equal(codes2.cpu(), u("codes")), name
        free2 = torch.from_numpy(g["pattern"][:, 1:] == -1)
        assert torch.equal(model.last_raw_ids.cpu()[:, 1:][free2], u("raw_ids")[:, 1:][free2]), name


def test_lookahead_cross_q_two_m_tiles_bf16(hip):
    """Look-ahead cross-q with more than one 32-row m-tile (batch 40; the same layout guidance gives at batch 32):
    teacher-forced logits of the mini_v0.1 decoder, look-ahead on vs off."""
    from parler_tts_amd import _hip
    from parler_tts_amd.configuration import mini_v01_config
    from parler_tts_amd.delay_pattern import build_delay_pattern_mask
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration
    from parler_tts_amd.weights import synthetic_state_dict

    cfg = mini_v01_config()
    cfg.decoder.num_hidden_layers = 4            # same widths, fewer layers
    sd = synthetic_state_dict(cfg, seed=0, never_eos=True, parts=("text_encoder", "decoder"))
    B, T, P, steps = 40, 12, 6, 6
    L = steps + 2
    K = cfg.decoder.num_codebooks
    g = torch.Generator().manual_seed(7)
    desc = torch.randint(2, 32100, (B, T), generator=g)
    prompt = torch.randint(2, 32100, (B, P), generator=g)
    forced = torch.randint(0, 1024, (steps, B * K), generator=g)
    _models.clear()
    torch.cuda.empty_cache()

    def run(flag):
        old = os.environ.get("PTTS_LOOKAHEAD")
        os.environ["PTTS_LOOKAHEAD"] = flag
        try:
            m = ParlerTTSForConditionalGeneration(cfg, sd, dtype="bf16", device="cuda:0")
            eng = m.engine(B, T, P, L)
            enc = eng.t5_encode(desc.cuda(), torch.ones(B, T, dtype=torch.long, device="cuda"))
            ids = torch.full((B * K, 1), cfg.decoder.bos_token_id, dtype=torch.long)
            ids, pattern = build_delay_pattern_mask(ids, cfg.decoder.bos_token_id, cfg.decoder.pad_token_id, L, K)
            pattern = pattern.cuda().contiguous()
            raw = torch.full((B * K, L), cfg.decoder.pad_token_id, dtype=torch.long, device="cuda")
            raw[:, :1] = ids.cuda()
            eng.prefill(B, B, enc.contiguous(), torch.ones(B, T, dtype=torch.int32, device="cuda"), prompt.cuda().contiguous(),
                        None, pattern, raw, 1)
            sp = _hip.PttsSampling(0, 1.0, 0, 1.0, 0)
            out = []
            for i in range(steps):
                eng.set_forced_tokens(forced[i])
                eng.decode_steps(1, sp)
                out.append(eng.last_logits(B * K).cpu())
            eng.set_forced_tokens(None)
            return torch.stack(out)
        finally:
            if old is None:
                os.environ.pop("PTTS_LOOKAHEAD", None)
            else:
                os.environ["PTTS_LOOKAHEAD"] = old

    a, b = run("1"), run("0")
    assert not torch.isnan(a).any()
    assert (a - b).abs().max().item() < 0.05
    assert (a.argmax(-1) == b.argmax(-1)).float().mean().item() > 0.97


# ------------------------------------------------------------------------------------------------
# generation-config semantics added in round 2
# ------------------------------------------------------------------------------------------------
def test_num_return_sequences_reproduces_the_reference_expansion(hip):
    """M:2600-2606: `_expand_inputs_for_generation` interleaves the id rows, the kwargs and the stashed delay pattern.  The
    engine is teacher-forced (device-side) along the REFERENCE's sampled ids of the fixture, so generate() must return the
    reference's [B, K, n * F] codes and its raw ids, and the warped probabilities of the last step must match."""
    from oracle import cpu_ref

    f, meta = load_case("tiny_sample_nrs2")
    cfg, model = model_for("tiny_sample_nrs2", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    L, K, n = int(f["max_length"]), cfg.decoder.num_codebooks, 2
    B = desc.shape[0]
    eng = model.engine(B * n, desc.shape[1], prompt.shape[1], L)
    forced = torch.from_numpy(f["raw_ids"]).to("cuda:0")
    assert forced.shape == (B * K * n, L)
    eng.set_forced_sequence(forced)
    try:
        codes = model.generate(encoder_outputs=(torch.from_numpy(f["enc_t5"]),), attention_mask=amask, prompt_input_ids=prompt,
                               do_sample=True, temperature=0.7, top_k=50, num_return_sequences=n, max_length=L, seed=5,
                               return_raw_ids=True)
        assert model.engine(B * n, desc.shape[1], prompt.shape[1], L) is eng
        probs = eng.last_logits(B * K * n, probs=True).cpu()
    finally:
        eng.set_forced_sequence(None)
    assert torch.equal(model.last_raw_ids.cpu(), torch.from_numpy(f["raw_ids"]))
    assert codes.shape == (1, B, K, n * (L - K)) and np.array_equal(codes[0].cpu().numpy(), f["codes"])
    ref = torch.softmax(cpu_ref.warp_logits(torch.from_numpy(f["step_logits"][-1]), 0.7, 50), dim=-1)
    assert (probs - ref).abs().max() < 1e-5
    with pytest.raises(ValueError):        # greedy keeps the reference's refusal (M:2575-2579)
        model.generate(input_ids=desc, prompt_input_ids=prompt, do_sample=False, num_return_sequences=2, max_length=L)


def test_min_length_and_min_new_tokens_suppress_eos(hip):
    """MinLength / MinNewTokensLength processors of the third-party `_get_logits_processor` (called at M:2561-2567):
    score[eos] = -inf while the id matrix is shorter than the bound.  A tiny model whose codebook-0 head prefers EOS from
    step 6 on (its EOS row is the row of the token the free run picks there, times 1.05): without a bound utterance 0 ends
    early; with one the engine must follow the oracle run that carries the same processor."""
    from oracle import cpu_ref
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    f, meta = load_case("tiny_eos_free")
    cfg, sd0 = state_dict_for("tiny_eos_free", meta)
    desc, amask, prompt, _ = case_inputs(f)
    L, eos, K = int(f["max_length"]), cfg.decoder.eos_token_id, cfg.decoder.num_codebooks
    enc = torch.from_numpy(f["enc_t5"])
    col = 7                                               # a free column of codebook 0 (its BOS run ends at column 0)
    tok = int(f["raw_ids"][0, col])
    sd = dict(sd0)
    h0 = sd["decoder.lm_heads.0.weight"].clone()
    h0[eos] = h0[tok] * 1.05
    sd["decoder.lm_heads.0.weight"] = h0
    plain = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, None, L)["raw_ids"]
    assert bool((plain[0, 1:col + 1] == eos).any())      # the modified head does emit EOS early
    bound = col + 6

    def min_len_proc(ids, scores):
        if ids.shape[-1] < bound:
            scores = scores.clone()
            scores[:, eos] = -float("inf")
        return scores

    want = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, None, L, logits_processor=[min_len_proc])["raw_ids"]
    assert not bool((want[:, 1:bound] == eos).any()) and not torch.equal(want, plain)
    _models.clear()
    torch.cuda.empty_cache()
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    kw0 = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=L, return_raw_ids=True)
    model.generate(**kw0)
    assert torch.equal(model.last_raw_ids.cpu(), plain)
    for kw in (dict(min_length=bound), dict(min_new_tokens=bound - 1)):                 # start_len = 1
        model.generate(**kw0, **kw)
        got = model.last_raw_ids.cpu()
        assert got.shape == want.shape and torch.equal(got, want), kw
    with pytest.raises(NotImplementedError):       # the reference's processor order makes the guidance combine NaN there
        model.generate(**{**kw0, "guidance_scale": 3.0, "min_length": bound})
    del model
    torch.cuda.empty_cache()


def test_eos_terminated_call_stops_within_the_advertised_poll_window(hip):
    """VERDICT r4 item 6.  The third-party loop tests `unfinished_sequences.max() == 0` after EVERY step (M:2582 / M:2608); here the
    loop state lives on the device and generate() reads it through pipelined snapshots every `eos_poll_steps` steps
    (ptts_state_snapshot / ptts_state_poll: chunk i + 1 is queued before chunk i's snapshot is awaited), so a call that ends on
    EOS runs at most 2 * eos_poll_steps - 1 steps past its last column (rounds 1-4: a blocking read every 128 steps).  A tiny model
    whose every head prefers EOS as soon as `min_length` allows it: the ids must equal the oracle run with the same processor, for
    every poll interval, and the device's step counter must stay inside the window."""
    from oracle import cpu_ref
    from parler_tts_amd.modeling import ParlerTTSForConditionalGeneration

    f, meta = load_case("tiny_eos_free")
    cfg, sd0 = state_dict_for("tiny_eos_free", meta)
    desc, amask, prompt, _ = case_inputs(f)
    eos, K, D = cfg.decoder.eos_token_id, cfg.decoder.num_codebooks, cfg.decoder.hidden_size
    enc = torch.from_numpy(f["enc_t5"])
    sd = dict(sd0)
    sd["decoder.model.decoder.layer_norm.bias"] = sd["decoder.model.decoder.layer_norm.bias"] + 1.0
    for k in range(K):                                   # logit[eos] ~ sum(final LayerNorm output) ~ D: far above every other logit
        h = sd[f"decoder.lm_heads.{k}.weight"].clone()
        h[eos] = 1.0
        sd[f"decoder.lm_heads.{k}.weight"] = h
    L, bound = 200, 24

    def min_len_proc(ids, scores):
        if ids.shape[-1] < bound:
            scores = scores.clone()
            scores[:, eos] = -float("inf")
        return scores

    want = cpu_ref.generate_ids(cfg, sd, enc, amask, prompt, None, L, logits_processor=[min_len_proc])["raw_ids"]
    done_len = want.shape[1]
    assert done_len == bound + 1 and bool((want[:, -1] == eos).all())          # every row ends with the first allowed EOS
    _models.clear()
    torch.cuda.empty_cache()
    model = ParlerTTSForConditionalGeneration(cfg, sd, dtype="fp32", device="cuda:0")
    kw = dict(encoder_outputs=(enc,), attention_mask=amask, prompt_input_ids=prompt, do_sample=False, max_length=L,
              min_length=bound, return_raw_ids=True)
    assert model.eos_poll_steps == 16
    for poll in (16, 4, 1, 128):
        model.eos_poll_steps = poll
        model.generate(**kw)
        got = model.last_raw_ids.cpu()
        assert got.shape == want.shape and torch.equal(got, want), poll
        st = model.engine().state()
        steps_needed = done_len - 1
        assert st.done_len == done_len and steps_needed <= st.step_count <= min(steps_needed + 2 * poll - 1, L - 1), (poll, st.step_count)
    # the blocking read is unchanged and a slot with no snapshot pending is an error, not a stale answer
    eng = model.engine()
    eng.state_snapshot(2)
    assert eng.state_poll(2).done_len == done_len
    with pytest.raises(Exception):
        eng.state_poll(2)
    del model
    torch.cuda.empty_cache()


def test_unseeded_sampling_consumes_the_global_generator(hip):
    """do_sample without `seed`: like torch.multinomial in the reference, every call draws from torch's global CPU generator,
    so two calls differ and torch.manual_seed reproduces; an explicit seed stays deterministic."""
    f, meta = load_case("tiny_sample")
    cfg, model = model_for("tiny_sample", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    kw = dict(encoder_outputs=(torch.from_numpy(f["enc_t5"]),), attention_mask=amask, prompt_input_ids=prompt, do_sample=True,
              top_k=50, temperature=1.0, max_length=int(f["max_length"]), return_raw_ids=True)
    torch.manual_seed(1234)
    a = model.generate(**kw).clone()
    b = model.generate(**kw).clone()
    assert not torch.equal(a, b)
    torch.manual_seed(1234)
    assert torch.equal(model.generate(**kw), a)
    assert torch.equal(model.generate(seed=3, **kw), model.generate(seed=3, **kw))


def test_model_on_its_own_device_while_another_is_current(hip):
    """generate() and the codec wrappers select the model's device themselves (streams, lazy workspaces and launches follow
    the CURRENT device): with one GPU this checks that the guard is in place and harmless."""
    f, meta = load_case("tiny_greedy")
    cfg, model = model_for("tiny_greedy", meta, "fp32")
    desc, amask, prompt, _ = case_inputs(f)
    codes = model.generate(input_ids=desc, attention_mask=amask, prompt_input_ids=prompt, do_sample=False,
                           max_length=int(f["max_length"]), return_raw_ids=True)
    assert np.array_equal(codes[0].cpu().numpy(), f["codes"])
    assert torch.cuda.current_device() == 0
