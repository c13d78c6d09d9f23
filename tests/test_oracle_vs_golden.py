"""Pins oracle/cpu_ref.py against fixtures produced by the REAL reference code (oracle/gen_golden.py).

CPU only.  The decoder / delay-pattern / un-delay / codec restatements use the same stock torch ops in
the same order as the reference, so on the machine that generated the fixtures they are bitwise equal;
the tolerances below only absorb a different CPU's GEMM blocking.
"""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from parler_tts_amd.delay_pattern import apply_delay_pattern_mask, build_delay_pattern_mask, frames_kept
from parler_tts_amd.weights import sinusoid_table, synthetic_state_dict
from tests.helpers import case_inputs, config_for, forced_tokens, free_positions, load_case, state_dict_for

GREEDY = ["tiny_greedy", "tiny_greedy_ragged", "tiny_eos_free", "tiny_eos_forced", "tiny_cfg",
          "mini_greedy", "mini_greedy_ragged"]


def test_delay_masks_bit_exact():
    f, _ = load_case("delay_masks")
    for i in range(int(f["n_cases"])):
        B, K, L, s = (int(f[f"c{i}_{x}"]) for x in "BKLs")
        cut, pat = build_delay_pattern_mask(torch.from_numpy(f[f"c{i}_ids"]), 1025, 1024, L, K)
        assert np.array_equal(cut.numpy(), f[f"c{i}_cut"]), i
        assert np.array_equal(pat.numpy(), f[f"c{i}_pattern"]), i
        # apply: same random ids are not stored, so check the defining property on the stored result
        applied = f[f"c{i}_applied"]
        n = applied.shape[1]
        forced = f[f"c{i}_pattern"][:, :n] != -1
        assert np.array_equal(applied[forced], f[f"c{i}_pattern"][:, :n][forced])


def test_delay_mask_docstring_example():
    # the 4-codebook / length-8 picture printed at M:84-87
    ids = torch.full((4, 1), 7)
    _, pat = build_delay_pattern_mask(ids, 7, 9, 8, 4)
    want = [[7, -1, -1, -1, -1, 9, 9, 9], [7, 7, -1, -1, -1, -1, 9, 9], [7, 7, 7, -1, -1, -1, -1, 9],
            [7, 7, 7, 7, -1, -1, -1, -1]]
    assert pat.tolist() == want
    x = torch.arange(32).reshape(4, 8)
    y = apply_delay_pattern_mask(x[:, :5], pat)
    assert y[0].tolist() == [7, 1, 2, 3, 4] and y[3].tolist() == [7, 7, 7, 7, 28]
    assert frames_kept(8, 4) == 4 and frames_kept(6, 4) == 6


def test_delay_pattern_round_trip_property():
    """Size-independent property used at sizes no fixture covers: frames [B, K, F] written through the delay pattern
    (codebook k shifted by k columns behind its BOS run) and read back by the un-delay of M:2633-2648 are the same frames,
    for random shapes including the BASELINE length (K = 9, L = 2580)."""
    g = torch.Generator().manual_seed(3)
    for (B, K, L) in [(1, 4, 8), (2, 9, 17), (3, 9, 40), (2, 2, 3), (1, 9, 2580), (4, 5, 33)]:
        bos, pad = 1025, 1024
        start = torch.full((B * K, 1), bos, dtype=torch.long)
        _, pat = build_delay_pattern_mask(start, bos, pad, L, K)
        F = frames_kept(L, K)
        frames = torch.randint(0, 1024, (B, K, F), generator=g)
        raw = torch.randint(0, 1024, (B * K, L), generator=g)          # whatever the loop sampled where the pattern overrides
        free = pat == -1
        assert int(free.sum()) == B * K * F
        raw[free] = frames.reshape(B * K, F).reshape(-1)
        out = apply_delay_pattern_mask(raw, pat)
        keep = (pat != bos) & (pat != pad)
        back = out[keep].reshape(B, K, -1)
        assert torch.equal(back, frames)
        # row k is free exactly on columns k+1 .. k+F
        for k in range(K):
            cols = torch.nonzero(free[k]).flatten()
            assert cols[0].item() == k + 1 and cols[-1].item() == k + F and len(cols) == F


def test_delay_mask_prompt_too_long_raises():
    with pytest.raises(RuntimeError):
        build_delay_pattern_mask(torch.zeros(4, 7, dtype=torch.long), 7, 9, 8, 4)


def test_sinusoid_table_matches_reference_formula():
    f, _ = load_case("sinusoid")
    ours = sinusoid_table(96, 128).numpy()
    assert np.abs(ours - f["table"]).max() < 1e-5  # fp32 exp()/product rounding of the reference build


@pytest.mark.parametrize("name", GREEDY)
def test_decoder_loop_matches_reference(name):
    f, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    gen = {k: v for k, v in meta["gen"].items() if k != "forced_tokens"}
    n_rec = f["step_logits"].shape[0]
    r = cpu_ref.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                             record_logits=n_rec, forced_tokens=forced_tokens(f), **gen)
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5
    assert np.array_equal(r["pattern"].numpy(), f["pattern"])
    free = free_positions(f["pattern"], r["raw_ids"].shape[1] - 1).T
    assert r["raw_ids"].shape == f["raw_ids"].shape
    assert np.array_equal(r["raw_ids"].numpy()[:, 1:][free], f["raw_ids"][:, 1:][free])
    codes = cpu_ref.undelay(cfg, r["raw_ids"], r["pattern"])[0]
    assert np.array_equal(codes.numpy(), f["codes"])
    assert codes.shape[-1] == frames_kept(r["raw_ids"].shape[1], cfg.decoder.num_codebooks)


@pytest.mark.parametrize("name", ["tiny_prefix", "tiny_prefix_masked"])
def test_decoder_prefix_and_decoder_attention_mask_match_reference(name):
    """`decoder_input_ids` (M:2150-2195) and `decoder_attention_mask` (M:845-846; left-padded prefix rows) through the real
    reference forward: the restated loop reproduces its logits and ids."""
    f, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    prefix = torch.from_numpy(f["decoder_input_ids"])
    dmask = torch.from_numpy(f["decoder_attention_mask"]) if "decoder_attention_mask" in f else None
    assert (dmask is not None) == name.endswith("masked")
    r = cpu_ref.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                             record_logits=f["step_logits"].shape[0], decoder_input_ids=prefix, decoder_attention_mask=dmask)
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5
    assert np.array_equal(r["pattern"].numpy(), f["pattern"])
    assert r["raw_ids"].shape == f["raw_ids"].shape
    start_len = prefix.shape[1] + 1
    free = f["pattern"][:, start_len:] == -1
    assert np.array_equal(r["raw_ids"].numpy()[:, start_len:][free], f["raw_ids"][:, start_len:][free])
    # every kept argmax of the reference run is far from a tie (step j fills column start_len + j)
    m = f["margins"]
    assert float(m[free.T[: m.shape[0]]].min()) >= 1e-3
    assert np.array_equal(cpu_ref.undelay(cfg, r["raw_ids"], r["pattern"])[0].numpy(), f["codes"])


def test_decoder_only_driver_matches_reference():
    """ParlerTTSForCausalLM.generate (M:1281-1552) on the real class: encoder / prompt hidden states handed in directly."""
    f, meta = load_case("tiny_causal")
    cfg, sd = state_dict_for("tiny_causal", meta)
    t = lambda k: torch.from_numpy(f[k])
    r = cpu_ref.generate_ids(cfg, sd, t("enc_h"), t("emask"), None, t("pmask"), int(f["max_length"]), record_logits=64,
                             prompt_hidden_states=t("prompt_h"), decoder_only=True)
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5
    assert np.array_equal(r["raw_ids"].numpy(), f["raw_ids"])
    assert np.array_equal(cpu_ref.undelay(cfg, r["raw_ids"], r["pattern"])[0].numpy(), f["codes"])
    assert float(f["free_min_margin"]) >= 1e-3


@pytest.mark.parametrize("name", ["tiny_causal_cfg", "tiny_causal_noenc"])
def test_decoder_only_driver_guidance_and_no_encoder_match_reference(name):
    """The two call shapes of ParlerTTSForCausalLM.generate round 1 refused, on the real class: guidance_scale > 1 (the driver
    doubles ids and prompt - zeroed prompt, zero prompt mask - and the caller the encoder states, M:1218-1232) and no
    encoder_hidden_states (layers without their cross-attention block, M:465)."""
    f, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    t = lambda k: torch.from_numpy(f[k]) if k in f else None
    r = cpu_ref.generate_ids(cfg, sd, t("enc_h"), t("emask"), None, t("pmask"), int(f["max_length"]), record_logits=64,
                             prompt_hidden_states=t("prompt_h"), decoder_only=True, guidance_scale=meta["guidance_scale"])
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5
    assert np.array_equal(r["raw_ids"].numpy(), f["raw_ids"])
    assert np.array_equal(cpu_ref.undelay(cfg, r["raw_ids"], r["pattern"])[0].numpy(), f["codes"])
    assert float(f["free_min_margin"]) >= 1e-3


def test_static_kv_equals_cat_kv():
    f, meta = load_case("tiny_greedy")
    cfg, sd = state_dict_for("tiny_greedy", meta)
    desc, amask, prompt, pmask = case_inputs(f)
    r = cpu_ref.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                             kv_mode="static", record_logits=64)
    assert np.array_equal(r["raw_ids"].numpy(), f["raw_ids"])
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5


def test_eos_latching_quirk_is_in_the_fixture():
    # utterance 1 / codebook 3 was forced to emit EOS at step 1 while still inside its BOS region:
    # the row stays PAD forever in the raw ids (SURVEY.md 8a note (i))
    f, _ = load_case("tiny_eos_forced")
    K = 4
    row = 1 * K + 3
    assert (f["raw_ids"][row, 2:] == 64).all()
    row0 = 2 * K + 0
    assert (f["raw_ids"][row0, 2:] == 64).all()


def test_sampling_path_matches_reference():
    f, meta = load_case("tiny_sample")
    cfg, sd = state_dict_for("tiny_sample", meta)
    desc, amask, prompt, pmask = case_inputs(f)
    g = torch.Generator().manual_seed(2024)
    r = cpu_ref.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                             do_sample=True, temperature=0.7, top_k=50, generator=g, record_logits=64)
    assert np.array_equal(r["raw_ids"].numpy(), f["raw_ids"])
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5


def test_num_return_sequences_matches_reference():
    """M:2600-2606: the reference's own expansion (third-party `_expand_inputs_for_generation`) interleaves the id rows and
    the stashed delay pattern; the restatement reproduces its sampled ids, logits and the [B, K, n * F] result."""
    f, meta = load_case("tiny_sample_nrs2")
    cfg, sd = state_dict_for("tiny_sample_nrs2", meta)
    desc, amask, prompt, pmask = case_inputs(f)
    g = torch.Generator().manual_seed(2025)
    r = cpu_ref.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                             do_sample=True, temperature=0.7, top_k=50, generator=g, record_logits=64, num_return_sequences=2)
    assert r["raw_ids"].shape == f["raw_ids"].shape == (2 * cfg.decoder.num_codebooks * 2, int(f["max_length"]))
    assert np.array_equal(r["pattern"].numpy(), f["pattern"])
    assert np.array_equal(r["raw_ids"].numpy(), f["raw_ids"])
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5
    codes = cpu_ref.undelay(cfg, r["raw_ids"], r["pattern"], num_return_sequences=2)[0]
    assert codes.shape == f["codes"].shape and codes.shape[-1] == 2 * frames_kept(int(f["max_length"]), cfg.decoder.num_codebooks)
    assert np.array_equal(codes.numpy(), f["codes"])


@pytest.mark.parametrize("name", ["tinylong_greedy", "minilong_greedy", "minilong2580_greedy"])
def test_long_horizon_fixtures_pin_the_oracle(name):
    """Depth: the reference run over 2580 (tiny shape) / 512 (mini_v0.1 shape) columns.  The fp32 restatement, teacher-
    forced along the reference's raw ids, reproduces the sampled-step logits (incl. the last 8 steps, S > 2048 on tiny) and
    every argmax whose top-2 margin clears fp32 re-association noise.  The mini case is minutes of CPU: tiny runs here,
    mini only its static checks (the same comparison runs on the GPU box against the HIP engine)."""
    f, meta = load_case(name)
    L = int(f["max_length"])
    steps = f["logit_steps"]
    assert f["raw_ids"].shape[1] == L and f["margins"].shape[0] == L - 1 and f["step_logits"].shape[0] == len(steps)
    assert steps[-1] == L - 2 and (np.diff(steps) > 0).all()
    free = free_positions(f["pattern"], L - 1)
    # how often the reference's own argmax is closer than 1e-4 to a tie over the whole horizon (documented, not a bound)
    close = (f["margins"][free] < 1e-4).mean()
    assert close < 0.01
    if name == "minilong2580_greedy":        # the benchmark's horizon: a logit sample at EVERY KV page the run crosses
        pages = {int(s_) for s_ in range(58, L - 1, 64)}
        assert pages <= set(int(x) for x in steps) and {1024, 2048, L - 2} <= set(int(x) for x in steps)
        assert f["raw_ids"].shape == (2 * 9, 2580)
    if name.startswith("mini"):
        return
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    from oracle import cpu_ref_bf16

    gold = torch.from_numpy(f["raw_ids"])
    r = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, L, exact=True, forced=gold,
                                  record=set(int(s) for s in steps))
    assert np.abs(r["step_logits"].numpy() - f["step_logits"]).max() < 2e-5
    own = r["own"].numpy()                               # [steps, rows]
    ok = free & (f["margins"] >= 1e-4)
    assert np.array_equal(own[ok], f["raw_ids"][:, 1:].T[ok])


@pytest.mark.parametrize("name", ["tiny_greedy", "tiny_greedy_ragged", "tiny_cfg", "tiny_eos_free"])
def test_engine_numerics_restatement_is_the_reference_when_rounding_is_off(name):
    """oracle/cpu_ref_bf16.py is the decoder in the engine's bf16 mode (bf16 rounding at the engine's rounding points,
    folded LayerNorm, look-ahead query).  With `exact=True` every rounding is the identity: it must then reproduce the
    reference fixtures like cpu_ref does, which pins everything in it except where the roundings sit."""
    from oracle import cpu_ref_bf16

    f, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    gs = meta["gen"].get("guidance_scale")
    r = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                                  exact=True, record=64, guidance_scale=gs)
    n = min(r["step_logits"].shape[0], f["step_logits"].shape[0])
    assert np.abs(r["step_logits"].numpy()[:n] - f["step_logits"][:n]).max() < 2e-5
    free = free_positions(f["pattern"], r["raw_ids"].shape[1] - 1).T
    assert np.array_equal(r["raw_ids"].numpy()[:, 1:][free], f["raw_ids"][:, 1:][free])
    # and with the roundings on, the same model stays within bf16 noise of it (sanity of the rounding points)
    rb = cpu_ref_bf16.generate_ids(cfg, sd, torch.from_numpy(f["enc_t5"]), amask, prompt, pmask, int(f["max_length"]),
                                   exact=False, record=64, guidance_scale=gs, forced=torch.from_numpy(f["raw_ids"]))
    d = np.abs(rb["step_logits"].numpy()[:n] - f["step_logits"][:n])
    assert 1e-5 < d.max() < 0.05


@pytest.mark.parametrize("name", ["tiny_greedy", "tiny_greedy_ragged", "mini_greedy_ragged"])
def test_t5_encoder_matches_transformers(name):
    f, meta = load_case(name)
    cfg, sd = state_dict_for(name, meta)
    desc, amask, _, _ = case_inputs(f)
    enc = cpu_ref.t5_encode(cfg, sd, desc, amask)
    valid = amask.bool().numpy()
    assert np.abs(enc.numpy() - f["enc_t5"])[valid].max() < 5e-5


@pytest.mark.parametrize("name", ["dac_tiny", "dac_mini"])
def test_dac_decode_matches_transformers_dac(name):
    f, _ = load_case(name)
    cfg = config_for(name)
    sd = synthetic_state_dict(cfg, seed=int(f["seed"]), parts=("audio_encoder",))
    codes = torch.from_numpy(f["codes"])
    z = cpu_ref.dac_from_codes(cfg, sd, codes)
    assert np.abs(z.numpy() - f["latent"]).max() < 1e-5
    wav = cpu_ref.dac_decode(cfg, sd, codes[None])
    assert wav.shape[-1] == codes.shape[-1] * cfg.audio_encoder.hop_length
    err = wav.numpy() - f["wav"]
    assert np.sqrt((err ** 2).mean()) < 1e-5
    rms = float(np.sqrt((f["wav"] ** 2).mean()))
    assert 0.05 < rms < 0.95, "fixture must not be saturated / silent"


def test_dac_decode_oracle_at_300_frames_matches_transformers_dac():
    """the production-size codec fixture (oracle/gen_golden.py --only-dac-long): the CPU restatement reproduces the
    transformers DacModel waveform at 300 frames too (first utterance; the GPU test compares both and the 2571-frame clip)"""
    f, _ = load_case("dac_mini_f300")
    cfg = config_for("dac_mini")
    sd = synthetic_state_dict(cfg, seed=int(f["seed"]), parts=("audio_encoder",))
    # the second utterance's first 96 frames, decoded from a 112-frame prefix (the codec is local: receptive field ~10 frames)
    codes = torch.from_numpy(f["codes"])[1:2, :, :112]
    n = 96 * cfg.audio_encoder.hop_length
    wav = cpu_ref.dac_decode(cfg, sd, codes[None])
    err = wav.numpy().reshape(-1)[:n] - f["wav"][1].reshape(-1)[:n]
    assert np.sqrt((err ** 2).mean()) < 1e-5 and np.abs(err).max() < 1e-4
    g, _ = load_case("dac_mini_f2571")
    assert int(g["n_samples"]) == 2571 * cfg.audio_encoder.hop_length and g["wav_stride"].shape[1] * int(g["stride"]) >= int(g["n_samples"])
    assert g["block_sums"].shape[1] == -(-int(g["n_samples"]) // int(g["block"])) and int(g["n_windows"]) == 4


def test_codes_to_waveform_sequential_path():
    # a special id inside the codes triggers the per-sample path (M:2654-2680): frames holding an id
    # >= codebook_size are dropped per utterance, results are zero padded to the longest
    cfg = config_for("dac_tiny")
    sd = synthetic_state_dict(cfg, seed=11, parts=("audio_encoder",))
    g = torch.Generator().manual_seed(0)
    codes = torch.randint(0, 64, (1, 2, 4, 6), generator=g)
    codes[0, 1, 2, 4] = cfg.decoder.eos_token_id
    wav = cpu_ref.codes_to_waveform(cfg, sd, codes)
    hop = cfg.audio_encoder.hop_length
    assert wav.shape == (2, 6 * hop)
    assert (wav[1, 5 * hop:] == 0).all() and wav[1, : 5 * hop].abs().sum() > 0
    full = cpu_ref.dac_decode(cfg, sd, codes[:, :1])[0, 0]
    assert torch.allclose(wav[0], full, atol=1e-6)


@pytest.mark.parametrize("name", ["dac_enc_tiny", "dac_enc_mini"])
def test_dac_encode_oracle_matches_hf_dac_fixture(name):
    """Codec ENCODE side (audio prompts, dac_wrapper/modeling_dac.py:25-96): the CPU restatement of the
    encoder conv stack + residual vector quantisation reproduces the codes that transformers' DacModel.encode produced for
    the same synthetic weights (fixture written by oracle/gen_golden.py --only-dac-encode).  This is groundwork: the HIP
    path does not serve `input_values` yet, the oracle it will be checked against is already pinned."""
    from oracle import cpu_ref
    from parler_tts_amd.weights import dac_encoder_keys, synthetic_dac_encoder_state_dict, synthetic_state_dict
    from tests.helpers import config_for

    f, _ = load_case(name)
    cfg = config_for("mini" if name.endswith("mini") else "tiny")
    seed = int(f["seed"])
    sd = synthetic_state_dict(cfg, seed=seed, parts=("audio_encoder",))
    enc = synthetic_dac_encoder_state_dict(cfg.audio_encoder, seed=seed)
    assert set(enc) == set(dac_encoder_keys(cfg.audio_encoder)) and not (set(enc) & set(sd))
    sd.update(enc)
    audio = torch.from_numpy(f["audio"])
    codes = cpu_ref.dac_encode(cfg, sd, audio)
    assert codes.shape == f["codes"].shape
    assert np.array_equal(codes.numpy(), f["codes"])
    assert float(f["min_margin"]) >= 1e-4
    # and the decode side inverts the quantiser's bookkeeping: from_codes(codes) is the quantised latent the encoder saw
    z = cpu_ref.dac_from_codes(cfg, sd, codes[0])
    assert z.shape == f["latent"].shape


def test_the_oracle_in_bfloat16_is_the_reference_in_bfloat16():
    """`<name>_refbf16` fixtures hold the REAL reference run with `model.to(torch.bfloat16)` (oracle/gen_golden.py
    --only-ref-bf16).  oracle/cpu_ref.py uses the reference's op order with stock torch ops and follows the dtype of what it is
    given, so evaluated on bf16 weights and bf16 encoder states it must reproduce those logits - which pins a PORTABLE statement
    of the reference's bf16 arithmetic (usable on the GPU box, where the reference is absent).  First 5 steps of the ragged
    mini_v0.1 case (24 layers; ~20 s); bitwise on the machine that wrote the fixture, one bf16 ulp of slack elsewhere."""
    from tests.helpers import bf16_bits_to_float

    name = "mini_greedy_ragged"
    f, meta = load_case(name)
    g, gmeta = load_case(name + "_refbf16")
    cfg, sd = state_dict_for(name, meta)
    desc, amask, prompt, pmask = case_inputs(f)
    gold = torch.from_numpy(f["raw_ids"])
    n = 5
    sdb = {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in sd.items() if not k.startswith("text_encoder.")}
    seen = []

    class _Enough(Exception):
        pass

    def grab(ids, scores):          # a logits processor sees every step's scores: keep n of them, then leave the loop
        seen.append(scores.float().clone())
        if len(seen) == n:
            raise _Enough
        return scores

    try:                            # the fixture's own max_length: the delay pattern depends on it
        cpu_ref.generate_ids(cfg, sdb, torch.from_numpy(f["enc_t5"]).to(torch.bfloat16), amask, prompt, pmask, int(f["max_length"]),
                             forced_tokens={s: gold[:, s + 1] for s in range(n)}, logits_processor=[grab])
    except _Enough:
        pass
    got = torch.stack(seen).numpy()
    want = bf16_bits_to_float(g["step_logits_bf16_bits"])[:n]
    assert np.array_equal(g["logit_steps"][:n], np.arange(n))
    assert np.abs(got - want).max() <= 2.0 ** -6, np.abs(got - want).max()          # logits < 4: one bf16 ulp is 2^-6 at most
    assert (got == want).mean() > 0.99
    # the fixture's own summary is what its arrays say
    d = np.abs(bf16_bits_to_float(g["step_logits_bf16_bits"]) - g["step_logits_fp32"])
    summ = gmeta["summary"]["bf16"]
    assert abs(float(d.max()) - summ["max_abs_dlogit"]) < 1e-6 and abs(float(d.mean()) - summ["mean_abs_dlogit"]) < 1e-7
    assert np.abs(g["step_logits_fp32"][: f["step_logits"].shape[0]] - f["step_logits"]).max() == 0.0   # same fp32 run as the base fixture
