"""world-size-2 gloo tests of the data-parallel plumbing (CPU): the DP result must equal the single-process result row
for row, with the call shape of the real API - ranks come back with DIFFERENT frame counts (a rank stops when its own rows
have emitted EOS), a rank may own no row at all, and both gathers (codec ids, waveforms) carry per-row lengths."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parler_tts_amd.distributed import generate_data_parallel, shard_bounds

PAD = 1024


def fake_frames(input_ids):
    """frames a slice 'generates': depends on its rows, so two ranks differ (like an EOS-dependent L_out)"""
    return 3 + int(input_ids.sum()) % 5


def fake_generate(input_ids, attention_mask=None, prompt_input_ids=None, prompt_attention_mask=None, K=3):
    """deterministic stand-in for model.generate(return_raw_ids=True)[0]: [b, K, F] codes that depend only on the row,
    F on the slice"""
    b = input_ids.shape[0]
    frames = fake_frames(input_ids)
    base = input_ids.sum(dim=1, keepdim=True) + (0 if prompt_input_ids is None else prompt_input_ids.sum(dim=1, keepdim=True))
    if attention_mask is not None:
        base = base + attention_mask.sum(dim=1, keepdim=True)
    k = torch.arange(K).view(1, K, 1)
    f = torch.arange(frames).view(1, 1, frames)
    return (base.view(b, 1, 1) + 7 * k + 3 * f) % 1024


def fake_waveform(input_ids, attention_mask=None, prompt_input_ids=None, prompt_attention_mask=None, hop=4):
    codes = fake_generate(input_ids, attention_mask, prompt_input_ids, prompt_attention_mask)
    return codes[:, 0, :].repeat_interleave(hop, dim=1).float() / 1024.0


def _inputs(B):
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, 100, (B, 6), generator=g)
    mask = (torch.rand(B, 6, generator=g) > 0.2).long()
    prompt = torch.randint(2, 100, (B, 4), generator=g)
    return ids, mask, prompt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, mask, prompt = _inputs(B)
    # only the source rank "has" the inputs
    args = dict(input_ids=ids, attention_mask=mask, prompt_input_ids=prompt) if rank == 0 else \
        dict(input_ids=None, attention_mask=None, prompt_input_ids=None)
    codes, frames = generate_data_parallel(fake_generate, **args, src=0, device="cpu", gather="codes", pad_token_id=PAD, K=3)
    wav, samples = generate_data_parallel(fake_waveform, **args, src=0, device="cpu", gather="waveform", hop=4)
    out_q.put((rank, codes.clone(), frames.clone(), wav.clone(), samples.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2, 1])
def test_dp_equals_single_process(B):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids, mask, prompt = _inputs(B)
    # what each slice produces on its own, padded by hand
    parts, lens = [], []
    for r in range(world):
        lo, hi = shard_bounds(B, world, r)
        if hi > lo:
            c = fake_generate(ids[lo:hi], mask[lo:hi], prompt[lo:hi])
            parts.append(c)
            lens += [c.shape[-1]] * (hi - lo)
    fmax = max(lens)
    if B == 5:
        assert len(set(lens)) > 1, "the test must exercise unequal lengths"
    ref = torch.cat([torch.nn.functional.pad(c, (0, fmax - c.shape[-1]), value=PAD) for c in parts])
    for r in range(world):
        codes, frames, wav, samples = got[r]
        assert codes.dtype == torch.long and torch.equal(codes, ref)
        assert frames.tolist() == lens
        assert samples.tolist() == [4 * n for n in lens] and wav.shape == (B, 4 * fmax)
        for b in range(B):
            assert torch.equal(wav[b, : samples[b]], (ref[b, 0, : lens[b]].repeat_interleave(4).float() / 1024.0))
            assert (wav[b, samples[b]:] == 0).all() and (codes[b, :, frames[b]:] == PAD).all()


def test_shard_bounds_cover_the_batch():
    for n in (1, 7, 32, 255):
        for world in (1, 2, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ---- bench.py's N > 1 path: the one_pass() closure and the checkable fields of its JSON line, with a fake model ----
class _FakeCfg:
    class decoder:
        num_codebooks = 3
        pad_token_id = PAD


class _FakeModel:
    """stands for ParlerTTSForConditionalGeneration: generate() leaves codec ids [1, b, K, F] in .last_codes"""

    def __init__(self, L):
        self.L, self.seen = L, []

    def generate(self, input_ids, attention_mask=None, prompt_input_ids=None, **kw):
        self.seen.append(tuple(input_ids.shape))
        K, F = _FakeCfg.decoder.num_codebooks, self.L - _FakeCfg.decoder.num_codebooks
        base = input_ids.sum(dim=1).view(-1, 1, 1) + prompt_input_ids.sum(dim=1).view(-1, 1, 1)
        self.last_codes = (((base + torch.arange(K).view(1, K, 1) + 5 * torch.arange(F).view(1, 1, F)) % 1024)[None])[0][None]
        return None


def _bench_worker(rank, world, port, B, L, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    g = torch.Generator().manual_seed(9)
    desc = torch.randint(2, 100, (B * world, 6), generator=g)
    prompt = torch.randint(2, 100, (B * world, 4), generator=g)
    amask = torch.ones(B * world, 6, dtype=torch.long)
    model, timings = _FakeModel(L), {}
    cpu = torch.device("cpu")
    one_pass = bench.make_one_pass(model, _FakeCfg, desc, amask, prompt, dict(do_sample=False, max_length=L), dist, rank, world,
                                   cpu, cpu, B, L, timings)
    codes = one_pass()
    phase = {"t5_prefill_ms": 1.0 + rank, "decode_ms": 100.0 * (1 + rank), "post_ms": 10.0}
    rep = bench.dp_report(dist, world, cpu, phase, timings, 1, 1000.0, root="/nonexistent")
    out_q.put((rank, codes.clone(), list(model.seen), rep, dict(timings)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_one_pass_and_dp_report_world2():
    world, B, L = 2, 3, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, B, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(9)
    desc = torch.randint(2, 100, (B * world, 6), generator=g)
    prompt = torch.randint(2, 100, (B * world, 4), generator=g)
    ref_model = _FakeModel(L)
    ref_model.generate(desc, None, prompt)
    for r in range(world):
        codes, seen, rep, timings = got[r]
        assert torch.equal(codes, ref_model.last_codes[0])          # whole job's codes on every rank
        assert seen == [(B, 6)]                                      # a rank generated ITS slice only (scatter, not broadcast)
        assert rep["ranks"] == world and rep["backend"] == "gloo"
        assert rep["per_rank_min_max"]["decode_ms"] == [100.0, 200.0] and rep["slowest_rank_by_decode"] == 1
        assert rep["efficiency_vs_n1"] is None and rep["n1_source"] is None
        assert set(timings) == {"scatter_ms", "generate_ms", "gather_ms"}


def _bad_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, mask, prompt = _inputs(4)
    args = dict(input_ids=ids, attention_mask=mask, prompt_input_ids=prompt) if rank == 0 else \
        dict(input_ids=None, attention_mask=None, prompt_input_ids=None)

    def broken(**kw):      # rank 1 returns one row too many
        c = fake_generate(**kw)
        return torch.cat([c, c[:1]]) if rank == 1 else c

    try:
        generate_data_parallel(broken, **args, src=0, device="cpu", gather="codes", pad_token_id=PAD)
        out_q.put((rank, "no error"))
    except ValueError as e:
        out_q.put((rank, "ValueError"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_wrong_row_count_raises_on_every_rank_instead_of_hanging():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: "ValueError", 1: "ValueError"}


def _unsupported_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from parler_tts_amd.distributed import broadcast_inputs, scatter_inputs

    ids, mask, prompt = _inputs(4)
    got = []
    for fn in (scatter_inputs, broadcast_inputs):
        for bad in (ids.reshape(4, 1, 1, 1, 6), ids.to(torch.complex64)):      # 5 dims; a dtype the wire does not carry
            t = dict(input_ids=bad, attention_mask=mask) if rank == 0 else dict(input_ids=None, attention_mask=None)
            try:
                fn(t, src=0, device="cpu")
                got.append("no error")
            except ValueError:
                got.append("ValueError")
    # the group is still usable: nobody is stuck inside a collective
    mine, total = scatter_inputs(dict(input_ids=ids if rank == 0 else None), src=0, device="cpu")
    got.append((total, tuple(mine["input_ids"].shape)))
    out_q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_an_unsupported_tensor_on_the_source_rank_raises_on_every_rank_instead_of_hanging():
    """_meta_tensor used to raise on the source rank alone, in front of a broadcast the others were already in."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unsupported_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r][:4] == ["ValueError"] * 4 and got[r][4] == (4, (2, 6))
