"""world-size-2 gloo test of the data-parallel plumbing (CPU): the DP result must equal the single-process
result row for row."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parler_tts_amd.distributed import generate_data_parallel, shard_bounds


def fake_generate(input_ids, attention_mask=None, prompt_input_ids=None, prompt_attention_mask=None, frames=5, K=3):
    """deterministic stand-in for model.generate(return_raw_ids=True)[0]: codes depend only on the row"""
    b = input_ids.shape[0]
    base = input_ids.sum(dim=1, keepdim=True) + (0 if prompt_input_ids is None else prompt_input_ids.sum(dim=1, keepdim=True))
    if attention_mask is not None:
        base = base + attention_mask.sum(dim=1, keepdim=True)
    k = torch.arange(K).view(1, K, 1)
    f = torch.arange(frames).view(1, 1, frames)
    return (base.view(b, 1, 1) + 7 * k + 3 * f) % 1024


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
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, 100, (B, 6), generator=g)
    mask = (torch.rand(B, 6, generator=g) > 0.2).long()
    prompt = torch.randint(2, 100, (B, 4), generator=g)
    # only the source rank "has" the inputs
    args = dict(input_ids=ids, attention_mask=mask, prompt_input_ids=prompt) if rank == 0 else \
        dict(input_ids=None, attention_mask=None, prompt_input_ids=None)
    codes = generate_data_parallel(fake_generate, **args, src=0, device="cpu", frames=5, K=3)
    out_q.put((rank, codes.clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [5, 2, 1])
def test_dp_equals_single_process(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(2, 100, (B, 6), generator=g)
    mask = (torch.rand(B, 6, generator=g) > 0.2).long()
    prompt = torch.randint(2, 100, (B, 4), generator=g)
    ref = fake_generate(ids, mask, prompt)
    assert torch.equal(got[0], ref) and torch.equal(got[1], ref)


def test_shard_bounds_cover_the_batch():
    for n in (1, 7, 32, 255):
        for world in (1, 2, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
