"""Data-parallel plumbing for batched generation across the GPUs of one node.

Utterances are independent (every op of the hot path is per batch row), so the
batch is cut into contiguous slices, weights are replicated, and there is NO collective inside the
token loop.  RCCL over xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests) is
used only for (i) broadcasting the small int64 id tensors from the source rank and (ii) gathering the
codec ids / waveform lengths afterwards.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `n` rows owned by `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_inputs(tensors: Dict[str, Optional[torch.Tensor]], src: int = 0, device=None, group=None):
    """Rank `src` holds the full-batch id/mask tensors; every rank receives them (they are KBs)."""
    rank = dist.get_rank(group)
    meta = [{k: (None if v is None else (tuple(v.shape), str(v.dtype).replace("torch.", ""))) for k, v in tensors.items()}] \
        if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, group=group)
    out = {}
    for k, m in meta[0].items():
        if m is None:
            out[k] = None
            continue
        shape, dt = m
        t = tensors[k].to(device) if rank == src else torch.empty(shape, dtype=getattr(torch, dt), device=device)
        dist.broadcast(t, src=src, group=group)
        out[k] = t
    return out


def gather_codes(local_codes: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All ranks receive the codec ids of the whole batch [total, K, F] (int32 on the wire: neither gloo nor RCCL carries int16)."""
    world = dist.get_world_size(group)
    K, F = local_codes.shape[1], local_codes.shape[2]
    per = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world))
    buf = torch.zeros(per, K, F, dtype=torch.int32, device=local_codes.device)
    buf[: local_codes.shape[0]] = local_codes.to(torch.int32)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(total, world, r)
        rows.append(parts[r][: hi - lo])
    return torch.cat(rows, dim=0).to(torch.long)


def generate_data_parallel(local_generate: Callable[..., torch.Tensor], input_ids, attention_mask=None,
                           prompt_input_ids=None, prompt_attention_mask=None, src: int = 0, device=None, group=None,
                           **gen_kwargs) -> torch.Tensor:
    """Run `local_generate(input_ids=..., attention_mask=..., prompt_input_ids=..., prompt_attention_mask=...,
    **gen_kwargs) -> codes [b, K, F]` on this rank's slice and return the codes of the whole batch on
    every rank.  `local_generate` is typically
    ``lambda **kw: model.generate(return_raw_ids=True, **kw)[0]``."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    full = broadcast_inputs(dict(input_ids=input_ids, attention_mask=attention_mask, prompt_input_ids=prompt_input_ids,
                                 prompt_attention_mask=prompt_attention_mask), src=src, device=device, group=group)
    total = full["input_ids"].shape[0]
    lo, hi = shard_bounds(total, world, rank)
    kw = {k: (None if v is None else v[lo:hi]) for k, v in full.items()}
    if hi > lo:
        codes = local_generate(**{k: v for k, v in kw.items() if v is not None}, **gen_kwargs)
    else:
        codes = None
    # ranks with an empty slice still take part in the gather
    shape = [None]
    if rank == src:
        shape = [tuple(codes.shape[1:])]
    dist.broadcast_object_list(shape, src=src, group=group)
    if codes is None:
        codes = torch.zeros((0,) + shape[0], dtype=torch.long, device=device)
    return gather_codes(codes, total, group=group)
