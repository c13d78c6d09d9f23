"""Data-parallel plumbing for batched generation across the GPUs of one node.

Utterances are independent (every op of the hot path is per batch row), so the
batch is cut into contiguous slices, weights are replicated, and there is NO collective inside the
token loop.  RCCL over xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests) is
used only for (i) broadcasting the small int64 id tensors from the source rank and (ii) gathering the
codec ids or waveforms afterwards.

A rank's loop ends when ITS rows have all emitted EOS, so ranks come back with different frame counts; the gathers
pad to the longest slice and carry every row's length, as the reference's evaluation loop pads across processes
(training/run_parler_tts_training.py:1555, 1684-1689: `accelerator.pad_across_processes(..., pad_index=0)`).

Launch (one process per GPU; the launcher runs BEFORE anything touches a GPU):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 your_script.py
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

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


def _gather_padded(local: torch.Tensor, total: int, pad_value, wire_dtype, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`local` [b, ..., n_local] (b = this rank's slice of `total` rows, possibly 0; the LAST dim is the ragged one):
    every rank receives ([total, ..., n_max] padded with pad_value, lengths [total] = n of the rank that produced the row)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local.device
    lo, hi = shard_bounds(total, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} owns rows [{lo}, {hi}) but produced {local.shape[0]} rows")
    # the middle dims must agree across ranks; a rank with an empty slice learns them from the others
    info = torch.tensor([local.shape[-1] if local.shape[0] else 0] + [int(d) for d in local.shape[1:-1]], dtype=torch.int64, device=dev)
    mx = info.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    n_max, mid = int(mx[0]), tuple(int(d) for d in mx[1:])
    if local.shape[0] and tuple(local.shape[1:-1]) != mid:
        raise ValueError(f"rank {rank}: inner dims {tuple(local.shape[1:-1])} differ from the other ranks' {mid}")
    per = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world))
    buf = torch.full((per,) + mid + (n_max,), pad_value, dtype=wire_dtype, device=dev)
    if local.shape[0]:
        buf[: local.shape[0], ..., : local.shape[-1]] = local.to(wire_dtype)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    lens = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, info[:1].clone(), group=group)
    rows, lengths = [], []
    for r in range(world):
        rlo, rhi = shard_bounds(total, world, r)
        rows.append(parts[r][: rhi - rlo])
        lengths.append(lens[r].expand(rhi - rlo))
    return torch.cat(rows, dim=0), torch.cat(lengths, dim=0)


def gather_codes(local_codes: torch.Tensor, total: int, pad_token_id: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Codec ids of the whole batch on every rank: ([total, K, F_max] int64 padded with pad_token_id, frames [total]).
    int32 on the wire (neither gloo nor RCCL carries int16)."""
    codes, lengths = _gather_padded(local_codes, total, int(pad_token_id), torch.int32, group)
    return codes.to(torch.long), lengths


def gather_waveforms(local_wav: torch.Tensor, total: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Waveforms of the whole batch on every rank: ([total, S_max] fp32 zero padded, samples [total])."""
    return _gather_padded(local_wav.to(torch.float32), total, 0.0, torch.float32, group)


def generate_data_parallel(local_generate: Callable[..., torch.Tensor], input_ids, attention_mask=None,
                           prompt_input_ids=None, prompt_attention_mask=None, src: int = 0, device=None, group=None,
                           gather: str = "codes", pad_token_id: int = 1024, **gen_kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run `local_generate(input_ids=..., attention_mask=..., prompt_input_ids=..., prompt_attention_mask=...,
    **gen_kwargs)` on this rank's slice of the batch that rank `src` holds, and return the whole batch's result on every
    rank together with every row's length:
      gather = "codes"    : local_generate returns codes [b, K, F]    -> ([B, K, F_max] int64, frames [B]);  typically
                            ``lambda **kw: model.generate(return_raw_ids=True, **kw)[0]``
      gather = "waveform" : local_generate returns audio [b, samples] -> ([B, S_max] fp32, samples [B]);      ``model.generate``
    Ranks whose slice is empty (B < world size) skip the generation and still take part in the collectives."""
    if gather not in ("codes", "waveform"):
        raise ValueError("gather must be 'codes' or 'waveform'")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    full = broadcast_inputs(dict(input_ids=input_ids, attention_mask=attention_mask, prompt_input_ids=prompt_input_ids,
                                 prompt_attention_mask=prompt_attention_mask), src=src, device=device, group=group)
    total = full["input_ids"].shape[0]
    lo, hi = shard_bounds(total, world, rank)
    if hi > lo:
        kw = {k: v[lo:hi] for k, v in full.items() if v is not None}
        out = local_generate(**kw, **gen_kwargs)
    else:       # inner dims are learnt from the other ranks inside the gather
        out = torch.zeros((0, 0, 0) if gather == "codes" else (0, 0), dtype=torch.long if gather == "codes" else torch.float32,
                          device=device)
    if gather == "codes":
        return gather_codes(_with_inner(out, group), total, pad_token_id, group)
    return gather_waveforms(out, total, group)


def _with_inner(codes: torch.Tensor, group=None) -> torch.Tensor:
    """give an empty [0, 0, 0] slice the K of the other ranks (so that its inner dims agree in the gather)"""
    k = torch.tensor([codes.shape[1]], dtype=torch.int64, device=codes.device)
    dist.all_reduce(k, op=dist.ReduceOp.MAX, group=group)
    if codes.shape[0] == 0:
        return torch.zeros((0, int(k), 0), dtype=codes.dtype, device=codes.device)
    return codes
