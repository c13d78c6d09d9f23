"""Data-parallel plumbing for batched generation across the GPUs of one node.

Utterances are independent (every op of the hot path is per batch row), so the
batch is cut into contiguous slices, weights are replicated, and there is NO collective inside the
token loop.  RCCL over xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests) is
used only for (i) scattering each rank's slice of the small int64 id tensors from the source rank and
(ii) gathering the codec ids or waveforms afterwards.

A rank's loop ends when ITS rows have all emitted EOS, so ranks come back with different frame counts; the gathers
pad to the longest slice and carry every row's length, as the reference's evaluation loop pads across processes
(training/run_parler_tts_training.py:1555, 1684-1689: `accelerator.pad_across_processes(..., pad_index=0)`).

Every shape disagreement is exchanged INSIDE a collective and raised on all ranks after it: a rank never raises in
front of a collective the others have already entered.

Launch (one process per GPU; the launcher runs BEFORE anything touches a GPU):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 your_script.py
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist

_DTYPES = [torch.int64, torch.int32, torch.float32, torch.bool, torch.int16, torch.uint8, torch.float64, torch.bfloat16,
           torch.float16]
_MAX_DIMS = 4
_BIG = 1 << 40


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `n` rows owned by `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _meta_tensor(tensors: Dict[str, Optional[torch.Tensor]], names, device) -> torch.Tensor:
    """fixed-size description of the named tensors: per name [present, dtype code, ndim, dims...] (no pickling, so the
    exchange is one small int64 broadcast on the collective's own device).  A tensor the scatter cannot carry (more than
    _MAX_DIMS dims, 0-dim, a dtype outside _DTYPES) is encoded as present = -1 and NOT raised here: only the source rank runs
    this function, in front of a broadcast the other ranks have already entered; `_check_meta` raises on every rank after it."""
    rows = []
    for k in names:
        v = tensors.get(k)
        if v is None:
            rows.append([0] * (3 + _MAX_DIMS))
        elif v.dim() > _MAX_DIMS or v.dim() < 1 or v.dtype not in _DTYPES:
            rows.append([-1, -1, min(v.dim(), _MAX_DIMS)] + list(v.shape[:_MAX_DIMS]) + [0] * (_MAX_DIMS - min(v.dim(), _MAX_DIMS)))
        else:
            rows.append([1, _DTYPES.index(v.dtype), v.dim()] + list(v.shape) + [0] * (_MAX_DIMS - v.dim()))
    return torch.tensor(rows, dtype=torch.int64, device=device)


def _check_meta(names, meta) -> None:
    """after the broadcast: the same meta on every rank, so every rank raises (or none does)"""
    bad = [k for k, m in zip(names, meta) if m[0] < 0]
    if bad:
        raise ValueError(f"{', '.join(bad)}: unsupported tensor for the batch scatter (1 to {_MAX_DIMS} dims, dtypes "
                         f"{[str(d).replace('torch.', '') for d in _DTYPES]})")


def scatter_inputs(tensors: Dict[str, Optional[torch.Tensor]], src: int = 0, device=None, group=None):
    """Rank `src` holds the full-batch id / mask tensors (dim 0 = utterances); every rank receives ITS contiguous slice
    (`shard_bounds`) and the batch size: ({name: slice or None}, total).  One int64 broadcast of the shapes, then one
    `scatter` per tensor of equal-sized (padded) chunks - a rank never sees the other ranks' rows."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    names = sorted(tensors.keys())
    meta = _meta_tensor(tensors, names, device) if rank == src else torch.zeros((len(names), 3 + _MAX_DIMS), dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src, group=group)
    meta = meta.cpu().tolist()
    _check_meta(names, meta)
    total = max([m[3] for m in meta if m[0]] + [0])
    per = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)) if total else 0
    lo, hi = shard_bounds(total, world, rank)
    out = {}
    for k, m in zip(names, meta):
        if not m[0]:
            out[k] = None
            continue
        dtype, shape = _DTYPES[m[1]], tuple(m[3:3 + m[2]])
        if shape[0] != total:
            raise ValueError(f"{k}: {shape[0]} rows, the batch has {total}")       # same meta on every rank: raised by all
        wire = torch.int32 if dtype in (torch.bool, torch.int16, torch.uint8) else dtype
        recv = torch.empty((per,) + shape[1:], dtype=wire, device=device)
        chunks = None
        if rank == src:
            full = tensors[k].to(device=device, dtype=wire)
            chunks = []
            for r in range(world):
                rlo, rhi = shard_bounds(total, world, r)
                c = torch.zeros((per,) + shape[1:], dtype=wire, device=device)
                c[: rhi - rlo] = full[rlo:rhi]
                chunks.append(c)
        dist.scatter(recv, chunks, src=src, group=group)
        out[k] = recv[: hi - lo].to(dtype)
    return out, total


def broadcast_inputs(tensors: Dict[str, Optional[torch.Tensor]], src: int = 0, device=None, group=None):
    """Every rank receives the FULL tensors of rank `src` (kept for callers that need the whole batch everywhere;
    `generate_data_parallel` scatters slices instead)."""
    rank = dist.get_rank(group)
    names = sorted(tensors.keys())
    meta = _meta_tensor(tensors, names, device) if rank == src else torch.zeros((len(names), 3 + _MAX_DIMS), dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src, group=group)
    meta = meta.cpu().tolist()
    _check_meta(names, meta)
    out = {}
    for k, m in zip(names, meta):
        if not m[0]:
            out[k] = None
            continue
        dtype, shape = _DTYPES[m[1]], tuple(m[3:3 + m[2]])
        t = tensors[k].to(device) if rank == src else torch.empty(shape, dtype=dtype, device=device)
        dist.broadcast(t, src=src, group=group)
        out[k] = t
    return out


def _gather_padded(local: torch.Tensor, total: int, pad_value, wire_dtype, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`local` [b, ..., n_local] (b = this rank's slice of `total` rows, possibly 0; the LAST dim is the ragged one):
    every rank receives ([total, ..., n_max] padded with pad_value, lengths [total] = n of the rank that produced the row).
    A rank with an empty slice learns the middle dims from the others.  A wrong row count or disagreeing middle dims on ANY
    rank travel inside the first all-reduce and raise ValueError on EVERY rank after it."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local.device
    lo, hi = shard_bounds(total, world, rank)
    nmid = local.dim() - 2
    if nmid < 0:
        raise ValueError("the gathered tensor needs at least [rows, n]")        # static property of the call, same on every rank
    have = local.shape[0] > 0
    mid = [int(d) for d in local.shape[1:-1]]
    # one MAX all-reduce carries: ragged length, "my row count is wrong", max of the middle dims, -min of the middle dims
    info = torch.tensor([local.shape[-1] if have else 0, int(local.shape[0] != hi - lo)] + (mid if have else [0] * nmid) +
                        ([-d for d in mid] if have else [-_BIG] * nmid), dtype=torch.int64, device=dev)
    mx = info.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    mx = mx.cpu().tolist()
    n_max, bad_rows = int(mx[0]), int(mx[1])
    mid_max, mid_min = tuple(mx[2:2 + nmid]), tuple(-d for d in mx[2 + nmid:])
    if bad_rows:
        raise ValueError(f"a rank produced a row count that is not its slice (rank {rank}: rows [{lo}, {hi}), produced {local.shape[0]})")
    if any(a != b for a, b in zip(mid_max, mid_min) if b != _BIG):
        raise ValueError(f"inner dims differ across ranks: between {mid_min} and {mid_max} (rank {rank}: {tuple(mid)})")
    per = max(shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world))
    buf = torch.full((per,) + mid_max + (n_max,), pad_value, dtype=wire_dtype, device=dev)
    if have:
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
                           gather: str = "codes", pad_token_id: int = 1024, timings: Optional[dict] = None,
                           **gen_kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run `local_generate(input_ids=..., attention_mask=..., prompt_input_ids=..., prompt_attention_mask=...,
    **gen_kwargs)` on this rank's slice of the batch that rank `src` holds, and return the whole batch's result on every
    rank together with every row's length (a 2-tuple; callers of the round-1 form that returned the tensor alone take [0]):
      gather = "codes"    : local_generate returns codes [b, K, F]    -> ([B, K, F_max] int64, frames [B]);  typically
                            ``lambda **kw: model.generate(return_raw_ids=True, **kw)[0]``
      gather = "waveform" : local_generate returns audio [b, samples] -> ([B, S_max] fp32, samples [B]);      ``model.generate``
    Ranks whose slice is empty (B < world size) skip the generation and still take part in the collectives.
    `timings` (optional dict) receives this rank's host wall times in ms: scatter_ms, generate_ms (until local_generate
    returned), gather_ms."""
    if gather not in ("codes", "waveform"):
        raise ValueError("gather must be 'codes' or 'waveform'")
    t0 = time.perf_counter()
    mine, total = scatter_inputs(dict(input_ids=input_ids, attention_mask=attention_mask, prompt_input_ids=prompt_input_ids,
                                      prompt_attention_mask=prompt_attention_mask), src=src, device=device, group=group)
    t1 = time.perf_counter()
    if mine["input_ids"] is not None and mine["input_ids"].shape[0] > 0:
        out = local_generate(**{k: v for k, v in mine.items() if v is not None}, **gen_kwargs)
    else:       # inner dims are learnt from the other ranks inside the gather
        out = torch.zeros((0, 0, 0) if gather == "codes" else (0, 0), dtype=torch.long if gather == "codes" else torch.float32,
                          device=device)
    t2 = time.perf_counter()
    res = gather_codes(out, total, pad_token_id, group) if gather == "codes" else gather_waveforms(out, total, group)
    if timings is not None:
        timings.update(scatter_ms=(t1 - t0) * 1e3, generate_ms=(t2 - t1) * 1e3, gather_ms=(time.perf_counter() - t2) * 1e3)
    return res
