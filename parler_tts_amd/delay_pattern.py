"""Delay-pattern mask: host-side integer logic of the hot path.

Same two public functions as the reference exports (``parler_tts/__init__.py:6-11``):

* :func:`build_delay_pattern_mask`  - behaviour of modeling_parler_tts.py:77-139
* :func:`apply_delay_pattern_mask`  - behaviour of modeling_parler_tts.py:68-74

Codebook ``k`` of every utterance is delayed by ``k`` steps.  For ``K`` codebooks and ``max_length`` L
the pattern of row ``k`` is::

      t <= k              -> BOS          (forced start region)
      t >= L - K + 1 + k  -> PAD          (forced tail region)
      k <= t < s + k      -> prompt[t-k]  (decoder prompt of length s, shifted right by k)
      otherwise           -> -1           (free: the model's prediction is kept)

Written in closed form with index comparisons (no triangular matrices).  The pattern matrix itself goes to the
engine (``ptts_prefill``'s ``pattern`` argument): ``select_kernel`` (csrc/misc.hip) reads its column ``cur_len`` to feed
the next step the forced value where it is not -1, which is ``apply_delay_pattern_mask`` on the last column.
"""
from __future__ import annotations

from typing import Tuple

import torch


def build_delay_pattern_mask(
    input_ids: torch.Tensor, bos_token_id: int, pad_token_id: int, max_length: int, num_codebooks: int
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns ``(input_ids_cut, pattern_mask)``.

    ``input_ids``: ``[B*K, s]`` int64.  ``pattern_mask``: ``[B*K, max_length]`` with -1 where the
    position is free.  ``input_ids_cut``: the pattern restricted to the columns before the first
    free position of codebook 0 (that is what generation starts from).
    """
    K = int(num_codebooks)
    L = int(max_length)
    ids = input_ids.reshape(-1, K, input_ids.shape[-1])
    B, _, s = ids.shape
    dev = ids.device

    # too short for a delay pattern: everything is free and the ids pass through (M:105-106)
    if L < 2 * K - 1:
        free = torch.full((B * K, L), -1, dtype=torch.long, device=dev)
        return ids.reshape(B * K, -1), free

    t = torch.arange(L, device=dev).view(1, 1, L)
    k = torch.arange(K, device=dev).view(1, K, 1)
    in_bos = t <= k
    in_pad = t >= (L - K + 1 + k)

    # prompt entries shifted right by the codebook index; a prompt that does not fit is an error,
    # exactly as the reference's slice assignment (M:111) raises
    if s + K - 1 > L:
        raise RuntimeError(
            f"decoder prompt of length {s} shifted by up to {K - 1} does not fit max_length={L}"
        )
    shifted = torch.full((B, K, L), -1, dtype=torch.long, device=dev)
    src_col = t - k  # column of the prompt that lands on (k, t)
    has_prompt = (src_col >= 0) & (src_col < s)
    gathered = torch.gather(ids, 2, src_col.clamp(0, max(s - 1, 0)).expand(B, K, L)) if s > 0 else shifted
    shifted = torch.where(has_prompt.expand(B, K, L), gathered, shifted)

    pattern = torch.where(in_bos, torch.full_like(shifted, bos_token_id), shifted)
    pattern = torch.where(in_pad, torch.full_like(shifted, pad_token_id), pattern)

    # first free column of codebook 0 (min over the batch); none -> keep all s columns
    free0 = pattern[:, 0, :] == -1
    if bool(free0.any()):
        first_free = int(torch.where(free0, t.view(1, L).expand(B, L), torch.full((B, L), L, device=dev)).min())
    else:
        first_free = s

    return pattern[..., :first_free].reshape(B * K, -1), pattern.reshape(B * K, L)


def apply_delay_pattern_mask(input_ids: torch.Tensor, decoder_pad_token_mask: torch.Tensor) -> torch.Tensor:
    """Keep ``input_ids`` where the pattern is -1, otherwise take the pattern's value (M:68-74)."""
    n = input_ids.shape[-1]
    pat = decoder_pad_token_mask[..., :n]
    return torch.where(pat == -1, input_ids, pat)


def frames_kept(out_len: int, num_codebooks: int) -> int:
    """Number of codec frames that survive the un-delay filter of M:2636-2645 for ``out_len`` columns."""
    if out_len < 2 * num_codebooks - 1:
        return out_len
    return out_len - num_codebooks
