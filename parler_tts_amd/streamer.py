"""Audio streaming on top of the reference's streamer protocol.

The reference hands a `streamer` object the start ids once and then one column of ids per decoding step
(modeling_parler_tts.py:2540-2541 + the third-party loop), and `end()` at the end; it decodes audio only after the last
step.  `AudioStreamer` is a streamer for that protocol that turns the id columns into audio WHILE the loop runs: it
un-delays the columns as they arrive (codebook k of frame f sits in column f + k + 1, M:77-139), and whenever
`play_frames` new complete frames are available it decodes them with the codec in a window that carries
`context_frames` extra frames on both sides and keeps the middle - the DAC decoder is a stack of finite convolutions
(receptive field: 10 frames per side for the 44 kHz model: 3 for the first conv7, then (3 + 9 + 27) taps of the three
dilated residual units and one tap of the transposed conv per block, divided by the block's stride), so with >= 10
frames of context the chunked waveform equals the one-shot DACModel.decode (dac_wrapper/modeling_dac.py:98-134) of the
whole utterance sample for sample.  Latency = context_frames codec frames (16 frames = 0.19 s at 86 frames/s).

Finished rows: once a (utterance, codebook) row has emitted EOS the loop pads it, so from some frame on an utterance's
frames hold a special id.  The reference drops such frames per utterance before it decodes (M:2668-2670) and zero-pads the
shorter waveforms (M:2676-2680).  The streamer does the same per window: an utterance whose first special frame `fb` falls
inside the window is decoded ALONE over [w0, fb) - its last valid frames then see the zero padding of the clip's end, as
in the one-shot decode, not clamped pad ids - and its samples from fb on are zero.
`num_return_sequences > 1` is not served (the reference interleaves the expanded rows, M:2600-2606 / M:2645: there is no
per-utterance frame order to stream): `put` raises ValueError when the row count changes after the start column.
"""
from __future__ import annotations

import queue
from typing import Callable, List, Optional

import torch


def dac_context_frames(upsampling_ratios, first_kernel: int = 7, dilations=(1, 3, 9), res_kernel: int = 7, last_kernel: int = 7) -> int:
    """one-sided receptive field of the DAC decoder, in codec frames (ceil), walking the stack from the waveform back"""
    need = (last_kernel - 1) // 2                                # samples at the output rate
    for s in reversed(list(upsampling_ratios)):
        need += sum((res_kernel - 1) // 2 * d for d in dilations)    # the block's three residual units (the 1x1 convs add none)
        need = -(-need // s) + 1                                 # transposed conv, kernel 2s / stride s: one more input sample
    return need + (first_kernel - 1) // 2


class AudioStreamer:
    """`streamer=` argument of generate(): collects id columns, emits waveform chunks [B, samples] as torch tensors.

    Iterate over it from another thread (`for chunk in streamer`), or pass `on_audio(chunk, is_last)`.
    `poll_steps` tells generate() how many decoding steps to run between two host polls (batched polling)."""

    def __init__(self, model, play_frames: int = 43, context_frames: Optional[int] = None, poll_steps: Optional[int] = None,
                 on_audio: Optional[Callable] = None, timeout: Optional[float] = None):
        self.model = model
        self.cfg = model.config
        self.K = self.cfg.decoder.num_codebooks
        self.play = int(play_frames)
        need = dac_context_frames(self.cfg.audio_encoder.upsampling_ratios)
        self.ctx = int(context_frames) if context_frames is not None else max(16, need)
        if self.ctx < need:
            raise ValueError(f"context_frames={self.ctx} is below the codec decoder's receptive field ({need} frames per side): "
                             "chunks would not add up to the one-shot waveform")
        self.poll_steps = int(poll_steps) if poll_steps is not None else max(1, min(self.play, 16))
        self.on_audio = on_audio
        self.timeout = timeout
        self.q: "queue.Queue" = queue.Queue()
        self.cols: List[torch.Tensor] = []        # every id column seen so far, each [B*K]
        self.emitted = 0                          # frames already turned into audio
        self.chunks: List[torch.Tensor] = []      # every chunk emitted (for tests / late consumers)
        self._ended = False

    # ---- streamer protocol (M:2540-2541: start ids once, then one column per step; end() after the loop) ----
    def put(self, value: torch.Tensor):
        v = value.detach().to("cpu")
        if v.dim() == 1:
            v = v[:, None]
        if self.cols and v.shape[0] != self.cols[0].shape[0]:
            raise ValueError("AudioStreamer does not serve num_return_sequences > 1 (the id rows were expanded after the start column)")
        for c in range(v.shape[1]):
            self.cols.append(v[:, c])
        self._emit(final=False)

    def end(self):
        self._emit(final=True)
        self._ended = True
        self.q.put(None)

    # ---- un-delay + windowed decode ----
    def _frames_complete(self) -> int:
        return max(0, len(self.cols) - self.K)     # frame f needs columns f + 1 .. f + K

    def _codes(self, f0: int, f1: int) -> torch.Tensor:
        """[B, K, f1 - f0]: codebook k of frame f is row b*K + k of column f + k + 1"""
        rows = self.cols[0].shape[0]
        B = rows // self.K
        out = torch.empty(B, self.K, f1 - f0, dtype=torch.long)
        for k in range(self.K):
            out[:, k, :] = torch.stack([self.cols[f + k + 1].view(B, self.K)[:, k] for f in range(f0, f1)], dim=-1)
        return out

    def _emit(self, final: bool):
        n = self._frames_complete()
        hop = self.cfg.audio_encoder.hop_length
        size = self.cfg.audio_encoder.codebook_size
        while True:
            avail = n - self.emitted
            if avail <= 0:
                return
            if not final and avail < self.play + self.ctx:
                return
            e0 = self.emitted
            e1 = n if (final and avail <= self.play + self.ctx) else e0 + self.play
            w0, w1 = max(0, e0 - self.ctx), (e1 if e1 == n and final else min(n, e1 + self.ctx))
            codes = self._codes(w0, w1)
            W = w1 - w0
            special = ((codes >= size) | (codes < 0)).any(dim=1)                       # [B, frames]
            # first special frame of every utterance inside the window (W = none)
            first_bad = torch.where(special.any(dim=1), special.float().argmax(dim=1), torch.full((codes.shape[0],), W))
            whole = (first_bad == W).nonzero().flatten()
            wav = torch.zeros(codes.shape[0], W * hop, dtype=torch.float32)
            if whole.numel():
                out = self.model.audio_encoder.decode(codes[whole][None].to(self.model.device), [None]).audio_values[:, 0]
                wav[whole] = out.to("cpu", torch.float32)
            for bidx in (first_bad < W).nonzero().flatten().tolist():
                fb = int(first_bad[bidx])
                if fb > 0:      # the utterance ends at fb: decoded alone, its tail sees the end of the clip
                    out = self.model.audio_encoder.decode(codes[bidx:bidx + 1, :, :fb][None].to(self.model.device), [None]).audio_values[:, 0]
                    wav[bidx, : fb * hop] = out[0].to("cpu", torch.float32)
            chunk = wav[:, (e0 - w0) * hop:(e1 - w0) * hop].clone()
            self.emitted = e1
            self.chunks.append(chunk)
            last = final and e1 == n
            if self.on_audio is not None:
                self.on_audio(chunk, last)
            self.q.put(chunk)

    # ---- consumer side ----
    def __iter__(self):
        return self

    def __next__(self) -> torch.Tensor:
        item = self.q.get(timeout=self.timeout)
        if item is None:
            raise StopIteration
        return item

    def waveform(self) -> torch.Tensor:
        """every chunk so far concatenated: [B, samples]"""
        return torch.cat(self.chunks, dim=-1) if self.chunks else torch.zeros(0)
