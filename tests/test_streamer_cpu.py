"""AudioStreamer's window logic on the CPU, with a stand-in codec (a finite filter over frames with zero padding at the clip
ends - what makes a chunked decode differ from a one-shot decode if windows are cut wrongly): chunks must add up to the
per-utterance one-shot decode, also when utterances finish at different frames (the reference drops frames holding special
ids per utterance, modeling_parler_tts.py:2668-2670, and zero-pads the shorter clips, :2676-2680)."""
import types

import pytest
import torch

from parler_tts_amd.streamer import AudioStreamer

K, HOP, SIZE, PAD, BOS = 3, 4, 64, 64, 65


class _Codec:
    def decode(self, codes, scales):
        c = codes[0].float()                                   # [B, K, F]
        v = c.sum(dim=1)                                       # per-frame value
        z = torch.zeros_like(v[:, :1])
        prev, nxt = torch.cat([z, v[:, :-1]], 1), torch.cat([v[:, 1:], z], 1)
        frame = v + 0.5 * prev + 0.25 * nxt + 1.0              # + 1: even an all-zero frame is not silence
        wav = frame.repeat_interleave(HOP, dim=1) + torch.arange(HOP).repeat(v.shape[1])[None] * 0.01
        return types.SimpleNamespace(audio_values=wav[:, None, :])


class _Model:
    device = torch.device("cpu")
    audio_encoder = _Codec()
    config = types.SimpleNamespace(decoder=types.SimpleNamespace(num_codebooks=K),
                                   audio_encoder=types.SimpleNamespace(upsampling_ratios=(2,), hop_length=HOP, codebook_size=SIZE))


def _columns(codes):
    """the loop's id columns for codes [B, K, F]: codebook k of frame f sits in column f + k + 1 (delay pattern)"""
    B, _, F = codes.shape
    cols = [torch.full((B * K,), BOS, dtype=torch.long)]
    for c in range(1, F + K):
        col = torch.full((B, K), PAD, dtype=torch.long)
        for k in range(K):
            f = c - k - 1
            col[:, k] = codes[:, k, f] if 0 <= f < F else (BOS if f < 0 else PAD)
        cols.append(col.reshape(-1))
    return cols


@pytest.mark.parametrize("ends", [(80, 80, 80), (80, 50, 23), (80, 0, 79)])
def test_chunks_equal_the_per_utterance_one_shot_decode(ends):
    g = torch.Generator().manual_seed(3)
    F = 80
    codes = torch.randint(0, SIZE, (3, K, F), generator=g)
    for b, fb in enumerate(ends):
        codes[b, 0, fb:] = PAD                                  # codebook 0 finished at frame fb: every later frame is special
        if fb + 2 < F:
            codes[b, 1:, fb + 2:] = PAD                         # the other codebooks a little later
    st = AudioStreamer(_Model(), play_frames=5)
    for col in _columns(codes):
        st.put(col)
    st.end()
    got = st.waveform()
    assert got.shape == (3, F * HOP)
    codec = _Codec()
    for b, fb in enumerate(ends):
        want = torch.zeros(F * HOP)
        if fb > 0:
            want[: fb * HOP] = codec.decode(codes[b:b + 1, :, :fb][None], None).audio_values[0, 0]
        assert torch.equal(got[b], want), (b, fb, int((got[b] != want).nonzero()[0]))
    assert len(st.chunks) > 5 and list(st)[-1] is not None


def test_expanded_rows_are_refused():
    st = AudioStreamer(_Model(), play_frames=5)
    st.put(torch.full((2 * K,), BOS))
    with pytest.raises(ValueError):
        st.put(torch.zeros(4 * K, dtype=torch.long))
