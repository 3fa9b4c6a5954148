"""Streaming (time-tiled) HiFi-GAN equals the whole-utterance call (tacotron2_amd/streaming.py)."""
import numpy as np
import pytest
import torch

from tacotron2_amd import streaming, synth
from tacotron2_amd.hifigan import AttrDict, Generator

pytestmark = pytest.mark.gpu


def _gen(prec):
    g = Generator(AttrDict(synth.HIFIGAN_V1), precision=prec)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
    return g.eval().to("cuda:0")


def test_halo_of_config_v1():
    # 3 + (1 + 60/8) + (1/8 + 60/64) + (1/64 + 60/128) + (1/128 + 60/256) + 3/256 = 13.3 -> 14, + 1 margin
    assert streaming.receptive_halo_frames(synth.HIFIGAN_V1) == 15


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("bf16", 1e-5)])
def test_chunks_concatenate_to_the_whole_call(prec, tol):
    g = _gen(prec)
    mel = torch.from_numpy(synth.synthetic_mel(203, 11)).cuda()
    whole = g(mel[None])[0, 0].cpu().numpy()
    parts = [w.cpu().numpy() for w in streaming.stream_hifigan(g, mel, chunk_frames=48)]
    got = np.concatenate(parts)
    assert [len(p) for p in parts] == [48 * 256] * 4 + [11 * 256]
    assert got.shape == whole.shape
    # identical inputs reach every kept sample, so the arithmetic is the same element by element
    assert np.abs(got - whole).max() <= tol


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_chunks_against_the_reference_golden(prec, golden_dir):
    """Pinned to the reference, not to this library's own whole call: the concatenated chunks of the 80x500 golden
    mel against the waveform the reference Generator produced for it (tests/golden/hifigan_v1_T500.npz)."""
    import os

    d = np.load(os.path.join(golden_dir, "hifigan_v1_T500.npz"))
    g = _gen(prec)
    mel = torch.from_numpy(synth.synthetic_mel(500, int(d["mel_seed"]))).cuda()
    got = np.concatenate([w.cpu().numpy() for w in streaming.stream_hifigan(g, mel, chunk_frames=64)])
    ref = d["wav"].astype(np.float64)
    assert got.shape == ref.shape
    if prec == "fp32":
        assert np.abs(got - ref).max() <= 1e-3
    else:
        assert 10 * np.log10((ref ** 2).sum() / ((ref - got) ** 2).sum()) >= 40.0


def test_too_small_a_halo_is_visible():
    g = _gen("fp32")
    mel = torch.from_numpy(synth.synthetic_mel(160, 3)).cuda()
    whole = g(mel[None])[0, 0].cpu().numpy()
    got = np.concatenate([w.cpu().numpy() for w in streaming.stream_hifigan(g, mel, chunk_frames=40, halo=2)])
    assert np.abs(got - whole).max() > 1e-3


def test_pcm16_chunks():
    g = _gen("bf16")
    mel = torch.from_numpy(synth.synthetic_mel(100, 5)).cuda()
    whole = g(mel[None], out="pcm16")[0, 0].cpu().numpy()
    got = np.concatenate([w.cpu().numpy() for w in streaming.stream_hifigan(g, mel, chunk_frames=64, out="pcm16")])
    assert got.dtype == np.int16 and np.array_equal(got, whole)
