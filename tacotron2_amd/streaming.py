"""Time-tiled (streaming) HiFi-GAN vocoding for the interactive driver (do_tts.py:347-353
vocodes a whole text part before the first sample can be played).

A chunk of `chunk_frames` mel frames is vocoded together with `halo` frames of context on either side and only
the samples of the chunk itself are kept.  With halo >= `receptive_halo_frames(cfg)` every kept sample sees
exactly the inputs it sees in the whole-utterance call (zero padding only at the true ends of the utterance), so
the concatenation of the chunks equals `Generator.forward(mel)`; tests/test_gpu_streaming.py checks that.

    for wav in stream_hifigan(gen, mel, chunk_frames=64):     # wav: [chunk*256] float32 (or int16 with out="pcm16")
        play(wav)

    python -m tacotron2_amd.streaming          # first-audio latency vs whole-utterance latency on cuda:0
"""
from __future__ import annotations

import math
from fractions import Fraction

import torch


def receptive_halo_frames(cfg) -> int:
    """Mel frames of context one output sample of Generator.forward depends on, per side (models.py:118-163):
    conv_pre (k7) + per stage [transposed conv: ceil(k/stride) - 1 input rows; MRF: the widest ResBlock,
    (k-1)/2 * sum(d+1) rows] + conv_post (k7), each converted to frames at the rate it runs at."""
    need = Fraction(3)
    rate = Fraction(1)
    for u, k in zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"]):
        need += Fraction(math.ceil(k / u) - 1) / rate
        rate *= u
        rb = max((kr - 1) // 2 * sum(d + 1 for d in dil)
                 for kr, dil in zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]))
        need += Fraction(rb) / rate
    need += Fraction(3) / rate
    return int(math.ceil(need)) + 1


def stream_hifigan(gen, mel, chunk_frames=64, halo=None, out="float"):
    """mel [80,T] or [1,80,T] on the generator's device -> yields the waveform chunk by chunk."""
    if mel.dim() == 3:
        if mel.size(0) != 1:
            raise ValueError("stream_hifigan takes one utterance")
        mel = mel[0]
    T = mel.size(1)
    hop = gen.hop
    h = receptive_halo_frames(gen.h) if halo is None else int(halo)
    for f0 in range(0, T, chunk_frames):
        f1 = min(T, f0 + chunk_frames)
        a, b = max(0, f0 - h), min(T, f1 + h)
        wav = gen(mel[:, a:b].contiguous()[None], out=out)[0, 0]
        yield wav[(f0 - a) * hop:(f1 - a) * hop]


def first_audio_latency(gen, mel, chunk_frames=64, repeats=5):
    """(seconds until the first chunk is on the host, seconds for the whole utterance in one call), medians."""
    import time

    def timed(fn):
        ts = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    first = timed(lambda: next(iter(stream_hifigan(gen, mel, chunk_frames, out="pcm16"))).cpu())
    whole = timed(lambda: gen(mel[None] if mel.dim() == 2 else mel, out="pcm16").cpu())
    return first, whole


def main():
    import json

    import numpy as np

    from . import synth
    from .hifigan import AttrDict, Generator

    dev = torch.device("cuda:0")
    out = {}
    for prec in ("bf16", "fp32"):
        g = Generator(AttrDict(synth.HIFIGAN_V1), precision=prec)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
        g = g.eval().to(dev)
        mel = torch.from_numpy(synth.synthetic_mel(861, 7)).to(dev)  # 10 s
        for _ in stream_hifigan(g, mel):
            pass
        g(mel[None])
        first, whole = first_audio_latency(g, mel)
        out[prec] = {"first_chunk_ms": round(first * 1e3, 3), "whole_utterance_ms": round(whole * 1e3, 3),
                     "chunk_audio_s": round(64 * 256 / 22050, 3), "utterance_audio_s": round(861 * 256 / 22050, 3)}
    out["halo_frames"] = receptive_halo_frames(synth.HIFIGAN_V1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
