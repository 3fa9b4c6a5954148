#!/usr/bin/env python3
"""Probe: how much of WaveGlow's / HiFi-GAN's step time is launch gaps?  Captures one forward in a HIP graph
(torch.cuda.CUDAGraph) and times replays against eager calls.  usage: tools/graph_probe.py [waveglow|hifigan]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tacotron2_amd import synth  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "waveglow"
    dev = torch.device("cuda:0")
    if which == "waveglow":
        from tacotron2_amd.waveglow import WaveGlow
        m = WaveGlow.remove_weightnorm(WaveGlow(**synth.WAVEGLOW_512))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(synth.WAVEGLOW_512, 0).items()})
        m.eval().to(dev)
        m.precision = "bf16"
        mel = torch.from_numpy(synth.synthetic_mel(500, 1234))[None].to(dev)
        noise = [torch.from_numpy(n).to(dev) for n in synth.waveglow_noise(synth.WAVEGLOW_512, 1, 500, 1)]
        run = lambda: m.infer(mel, sigma=0.666, noise=noise)  # noqa: E731
        audio = 500 * 256 / 22050
    else:
        from tacotron2_amd.hifigan import Generator, AttrDict
        m = Generator(AttrDict(synth.HIFIGAN_V1), precision="bf16")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
        m.eval().to(dev)
        lens = np.sort(synth.synthetic_lengths(32, 1234))[::-1].copy()
        T = int(lens.max())
        x = torch.full((32, 80, T), synth.MEL_PAD_VALUE)
        for b, n in enumerate(lens):
            x[b, :, :n] = torch.from_numpy(np.asarray(synth.synthetic_mel(int(n), 1234 + b)))
        mel = x.to(dev)
        ln = torch.from_numpy(lens.astype(np.int32)).to(dev)
        run = lambda: m(mel, lengths=ln)  # noqa: E731
        audio = float(lens.sum()) * 256 / 22050

    def timed(f, n=20):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    te = timed(run)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            y = run()
    tg = timed(g.replay)
    print(f"{which}: eager {te * 1e3:.3f} ms/step ({audio / te:.1f} audio-s/s), graph replay {tg * 1e3:.3f} ms/step "
          f"({audio / tg:.1f} audio-s/s), out {tuple(y.shape)}")


if __name__ == "__main__":
    main()
