"""Steady-state sweep of the e2e driver's decode bucket size (tacotron2_amd/do_syn_e2e.py) on one MI355X.

Each (max_batch, width) setting runs the corpus twice: the first pass captures the producer's decode step for every
bucket shape, the second is the figure printed.  Columns: max_batch, width, audio-s/s, decode seconds, vocoder seconds,
vocoder calls.

    python tools/e2e_sweep.py [--n 1024] [--settings 64:128,128:128,256:256]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tacotron2_amd import do_syn_batch, do_syn_e2e  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--settings", default="64:128,128:128,128:256,256:256")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    producer = do_syn_e2e.build_producer(dev)
    postnet = do_syn_e2e.build_postnet(dev, "bf16")
    voc = do_syn_batch.build_vocoder("synthetic-hifigan", dev, "bf16")
    texts, steps = do_syn_e2e.synthetic_texts(args.n, 1234)
    for setting in args.settings.split(","):
        mb, w = (int(x) for x in setting.split(":"))
        for _ in range(2):
            _, st = do_syn_e2e.synthesize_shard(producer, postnet, voc, texts, steps, device=dev, max_batch=mb, width=w)
        print(mb, w, round(st["audio_seconds"] / st["wall"], 1), round(st["t_tc2"], 3), round(st["t_voc"], 3),
              st["calls"], flush=True)


if __name__ == "__main__":
    main()
