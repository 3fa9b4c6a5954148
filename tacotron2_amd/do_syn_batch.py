"""Batch vocoding driver: the vocoder half of do_syn.py (do_syn.py:144-174,385-396,484) over a corpus
of mels, length-bucketed and sharded by utterance over the ranks of one node.

    python -m tacotron2_amd.do_syn_batch --vocoder synthetic-hifigan --n 10000 --dtype bf16
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m tacotron2_amd.do_syn_batch --vocoder path/to/hifigan_ckpt.pth.tar --mels DIR --out_dir OUT

`--vocoder` is a checkpoint path (kind by the reference's substring rule) or `synthetic-hifigan` /
`synthetic-waveglow` (deterministic random weights, BASELINE.json config 4).  `--mels DIR` reads the
reference's parameter files (16-byte int32 header frames, dim, num, den + float32 frames:
README.md:14-17, do_syn.py:468-478); without it a synthetic corpus of 2-10 s mels is used.
With `--out_dir` each utterance is written as mono 16-bit WAV (do_syn.py:285,394); `--seg` adds the `.seg`
label track and `--parameter_files` the vocoded frames in the reference's format (do_syn.py:286,396,466-478).
One JSON line of gathered statistics is printed by rank 0; RCCL is used for that gather only.
"""
from __future__ import annotations

import argparse
import glob
import json
import os

import numpy as np
import torch

from . import batch, outputs, synth, vocoder


def read_parameter_file(path):
    """reference parameter file -> (mel [dim, frames], (num, den)); format in tacotron2_amd/outputs.py"""
    hdr = np.fromfile(path, dtype=np.int32, count=4)
    frames, _ = outputs.read_parameter_file(path)
    return np.ascontiguousarray(frames.T), (int(hdr[2]), int(hdr[3]))


def write_parameter_file(path, mel, num=22050, den=256):
    """mel [dim, frames] -> reference parameter file at num/den frames per second"""
    outputs.write_parameter_file(path, np.asarray(mel, np.float32).T, num / den)


def build_vocoder(name, device, dtype, unsafe_pickle=False):
    if name == "synthetic-hifigan":
        from .hifigan import AttrDict, Generator

        g = Generator(AttrDict(synth.HIFIGAN_V1), precision=dtype)
        g.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state(synth.HIFIGAN_V1, 0).items()})
        g.eval()
        return vocoder.Vocoder("HIFIGAN", g.to(device))
    if name == "synthetic-waveglow":
        from .waveglow import WaveGlow

        m = WaveGlow.remove_weightnorm(WaveGlow(**synth.WAVEGLOW_512))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.waveglow_state(synth.WAVEGLOW_512, 0).items()})
        m.eval().to(device)
        m.precision = dtype
        return vocoder.Vocoder("WAVEGLOW", m, sigma=0.666)
    v = vocoder.load_vocoder(name, device=device, precision=dtype, unsafe_pickle=unsafe_pickle)
    if v is None:
        raise SystemExit(1)
    return v


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--vocoder", default="synthetic-hifigan")
    ap.add_argument("--mels", default=None, help="directory of parameter files (*.WAVEGLOW / *.mel)")
    ap.add_argument("--n", type=int, default=1000, help="synthetic corpus size when --mels is not given")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "fp32", "bf16x3"])
    ap.add_argument("--unsafe_pickle", action="store_true",
                    help="load a WaveGlow checkpoint as the reference does (torch.load weights_only=False, do_syn.py:148) "
                         "instead of under the allow-listed unpickler: only for files you trust")
    ap.add_argument("--max_batch", type=int, default=32)
    ap.add_argument("--width", type=int, default=64, help="max frame spread inside a bucket")
    ap.add_argument("--out_dir", default=None)
    ap.add_argument("--fe_wav", type=int, default=22050)
    ap.add_argument("--device_mels", action="store_true",
                    help="synthetic corpus only: generate each bucket's mels on the device (BASELINE config 4), no H2D")
    ap.add_argument("--no_d2h", action="store_true", help="leave the audio on the device (kernel-side rate)")
    ap.add_argument("--no_pipeline", action="store_true", help="pad / upload / vocode / download strictly in turn")
    ap.add_argument("--seg", action="store_true", help="with --out_dir: also write the .seg label track")
    ap.add_argument("--parameter_files", action="store_true", help="with --out_dir: also write the frames (.WAVEGLOW)")
    args = ap.parse_args(argv)

    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("do_syn_batch needs an MI355X; the native vocoders have no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    voc = build_vocoder(args.vocoder, dev, args.dtype)
    names = None
    if args.mels:
        names = sorted(glob.glob(os.path.join(args.mels, "*.WAVEGLOW")) + glob.glob(os.path.join(args.mels, "*.mel")))
        mels = [read_parameter_file(p)[0] for p in names]
        lengths = [m.shape[1] for m in mels]
    else:
        mels, lengths = batch.synthetic_corpus(args.n, args.seed)
    if not lengths or max(lengths) <= 0:
        raise SystemExit("do_syn_batch: no utterances (empty --mels directory or --n 0)")
    # warm-up at the largest bucket's shape, so weight upload and the growth of the workspace to its final size stay
    # out of the timing
    bk = max(batch.make_buckets(lengths, args.max_batch, args.width), key=lambda b: len(b.indices) * b.tmax)
    voc(torch.full((len(bk.indices), mels[0].shape[0], bk.tmax), synth.MEL_PAD_VALUE, device=dev), lengths=bk.lengths)
    keep = args.out_dir is not None
    if args.device_mels and args.mels:
        raise SystemExit("--device_mels is for the synthetic corpus")
    res, st = batch.synthesize_shard(voc, mels, lengths, rank=rank, world=world, device=dev, fe_wav=args.fe_wav,
                                     hop=voc.hop, max_batch=args.max_batch, width=args.width, keep_audio=keep,
                                     pipeline=not args.no_pipeline, d2h=not args.no_d2h,
                                     device_mels=args.seed if args.device_mels else None)
    if keep:
        os.makedirs(args.out_dir, exist_ok=True)
        for i, pcm in res.items():
            base = os.path.splitext(os.path.basename(names[i]))[0] if names else "utt%06d" % i
            stem = os.path.join(args.out_dir, base)
            if args.seg or args.parameter_files:
                w = outputs.UtteranceWriter(stem, fe_wav=args.fe_wav, fe_data=args.fe_wav / voc.hop,
                                            dim=mels[i].shape[0], parameter_file=args.parameter_files)
                w.add_chunk(pcm, np.asarray(mels[i], np.float32).T, base)
                w.close()
                if not args.seg:
                    os.remove(stem + ".seg")
            else:
                vocoder.write_wav(stem + ".wav", pcm, args.fe_wav)
    total = batch.gather_stats(st, dev)
    if rank == 0:
        print(json.dumps({
            "vocoder": voc.kind, "dtype": args.dtype, "n_gpus": world, "utterances": total.utterances,
            "audio_seconds": round(total.audio_seconds, 3), "wall_seconds": round(total.wall_seconds, 4),
            "audio_s_per_s": round(total.audio_seconds / total.wall_seconds, 2),
            "rtf_vocoder": round(total.wall_seconds / total.audio_seconds, 6), "calls": total.calls,
            "checksum": total.checksum,
            "timing": "wall, device-synchronised; mel %s; PCM %s; %s" % (
                "generated on the device" if args.device_mels else "padded on the host + H2D",
                "left on the device" if args.no_d2h else "D2H into pinned memory",
                "serial" if args.no_pipeline else "3-stage pipeline (prep / copy-in / compute / copy-out overlap)"),
            "per_rank_wall": [round(p.wall_seconds, 4) for p in total.per_rank]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
