#!/usr/bin/env python3
"""bench.py -- audio-seconds synthesized per wall-second (22.05 kHz) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--model hifigan|waveglow]

Workload (BASELINE.json configs[1]): HiFi-GAN V1 generator, one batch of
32 synthetic 80-bin mels of 2-10 s (T_i = 2*U{86..430} frames, numpy default_rng(1234+rank)),
sorted by length, right-padded with the mel floor, vocoded in ONE call with a `lengths` vector
(tiles past a sequence's end exit at once, so padding costs no arithmetic).  A "step" is one such
call; weights are synthetic (tacotron2_amd.synth, seed 0); mels are resident in HBM before the
timed region.  Audio seconds are counted on TRUE lengths only.  For N > 1 every rank vocodes its
own batch (weak scaling, no data-path collective); RCCL only reduces the timing.

The JSON line also carries
  roofline      dominant MFMA kernel: algorithmic FLOP/launch / mean launch time from HIP events
                recorded by the library around every launch (mv_profile_begin/end)
  cpu_baseline  the oracle's PyTorch-CPU restatement of the reference generator (oracle/torch_ref.py)
                timed on this host on the configs[0] input (one 80x500 mel, batch 1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 22050.0
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default=os.environ.get("MELVOC_BENCH_DTYPE", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--model", default="hifigan", choices=["hifigan", "waveglow", "postnet"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    return ap.parse_args()


def make_batch(torch, B, seed, dev):
    from tacotron2_amd import synth

    lengths = np.sort(synth.synthetic_lengths(B, seed))[::-1].copy()
    Tmax = int(lengths.max())
    mel = np.full((B, 80, Tmax), synth.MEL_PAD_VALUE, np.float32)
    for b, T in enumerate(lengths):
        mel[b, :, :T] = synth.synthetic_mel(int(T), seed * 1000 + b)
    return torch.from_numpy(mel).to(dev), torch.from_numpy(lengths.astype(np.int32)).to(dev), lengths


def cpu_baseline(torch):
    """Reference-equivalent PyTorch CPU generator on configs[0]: one 80x500 mel, batch 1, fp32."""
    from oracle import torch_ref
    from tacotron2_amd import synth

    cores = os.cpu_count() or 1
    # the GPU box exports OMP_NUM_THREADS for its CPU share; honour it
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    torch.set_num_threads(threads)
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    mel = synth.synthetic_mel(500, 1234)[None]
    torch_ref.hifigan_forward(st, synth.HIFIGAN_V1, mel)  # warm-up
    times = []
    t_end = time.perf_counter() + 12.0
    while len(times) < 10 and (time.perf_counter() < t_end or len(times) < 3):
        t0 = time.perf_counter()
        torch_ref.hifigan_forward(st, synth.HIFIGAN_V1, mel)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(500 * 256 / SR / med, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
            "sample": f"oracle/torch_ref.py (PyTorch CPU fp32 restatement of hifigan/models.py Generator), one 80x500 "
                      f"synthetic mel, batch 1, median of {len(times)} calls, {threads} threads of {cores} host CPUs"}


def profiled_traffic(kernel_name, args):
    """HBM-side bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (profiles/*_traffic.json),
    valid for the default bf16 HiFi-GAN batch only; None when there is no matching measurement."""
    if args.model != "hifigan" or args.dtype != "bf16" or args.batch != 32:
        return None
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        table = json.load(f)
    for k, v in table.items():
        if not k.startswith("_") and kernel_name.startswith(k):
            return v["bytes_per_launch"]
    return None


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from tacotron2_amd import synth

    if args.model == "hifigan":
        from tacotron2_amd.hifigan import AttrDict, Generator

        model = Generator(AttrDict(synth.HIFIGAN_V1), precision=args.dtype)
        sd = synth.hifigan_state(synth.HIFIGAN_V1, 0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval()
        model = model.to(dev)
        mel, lengths_d, lengths = make_batch(torch, args.batch, 1234 + rank, dev)
        run = lambda: model(mel, lengths=lengths_d)  # noqa: E731
        workload = (f"HiFi-GAN V1 (config_v1), batch {args.batch} synthetic 80-bin mels of 2-10 s, one "
                    f"length-masked call per step")
    elif args.model == "postnet":
        # the Tacotron2 Postnet (+ input) of model.py:717 on the same batch of mels; its
        # "audio seconds" are the seconds of speech the frames stand for (256 samples each)
        from tacotron2_amd.postnet import Postnet

        model = Postnet(synth.POSTNET_TC2, 0, precision=args.dtype)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.postnet_state(synth.POSTNET_TC2, 0).items()},
                              strict=False)
        model = model.eval().to(dev)
        mel, lengths_d, lengths = make_batch(torch, args.batch, 1234 + rank, dev)
        run = lambda: model(mel, lengths=lengths_d, add_input=True)  # noqa: E731
        workload = (f"Tacotron2 Postnet (5 x Conv1d k5 512 ch + folded BatchNorm + tanh, + input), batch "
                    f"{args.batch} synthetic 80-bin mels of 2-10 s")
    else:
        from tacotron2_amd.waveglow import WaveGlow

        model = WaveGlow(**synth.WAVEGLOW_512)
        model = WaveGlow.remove_weightnorm(model)
        sd = synth.waveglow_state(synth.WAVEGLOW_512, 0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.eval().to(dev)
        model.precision = args.dtype
        B = 1 if args.batch == 32 else args.batch
        lengths = np.full(B, 500, np.int32)
        mel = torch.from_numpy(np.stack([synth.synthetic_mel(500, 1234 + rank * 100 + b) for b in range(B)])).to(dev)
        torch.manual_seed(1234)
        run = lambda: model.infer(mel, sigma=0.666)  # noqa: E731
        workload = f"WaveGlow 12 flows x WN 8 layers x 512 ch, {B} x 80x500 mel, sigma 0.666"
    frames = float(lengths.sum())
    audio_s = frames * 256 / SR

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        a = torch.tensor([audio_s], device=dev, dtype=torch.float64)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        total_audio = float(a.item())
    else:
        total_audio = audio_s
    value = total_audio * args.steps / dt

    roofline = None
    if rank == 0 and not args.no_profile:
        h = model.native_handle(dev)
        nprof = max(1, min(args.steps, 5))
        h.profile_begin()
        for _ in range(nprof):
            run()
        torch.cuda.synchronize()
        rows = h.profile_end()
        rows.sort(key=lambda r: -r["ms"])
        if os.environ.get("MELVOC_BENCH_ROWS"):
            for r in rows:
                print("# %-48s launches/step %3d  ms/step %8.3f  TFLOP/s %7.1f" % (
                    r["name"], r["launches"] // nprof, r["ms"] / nprof,
                    2.0 * r["macs_per_frame"] * frames / max(r["ms"], 1e-9) / 1e9), file=sys.stderr)
        top = rows[0]
        flops = 2.0 * top["macs_per_frame"] * frames  # over all recorded launches of that kernel
        achieved = flops / (top["ms"] * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        all_ms = sum(r["ms"] for r in rows)
        all_flops = sum(2.0 * r["macs_per_frame"] * frames for r in rows)
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": profiled_traffic(top["name"], args), "kernel": top["name"],
                    "launches_per_step": top["launches"] // nprof,
                    "avg_launch_us": round(top["ms"] * 1e3 / top["launches"], 2),
                    "flop_per_launch": round(flops / top["launches"]),
                    "share_of_mfma_kernel_time": round(top["ms"] / all_ms, 3),
                    "all_conv_kernels_tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2)}

    if rank == 0:
        macs = model.macs_per_frame(dev)
        peak_audio = PEAK_TFLOPS[args.dtype] * 1e12 / (2.0 * macs * SR / 256)
        line = {
            "metric": "audio-seconds synthesized per wall-second (22.05 kHz)",
            "value": round(value, 2),
            "unit": "audio-s/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic mels (clip(N(-5,2^2),-12.59,2.5)), synthetic fan-in-scaled weights (seed 0)",
            "config": {"workload": workload, "audio_s_per_step_per_gpu": round(audio_s, 3),
                       "frames_per_step_per_gpu": int(frames), "parallelism": f"replicated weights, {world} rank(s), "
                       "utterances sharded, no data-path collective"},
            "gflop_per_audio_s": round(2.0 * macs * SR / 256 / 1e9, 3),
            "frac_of_mfma_roofline": round(value / world / peak_audio, 4),
            "roofline": roofline,
            "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(torch),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
