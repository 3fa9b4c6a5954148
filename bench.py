#!/usr/bin/env python3
"""bench.py -- audio-seconds synthesized per wall-second (22.05 kHz) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|fp32] [--model hifigan|waveglow|postnet]

Headline workload (BASELINE.json configs[1]): HiFi-GAN V1 generator, one batch of
32 synthetic 80-bin mels of 2-10 s (T_i = 2*U{86..430} frames, numpy default_rng(1234+rank)), sorted by
length, right-padded with the mel floor, vocoded in ONE call with a `lengths` vector (windows past a
sequence's end are never visited, so padding costs no arithmetic).  A "step" is one such call; weights are
synthetic (tacotron2_amd.synth, seed 0); mels are resident in HBM before the timed region.  Audio seconds are
counted on TRUE lengths only.

Ranks: with WORLD_SIZE in the environment (torch.distributed.run) this process is one rank.  Without it,
`--gpus N` with N > 1 makes this process a launcher: it starts N fresh rank processes (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous) before anything touches a GPU, relays rank 0's JSON line and
exits non-zero if a rank fails or the node has fewer than N devices.  Every rank vocodes its own batch (weak
scaling, no data-path collective); RCCL only carries the barrier and the max-time / sum-audio reduction.

The JSON line carries, besides the contract's fields,
  roofline      dominant MFMA kernel BY ROCPROF SYMBOL: algorithmic FLOP of all its launches / their summed HIP-event
                time (recorded by the library around every launch on the launch stream, mv_profile_begin/end), so the
                figure can be re-derived from profiles/*_kernel_stats.csv; `by_detail` splits the symbol's launches
  cpu_baseline  the oracle's PyTorch-CPU restatement of the reference generator (oracle/torch_ref.py) timed on
                this host on the configs[0] input (one 80x500 mel, batch 1); rank 0, N = 1 only
  fp32          the same batch through MV_PREC_FP32 (the reference's own arithmetic, exact-fp32 MFMA; configs[1]
                "bf16 and fp32"); its `bf16x3` sub-record is the same batch through MV_PREC_BF16X3 -- fp32-grade
                results (<= 1e-3, the fp32 bar) from split bf16 operands, priced against bf16 peak / 3
  waveglow      BASELINE configs[2]: WaveGlow 12 flows x 8 layers x 512 ch, 80x500 mels, sigma 0.666, noise from
                torch.manual_seed(1234); records B1 and B8 (bf16 operands) and B1_fp32
  f16           the headline batch through MV_PREC_F16 (float16 operands, the bf16 kernels); snr_db_vs_fp32_mode = worst
                per-utterance SNR of bf16 / f16 on the batch against the library's fp32 mode
  sharded       configs[3] on the ranks of this run (every N, N = 1 included): 10 000 utterances bucketed, dealt to the
                ranks longest-processing-time-first, host mels -> PCM16 on the host; value = audio seconds / slowest rank
  e2e           configs[4] on this one GPU: batched Tacotron2-shaped decode (PyTorch-ROCm) -> native Postnet -> native
                vocoder, Tc2 and vocoder seconds separately
(all but `sharded` at N = 1 only; --no-extras skips them all).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 22050.0
# dense MFMA peaks, MI355X_MICROARCH.md.  bf16x3 (fp32-grade results from split bf16 operands, melvoc.h) executes three
# bf16 MFMAs per algorithmic multiply-accumulate: its ceiling for ALGORITHMIC flops is a third of the bf16 peak.
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "fp32", "bf16x3"])
    ap.add_argument("--model", default="hifigan", choices=["hifigan", "waveglow", "postnet"])
    ap.add_argument("--batch", type=int, default=None, help="utterances per step (default 32; waveglow 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 / f16 / waveglow / sharded / e2e records")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded-corpus record (configs[3])")
    ap.add_argument("--rows", action="store_true", help="per-kernel table on stderr")
    ap.add_argument("--options", default=None,
                    help='mv_options of the benched model as JSON, e.g. \'{"wg_gemm": 0}\' (A/B of kernel structures)')
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reduce path only: gloo on the CPU, the step is a no-op (tests)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run
def count_gpus_without_hip(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPU nodes of this machine WITHOUT touching HIP in this process (the launcher must not have initialised the GPU when
    it starts its children): the KFD topology in sysfs lists every node with its SIMD count (CPUs have none) -- narrowed
    by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when they are set; if sysfs is not readable, a throw-away child process
    asks torch."""
    import glob

    n = 0
    found = False
    for path in glob.glob(os.path.join(topology, "*", "properties")):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        found = True
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    if not found:
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                 capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
            n = int(out[-1]) if out else 0
        except (OSError, ValueError, subprocess.TimeoutExpired):
            n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launch_ranks(args, argv, deadline_s=1500.0):
    """Start one rank process per GPU, relay rank 0's line.  Every child is polled: the first one that exits non-zero
    ends the run (the others are killed, by PID) instead of leaving rank 0 to sit in the rendezvous until its
    store times out; an overall deadline bounds the whole launch."""
    if not args.dry_run:
        have = count_gpus_without_hip()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node shows {have} device(s)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        # (the environment is inherited as is: HSA_ENABLE_IPC_MODE_LEGACY and friends are the box's business)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained by a thread so that polling never blocks on a full pipe
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t_end = time.time() + deadline_s
    rcs = [None] * len(procs)
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = f"rank {r} exited with {rcs[r]}"
        if failed is None and time.time() > t_end:
            failed = f"deadline of {deadline_s:.0f} s passed"
        if failed is not None:
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()  # this exact child
                    rcs[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(10)
    sys.stdout.write("".join(x or "" for x in out0))
    sys.stdout.flush()
    if failed is not None or any(rc != 0 for rc in rcs):
        print(f"bench.py: {failed or 'ranks failed'}; exit codes {rcs}", file=sys.stderr)
        return 1
    return 0


# ---------------------------------------------------------------------------------------------------------
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


def profiled_traffic(symbol, model, dtype, batch):
    """HBM-side bytes per launch of `symbol` from the committed rocprofv3 PMC passes (newest profiles/*_traffic.json
    whose `_workload` matches); None when there is no matching measurement."""
    import glob

    want = f"{model}/{dtype}/B{batch}"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        with open(path) as f:
            table = json.load(f)
        if table.get("_workload") != want:
            continue
        v = table.get(symbol)
        if v:
            return v["bytes_per_launch"]
    return None


def build_model(torch, model, dtype, batch, rank, dev, options=None):
    """-> (module, run(), frames per step, workload string, batch)"""
    from tacotron2_amd import synth

    if model == "hifigan":
        from tacotron2_amd.hifigan import AttrDict, Generator

        B = batch or 32
        m = Generator(AttrDict(synth.HIFIGAN_V1), precision=dtype, options=options)
        sd = synth.hifigan_state(synth.HIFIGAN_V1, 0)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m.eval()
        m = m.to(dev)
        mel, lengths_d, lengths = make_batch(torch, B, 1234 + rank, dev)
        run = lambda: m(mel, lengths=lengths_d)  # noqa: E731
        workload = f"HiFi-GAN V1 (config_v1), batch {B} synthetic 80-bin mels of 2-10 s, one length-masked call per step"
    elif model == "postnet":
        # the Tacotron2 Postnet (+ input) of model.py:717 on the same batch of mels; its
        # "audio seconds" are the seconds of speech the frames stand for (256 samples each)
        from tacotron2_amd.postnet import Postnet

        B = batch or 32
        m = Postnet(synth.POSTNET_TC2, 0, precision=dtype)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.postnet_state(synth.POSTNET_TC2, 0).items()},
                          strict=False)
        m = m.eval().to(dev)
        mel, lengths_d, lengths = make_batch(torch, B, 1234 + rank, dev)
        run = lambda: m(mel, lengths=lengths_d, add_input=True)  # noqa: E731
        workload = (f"Tacotron2 Postnet (5 x Conv1d k5 512 ch + folded BatchNorm + tanh, + input), batch {B} "
                    f"synthetic 80-bin mels of 2-10 s")
    else:
        from tacotron2_amd.waveglow import WaveGlow

        B = batch or 1
        m = WaveGlow(**synth.WAVEGLOW_512)
        m = WaveGlow.remove_weightnorm(m)
        sd = synth.waveglow_state(synth.WAVEGLOW_512, 0)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        del sd
        m.eval().to(dev)
        m.precision = dtype
        m.options = options
        lengths = np.full(B, 500, np.int32)
        mel = torch.from_numpy(np.stack([synth.synthetic_mel(500, 1234 + rank * 100 + b) for b in range(B)])).to(dev)
        torch.manual_seed(1234)
        run = lambda: m.infer(mel, sigma=0.666)  # noqa: E731
        workload = (f"WaveGlow 12 flows x WN 8 layers x 512 ch, {B} x 80x500 mel, sigma 0.666, noise drawn by torch "
                    f"on the device after manual_seed(1234)")
    return m, run, float(lengths.sum()), workload, B


def waveglow_reference_macs_per_frame(cfg):
    """Multiply-accumulates per mel frame of WaveGlow.infer in the REFERENCE's formulation (SURVEY.md section 8a rows
    W1-W10: cond_layer, in_layers, res_skip with its skip halves, end, 1x1 convs, upsampler) -- 1.4407 TFLOP per audio
    second at 512 channels.  The library's bf16 mode executes less (folded skip path); mv_macs_per_frame reports that."""
    wn = cfg["WN_config"]
    C, nl, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
    ng, nm = cfg["n_group"], cfg["n_mel_channels"]
    ncond = nm * ng
    per_step, n_half, n_rem = 0.0, ng // 2, ng
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        per_step += (n_half * C + ncond * 2 * C * nl + nl * 2 * C * C * ks + (nl - 1) * 2 * C * C + C * C + C * 2 * n_half
                     + n_rem * n_rem)
    return per_step * (256 // ng) + nm * nm * 1024


def roofline_of(torch, m, run, frames, dev, model, dtype, batch, nprof, show_rows):
    h = m.native_handle(dev)
    h.profile_begin()
    for _ in range(nprof):
        run()
    torch.cuda.synchronize()
    rows = h.profile_end()
    if not rows:
        return None
    by_sym = {}
    for r in rows:
        by_sym.setdefault(r["name"], []).append(r)
    tot = lambda rs, k: sum(r[k] for r in rs)  # noqa: E731
    if show_rows:
        for r in sorted(rows, key=lambda r: -r["ms"]):
            print("# %-86s %-16s launches/step %3d  ms/step %8.3f  TFLOP/s %7.1f" % (
                r["name"][:86], r["detail"], r["launches"] // nprof, r["ms"] / nprof,
                2.0 * r["macs_per_frame"] * frames / max(r["ms"], 1e-9) / 1e9), file=sys.stderr)
    sym, rs = max(by_sym.items(), key=lambda kv: tot(kv[1], "ms"))
    ms, launches = tot(rs, "ms"), tot(rs, "launches")
    flops = 2.0 * tot(rs, "macs_per_frame") * frames  # algorithmic, over all recorded launches of the symbol
    achieved = flops / (ms * 1e-3) / 1e12
    peak = PEAK_TFLOPS[dtype]
    all_ms = tot(rows, "ms")
    all_flops = 2.0 * tot(rows, "macs_per_frame") * frames
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": profiled_traffic(sym, model, dtype, batch), "kernel": sym,
            "launches_per_step": launches // nprof, "avg_launch_us": round(ms * 1e3 / launches, 2),
            "flop_per_launch": round(flops / launches),
            "share_of_mfma_kernel_time": round(ms / all_ms, 3),
            "all_conv_kernels_tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2),
            "by_detail": [{"detail": r["detail"], "launches_per_step": r["launches"] // nprof,
                           "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                           "tflops": round(2.0 * r["macs_per_frame"] * frames / (r["ms"] * 1e-3) / 1e12, 2)}
                          for r in sorted(rs, key=lambda r: -r["ms"])]}


def timed(torch, run, steps, warmup, barrier):
    for _ in range(warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    barrier()
    return time.perf_counter() - t0


def side_record(torch, model, dtype, batch, dev, steps, warmup, show_rows):
    """One of the secondary records (N = 1): same measurement as the headline on another model / precision."""
    m, run, frames, workload, B = build_model(torch, model, dtype, batch, 0, dev)
    dt = timed(torch, run, steps, warmup, torch.cuda.synchronize)
    audio_s = frames * 256 / SR
    macs = m.macs_per_frame(dev)
    peak_audio = PEAK_TFLOPS[dtype] * 1e12 / (2.0 * macs * SR / 256)
    rec = {"value": round(audio_s * steps / dt, 2), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "warmup": warmup, "dtype": dtype, "workload": workload,
           "gflop_per_audio_s": round(2.0 * macs * SR / 256 / 1e9, 3),
           "frac_of_mfma_roofline": round(audio_s * steps / dt / peak_audio, 4),
           "roofline": roofline_of(torch, m, run, frames, dev, model, dtype, B, max(1, min(steps, 3)), show_rows)}
    if model == "waveglow":
        # the same throughput priced with the REFERENCE's formulation (1.4407 TFLOP per audio second),
        # beside the executed count above (bf16 mode folds the skip path and executes less)
        from tacotron2_amd import synth

        ref_macs = waveglow_reference_macs_per_frame(synth.WAVEGLOW_512)
        rec["gflop_per_audio_s_reference"] = round(2.0 * ref_macs * SR / 256 / 1e9, 3)
        rec["frac_of_mfma_roofline_reference_flops"] = round(
            audio_s * steps / dt * (2.0 * ref_macs * SR / 256) / (PEAK_TFLOPS[dtype] * 1e12), 4)
    del m, run
    torch.cuda.empty_cache()
    return rec


def sharded_record(torch, dist, dev, rank, world, dtype):
    """BASELINE configs[3] -- what north_star's scaling target is about -- on the ranks of this run: 10 000 synthetic
    utterances of 2-10 s, length-bucketed (<= 32 per call, <= 64 frames apart), buckets dealt to the ranks
    longest-processing-time-first on their frame totals (no data-path collective), and on every rank the whole host
    path: float32 mels in host memory -> pinned -> H2D -> vocode -> PCM16 -> D2H -> a per-utterance sink that is handed a
    view of the pinned slot (tacotron2_amd.batch.synthesize_shard).  value = sum of audio seconds / the slowest rank's
    wall time; never the headline `value` (it includes PCIe and host work).  Called by EVERY rank; rank 0 gets the record."""
    from tacotron2_amd import batch as vb
    from tacotron2_amd import do_syn_batch, synth

    n = 10000
    voc = do_syn_batch.build_vocoder("synthetic-hifigan", dev, dtype)
    lens = synth.synthetic_lengths(n, 1234)
    base = [synth.synthetic_mel(int(lens.max()), 7 + i) for i in range(16)]  # 16 distinct mels, cut to length: the host-side
    mels = [base[i % 16][:, : int(T)] for i, T in enumerate(lens)]  # generator is not what is being timed
    lengths = [int(t) for t in lens]
    seen = [0]

    def sink(i, pcm):  # what a WAV writer would be handed: a view of the pinned download slot
        seen[0] += int(pcm.shape[0])

    vb.synthesize_shard(voc, mels[:64], device=dev, sink=sink)  # warm-up (workspace growth, pinned slots)
    seen[0] = 0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _, st = vb.synthesize_shard(voc, mels, lengths, rank=rank, world=world, device=dev, sink=sink)
    assert seen[0] == st.frames * 256
    total = vb.gather_stats(st, device=dev)
    per = sorted(total.per_rank, key=lambda p: p.rank)
    frames = [p.frames for p in per]
    mean = sum(frames) / len(frames)
    return {"value": round(total.audio_seconds / total.wall_seconds, 2), "unit": "audio-s/s", "n_gpus": world,
            "utterances": total.utterances, "calls": total.calls, "audio_seconds": round(total.audio_seconds, 1),
            "wall_s": round(total.wall_seconds, 3), "per_rank_wall_s": [round(p.wall_seconds, 3) for p in per],
            "per_rank_audio_s": [round(p.audio_seconds, 1) for p in per],
            "lpt_load_spread": round((max(frames) - min(frames)) / mean, 5) if mean else 0.0,
            "checksum": total.checksum, "dtype": dtype, "scaling": "strong",
            "workload": "do_syn-style batch (BASELINE configs[3]): 10000 synthetic utterances of 2-10 s (lengths "
                        "synth.synthetic_lengths(10000, 1234)), host float32 mels -> int16 PCM in pinned host memory handed "
                        "to a per-utterance sink, length-bucketed (<= 32 per call), buckets dealt to the ranks "
                        "longest-processing-time-first, pad / H2D / vocode / D2H overlapped on every rank; value = total "
                        "audio seconds / slowest rank's wall"}


def snr_record(torch, dev):
    """Worst per-utterance SNR of the 16-bit-operand modes on the bench batch against the library's fp32 mode (which the
    goldens pin to the reference at <= 1e-3): BASELINE's bar is >= 40 dB."""
    from tacotron2_amd import synth

    lengths = np.sort(synth.synthetic_lengths(32, 1234))[::-1]  # the batch build_model makes for rank 0
    out, ref = {}, None
    for dtype in ("fp32", "bf16", "f16"):
        m, run, _, _, _ = build_model(torch, "hifigan", dtype, None, 0, dev)
        y = run()[:, 0].double()
        torch.cuda.synchronize()
        del m, run
        if ref is None:
            ref = y
            continue
        worst = 1e9
        for b, T in enumerate(lengths):
            n = int(T) * 256
            num = float((ref[b, :n] ** 2).sum())
            den = max(float(((ref[b, :n] - y[b, :n]) ** 2).sum()), 1e-300)
            worst = min(worst, 10.0 * np.log10(num / den))
        out[dtype] = round(float(worst), 2)
    torch.cuda.empty_cache()
    return out


def e2e_record(torch, dev):
    """BASELINE configs[4] on this one GPU: symbols -> batched Tacotron2-shaped decode (PyTorch-ROCm, imposed lengths)
    -> native Postnet -> native HiFi-GAN, per-stage seconds like do_syn.py:484 (tacotron2_amd/do_syn_e2e.py)."""
    from tacotron2_amd import do_syn_batch, do_syn_e2e

    producer = do_syn_e2e.build_producer(dev)
    postnet = do_syn_e2e.build_postnet(dev, "bf16")
    voc = do_syn_batch.build_vocoder("synthetic-hifigan", dev, "bf16")
    texts, steps = do_syn_e2e.synthetic_texts(1024, 1234)
    do_syn_e2e.synthesize_shard(producer, postnet, voc, texts[:512], [8] * 512, device=dev)  # warm-up
    do_syn_e2e.synthesize_shard(producer, postnet, voc, texts[:2], [max(steps)] * 2, device=dev)
    # first pass: every decode shape of the corpus is captured once (kept by the producer); second pass: a running service
    _, st1 = do_syn_e2e.synthesize_shard(producer, postnet, voc, texts, steps, device=dev)
    _, st = do_syn_e2e.synthesize_shard(producer, postnet, voc, texts, steps, device=dev)
    a = st["audio_seconds"]
    return {"value": round(a / st["wall"], 2), "first_pass_value": round(a / st1["wall"], 2), "unit": "audio-s/s",
            "utterances": st["utterances"],
            "audio_seconds": round(a, 2), "wall_s": round(st["wall"], 3), "tc2_s": round(st["t_tc2"], 3),
            "postnet_s": round(st["t_post"], 4), "vocoder_s": round(st["t_voc"], 4),
            "rtf_tc2": round(st["t_tc2"] / a, 6), "rtf_vocoder": round((st["t_voc"] + st["t_post"]) / a, 6),
            "workload": "1024 synthetic utterances of 24-120 symbols (1.7-10 s), tc2_french.yaml-shaped mel producer in "
                        "PyTorch-ROCm fp32 (batched decode, <= 512 per bucket, step replayed from a HIP graph captured once per "
                        "bucket shape, lengths imposed) -> native Postnet -> native HiFi-GAN V1 bf16 -> PCM16, all on the "
                        "device; value = second pass over the corpus, first_pass_value includes the captures"}


def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        # the launch / rendezvous / reduction path with no GPU: what tests/test_host_logic.py drives at world 2
        if os.environ.get("BENCH_DRY_RUN_FAIL_RANK") == str(rank):  # (test hook: a rank that dies before the rendezvous)
            return 3
        if world > 1:
            dist.init_process_group("gloo")
        t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        a = torch.tensor([1.0], dtype=torch.float64)
        if world > 1:
            dist.barrier()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(a, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"metric": "audio-seconds synthesized per wall-second (22.05 kHz)", "value": None,
                              "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "dry_run": True, "max_time": float(t.item()), "sum_audio": float(a.item())}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return 0
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    model, run, frames, workload, B = build_model(torch, args.model, args.dtype, args.batch, rank, dev,
                                                  json.loads(args.options) if args.options else None)
    audio_s = frames * 256 / SR

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dt = timed(torch, run, args.steps, args.warmup, barrier)
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

    # BASELINE configs[3] on the ranks of THIS run (every rank takes part; N = 1 included)
    sharded = None
    if args.model == "hifigan" and not args.no_extras and not args.no_sharded and args.batch in (None, 32):
        try:
            sharded = sharded_record(torch, dist, dev, rank, world, args.dtype)
        except Exception as e:  # secondary: never lose the headline to it
            if world > 1:
                raise  # (a rank that left the collective would hang the others: fail loudly instead)
            sharded = {"error": repr(e)[:200]}

    if rank == 0:
        roofline = None
        if not args.no_profile:
            roofline = roofline_of(torch, model, run, frames, dev, args.model, args.dtype, B,
                                   max(1, min(args.steps, 5)), args.rows)
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
            "cpu_baseline": None,
        }
        extras = world == 1 and not args.no_extras and args.model == "hifigan" and args.dtype == "bf16" \
            and args.batch in (None, 32)
        if extras:
            del model, run
            torch.cuda.empty_cache()
            line["fp32"] = side_record(torch, "hifigan", "fp32", None, dev, 10, 2, args.rows)
            line["fp32"]["bf16x3"] = side_record(torch, "hifigan", "bf16x3", None, dev, 20, 3, args.rows)
            # float16 operands (MV_PREC_F16): the bf16 kernels and rate with 11 significant bits per operand
            line["f16"] = side_record(torch, "hifigan", "f16", None, dev, 50, 5, args.rows)
            line["f16"]["vs_bf16_throughput"] = round(line["f16"]["value"] / value, 4)
            line["snr_db_vs_fp32_mode"] = snr_record(torch, dev)
            line["waveglow"] = {
                "B1": side_record(torch, "waveglow", "bf16", 1, dev, 60, 5, args.rows),
                "B8": side_record(torch, "waveglow", "bf16", 8, dev, 10, 2, args.rows),
                "B1_fp32": side_record(torch, "waveglow", "fp32", 1, dev, 5, 1, args.rows),
                "B1_bf16x3": side_record(torch, "waveglow", "bf16x3", 1, dev, 10, 2, args.rows),
            }
            try:
                line["e2e"] = e2e_record(torch, dev)
            except Exception as e:  # secondary: never lose the headline to it
                line["e2e"] = {"error": repr(e)[:200]}
        if sharded is not None:
            line["sharded"] = sharded
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(torch)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
