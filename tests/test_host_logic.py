"""CPU tests of the host side: the C ABI exports, checkpoint loading (D1), PCM / WAV (D2),
accounting (D3), bucketing and sharding, including a world_size-2 gloo run."""
import copy
import os
import re
import sys
import time
import wave

import numpy as np
import pytest
import torch

from tacotron2_amd import _native, batch, synth, vocoder
from tacotron2_amd.hifigan import AttrDict, Generator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "melvoc.h")).read()
    declared = re.findall(r"MV_API\s+[\w\s\*]+?\b(mv_\w+)\s*\(", hdr)
    assert len(declared) >= 13
    l = _native.lib()
    for name in declared:
        assert hasattr(l, name), name
    assert set(declared) == set(_native.EXPORTS)
    assert l.mv_backend() == 1
    assert l.mv_last_error() is not None


def test_null_arguments_are_rejected_without_a_gpu():
    l = _native.lib()
    assert l.mv_workspace_bytes(None, 1, 1) == 0
    assert l.mv_hop(None) == 0
    assert l.mv_pcm16(None, 4, None, None) == -1
    assert b"null" in l.mv_last_error()


def test_vocoder_kind_is_the_filename_substring_rule():
    assert vocoder.vocoder_kind("ckpt/waveglow_NEB.pt") == "WAVEGLOW"
    assert vocoder.vocoder_kind("hifigan/generator_universal.pth.tar") == "HIFIGAN"
    assert vocoder.vocoder_kind("model.pt") is None
    assert vocoder.load_vocoder("/nonexistent/hifigan.pt") is None  # do_syn.py:171-174: message, no synthesis


def test_generator_surface_and_state_dict_forms():
    g = Generator(AttrDict(synth.HIFIGAN_V1))
    wn = synth.hifigan_state(synth.HIFIGAN_V1, 0, weight_norm=True)
    assert len(g.state_dict()) == 234
    g.load_state_dict({k: torch.from_numpy(v) for k, v in wn.items()})
    g.eval()
    g.remove_weight_norm()
    g.remove_weight_norm()  # idempotent
    sd = g.state_dict()
    folded = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    assert len(sd) == 156 and set(sd) == set(folded)
    for k, v in folded.items():
        assert np.abs(sd[k].numpy() - v).max() < 1e-6, k
    n_params = sum(p.nelement() for p in g.parameters())
    assert n_params == 13926017
    assert sum(1 for _ in g.buffers()) == 0
    for p in g.parameters():
        p.requires_grad = False  # do_syn.py:165
    # folded checkpoint into a fresh (weight-normed) module, and g/v into a folded one
    g2 = Generator(AttrDict(synth.HIFIGAN_V1))
    g2.load_state_dict({k: torch.from_numpy(v) for k, v in folded.items()})
    assert len(g2.state_dict()) == 156
    g2.load_state_dict({k: torch.from_numpy(v) for k, v in wn.items()})
    assert len(g2.state_dict()) == 234
    with pytest.raises(RuntimeError):
        g2(torch.zeros(1, 80, 4))  # CPU tensor: loud failure, no fallback


def test_load_vocoder_hifigan_checkpoint(tmp_path):
    path = str(tmp_path / "hifigan_synth.pth.tar")
    wn = synth.hifigan_state(synth.HIFIGAN_V1, 3, weight_norm=True)
    torch.save({"generator": {k: torch.from_numpy(v) for k, v in wn.items()}}, path)
    v = vocoder.load_vocoder(path, device="cpu", precision="bf16")
    assert v.kind == "HIFIGAN" and v.model.precision == "bf16" and not v.model.training
    folded = synth.hifigan_state(synth.HIFIGAN_V1, 3)
    st = v.model.native_state()
    assert set(st) == set(folded)
    assert max(np.abs(st[k] - folded[k]).max() for k in folded) < 1e-6


def test_load_vocoder_waveglow_pickled_module(golden_dir):
    """A checkpoint pickled from the REFERENCE's glow.WaveGlow (weight-norm hooks on) unpickles into
    the native class because tacotron2_amd/waveglow is what `glow` resolves to."""
    path = os.path.join(golden_dir, "waveglow_tiny_ref.pt")
    v = vocoder.load_vocoder(path, device="cpu")
    assert v.kind == "WAVEGLOW" and v.sigma == 0.6
    m = v.model
    assert type(m).__module__ == "glow" and hasattr(m, "native_state")
    cfg = m.config()
    assert cfg["n_flows"] == 4 and cfg["WN_config"]["n_channels"] == 32 and cfg["n_mel_channels"] == 8
    want = synth.waveglow_state(cfg, 5)
    st = m.native_state()
    assert set(st) == set(want)
    assert max(np.abs(st[k] - want[k]).max() for k in want) < 1e-6
    assert m.n_remaining_channels == 6


def test_pcm16_and_wav(tmp_path, golden_dir):
    w = np.load(os.path.join(golden_dir, "hifigan_v1_T64.npz"))["wav"]
    g = np.load(os.path.join(golden_dir, "pcm16_v1_T64.npz"))["pcm"]
    assert np.array_equal(vocoder.to_pcm16(torch.from_numpy(w)), g)
    p = str(tmp_path / "a.wav")
    vocoder.write_wav(p, torch.from_numpy(w), 22050)
    with wave.open(p) as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()) == (1, 2, 22050, len(w))
        assert np.array_equal(np.frombuffer(wf.readframes(len(w)), np.int16), g)


def test_rtf_meter():
    m = vocoder.RtfMeter(22050)
    m.start()
    m.stop(22050 * 3, 2)
    assert m.dt_s == 3.0 and m.utterances == 2 and m.rtf == m.dt_voc / 3.0


def test_buckets_and_lpt_sharding():
    lengths = synth.synthetic_lengths(500, 7).tolist() + [0]
    bks = batch.make_buckets(lengths, max_batch=32, width=64, max_frames=16384)
    seen = sorted(i for b in bks for i in b.indices)
    assert seen == list(range(500))  # every non-empty utterance exactly once
    for b in bks:
        assert len(b.indices) <= 32 and b.lengths[0] - b.lengths[-1] <= 64 and b.tmax * len(b.indices) <= 16384
        assert b.lengths == sorted(b.lengths, reverse=True) and b.cost == sum(b.lengths)
    for world in (1, 2, 4, 8):
        shards, loads = batch.shard_buckets(bks, world)
        assert sorted(bi for s in shards for bi in s) == list(range(len(bks)))
        assert sum(loads) == sum(lengths)
        assert max(loads) - min(loads) <= max(b.cost for b in bks)  # LPT bound
    assert batch.make_buckets([], 4) == [] and batch.make_buckets([0, 0], 4) == []


class _FakeVocoder:
    """hop-256 stand-in: sample value = frame index of the utterance's own mel (so padding leaks show)."""

    def __call__(self, mel, lengths=None, out="pcm16"):
        B, _, T = mel.shape
        wav = torch.zeros(B, T * 256, dtype=torch.int16)
        for b in range(B):
            n = int(lengths[b])
            wav[b, : n * 256] = (mel[b, 0, :n].round().to(torch.int16)).repeat_interleave(256)
        return wav


def _corpus(n):
    lengths = synth.synthetic_lengths(n, 11, 2, 40).tolist()
    mels = [np.full((80, T), float(i % 100), np.float32) for i, T in enumerate(lengths)]
    return mels, lengths


def _rank_main(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mels, lengths = _corpus(57)
    res, st = batch.synthesize_shard(_FakeVocoder(), mels, lengths, rank=rank, world=world, max_batch=8, width=8)
    total = batch.gather_stats(st)
    q.put((rank, sorted(res), {i: int(a[0]) for i, a in res.items()}, total.utterances, total.frames,
           total.checksum, [p.utterances for p in total.per_rank]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_covers_the_corpus_once():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    mels, lengths = _corpus(57)
    got = sorted(i for o in outs for i in o[1])
    assert got == list(range(57))  # disjoint and complete
    for o in outs:
        for i, first in o[2].items():
            assert first == i % 100  # each waveform came from its own mel
        assert o[3] == 57 and o[4] == sum(lengths)  # both ranks hold the same gathered totals
        assert sum(o[6]) == 57
    assert outs[0][5] == outs[1][5]
    # single-process run gives the same corpus fingerprint
    res, st = batch.synthesize_shard(_FakeVocoder(), mels, lengths, max_batch=8, width=8)
    assert abs(batch.gather_stats(st).checksum - outs[0][5]) < 1e-6


def test_device_lengths_host_side_check():
    """Host-side lengths are validated before anything is launched (device-side ones are clamped on the device)."""
    import torch

    from tacotron2_amd import _native

    ok = _native.device_lengths([3, 0, 7], 3, 7, torch.device("cpu"))
    assert ok.dtype == torch.int32 and ok.tolist() == [3, 0, 7]
    for bad in ([3, 8, 1], [3, -1, 1], [3, 1]):
        with pytest.raises(RuntimeError):
            _native.device_lengths(bad, 3, 7, torch.device("cpu"))


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus N` (the driver's own command shape, no torch.distributed.run around it) must start N
    ranks itself: the launch / rendezvous / reduction path at world 2 over gloo, with the vocoder step skipped."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True
    assert rec["sum_audio"] == 2.0 and abs(rec["max_time"] - 0.002) < 1e-12  # SUM and MAX over both ranks
    # under torch.distributed.run (WORLD_SIZE set) the process is a rank, not a launcher
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=env1,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_launcher_ends_the_run_when_a_rank_dies():
    """A rank that exits before the rendezvous must not leave rank 0 waiting for the store's timeout: the launcher
    polls every child, kills the survivors and exits non-zero within seconds."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_DRY_RUN_FAIL_RANK"] = "1"
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 1 and "rank 1 exited with 3" in p.stderr, p.stderr[-500:]
    assert time.time() - t0 < 120
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]  # no result line from a broken run


def test_bench_counts_gpus_from_the_kfd_topology_without_hip(tmp_path, monkeypatch):
    """The launcher must know how many GPUs the node has BEFORE it starts its children and WITHOUT initialising HIP in
    its own process: it reads the KFD topology (CPU nodes have simd_count 0) and honours *_VISIBLE_DEVICES."""
    sys.path.insert(0, ROOT)
    import bench

    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):  # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.count_gpus_without_hip(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus_without_hip(str(tmp_path)) == 2
    assert "torch" not in bench.count_gpus_without_hip.__code__.co_names  # (no torch call on the sysfs path)


def test_bench_prices_waveglow_with_the_references_flops():
    """WaveGlow-512 costs 65 338 996 FLOP per sample = 1.4407 TFLOP per audio second in the
    reference's formulation; bench.py's `frac_of_mfma_roofline_reference_flops` uses exactly that count."""
    sys.path.insert(0, ROOT)
    import bench

    macs = bench.waveglow_reference_macs_per_frame(synth.WAVEGLOW_512)
    assert abs(2.0 * macs / 256 - 65338996) <= 64  # FLOP per sample
    assert abs(2.0 * macs * 22050 / 256 / 1e12 - 1.4407) < 1e-4


class _PassThroughVocoder:
    """What the host sees of a GPU vocoder: int16 [B, 256 T] handed back without computing anything (a view of one
    reusable buffer, as the pinned download slots are), so that only the host-side work of a rank is timed."""
    hop = 256

    def __init__(self):
        self.buf = torch.zeros(32 * 1024 * 256, dtype=torch.int16)

    def __call__(self, mel, lengths=None, out="pcm16"):
        B, _, T = mel.shape
        return self.buf[: B * T * 256].view(B, T * 256)


def _host_rate_main(rank, world, n, q):
    torch.set_num_threads(1)
    lens = synth.synthetic_lengths(n, 1234)
    base = [synth.synthetic_mel(int(lens.max()), 7 + i) for i in range(4)]
    mels = [base[i % 4][:, : int(T)] for i, T in enumerate(lens)]
    seen = [0]

    def sink(i, pcm):
        seen[0] += int(pcm.shape[0])

    t0 = time.perf_counter()
    _, st = batch.synthesize_shard(_PassThroughVocoder(), mels, [int(t) for t in lens], rank=rank, world=world, sink=sink)
    dt = time.perf_counter() - t0
    q.put((rank, st.utterances, st.frames, seen[0], dt))


def test_host_side_of_eight_ranks_on_the_configs3_corpus():
    """BASELINE configs[3] is 10 000 utterances over 8 ranks of ONE host.  Everything a rank does on the host per
    utterance -- bucket, pad into the staging buffer, hand the PCM view to the sink, account -- with the vocoder
    replaced by a pass-through: utterances per second per process at 1 and at 8 processes.  One MI355X consumes
    ~3 000 utterances of this corpus per second (17.5 k audio-s/s over 5.9 s each); the host path of a rank must
    stay well above that with all eight running, or it -- not the GPUs -- bends the scaling curve."""
    import torch.multiprocessing as mp

    n = 10000
    ctx = mp.get_context("spawn")
    rates = {}
    for world in (1, 8):
        q = ctx.Queue()
        procs = [ctx.Process(target=_host_rate_main, args=(r, world, n, q)) for r in range(world)]
        for p in procs:
            p.start()
        outs = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert sum(o[1] for o in outs) == n  # the shards cover the corpus exactly once ...
        assert all(o[3] == o[2] * 256 for o in outs)  # ... and every rank's sink saw every sample of its shard
        frames = [o[2] for o in outs]
        assert (max(frames) - min(frames)) / (sum(frames) / world) < 0.02  # LPT: loads within 2 % of each other
        rates[world] = sorted(o[1] / o[4] for o in outs)
    print("host path, utterances/s per process: 1 process %.0f; 8 processes min %.0f median %.0f (os.cpu_count() = %d)"
          % (rates[1][0], rates[8][0], rates[8][4], os.cpu_count() or 0))
    assert rates[1][0] >= 3000.0
    if (os.cpu_count() or 1) >= 8:  # (with fewer cores than ranks the processes time-share)
        assert rates[8][0] >= 1500.0
