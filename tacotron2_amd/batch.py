"""Length-bucketed, utterance-sharded batch vocoding (the do_syn.py outer loop, do_syn.py:235-394,
re-shaped for 1-8 MI355X).

The reference vocodes one chunk at a time on one GPU (do_syn.py:188-189,302-394).  Chunks are
independent, so the path shards with no data-path collective: utterances are sorted by frame
count, cut into buckets of similar length (the kernels mask every layer at each sequence's own
end, so a bucket is ONE length-masked call), and the buckets are dealt to ranks
longest-processing-time-first on their frame totals.  Each rank is one process with one GPU and
its own replica of the weights; `torch.distributed` (RCCL on GPUs, gloo in the CPU tests) is used
only after the work, to gather a fixed-size statistics record per rank.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import synth


@dataclass
class Bucket:
    indices: list            # utterance indices, longest first
    lengths: list            # frames per utterance
    cost: int = 0            # sum of frames (the arithmetic is proportional to it)

    @property
    def tmax(self):
        return max(self.lengths)


def make_buckets(lengths, max_batch=32, width=64, max_frames=32 * 1024):
    """Sort by length (longest first) and cut into buckets: at most `max_batch` utterances, at most
    `width` frames between the longest and the shortest, at most `max_frames` padded frames
    (B * Tmax, which bounds the workspace).  Zero-length utterances get no bucket."""
    order = sorted((i for i in range(len(lengths)) if lengths[i] > 0), key=lambda i: (-int(lengths[i]), i))
    buckets, cur = [], None
    for i in order:
        T = int(lengths[i])
        if cur is not None:
            tmax = cur.lengths[0]
            if len(cur.indices) >= max_batch or tmax - T > width or tmax * (len(cur.indices) + 1) > max_frames:
                buckets.append(cur)
                cur = None
        if cur is None:
            cur = Bucket([], [])
        cur.indices.append(i)
        cur.lengths.append(T)
        cur.cost += T
    if cur is not None:
        buckets.append(cur)
    return buckets


def shard_buckets(buckets, world):
    """Longest-processing-time-first: heaviest bucket to the least loaded rank.  Deterministic, so
    every rank computes the same assignment without communicating."""
    loads = [0] * world
    shards = [[] for _ in range(world)]
    for bi in sorted(range(len(buckets)), key=lambda b: (-buckets[b].cost, b)):
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(bi)
        loads[r] += buckets[bi].cost
    return shards, loads


def pad_bucket(mels, bucket, pad_value=synth.MEL_PAD_VALUE, device=None, into=None):
    """Right-pad the bucket's mels ([80,T] tensors / arrays) to [B,80,Tmax] plus an int32 length vector.
    `into`: a flat float32 tensor (e.g. a reusable pinned staging buffer) to build the batch in.
    Each row is written once -- its mel, then the pad value behind it -- through a numpy view of the batch (one slice
    copy per utterance with the GIL released for its duration: ~20 us per utterance, 50 k utterances/s per host thread,
    against ~3 k/s that one GPU consumes; measured by tests/test_host_logic.py at 1 and 8 processes)."""
    B, tmax = len(bucket.indices), bucket.tmax
    n_mels = int(mels[bucket.indices[0]].shape[0])
    if into is not None:
        x = into[: B * n_mels * tmax].view(B, n_mels, tmax)
    else:
        x = torch.empty((B, n_mels, tmax), dtype=torch.float32)
    xn = x.numpy()
    for b, (i, T) in enumerate(zip(bucket.indices, bucket.lengths)):
        m = mels[i]
        xn[b, :, :T] = (m.detach().cpu().numpy() if isinstance(m, torch.Tensor) else m)[:, :T]
        if T < tmax:
            xn[b, :, T:] = pad_value
    lengths = torch.tensor(bucket.lengths, dtype=torch.int32)
    if device is not None:
        x, lengths = x.to(device), lengths.to(device)
    return x, lengths


@dataclass
class ShardStats:
    rank: int = 0
    utterances: int = 0
    frames: int = 0
    audio_seconds: float = 0.0
    wall_seconds: float = 0.0
    calls: int = 0
    checksum: float = 0.0  # sum |pcm| over the shard: a cheap whole-corpus fingerprint
    per_rank: list = field(default_factory=list)

    def as_tensor(self):
        return torch.tensor([self.rank, self.utterances, self.frames, self.audio_seconds, self.wall_seconds,
                             self.calls, self.checksum, 0.0], dtype=torch.float64)

    @staticmethod
    def from_tensor(t):
        v = t.tolist()
        return ShardStats(int(v[0]), int(v[1]), int(v[2]), v[3], v[4], int(v[5]), v[6])


def device_mel_bucket(bucket, seed, device, n_mels=80, pad_value=synth.MEL_PAD_VALUE):
    """BASELINE.json config 4: the bucket's mels generated ON the device from one seed per utterance
    (clip(N(-5, 2^2), pad, 2.5) like synth.synthetic_mel), so no mel crosses PCIe."""
    B, tmax = len(bucket.indices), bucket.tmax
    x = torch.full((B, n_mels, tmax), float(pad_value), dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for b, (i, T) in enumerate(zip(bucket.indices, bucket.lengths)):
        g.manual_seed(int(seed) * 100003 + int(i))
        x[b, :, :T] = torch.empty((n_mels, T), dtype=torch.float32, device=device).normal_(-5.0, 2.0, generator=g) \
            .clamp_(float(pad_value), 2.5)
    return x, torch.tensor(bucket.lengths, dtype=torch.int32)


def synthesize_shard(vocoder, mels, lengths=None, rank=0, world=1, device=None, fe_wav=22050, hop=None,
                     max_batch=32, width=64, max_frames=32 * 1024, keep_audio=True, out="pcm16", pipeline=True,
                     device_mels=None, d2h=True, sink=None):
    """Vocode this rank's share of `mels` (list of [80,T]).  Returns ({utterance index: int16 / float
    numpy waveform}, ShardStats).  `vocoder(mel[B,80,T], lengths=int32[B], out=...) -> [B, hop*T]`.

    On a GPU the loop is a three-stage pipeline: a host thread pads bucket k+1 into pinned memory while the
    copy-in stream uploads bucket k, the compute stream vocodes it, and the copy-out stream drains bucket k-1
    into pinned memory (do_syn.py:380-394 does these strictly one after the other).  `device_mels=seed`
    generates the mels on the device instead (config 4); `d2h=False` leaves the audio on the device (then nothing
    is returned).  The checksum (sum |sample|) is accumulated on the device and read once at the end.

    `sink(i, pcm)`: called once per utterance with a VIEW of the pinned download slot (valid only during the call:
    write the WAV / hand the bytes on from there) -- with a sink no per-utterance host copy is made, which is what 8
    ranks sharing one host would otherwise contend on.  `keep_audio` without a sink copies each waveform out of the
    slot into the returned dict.  `hop` defaults to the vocoder's own samples-per-frame."""
    if hop is None:
        hop = int(getattr(vocoder, "hop", 256))
    if lengths is None:
        lengths = [int(m.shape[1]) for m in mels]
    buckets = make_buckets(lengths, max_batch, width, max_frames)
    shards, _ = shard_buckets(buckets, world)
    mine = shards[rank]
    stats = ShardStats(rank=rank)
    results = {}
    on_gpu = device is not None and torch.device(device).type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    sync()
    t0 = time.perf_counter()
    if not (pipeline and on_gpu):
        for bi in mine:
            bk = buckets[bi]
            x, lens = pad_bucket(mels, bk, device=device)
            wav = vocoder(x, lengths=lens, out=out)
            stats.calls += 1
            host = wav.cpu().numpy()  # the D->H copy is part of the driver's work (do_syn.py:393)
            for b, (i, T) in enumerate(zip(bk.indices, bk.lengths)):
                a = host[b, : T * hop]
                stats.checksum += (float(np.abs(a.astype(np.int32)).sum(dtype=np.int64)) if a.dtype == np.int16
                                   else float(np.abs(a.astype(np.float64)).sum()))
                if sink is not None:
                    sink(i, a)
                elif keep_audio:
                    results[i] = a.copy()
            stats.utterances += len(bk.indices)
            stats.frames += bk.cost
    else:
        from concurrent.futures import ThreadPoolExecutor

        dev = torch.device(device)
        compute = torch.cuda.current_stream(dev)
        copy_in, copy_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        # two host workers: one pads the next bucket into a pinned slot, one hands the previous bucket's audio to
        # the caller; the main thread only launches.  Three slots each way: slot k is free again once bucket k's
        # upload (in) / hand-over (out) is done, which the worker that reuses it waits for.
        pool_in, pool_out = ThreadPoolExecutor(1), ThreadPoolExecutor(1)
        csum = torch.zeros((), dtype=torch.float64, device=dev)
        odt = torch.int16 if out == "pcm16" else torch.float32
        NS = 3
        cap = max((len(buckets[bi].indices) * buckets[bi].tmax for bi in mine), default=0)
        n_mels = int(mels[0].shape[0]) if (device_mels is None and len(mels)) else 80
        out_slots = [torch.empty(cap * hop, dtype=odt).pin_memory() for _ in range(NS)] if (d2h and cap) else []
        in_slots = ([torch.empty(cap * n_mels, dtype=torch.float32).pin_memory() for _ in range(NS)]
                    if (device_mels is None and cap) else [])
        in_free = [None] * NS   # event: the upload that last read the slot
        out_free = [None] * NS  # future: the hand-over that last read the slot

        def prep(k, bi):
            if device_mels is not None:
                return None
            if in_free[k % NS] is not None:
                in_free[k % NS].synchronize()
            return pad_bucket(mels, buckets[bi], into=in_slots[k % NS])

        def finalize(bk, hb, ev):
            ev.synchronize()
            if sink is not None or keep_audio:
                host = hb.numpy()
                for b, (i, T) in enumerate(zip(bk.indices, bk.lengths)):
                    if sink is not None:
                        sink(i, host[b, : T * hop])  # a view of the pinned slot: no host copy
                    else:
                        results[i] = host[b, : T * hop].copy()

        fut = pool_in.submit(prep, 0, mine[0]) if mine else None
        for k, bi in enumerate(mine):
            bk = buckets[bi]
            staged = fut.result()
            if k + 1 < len(mine):
                fut = pool_in.submit(prep, k + 1, mine[k + 1])
            if device_mels is not None:
                x, lens = device_mel_bucket(bk, device_mels, dev)
            else:
                xh, lens = staged
                with torch.cuda.stream(copy_in):
                    x = xh.to(dev, non_blocking=True)
                    ev_in = copy_in.record_event()
                in_free[k % NS] = ev_in
                compute.wait_event(ev_in)
                x.record_stream(compute)
            wav = vocoder(x, lengths=lens, out=out)
            stats.calls += 1
            csum += wav.to(torch.float32).abs().sum(dtype=torch.float64)
            if d2h:
                ev_c = compute.record_event()
                B, n = wav.shape[0], wav.shape[-1]
                if out_free[k % NS] is not None:
                    out_free[k % NS].result()  # bucket k-3 has been handed over: its slot is free
                hb = out_slots[k % NS][: B * n].view(B, n)
                with torch.cuda.stream(copy_out):
                    copy_out.wait_event(ev_c)
                    hb.copy_(wav.view(B, n), non_blocking=True)
                    wav.record_stream(copy_out)
                    ev_o = copy_out.record_event()
                out_free[k % NS] = pool_out.submit(finalize, bk, hb, ev_o)
            stats.utterances += len(bk.indices)
            stats.frames += bk.cost
        for f in out_free:
            if f is not None:
                f.result()
        pool_in.shutdown()
        pool_out.shutdown()
        sync()
        stats.checksum = float(csum.item())
    sync()
    stats.wall_seconds = time.perf_counter() - t0
    stats.audio_seconds = stats.frames * hop / float(fe_wav)
    return results, stats


def gather_stats(stats: ShardStats, device=None):
    """One fixed-size all_gather per rank AFTER the work (the path's only collective)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        total = ShardStats(**{k: getattr(stats, k) for k in ("rank", "utterances", "frames", "audio_seconds",
                                                              "wall_seconds", "calls", "checksum")})
        total.per_rank = [stats]
        return total
    t = stats.as_tensor()
    if device is not None and dist.get_backend() == "nccl":
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    per = [ShardStats.from_tensor(o.cpu()) for o in out]
    total = ShardStats(rank=-1)
    total.per_rank = per
    total.utterances = sum(p.utterances for p in per)
    total.frames = sum(p.frames for p in per)
    total.audio_seconds = sum(p.audio_seconds for p in per)
    total.wall_seconds = max(p.wall_seconds for p in per)  # the job ends with its slowest rank
    total.calls = sum(p.calls for p in per)
    total.checksum = sum(p.checksum for p in per)
    return total


def synthetic_corpus(n, seed=1234, lo=86, hi=430):
    """Config 4 of BASELINE.json: n utterance lengths of 2-10 s; mels are generated lazily per index."""
    lengths = synth.synthetic_lengths(n, seed, lo, hi)

    class _Lazy:
        def __len__(self):
            return n

        def __getitem__(self, i):
            return synth.synthetic_mel(int(lengths[i]), seed * 100003 + i)

    return _Lazy(), [int(t) for t in lengths]
