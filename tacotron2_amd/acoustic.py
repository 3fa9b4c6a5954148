"""Mel producer for the end-to-end path (BASELINE.json configs[4]): a Tacotron2-shaped
acoustic model in plain PyTorch (ROCm), whose first decoder -- the 80-bin `WAVEGLOW` stream that the vocoders
consume -- decodes a whole length-bucket at once.

north_star keeps the acoustic model in PyTorch-ROCm; what changes is the loop.  The reference decodes one
utterance chunk at a time (`batch_size=1`, do_syn.py:188-189; `Decoder.inference`, model.py:532-569: a Python
`while True` that grows its outputs with `torch.cat` and reads the stop gate back to the host every step).
`MelProducer.infer` runs the same step (model.py:452-492) on `[B, .]` states with

  * the encoder on a padded batch (packed BiLSTM, as the reference's own training forward does, model.py:204-219),
  * attention masked at each text's own end (model.py:116-117, the `mask` the reference only uses in training),
  * per-utterance stopping: an utterance is finished at its first gate > threshold or at its imposed length
    (`n_steps`, the `duree_imposee` of model.py:551) -- finished rows keep stepping but their frames are dropped --
    and the loop reads the device once every `poll` steps instead of every step,
  * outputs written into a preallocated `[steps, B, 160]` buffer.

Module and parameter names follow the reference's `state_dict` (embedding, speaker_embedding, style_embedding,
encoder.convolutions.N.0.conv / .1, encoder.lstm, decoder.0.*, postnet.0.*), so a tc2 checkpoint's first decoder
loads with `load_state_dict(..., strict=False)`; the second decoder (the 37-dim `AU_60` stream, tc2_french.yaml:13-16)
and the phonetizer are not on the path to a waveform and are not built.  With `batch 1`, `mask=None` semantics and the
reference's order of random draws (prenet dropout is always on, model.py:136), `infer` reproduces
`Tacotron2.inference` bit for bit on the CPU: tests/test_acoustic.py pins that against a fixture generated from the
reference's own model.py (tests/gen_golden.py).

The Postnet (model.py:140-174,756) is NOT in this module: the end-to-end driver hands the decoder's `[B,80,T]`
to the native `tacotron2_amd.postnet.Postnet` (libmelvoc) and its output to the native vocoder, all on the device.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

# tc2_french.yaml (reference): the keys `MelProducer` reads, first decoder only
TC2_FRENCH = {
    "n_symbols": 96, "symbols_embedding_dim": 512, "nb_speakers": 13, "nb_styles": 13,
    "encoder_kernel_size": 5, "encoder_n_convolutions": 3, "encoder_embedding_dim": 512, "encoder_lstm_hidden_dim": 256,
    "dim_data": [80], "fe_data": [86.1328125], "n_frames_per_step": [2], "decoder_rnn_dim": [1024],
    "prenet_dim": [256], "gate_threshold": [0.5], "p_prenet_dropout": [0.5], "attention_rnn_dim": [1024],
    "attention_dim": [128], "attention_location_n_filters": [32], "attention_location_kernel_size": [31],
    "lgs_max": 20.0,
}


class _Linear(nn.Module):
    """`LinearNorm` key layout (model.py:29-36): <name>.linear_layer.weight"""

    def __init__(self, i, o, bias=True):
        super().__init__()
        self.linear_layer = nn.Linear(i, o, bias=bias)

    def forward(self, x):
        return self.linear_layer(x)


class _Conv(nn.Module):
    """`ConvNorm` key layout (model.py:38-50): <name>.conv.weight"""

    def __init__(self, i, o, k, bias=True):
        super().__init__()
        self.conv = nn.Conv1d(i, o, k, padding=(k - 1) // 2, bias=bias)

    def forward(self, x):
        return self.conv(x)


class _Location(nn.Module):
    def __init__(self, n_filters, k, dim):
        super().__init__()
        self.location_conv = _Conv(2, n_filters, k, bias=False)
        self.location_dense = _Linear(n_filters, dim, bias=False)

    def forward(self, w_cat):  # [B,2,Lin] -> [B,Lin,dim]
        return self.location_dense(self.location_conv(w_cat).transpose(1, 2))


class _Attention(nn.Module):
    """Location-sensitive attention, model.py:77-123."""

    def __init__(self, rnn_dim, emb_dim, dim, n_filters, k):
        super().__init__()
        self.query_layer = _Linear(rnn_dim, dim, bias=False)
        self.memory_layer = _Linear(emb_dim, dim, bias=False)
        self.v = _Linear(dim, 1, bias=False)
        self.location_layer = _Location(n_filters, k, dim)

    def forward(self, query, memory, processed_memory, w_cat, mask):
        e = self.v(torch.tanh(self.query_layer(query.unsqueeze(1)) + self.location_layer(w_cat) + processed_memory))
        e = e.squeeze(-1)
        if mask is not None:
            e = e.masked_fill(mask, -float("inf"))
        w = F.softmax(e, dim=1)
        return torch.bmm(w.unsqueeze(1), memory).squeeze(1), w


class _Prenet(nn.Module):
    def __init__(self, i, sizes, p):
        super().__init__()
        self.p = p
        self.layers = nn.ModuleList([_Linear(a, b, bias=False) for a, b in zip([i] + sizes[:-1], sizes)])

    def forward(self, x):
        for lin in self.layers:
            x = F.dropout(F.relu(lin(x)), p=self.p, training=True)  # always on, also at synthesis (model.py:136)
        return x


class _Encoder(nn.Module):
    """model.py:176-234 without the `stateful` carry-over (a per-chapter state of the batch-1 driver)."""

    def __init__(self, hps):
        super().__init__()
        d, k = hps["encoder_embedding_dim"], hps["encoder_kernel_size"]
        self.convolutions = nn.ModuleList(
            [nn.Sequential(_Conv(d, d, k), nn.BatchNorm1d(d)) for _ in range(hps["encoder_n_convolutions"])])
        self.lstm = nn.LSTM(d, hps["encoder_lstm_hidden_dim"], 1, batch_first=True, bidirectional=True)

    def forward(self, x, lengths=None):  # x [B,d,Lin]
        for conv in self.convolutions:
            x = F.relu(conv(x))
            if lengths is not None:  # a padded batch: keep the padding at zero so the next conv sees the batch-1 edge
                x = x * self._mask(lengths, x.size(2), x.device).unsqueeze(1).to(x.dtype)
        x = x.transpose(1, 2)
        self.lstm.flatten_parameters()
        if lengths is None:
            return self.lstm(x)[0]
        packed = nn.utils.rnn.pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
        out, _ = self.lstm(packed)
        return nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=x.size(1))[0]

    @staticmethod
    def _mask(lengths, L, device):
        return (torch.arange(L, device=device)[None, :] < lengths.to(device)[:, None]).to(torch.float32)


class _Decoder(nn.Module):
    """First decoder of model.py:331-373 (prenet_dim > 0 branch): the parameters and one step."""

    def __init__(self, hps, i=0):
        super().__init__()
        self.dim_data, self.nfs = hps["dim_data"][i], hps["n_frames_per_step"][i]
        self.fe_data, self.lgs_max = hps["fe_data"][i], hps["lgs_max"]
        self.gate_threshold = hps["gate_threshold"][i]
        emb, arnn, drnn = hps["encoder_embedding_dim"], hps["attention_rnn_dim"][i], hps["decoder_rnn_dim"][i]
        self.emb, self.arnn, self.drnn = emb, arnn, drnn
        pd = hps["prenet_dim"][i]
        self.prenet = _Prenet(self.dim_data * self.nfs, [pd, pd], hps["p_prenet_dropout"][i])
        self.attention_rnn = nn.LSTMCell(pd + emb, arnn)
        self.attention_layer = _Attention(arnn, emb, hps["attention_dim"][i], hps["attention_location_n_filters"][i],
                                          hps["attention_location_kernel_size"][i])
        self.decoder_rnn = nn.LSTMCell(arnn + emb, drnn, 1)
        self.gate_layer = _Linear(drnn + emb, 1, bias=True)
        self.linear_projection = _Linear(drnn + emb, self.dim_data * self.nfs)


class MelProducer(nn.Module):
    def __init__(self, hps=None):
        super().__init__()
        hps = dict(TC2_FRENCH if hps is None else hps)
        self.hps = hps
        self.embedding = nn.Embedding(hps["n_symbols"], hps["symbols_embedding_dim"])
        emb = hps["encoder_embedding_dim"]
        if hps.get("nb_speakers", 0) > 1:
            self.speaker_embedding = nn.Embedding(hps["nb_speakers"], emb)
        if hps.get("nb_styles", 0) > 1:
            self.style_embedding = nn.Embedding(hps["nb_styles"], emb)
        self.encoder = _Encoder(hps)
        self.decoder = nn.ModuleList([_Decoder(hps, 0)])

    @property
    def frames_per_step(self):
        return self.decoder[0].nfs

    @torch.no_grad()
    def encode(self, text, text_lengths=None, spk=None, style=None):
        """text [B,Lin] int64 (right-padded) -> memory [B,Lin,512] (model.py:734-747; padding rows are zero)."""
        B, L = text.shape
        x = self.embedding(text).transpose(1, 2)
        if text_lengths is not None:  # the padding symbols must not reach the first conv's receptive field
            x = x * _Encoder._mask(text_lengths, L, x.device).unsqueeze(1).to(x.dtype)
        mem = self.encoder(x, text_lengths)
        for table, idx in ((getattr(self, "speaker_embedding", None), spk), (getattr(self, "style_embedding", None), style)):
            if table is None:
                continue
            idx = torch.zeros(B, dtype=torch.long, device=text.device) if idx is None else idx.to(text.device).view(B)
            mem = mem + table(idx)[:, None, :]  # model.py:736-747: added to every position ...
        if text_lengths is not None:  # ... of the text itself (model.py:687 "respect du padding")
            mem = mem * _Encoder._mask(text_lengths, L, mem.device).unsqueeze(2).to(mem.dtype)
        return mem

    @torch.no_grad()
    def infer(self, text, text_lengths=None, spk=None, style=None, seed=None, n_steps=None, max_steps=None, poll=16,
              graph=None):
        """Batched `Tacotron2.inference` for the first decoder (model.py:725-761 -> 532-569).

        text          [B,Lin] symbol ids, right-padded; text_lengths [B] (None: all Lin, no masking -- batch-1 semantics)
        n_steps       None, or [B] decoder steps to run per utterance regardless of the gate (`duree_imposee`,
                      model.py:551,558: random-weight models never raise their gate)
        max_steps     cap for gate-stopped decoding (default int(lgs_max * fe_data), model.py:551)
        poll          the device is read once every `poll` steps to see whether every utterance has finished
        graph         replay the decode step from a captured HIP graph (default: on a GPU, when more than 8 steps are
                      known to run).  One step is ~40 small kernels whose launches, not their arithmetic, set the pace;
                      captured once per call (states in static buffers, the step index on the device), a step costs
                      one graph launch.  Same arithmetic; the dropout draws come from the graph-safe generator, so
                      the samples differ from the eager loop's for one seed.
        Returns (mel [B,80,Tmax] zero-padded, frames [B] int32 on the device, gate [B,steps_max]).
        `seed`: torch.manual_seed(seed) before anything is drawn, as model.py:733 does.
        """
        dec = self.decoder[0]
        if seed is not None:
            torch.manual_seed(seed)
        dev = text.device
        B, Lin = text.shape
        memory = self.encode(text, text_lengths, spk, style)
        mask = None
        if text_lengths is not None:
            mask = torch.arange(Lin, device=dev)[None, :] >= text_lengths.to(dev)[:, None]
        cap = int(dec.lgs_max * dec.fe_data) if max_steps is None else int(max_steps)
        if n_steps is not None:
            n_steps = torch.as_tensor(n_steps, device=dev, dtype=torch.int32).view(B)
            cap = int(n_steps.max().item())
        z = lambda d: memory.new_zeros(B, d)  # noqa: E731
        ah, ac, dh, dc = z(dec.arnn), z(dec.arnn), z(dec.drnn), z(dec.drnn)
        aw, awc, ctx = z(Lin), z(Lin), z(dec.emb)
        pmem = dec.attention_layer.memory_layer(memory)
        out_dim = dec.dim_data * dec.nfs
        mels = memory.new_zeros(cap, B, out_dim)
        gates = memory.new_zeros(cap, B)
        done_at = torch.full((B,), cap, dtype=torch.int32, device=dev)  # steps kept per utterance
        x = memory.new_zeros(B, out_dim)  # the go frame (model.py:545)
        steps = 0
        if graph is None:
            graph = memory.is_cuda and n_steps is not None and cap > 8
        if graph:
            return self._infer_graph(dec, memory, pmem, mask, n_steps, cap, poll)
        for t in range(cap):
            x = dec.prenet(x)
            ah, ac = dec.attention_rnn(torch.cat((x, ctx), -1), (ah, ac))
            ctx, aw = dec.attention_layer(ah, memory, pmem, torch.stack((aw, awc), 1), mask)
            awc = awc + aw
            dh, dc = dec.decoder_rnn(torch.cat((ah, ctx), -1), (dh, dc))
            hc = torch.cat((dh, ctx), 1)
            x = dec.linear_projection(hc)
            g = dec.gate_layer(hc).squeeze(1)
            mels[t], gates[t] = x, g
            steps = t + 1
            if n_steps is None:
                # first step whose gate fires ends the utterance, that step included (model.py:555-559)
                fired = (torch.sigmoid(g) > dec.gate_threshold) & (done_at == cap)
                done_at = torch.where(fired, torch.full_like(done_at, t + 1), done_at)
                if (t + 1) % poll == 0 or B == 1:
                    if bool((done_at < cap).all()):
                        break
        if n_steps is not None:
            done_at = n_steps
        mel = mels[:steps].transpose(0, 1).reshape(B, steps * dec.nfs, dec.dim_data).transpose(1, 2)  # model.py:447-449
        frames = (done_at.clamp(max=steps) * dec.nfs).to(torch.int32)
        keep = torch.arange(steps * dec.nfs, device=dev)[None, :] < frames[:, None]
        mel = (mel * keep.unsqueeze(1).to(mel.dtype)).contiguous()
        return mel, frames, gates[:steps].transpose(0, 1)


def _decode_step(dec, memory, pmem, mask, S):
    """One decoder step (model.py:452-492, 553-562) on the state dict S, in place."""
    x = dec.prenet(S["x"])
    ah, ac = dec.attention_rnn(torch.cat((x, S["ctx"]), -1), (S["ah"], S["ac"]))
    ctx, aw = dec.attention_layer(ah, memory, pmem, torch.stack((S["aw"], S["awc"]), 1), mask)
    S["awc"].add_(aw)
    dh, dc = dec.decoder_rnn(torch.cat((ah, ctx), -1), (S["dh"], S["dc"]))
    hc = torch.cat((dh, ctx), 1)
    xo = dec.linear_projection(hc)
    g = dec.gate_layer(hc).squeeze(1)
    for k, v in (("ah", ah), ("ac", ac), ("ctx", ctx), ("aw", aw), ("dh", dh), ("dc", dc), ("x", xo)):
        S[k].copy_(v)
    S["mels"].index_copy_(0, S["t"], xo.unsqueeze(0))
    S["gates"].index_copy_(0, S["t"], g.unsqueeze(0))
    # first step whose gate fires ends the utterance, that step included (model.py:555-559)
    fired = (torch.sigmoid(g) > dec.gate_threshold) & (S["done_at"] == S["cap"])
    S["done_at"].copy_(torch.where(fired, (S["t"] + 1).to(torch.int32).expand_as(S["done_at"]), S["done_at"]))
    S["t"].add_(1)


_GRAPH_CACHE_MAX = 8  # captured decode steps kept per MelProducer (one per (batch, text length) shape)


def _capture_step(dec, memory, pmem, mask, cap_alloc):
    """Static buffers + one captured decode step for inputs of this shape (capture costs ~0.1 s: warm-up steps on a
    side stream for library plans and allocator pools, then the capture itself)."""
    dev, B, Lin = memory.device, memory.size(0), memory.size(1)
    z = lambda *d: memory.new_zeros(*d)  # noqa: E731
    out_dim = dec.dim_data * dec.nfs
    E = {"memory": memory.clone(), "pmem": pmem.clone(), "mask": None if mask is None else mask.clone(),
         "cap_alloc": cap_alloc}
    S = {"x": z(B, out_dim), "ah": z(B, dec.arnn), "ac": z(B, dec.arnn), "dh": z(B, dec.drnn), "dc": z(B, dec.drnn),
         "aw": z(B, Lin), "awc": z(B, Lin), "ctx": z(B, dec.emb), "mels": z(cap_alloc + 1, B, out_dim),
         "gates": z(cap_alloc + 1, B), "t": torch.zeros(1, dtype=torch.long, device=dev),
         "cap": torch.full((B,), cap_alloc, dtype=torch.int32, device=dev),
         "done_at": torch.full((B,), cap_alloc, dtype=torch.int32, device=dev)}
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            _decode_step(dec, E["memory"], E["pmem"], E["mask"], S)
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _decode_step(dec, E["memory"], E["pmem"], E["mask"], S)
    # (the capture itself does not execute the step)
    E["S"], E["graph"] = S, g
    return E


def _graph_infer(self, dec, memory, pmem, mask, n_steps, cap, poll):
    """The decode loop as replays of one captured step.  Captures are kept per input shape (batch, text length, dtype,
    masked or not) with their static buffers, so a driver that pads its buckets to a few shapes pays for a capture
    once per shape, not once per bucket (tacotron2_amd/do_syn_e2e.py: that was most of its Tacotron2 time)."""
    dev, B = memory.device, memory.size(0)
    cache = self.__dict__.setdefault("_graph_cache", {})
    key = (str(dev), B, memory.size(1), memory.dtype, mask is None)
    E = cache.get(key)
    if E is None or E["cap_alloc"] < cap:
        cache.pop(key, None)
        while len(cache) >= _GRAPH_CACHE_MAX:
            cache.pop(next(iter(cache)))  # oldest first
        E = cache[key] = _capture_step(dec, memory, pmem, mask, (cap + 127) // 128 * 128)
    else:
        cache[key] = cache.pop(key)  # most recently used last
        E["memory"].copy_(memory)
        E["pmem"].copy_(pmem)
        if mask is not None:
            E["mask"].copy_(mask)
    S, g = E["S"], E["graph"]
    for k in ("x", "ah", "ac", "dh", "dc", "aw", "awc", "ctx", "t"):  # the go state
        S[k].zero_()
    S["cap"].fill_(cap)
    S["done_at"].fill_(cap)
    steps = 0
    for t in range(cap):
        g.replay()
        steps = t + 1
        if n_steps is None and (t + 1) % poll == 0 and bool((S["done_at"] < cap).all()):
            break
    done_at = S["done_at"] if n_steps is None else n_steps
    mel = S["mels"][:steps].transpose(0, 1).reshape(B, steps * dec.nfs, dec.dim_data).transpose(1, 2)  # model.py:447-449
    frames = (done_at.clamp(max=steps) * dec.nfs).to(torch.int32)
    keep = torch.arange(steps * dec.nfs, device=dev)[None, :] < frames[:, None]
    mel = (mel * keep.unsqueeze(1).to(mel.dtype)).contiguous()
    # (mel is a fresh tensor; the gate trace is copied out of the static buffer the next call overwrites)
    return mel, frames, S["gates"][:steps].transpose(0, 1).clone()


MelProducer._infer_graph = _graph_infer


def synthetic_state(module: nn.Module, seed: int = 0):
    """Deterministic weights for a `MelProducer` (or any module): numpy default_rng(seed), tensors visited in sorted key
    order, std = gain / sqrt(fan_in) so activations stay O(1); BatchNorm statistics near (0, 1) but not trivial."""
    import numpy as np

    rng = np.random.default_rng(seed)
    out = {}
    for k, v in sorted(module.state_dict().items()):
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
            continue
        if k.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shape)
        elif k.endswith("running_mean"):
            a = 0.1 * rng.standard_normal(shape)
        elif ".1.weight" in k and len(shape) == 1:  # BatchNorm gamma
            a = rng.uniform(0.8, 1.2, shape)
        elif k.endswith("gate_layer.linear_layer.bias"):
            a = np.full(shape, -6.0)  # a random-weight model's stop gate must stay shut
        elif len(shape) == 1:
            a = 0.05 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) / math.sqrt(max(fan_in, 1))
            if "embedding" in k:
                a = a * math.sqrt(max(fan_in, 1)) * 0.3
        out[k] = torch.from_numpy(np.asarray(a, np.float32))
    return out
