"""Vocoder selection, loading, PCM conversion and accounting as the reference drivers do them
(do_syn.py:144-174,270-282,385-396,484; do_tts.py:139-166,346-354), on the native vocoders.

D1  `vocoder_kind` / `load_vocoder`: the kind is the substring 'waveglow' / 'hifigan' of the
    checkpoint path (do_syn.py:145,153); a YAML `vocoder:` key, if present, names the path
    (README.md:36 promises the key; the shipped YAMLs carry none, so the path rule is the contract).
D2  `to_pcm16` / `write_wav`: 32768*y -> int16 by truncation, no clip; mono 16-bit WAV at fe_wav.
D3  `RtfMeter`: audio seconds = samples / fe_wav; RTF = vocoder seconds / audio seconds.  The
    reference times with un-synchronised time.process_time(); this meter uses wall time around a
    device synchronisation and says so.
"""
from __future__ import annotations

import json
import os
import sys
import time
import wave

import numpy as np
import torch

from . import synth

MAX_WAV_VALUE = 32768.0  # do_syn.py:11
DEFAULT_SIGMA = 0.6      # do_syn.py:146


def vocoder_kind(path: str):
    """'WAVEGLOW' | 'HIFIGAN' | None, by the reference's filename-substring rule."""
    if path is None:
        return None
    if path.find("waveglow") >= 0:
        return "WAVEGLOW"
    if path.find("hifigan") >= 0:
        return "HIFIGAN"
    return None


def _hifigan_config(config_path=None):
    for cand in (config_path, os.path.join("hifigan", "config.json")):  # do_syn.py:157 reads the latter
        if cand and os.path.exists(cand):
            with open(cand, "r") as f:
                return json.load(f)
    return dict(synth.HIFIGAN_V1)


class Vocoder:
    """Uniform call surface over the two native vocoders: mel [B,80,T] / [80,T] -> wav [B, 256T]."""

    def __init__(self, kind, model, sigma=DEFAULT_SIGMA, path=None):
        self.kind, self.model, self.sigma, self.path = kind, model, sigma, path
        self.hop = int(getattr(model, "hop", 256))  # HiFi-GAN: prod(upsample_rates); WaveGlow: 256 (glow.py:183-185)

    def __call__(self, mel, lengths=None, out="float"):
        if mel.dim() == 2:
            mel = mel.unsqueeze(0)
        with torch.no_grad():
            if self.kind == "WAVEGLOW":
                return self.model.infer(mel, sigma=self.sigma, lengths=lengths, out=out)
            return self.model(mel, lengths=lengths, out=out)[:, 0]

    def pcm16(self, mel, lengths=None):
        """do_syn.py:389-394 fused on device: int16 tensor [B, 256T]."""
        return self(mel, lengths=lengths, out="pcm16")


def _waveglow_pickle_allowlist(glow):
    """What a reference WaveGlow checkpoint (`{'model': <pickled glow.WaveGlow>, ...}`, waveglow/train.py:54-62) may
    resolve while it unpickles: the three glow classes, the torch.nn containers / layers they hold, the weight-norm
    hook and the plain-Python scaffolding of a module's __dict__.  Anything else is refused by torch's weights_only
    unpickler, so a doctored checkpoint cannot run code."""
    import collections

    allowed = [glow.WaveGlow, glow.WN, glow.Invertible1x1Conv, torch.nn.Conv1d, torch.nn.ConvTranspose1d,
               torch.nn.ModuleList, collections.OrderedDict, set]
    try:
        from torch.nn.utils.weight_norm import WeightNorm  # the hook object old-style weight_norm leaves on a module

        allowed.append(WeightNorm)
    except ImportError:
        pass
    try:
        from torch.nn.utils import parametrize
        from torch.nn.utils.parametrizations import _WeightNorm

        allowed += [_WeightNorm, parametrize.ParametrizationList]
    except ImportError:
        pass
    return allowed


def load_vocoder(path, device="cuda", hifigan_config=None, precision=None, sigma=DEFAULT_SIGMA, hps=None,
                 unsafe_pickle=False):
    """Load a vocoder checkpoint the way do_syn.py:144-174 does and return a `Vocoder`.

    `hps` may be the YAML hyper-parameter dict: its `vocoder` key (a path) is used when `path` is
    None, `vocoder_dtype` selects fp32 / bf16, `vocoder_backend` must be 'native' if given.
    Returns None (after the reference's message) when the file does not exist (do_syn.py:171-174).
    """
    if hps:
        path = path or hps.get("vocoder")
        precision = precision or hps.get("vocoder_dtype")
        if hps.get("vocoder_backend", "native") != "native":
            raise ValueError("this build has only the native (libmelvoc) vocoder backend")
    if path is None or not os.path.exists(path):
        print("VOCODER: MODEL {} not found. No synthesis".format(path), flush=True)
        return None
    kind = vocoder_kind(path)
    if kind == "WAVEGLOW":
        from .waveglow import GLOW_PATH, WaveGlow

        # do_syn.py:147 appends 'waveglow' to sys.path because the pickle names its classes as module `glow`; here the
        # native glow module stands under that name for the duration of the load only.  The load itself runs under
        # torch's restricted unpickler with an allow-list (the reference uses weights_only=False, do_syn.py:148);
        # `unsafe_pickle=True` is the explicit opt-in for checkpoints that carry other objects.
        import importlib

        had_path = GLOW_PATH in sys.path
        if not had_path:
            sys.path.insert(0, GLOW_PATH)
        try:
            _glow = importlib.import_module("glow")  # the native module under the name the pickle asks for
            if unsafe_pickle:
                ckpt = torch.load(path, map_location="cpu", weights_only=False)
            else:
                import pickle

                try:
                    with torch.serialization.safe_globals(_waveglow_pickle_allowlist(_glow)):
                        ckpt = torch.load(path, map_location="cpu", weights_only=True)
                except pickle.UnpicklingError as e:
                    # the reference loads the same file with weights_only=False (do_syn.py:148) and would accept it
                    raise RuntimeError(
                        f"{path}: the checkpoint pickles an object outside the WaveGlow allow-list ({e}).  If the file "
                        f"comes from a source you trust, load it as the reference does: load_vocoder(..., "
                        f"unsafe_pickle=True), or --unsafe_pickle on the do_syn_batch / do_syn_e2e command line") from e
        finally:
            if not had_path and GLOW_PATH in sys.path:
                sys.path.remove(GLOW_PATH)
        model = ckpt["model"]
        model = WaveGlow.remove_weightnorm(model)
        if not hasattr(model, "infer") or not hasattr(model, "native_state"):
            raise RuntimeError("checkpoint did not unpickle into the native glow.WaveGlow (is another `glow` module "
                               "ahead on sys.path?)")
        model.eval()
        if precision:
            model.precision = precision
        model.to(device)
        print("WAVEGLOW: MODEL {} loaded".format(path), flush=True)
        return Vocoder(kind, model, sigma, path)
    if kind == "HIFIGAN":
        from .hifigan import AttrDict, Generator

        cfg = AttrDict(_hifigan_config(hifigan_config))
        hgan = Generator(cfg, precision=precision)
        ckpt = torch.load(path, map_location=torch.device("cpu"), weights_only=True)
        hgan.load_state_dict(ckpt["generator"])
        hgan.eval()
        hgan.remove_weight_norm()
        size = sum(p.nelement() * p.element_size() for p in hgan.parameters())
        size += sum(b.nelement() * b.element_size() for b in hgan.buffers())
        for p in hgan.parameters():
            p.requires_grad = False
        print("HIFIGAN: MODEL {} : {:.2f} Mo".format(path, size / 1024 ** 2), flush=True)
        hgan.to(device)
        return Vocoder(kind, hgan, sigma, path)
    raise ValueError(f"cannot tell the vocoder type of {path!r}: the path must contain 'waveglow' or 'hifigan'")


def to_pcm16(audio) -> np.ndarray:
    """`MAX_WAV_VALUE * audio` -> host -> astype('int16') (do_syn.py:389-394): truncation, no clip."""
    a = audio.detach().to("cpu", torch.float32).numpy() if isinstance(audio, torch.Tensor) else np.asarray(audio)
    if a.dtype == np.int16:
        return a
    return (np.float32(MAX_WAV_VALUE) * a.astype(np.float32)).astype("int16")


def write_wav(path, pcm, fe_wav=22050):
    """Mono, 16-bit, `fe_wav` Hz, as do_syn.py:285 sets the parameters."""
    pcm = to_pcm16(pcm)
    with wave.open(path, "wb") as wf:
        wf.setparams((1, 2, int(fe_wav), 0, "NONE", "not compressed"))
        wf.writeframes(np.ascontiguousarray(pcm).tobytes())


class RtfMeter:
    """dt_VOC / dt_S accounting of do_syn.py:234,386-394,484, on synchronised wall time."""

    def __init__(self, fe_wav=22050):
        self.fe_wav = float(fe_wav)
        self.dt_voc = 0.0
        self.dt_s = 0.0
        self.utterances = 0
        self._t0 = None

    def start(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._t0 = time.perf_counter()

    def stop(self, n_samples, n_utterances=1):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.dt_voc += time.perf_counter() - self._t0
        self.dt_s += n_samples / self.fe_wav
        self.utterances += n_utterances

    @property
    def rtf(self):
        return self.dt_voc / self.dt_s if self.dt_s > 0 else float("nan")

    @property
    def audio_s_per_s(self):
        return self.dt_s / self.dt_voc if self.dt_voc > 0 else float("nan")

    def report(self, kind="VOCODER"):
        return "RTF {}={:.4f} ({:.2f} s audio in {:.3f} s wall, device-synchronised)".format(
            kind, self.rtf, self.dt_s, self.dt_voc)
