"""Files either side of the vocoder call, in the reference's formats.

* parameter files: 4 x int32 header (frames, dim, num, den) + frames x dim float32 (README.md:14-17; writer
  do_syn.py:468-478; reader hifi-gan-master/meldataset.py:147-148) and the 2 x int32-header view of the same
  files that hifi-gan-master/inference_e2e.py:55-57 takes (shape from the first 8 bytes, data from offset 8:
  the (num, den) pair is then read as the first two float32 of an extra leading frame -- reproduced as is);
* mono 16-bit WAV at fe_wav plus a `.seg` label track (do_syn.py:285-286,396,467);
* silence between the chunks of one utterance (do_syn.py:399-413): `short_pause` / `long_pause` seconds
  rounded to whole frames, written as samples of `sil_<speaker>_<fe_wav>.wav` when that file exists, zeros
  otherwise, and as repeats of the chunk's first frame in the parameter track.

Host-side only (numpy + wave): no device work here, so it is covered by the CPU tests.
"""
from __future__ import annotations

import os
import struct
import wave

import numpy as np


def fe_ratio(fe):
    """(num, den) of a frame rate, do_syn.py:470-473: ints keep den = 1, floats use as_integer_ratio()."""
    if isinstance(fe, (int, np.integer)):
        return int(fe), 1
    num, den = float(fe).as_integer_ratio()
    return int(num), int(den)


def write_parameter_file(path, frames, fe):
    """frames: [n_frames, dim] float32 (do_syn.py's out_par layout).  Header = shape + (num, den)."""
    frames = np.ascontiguousarray(np.asarray(frames, dtype=np.float32))
    if frames.ndim != 2:
        raise ValueError("parameter file: frames must be [n_frames, dim]")
    num, den = fe_ratio(fe)
    with open(path, "wb") as f:
        f.write(np.asarray(frames.shape + (num, den), dtype=np.int32).tobytes())
        f.write(frames.tobytes(order="C"))


def read_parameter_file(path, header_ints=4):
    """-> (frames [n_frames, dim] float32, fe or None).  header_ints=4: meldataset.py:147-148;
    header_ints=2: inference_e2e.py:55-57 (data from byte 8, so a 4-int file yields its (num, den) words
    reinterpreted as float32 at the head of the stream -- exactly what that script feeds the generator)."""
    if header_ints not in (2, 4):
        raise ValueError("header_ints must be 2 or 4")
    hdr = np.fromfile(path, dtype=np.int32, count=header_ints)
    if hdr.size < header_ints:
        raise ValueError(f"{path}: truncated header")
    n, dim = int(hdr[0]), int(hdr[1])
    if n < 0 or dim <= 0:
        raise ValueError(f"{path}: bad shape ({n}, {dim})")
    x = np.fromfile(path, dtype=np.float32, offset=4 * header_ints, count=n * dim)
    if x.size != n * dim:
        raise ValueError(f"{path}: expected {n * dim} values, found {x.size}")
    fe = None
    if header_ints == 4 and hdr[3] != 0:
        fe = int(hdr[2]) / int(hdr[3])
    return x.reshape(n, dim), fe


def silence_lengths(pause_s, fe_data, fe_wav):
    """(frames, samples) of an inserted pause, do_syn.py:401-405: nt = int(pause*fe_data); ne = int(nt*fe_wav/fe_data)."""
    nt = int(pause_s * fe_data)
    ne = int(nt * fe_wav / fe_data)
    return nt, ne


def silence_samples(ne, speaker=None, fe_wav=22050, directory="."):
    """int16[ne]: the head of sil_<speaker>_<fe_wav>.wav if it exists (do_syn.py:406-409), else zeros (:411)."""
    if speaker is not None:
        nm = os.path.join(directory, "sil_{}_{}.wav".format(speaker, fe_wav))
        if os.path.exists(nm):
            with wave.open(nm) as f:
                raw = f.readframes(ne)
            got = len(raw) // 2
            out = np.zeros(ne, dtype=np.int16)
            out[:got] = np.array(struct.unpack("%ih" % got, raw[:2 * got]), dtype=np.int16)
            return out
    return np.zeros(ne, dtype=np.int16)


class SegWriter:
    """The `.seg` label track next to a synthesised WAV (do_syn.py:286,396,467): sample positions of the
    leading half-silence, the speech onset (with the chunk's text), the chunk's middle and the trailing silence."""

    def __init__(self, path, fe_wav=22050, lgs_sil_sides=0.0):
        self.f = open(path, "w", encoding="utf-8")
        self.fe_wav = fe_wav
        self.sil = lgs_sil_sides
        self.pos = 0
        print("PTS:PHON|TEXT\nFE: {}\n0 .".format(fe_wav), file=self.f)

    def add_chunk(self, text, n_samples):
        s, fe, lg = self.sil, self.fe_wav, self.pos
        print("{} __\n{} .|{}\n{} utt\n{} .".format(lg + int(s * fe / 2), lg + int(s * fe), text,
                                                    lg + int(n_samples / 2), lg + n_samples - int(s * fe)), file=self.f)
        self.pos += n_samples

    def add_silence(self, n_samples):
        self.pos += n_samples

    def close(self):
        if self.f:
            print("{} __\n{} .".format(self.pos - int(self.sil * self.fe_wav / 2), self.pos), file=self.f)
            self.f.close()
            self.f = None


class UtteranceWriter:
    """One utterance = several vocoded chunks with pauses between them: WAV + .seg (+ parameter file), written
    the way do_syn.py:283-286,393-413,466-478 does.  Chunks arrive as int16 PCM (already truncated by the
    vocoder's PCM path) together with the frames they were vocoded from."""

    def __init__(self, stem, fe_wav=22050, fe_data=22050 / 256, dim=80, lgs_sil_sides=0.0, speaker=None,
                 sil_dir=".", ext="WAVEGLOW", parameter_file=True):
        self.stem, self.fe_wav, self.fe_data, self.ext = stem, fe_wav, fe_data, ext
        self.speaker, self.sil_dir, self.parameter_file = speaker, sil_dir, parameter_file
        self.wav = wave.open(stem + ".wav", "wb")
        self.wav.setparams((1, 2, fe_wav, 0, "NONE", "not compressed"))
        self.seg = SegWriter(stem + ".seg", fe_wav, lgs_sil_sides)
        self.out_par = np.empty((0, dim), dtype=np.float32)
        self.n_frames = 0

    def add_chunk(self, pcm, frames, text=""):
        pcm = np.ascontiguousarray(np.asarray(pcm, dtype=np.int16))
        frames = np.asarray(frames, dtype=np.float32)
        self.wav.writeframes(pcm.tobytes())
        self.seg.add_chunk(text, len(pcm))
        self.out_par = np.concatenate((self.out_par, frames))
        self.n_frames += len(frames)
        self._first = frames[0, :] if len(frames) else None

    def add_pause(self, pause_s):
        """after a chunk that is not the utterance's last (do_syn.py:399-413)"""
        nt, ne = silence_lengths(pause_s, self.fe_data, self.fe_wav)
        self.n_frames += nt
        self.wav.writeframes(silence_samples(ne, self.speaker, self.fe_wav, self.sil_dir).tobytes())
        self.seg.add_silence(ne)
        if self._first is not None:
            self.out_par = np.concatenate((self.out_par, np.tile(self._first, (nt, 1))))
        return nt, ne

    def close(self):
        self.wav.close()
        self.seg.close()
        if self.parameter_file:
            write_parameter_file(self.stem + "." + self.ext, self.out_par, self.fe_data)
        return self.n_frames / self.fe_data
