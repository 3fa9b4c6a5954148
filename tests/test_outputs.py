"""Host-side file formats around the vocoder call (tacotron2_amd/outputs.py).
The reference has no tests or fixtures for these files and its drivers cannot be imported here (missing
py3nvml, librosa, ...): expected values below are worked out by hand from do_syn.py:286,396-413,466-478 and
the two readers (meldataset.py:147-148, inference_e2e.py:55-57) -- parity unpinned by the reference."""
import wave

import numpy as np
import pytest

from tacotron2_amd import outputs


def test_fe_ratio_matches_as_integer_ratio():
    assert outputs.fe_ratio(22050) == (22050, 1)
    assert outputs.fe_ratio(22050 / 256) == (11025, 128)      # 86.1328125 is exact in binary
    assert outputs.fe_ratio(np.int64(100)) == (100, 1)


def test_parameter_file_round_trip_and_layout(tmp_path):
    x = np.arange(15, dtype=np.float32).reshape(5, 3) * 0.5
    p = tmp_path / "a.WAVEGLOW"
    outputs.write_parameter_file(p, x, 22050 / 256)
    raw = p.read_bytes()
    assert len(raw) == 16 + 15 * 4
    assert tuple(np.frombuffer(raw[:16], dtype=np.int32)) == (5, 3, 11025, 128)
    assert np.array_equal(np.frombuffer(raw[16:], dtype=np.float32), x.ravel())
    y, fe = outputs.read_parameter_file(p)
    assert np.array_equal(x, y) and fe == 22050 / 256
    # the 2-int reader of inference_e2e.py sees the same shape but starts 8 bytes early
    z, fe2 = outputs.read_parameter_file(p, header_ints=2)
    assert fe2 is None and z.shape == (5, 3)
    assert np.array_equal(z.ravel()[2:], x.ravel()[:13])
    assert z.ravel()[:2].view(np.int32).tolist() == [11025, 128]


def test_parameter_file_errors(tmp_path):
    p = tmp_path / "bad.mel"
    p.write_bytes(np.array([4, 2, 1, 1], np.int32).tobytes() + b"\0" * 12)
    with pytest.raises(ValueError):
        outputs.read_parameter_file(p)
    with pytest.raises(ValueError):
        outputs.write_parameter_file(tmp_path / "c.mel", np.zeros(3, np.float32), 10)
    (tmp_path / "empty.mel").write_bytes(b"")
    with pytest.raises(ValueError):
        outputs.read_parameter_file(tmp_path / "empty.mel")


def test_silence_lengths_truncate_like_the_reference():
    fe = 22050 / 256
    assert outputs.silence_lengths(0.2, fe, 22050) == (17, 4352)     # int(17.2265625), 17 frames * 256
    assert outputs.silence_lengths(0.5, fe, 22050) == (43, 11008)
    assert outputs.silence_lengths(0.0, fe, 22050) == (0, 0)


def test_silence_samples_from_file_or_zeros(tmp_path):
    assert not outputs.silence_samples(5, "NEB", 22050, str(tmp_path)).any()
    with wave.open(str(tmp_path / "sil_NEB_22050.wav"), "wb") as f:
        f.setparams((1, 2, 22050, 0, "NONE", "not compressed"))
        f.writeframes(np.arange(1, 9, dtype=np.int16).tobytes())
    assert outputs.silence_samples(5, "NEB", 22050, str(tmp_path)).tolist() == [1, 2, 3, 4, 5]
    assert outputs.silence_samples(10, "NEB", 22050, str(tmp_path)).tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 0, 0]


def test_utterance_writer_wav_seg_and_parameters(tmp_path):
    stem = str(tmp_path / "utt_syn")
    fe = 22050 / 256
    w = outputs.UtteranceWriter(stem, fe_wav=22050, fe_data=fe, dim=4, lgs_sil_sides=0.1, speaker=None)
    f1 = np.full((10, 4), 1.0, np.float32)
    f1[0] = 7.0
    f2 = np.full((6, 4), 2.0, np.float32)
    w.add_chunk(np.full(2560, 100, np.int16), f1, "bonjour")
    assert w.add_pause(0.2) == (17, 4352)
    w.add_chunk(np.full(1536, -5, np.int16), f2, "monde")
    dur = w.close()
    assert dur == pytest.approx(33 / fe)
    with wave.open(stem + ".wav") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 22050, 2560 + 4352 + 1536)
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    assert (pcm[:2560] == 100).all() and not pcm[2560:6912].any() and (pcm[6912:] == -5).all()
    # .seg: header; per chunk lg+int(s*fe/2), lg+int(s*fe)|text, lg+n/2, lg+n-int(s*fe); final lg-int(s*fe/2), lg
    assert open(stem + ".seg", encoding="utf-8").read().split("\n") == [
        "PTS:PHON|TEXT", "FE: 22050", "0 .",
        "1102 __", "2205 .|bonjour", "1280 utt", "355 .",
        "8014 __", "9117 .|monde", "7680 utt", "6243 .",
        "7346 __", "8448 .", ""]
    par, fe_r = outputs.read_parameter_file(stem + ".WAVEGLOW")
    assert fe_r == fe and par.shape == (33, 4)
    assert (par[10:27] == 7.0).all() and (par[27:] == 2.0).all()
