"""GPU tests of the driver-level path: checkpoint loading (D1), PCM (D2), bucketed / sharded batch
vocoding against per-utterance batch-1 oracle output (the parity definition of SURVEY.md §0)."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
from tacotron2_amd import batch, do_syn_batch, synth, vocoder  # noqa: E402


def test_pickled_reference_waveglow_runs_natively(golden_dir):
    d = np.load(os.path.join(golden_dir, "waveglow_tiny_ref.npz"))
    cfg = json.loads(str(d["cfg"]))
    v = vocoder.load_vocoder(os.path.join(golden_dir, "waveglow_tiny_ref.pt"), device="cuda")
    T = int(d["T"])
    mel = torch.from_numpy(synth.synthetic_mel(T, int(d["mel_seed"]), n_mels=8)[None]).cuda()
    noise = [torch.from_numpy(n) for n in synth.waveglow_noise(cfg, 1, T, int(d["noise_seed"]))]
    y = v.model.infer(mel, sigma=float(d["sigma"]), noise=noise).cpu().numpy()
    assert np.abs(y - d["wav"]).max() <= 1e-3


def test_hifigan_checkpoint_to_pcm(tmp_path, golden_dir):
    path = str(tmp_path / "hifigan_v1_synth.pth.tar")
    wn = synth.hifigan_state(synth.HIFIGAN_V1, 0, weight_norm=True)
    torch.save({"generator": {k: torch.from_numpy(a) for k, a in wn.items()}}, path)
    v = vocoder.load_vocoder(path, device="cuda", precision="fp32")
    mel = torch.from_numpy(synth.synthetic_mel(64, 1234)).cuda()
    # the driver's own sequence: MAX_WAV_VALUE * hgan(mel)[0] -> cpu -> int16 (do_syn.py:391-394)
    audio = vocoder.MAX_WAV_VALUE * v.model(mel)[0]
    pcm = audio.cpu().numpy().astype("int16")
    g = np.load(os.path.join(golden_dir, "pcm16_v1_T64.npz"))["pcm"]
    assert np.abs(pcm.astype(np.int32) - g.astype(np.int32)).max() <= 33
    assert np.array_equal(v.pcm16(mel)[0].cpu().numpy(), pcm)


def test_bucketed_sharded_corpus_equals_batch1():
    v = do_syn_batch.build_vocoder("synthetic-hifigan", torch.device("cuda"), "fp32")
    lengths = [3, 17, 18, 40, 41, 9, 64, 2]
    mels = [synth.synthetic_mel(T, 300 + i) for i, T in enumerate(lengths)]
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    merged, total_utts = {}, 0
    for rank in range(2):  # two "ranks" run one after the other on the one GPU of the test box
        res, stats = batch.synthesize_shard(v, mels, lengths, rank=rank, world=2, device="cuda", max_batch=3,
                                            width=8, out="float")
        assert not (set(res) & set(merged))
        merged.update(res)
        total_utts += stats.utterances
    assert sorted(merged) == list(range(len(lengths))) and total_utts == len(lengths)
    for i in (0, 2, 4, 7):  # oracle runs take a second each; four cover short, long and bucket-mates
        ref = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mels[i])
        assert merged[i].shape == ref.shape
        assert np.abs(merged[i] - ref).max() <= 1e-3


def test_parameter_file_round_trip(tmp_path):
    mel = synth.synthetic_mel(37, 5)
    p = str(tmp_path / "a.WAVEGLOW")
    do_syn_batch.write_parameter_file(p, mel)
    back, (num, den) = do_syn_batch.read_parameter_file(p)
    assert np.array_equal(back, mel) and (num, den) == (11025, 128)  # as_integer_ratio of 22050/256, do_syn.py:473
