"""Pins the two oracles (oracle/vocoder_oracle.c via ctypes, oracle/torch_ref.py)
against the golden vectors that tests/gen_golden.py captured from the REAL
reference modules (hifigan/models.py, waveglow/glow.py).  CPU only.

Tolerance: the reference is fp32 (oneDNN / native conv); the C oracle accumulates
in double.  1e-4 absolute on O(1) tensors is two orders tighter than the 1e-3
parity bar of BASELINE.json and two looser than the observed 4e-6.
"""
import copy
import os

import numpy as np
import pytest

from oracle import c_oracle, torch_ref
from tacotron2_amd import synth

TOL = 1e-4


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _wg_cfg(c):
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = c
    return cfg


def _small_hifigan():
    cfg = copy.deepcopy(synth.HIFIGAN_V1)
    cfg["upsample_initial_channel"] = 64
    return cfg


def test_c_oracle_hifigan_v1_T64(golden_dir):
    d = _load(golden_dir, "hifigan_v1_T64")
    st = synth.hifigan_state(synth.HIFIGAN_V1, int(d["weight_seed"]))
    mel = synth.synthetic_mel(int(d["T"]), int(d["mel_seed"]))
    wav, stage = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mel, want_stage=True)
    assert wav.shape == d["wav"].shape == (64 * 256,)
    assert np.abs(wav - d["wav"]).max() < TOL
    assert np.abs(stage[:, :32] - d["stage.3.head"]).max() < TOL
    assert np.abs(stage[:, -32:] - d["stage.3.tail"]).max() < TOL
    assert abs(stage.astype(np.float64).sum() - float(d["stage.3.sum"])) < 1e-2 * max(1.0, abs(float(d["stage.3.abs"])) * 1e-4)


def test_torch_ref_hifigan_all_taps(golden_dir):
    d = _load(golden_dir, "hifigan_c64_T48")
    cfg = _small_hifigan()
    st = synth.hifigan_state(cfg, int(d["weight_seed"]))
    mel = synth.synthetic_mel(int(d["T"]), int(d["mel_seed"]))
    taps = {}
    y = torch_ref.hifigan_forward(st, cfg, mel[None], taps=taps)
    assert np.abs(y[0, 0].numpy() - d["wav"]).max() < TOL
    for k, v in taps.items():
        assert np.abs(v[0].numpy() - d[k]).max() < TOL, k
    # 2-D input contract of do_syn.py:391: [80,T] -> [1,256T], same values
    y2 = torch_ref.hifigan_forward(st, cfg, mel)
    assert y2.shape == (1, 48 * 256)
    assert np.array_equal(y2.numpy(), y[0].numpy())


def test_c_oracle_hifigan_small(golden_dir):
    d = _load(golden_dir, "hifigan_c64_T48")
    cfg = _small_hifigan()
    st = synth.hifigan_state(cfg, int(d["weight_seed"]))
    mel = synth.synthetic_mel(int(d["T"]), int(d["mel_seed"]))
    wav, stage = c_oracle.hifigan_forward(st, cfg, mel, want_stage=True)
    assert np.abs(wav - d["wav"]).max() < TOL
    assert np.abs(stage - d["stage.3"]).max() < TOL


def test_c_oracle_conv_primitives(golden_dir):
    """conv_pre and ups.0 of the reduced-width generator against the reference's own tensors."""
    d = _load(golden_dir, "hifigan_c64_T48")
    cfg = _small_hifigan()
    st = synth.hifigan_state(cfg, int(d["weight_seed"]))
    mel = synth.synthetic_mel(int(d["T"]), int(d["mel_seed"]))
    pre = c_oracle.conv1d(mel, st["conv_pre.weight"], st["conv_pre.bias"], 1, 3)
    assert np.abs(pre - d["conv_pre"]).max() < TOL
    x = np.where(pre >= 0, pre, pre * np.float32(0.1)).astype(np.float32)
    up = c_oracle.conv_transpose1d(x, st["ups.0.weight"], st["ups.0.bias"], 8, 4)
    assert up.shape == d["ups.0"].shape
    assert np.abs(up - d["ups.0"]).max() < TOL


def test_c_oracle_ragged_is_per_utterance(golden_dir):
    d = _load(golden_dir, "hifigan_v1_ragged")
    st = synth.hifigan_state(synth.HIFIGAN_V1, int(d["weight_seed"]))
    b = 0  # one is enough on CPU; all three run in the GPU parity test
    T = int(d["lengths"][b])
    mel = synth.synthetic_mel(T, int(d["mel_seed"]) + b)
    wav = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mel)
    assert np.abs(wav - d[f"wav.{b}"]).max() < TOL


def test_fold_weight_norm_matches_reference(golden_dir):
    d = _load(golden_dir, "hifigan_v1_fold")
    wn = synth.hifigan_state(synth.HIFIGAN_V1, 0, weight_norm=True)
    assert len(wn) == 234
    folded = synth.fold_weight_norm(wn)
    assert len(folded) == int(d["n_keys_folded"]) == 156
    for k in ("conv_pre.weight", "ups.0.weight", "ups.3.weight", "resblocks.4.convs1.1.weight", "conv_post.weight"):
        a = folded[k]
        assert np.abs(a.ravel()[:64] - d[k + ".head"]).max() < 1e-6, k
        assert abs(a.astype(np.float64).sum() - float(d[k + ".sum"])) < 1e-3, k
        assert abs(np.abs(a.astype(np.float64)).sum() - float(d[k + ".abs"])) < 1e-4 * float(d[k + ".abs"]), k


@pytest.mark.parametrize("c,name", [(64, "waveglow_c64_T16"), (512, "waveglow_c512_T16")])
def test_c_oracle_waveglow(golden_dir, c, name):
    d = _load(golden_dir, name)
    cfg = _wg_cfg(c)
    st = synth.waveglow_state(cfg, int(d["weight_seed"]))
    B, T = int(d["B"]), int(d["T"])
    noise = synth.waveglow_noise(cfg, B, T, int(d["noise_seed"]))
    chans = synth.waveglow_flow_channels(cfg)
    for b in range(B):
        mel = synth.synthetic_mel(T, int(d["mel_seed"]) + b)
        wav, flows = c_oracle.waveglow_infer(st, cfg, mel, [n[b] for n in noise], float(d["sigma"]), want_flows=True)
        assert np.abs(wav - d["wav"][b]).max() < TOL
        for k in range(cfg["n_flows"]):
            ck = chans[k][0]
            assert np.abs(flows[k][:ck] - d[f"flow.{k}"][b]).max() < TOL, k
        if "spect" in d:
            sp = c_oracle.waveglow_upsample(st, cfg, mel)
            assert np.abs(sp - d["spect"][b]).max() < TOL


@pytest.mark.parametrize("c,name", [(256, "waveglow_c256_T64"), (512, "waveglow_c512_T64")])
def test_torch_ref_waveglow(golden_dir, c, name):
    d = _load(golden_dir, name)
    cfg = _wg_cfg(c)
    st = synth.waveglow_state(cfg, int(d["weight_seed"]))
    B, T = int(d["B"]), int(d["T"])
    mel = np.stack([synth.synthetic_mel(T, int(d["mel_seed"]) + b) for b in range(B)])
    noise = synth.waveglow_noise(cfg, B, T, int(d["noise_seed"]))
    taps = {}
    y = torch_ref.waveglow_infer(st, cfg, mel, noise, float(d["sigma"]), taps=taps)
    assert y.shape == (B, 256 * T)
    assert np.abs(y.numpy() - d["wav"]).max() < TOL
    for k in range(cfg["n_flows"]):
        assert np.abs(taps[f"flow.{k}"].numpy() - d[f"flow.{k}"]).max() < TOL


def test_pcm16_contract(golden_dir):
    """D2 (do_syn.py:389-394): 32768*y, truncate toward zero, no clip."""
    w = _load(golden_dir, "hifigan_v1_T64")["wav"]
    g = _load(golden_dir, "pcm16_v1_T64")["pcm"]
    assert np.array_equal(c_oracle.pcm16(w), g)
    assert np.array_equal(torch_ref.pcm16(w), g)
    edge = np.array([0.99999, -0.99999, 0.5 / 32768, -0.5 / 32768, -1.0], np.float32)
    assert c_oracle.pcm16(edge).tolist() == [32767, -32767, 0, 0, -32768]


def test_bf16_operand_model_meets_snr():
    """The bf16 path rounds MFMA operands only (fp32 accumulate, fp32 residual stream);
    the oracle's model of that must clear BASELINE.json's 40 dB bar, else the design cannot."""
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    mel = synth.synthetic_mel(24, 5)
    a = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mel)
    b = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mel, bf16_operands=True)
    snr = 10 * np.log10((a.astype(np.float64) ** 2).sum() / ((a - b).astype(np.float64) ** 2).sum())
    assert snr >= 40.0, snr


def test_f16_operand_model():
    """MV_PREC_F16's operand conversion as the oracle models it (mode 2 of `bf16_operands`): float32 -> nearest float16,
    ties to even, saturating at +-65504, denormals kept -- bit for bit numpy's astype(float16) on clamped input -- and a
    generator built on it clears the 50 dB bar the float16 path is held to (the bf16 model: 42.9 dB)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4000) * np.logspace(-9, 6, 4000),
                        [7e4, -7e4, 65504, 65519.9, 65520, 1e-8, -3e-6, 6.1e-5, 0.0, -0.0]]).astype(np.float32)
    y = c_oracle.conv1d(x[None, :], np.ones((1, 1, 1), np.float32), np.zeros(1, np.float32), 1, 0, bf16_operands=2)[0]
    assert np.array_equal(y, np.clip(x, -65504, 65504).astype(np.float16).astype(np.float32))
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    mel = synth.synthetic_mel(24, 5)
    a = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mel)
    b = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, mel, bf16_operands=2)
    snr = 10 * np.log10((a.astype(np.float64) ** 2).sum() / ((a - b).astype(np.float64) ** 2).sum())
    assert snr >= 50.0, snr


def test_postnet_oracles_match_reference(golden_dir):
    """model.py:140-174,717 (eval mode) captured from the real reference Postnet, per sequence."""
    d = _load(golden_dir, "postnet_tc2_ragged")
    st = synth.postnet_state(synth.POSTNET_TC2, int(d["seed"]))
    for i, T in enumerate(d["lengths"]):
        x = synth.synthetic_mel(int(T), int(d["mel_seed"]) + i)
        y = c_oracle.postnet_forward(st, synth.POSTNET_TC2, x)
        assert np.abs(y - d[f"post.{i}"]).max() < TOL
        s = c_oracle.postnet_forward(st, synth.POSTNET_TC2, x, add_input=True)
        assert np.abs(s - d[f"sum.{i}"]).max() < TOL
        t = torch_ref.postnet_forward(st, synth.POSTNET_TC2, x, add_input=True).numpy()
        assert np.abs(t - d[f"sum.{i}"]).max() < TOL


def test_pcm16_matches_numpy_outside_the_int16_range():
    """do_syn.py:389-394 is numpy's astype('int16') of 32768*y: no clip, wrap-around, and 0 beyond the int32 range or
    for NaN (the x86 conversion's 'integer indefinite').  WaveGlow's output is unbounded, so the wrap is reachable."""
    import warnings

    y = np.array([0.99999, -1.0, 1.0, 1.5, -1.5, 3.0, 1234.5678, 65535.9, 70000.0, -70000.0, 1e10, -1e10, np.inf, -np.inf,
                  np.nan, 0.0, -0.00001], np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = (np.float32(32768.0) * y).astype("int16")
    assert np.array_equal(c_oracle.pcm16(y), want)
