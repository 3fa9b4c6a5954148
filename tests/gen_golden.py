#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference modules.

Runs only in the build container (needs /root/reference); the fixtures it
writes are committed, the reference never travels.  Weights are NOT stored:
they are regenerated from seeds by tacotron2_amd.synth.

    python tests/gen_golden.py

Reference entry points exercised:
  hifigan/models.py:112-174  Generator (load_state_dict with weight_g/weight_v,
                             remove_weight_norm, forward on 3-D and 2-D input)
  waveglow/glow.py:178-313   WaveGlow (remove_weightnorm, upsample, WN.forward,
                             Invertible1x1Conv reverse) walked as infer() does at
                             glow.py:251-295 with explicit noise, because infer()
                             itself allocates torch.cuda tensors (glow.py:262-269)
  model.py:140-174,717       Postnet (eval mode) and the sum with its input
"""
import copy
import json
import os
import sys
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "waveglow"))

from tacotron2_amd import synth  # noqa: E402

GOLD = os.path.join(HERE, "golden")
torch.manual_seed(0)
torch.set_grad_enabled(False)


def ref_generator(cfg, seed):
    from hifigan import AttrDict
    from hifigan.models import Generator

    g = Generator(AttrDict(cfg))
    sd = synth.hifigan_state(cfg, seed, weight_norm=True)
    g.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    g.eval()
    g.remove_weight_norm()
    return g


def hifigan_case(name, cfg, seed, T, mel_seed, slices=True):
    g = ref_generator(cfg, seed)
    mel = synth.synthetic_mel(T, mel_seed)
    y3 = g(torch.from_numpy(mel)[None])  # [1,1,256T]
    y2 = g(torch.from_numpy(mel))  # do_syn.py:391 passes 2-D -> [1,256T]
    assert torch.equal(y3[0], y2)
    out = {"wav": y3[0, 0].numpy(), "T": np.int32(T), "mel_seed": np.int32(mel_seed), "weight_seed": np.int32(seed)}
    # per-stage tensors via forward hooks on the reference's own modules
    acts = {}
    hooks = [g.conv_pre.register_forward_hook(lambda m, i, o: acts.__setitem__("conv_pre", o))]
    for i, up in enumerate(g.ups):
        hooks.append(up.register_forward_hook(lambda m, i_, o, i=i: acts.__setitem__(f"ups.{i}", o)))
    nk = g.num_kernels
    for n, rb in enumerate(g.resblocks):
        # clone: Generator.forward accumulates the MRF sum in place into the first block's output (models.py:159)
        hooks.append(rb.register_forward_hook(lambda m, i_, o, n=n: acts.__setitem__(f"rb.{n}", o.clone())))
    g(torch.from_numpy(mel)[None])
    for h in hooks:
        h.remove()
    for i in range(g.num_upsamples):
        st = sum(acts[f"rb.{i * nk + j}"] for j in range(nk)) / nk
        acts[f"stage.{i}"] = st
    for k, v in acts.items():
        if k.startswith("rb."):
            continue
        a = v[0].numpy()
        if slices and a.size > 70000:
            # head / tail 32 columns + checksums keep the fixture small
            out[f"{k}.head"] = a[:, :32].copy()
            out[f"{k}.tail"] = a[:, -32:].copy()
            out[f"{k}.sum"] = np.float64(a.astype(np.float64).sum())
            out[f"{k}.abs"] = np.float64(np.abs(a.astype(np.float64)).sum())
        else:
            out[k] = a
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "wav rms %.4f" % float(y3.pow(2).mean().sqrt()), "T", T)


def hifigan_ragged(name, cfg, seed, lengths, mel_seed):
    g = ref_generator(cfg, seed)
    out = {"lengths": np.asarray(lengths, np.int32), "weight_seed": np.int32(seed), "mel_seed": np.int32(mel_seed)}
    for b, T in enumerate(lengths):
        mel = synth.synthetic_mel(int(T), mel_seed + b)
        out[f"wav.{b}"] = g(torch.from_numpy(mel))[0].numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, lengths)


def fold_case(name, cfg, seed):
    """Folded-weight checksums incl. the ConvTranspose per-input-channel case."""
    g = ref_generator(cfg, seed)
    sd = g.state_dict()
    out = {}
    for k in ("conv_pre.weight", "ups.0.weight", "ups.3.weight", "resblocks.4.convs1.1.weight", "conv_post.weight"):
        a = sd[k].numpy().astype(np.float64)
        out[k + ".sum"] = np.float64(a.sum())
        out[k + ".abs"] = np.float64(np.abs(a).sum())
        out[k + ".head"] = sd[k].numpy().ravel()[:64].copy()
    out["n_keys_folded"] = np.int32(len(sd))
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "folded keys", len(sd))


def waveglow_case(name, cfg, seed, T, B, mel_seed, noise_seed, sigma):
    from glow import WaveGlow

    m = WaveGlow(**cfg)
    m = WaveGlow.remove_weightnorm(m)
    st = synth.waveglow_state(cfg, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    m.eval()
    mel = np.stack([synth.synthetic_mel(T, mel_seed + b) for b in range(B)])
    noise = synth.waveglow_noise(cfg, B, T, noise_seed)
    # walk of WaveGlow.infer (glow.py:251-295) on the reference's own sub-modules
    spect = m.upsample(torch.from_numpy(mel))
    time_cutoff = m.upsample.kernel_size[0] - m.upsample.stride[0]
    spect = spect[:, :, :-time_cutoff]
    spect = spect.unfold(2, m.n_group, m.n_group).permute(0, 2, 1, 3)
    spect = spect.contiguous().view(spect.size(0), spect.size(1), -1).permute(0, 2, 1)
    audio = sigma * torch.from_numpy(noise[0])
    ni = 1
    out = {"T": np.int32(T), "B": np.int32(B), "sigma": np.float32(sigma), "weight_seed": np.int32(seed),
           "mel_seed": np.int32(mel_seed), "noise_seed": np.int32(noise_seed)}
    small = cfg["WN_config"]["n_channels"] <= 64
    if small:
        out["spect"] = spect.contiguous().numpy()
    else:
        out["spect.head"] = spect[:, :, :16].contiguous().numpy()
        out["spect.sum"] = np.float64(spect.double().sum())
    for k in reversed(range(m.n_flows)):
        n_half = int(audio.size(1) / 2)
        audio_0 = audio[:, :n_half, :]
        audio_1 = audio[:, n_half:, :]
        output = m.WN[k]((audio_0, spect))
        s = output[:, n_half:, :]
        b = output[:, :n_half, :]
        audio_1 = (audio_1 - b) / torch.exp(s)
        audio = torch.cat([audio_0, audio_1], 1)
        audio = m.convinv[k](audio, reverse=True)
        out[f"flow.{k}"] = audio.contiguous().numpy()
        if k == m.n_flows - 1:
            out["wn_out.last"] = output.contiguous().numpy()
        if k % m.n_early_every == 0 and k > 0:
            audio = torch.cat((sigma * torch.from_numpy(noise[ni]), audio), 1)
            ni += 1
    audio = audio.permute(0, 2, 1).contiguous().view(audio.size(0), -1)
    out["wav"] = audio.numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, "wav rms %.4f" % float(audio.pow(2).mean().sqrt()), "max %.3f" % float(audio.abs().max()))


def waveglow_pickle_case(name):
    """A tiny WaveGlow saved the way waveglow/train.py:54-62 saves it (the whole module, weight-norm
    hooks still on, under 'model') plus the walk of infer() on it: pins D1's pickled-module load."""
    import copy as _copy

    from glow import WaveGlow

    cfg = {"n_mel_channels": 8, "n_flows": 4, "n_group": 8, "n_early_every": 2, "n_early_size": 2,
           "WN_config": {"n_layers": 2, "n_channels": 32, "kernel_size": 3}}
    st = synth.waveglow_state(cfg, 5)
    m = WaveGlow(**cfg)
    sd = m.state_dict()
    for k, v in st.items():
        t = torch.from_numpy(v)
        if k in sd:
            sd[k] = t
        else:  # weight-normalised: weight_v = w, weight_g = ||w|| per output channel
            sd[k + "_v"] = t
            sd[k + "_g"] = t.flatten(1).norm(dim=1).view(-1, 1, 1)
    m.load_state_dict(sd)
    torch.save({"model": m, "iteration": 1, "learning_rate": 1e-4}, os.path.join(GOLD, name + ".pt"))
    ref = WaveGlow.remove_weightnorm(_copy.deepcopy(m)).eval()
    T, sigma = 6, 0.6
    mel = synth.synthetic_mel(T, 21, n_mels=8)[None]
    noise = synth.waveglow_noise(cfg, 1, T, 22)
    spect = ref.upsample(torch.from_numpy(mel))[:, :, :-(1024 - 256)]
    spect = spect.unfold(2, 8, 8).permute(0, 2, 1, 3)
    spect = spect.contiguous().view(spect.size(0), spect.size(1), -1).permute(0, 2, 1)
    audio = sigma * torch.from_numpy(noise[0])
    ni = 1
    for k in reversed(range(ref.n_flows)):
        n_half = int(audio.size(1) / 2)
        a0, a1 = audio[:, :n_half, :], audio[:, n_half:, :]
        out = ref.WN[k]((a0, spect))
        a1 = (a1 - out[:, :n_half, :]) / torch.exp(out[:, n_half:, :])
        audio = ref.convinv[k](torch.cat([a0, a1], 1), reverse=True)
        if k % ref.n_early_every == 0 and k > 0:
            audio = torch.cat((sigma * torch.from_numpy(noise[ni]), audio), 1)
            ni += 1
    wav = audio.permute(0, 2, 1).contiguous().view(1, -1).numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), wav=wav, T=np.int32(T), sigma=np.float32(sigma),
                        mel_seed=np.int32(21), noise_seed=np.int32(22), cfg=json.dumps(cfg))
    print(name, "wav rms %.4f" % float(np.sqrt((wav ** 2).mean())))


def pcm_case(name):
    """D2: 32768*y -> astype('int16') exactly as do_syn.py:389-394 executes it (numpy on host)."""
    d = np.load(os.path.join(GOLD, "hifigan_v1_T64.npz"))
    y = torch.from_numpy(d["wav"])
    audio = 32768.0 * y
    pcm = audio.cpu().numpy().astype("int16")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), pcm=pcm)
    print(name, pcm[:8])


def postnet_case(name, hps, seed, lengths, mel_seed):
    """model.py:140-174 Postnet in eval mode on each sequence alone (batch 1), and model.py:717's sum."""
    import model  # the reference's model.py

    ref = model.Postnet(hps, 0)
    st = synth.postnet_state(hps, seed)
    missing = ref.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys) and not missing.unexpected_keys
    ref.eval()
    out = {"lengths": np.asarray(lengths, np.int32), "seed": np.int32(seed), "mel_seed": np.int32(mel_seed)}
    for i, T in enumerate(lengths):
        x = torch.from_numpy(synth.synthetic_mel(int(T), mel_seed + i))[None]
        y = ref(x)
        out[f"post.{i}"] = y[0].numpy()
        out[f"sum.{i}"] = (x + y)[0].numpy()
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if k.startswith("post")})


def tacotron2_case(name, n_text=23, seed=1234, lgs_max=0.3):
    """model.py:725-761 `Tacotron2.inference` of the REFERENCE on the CPU (tc2_french.yaml shapes, batch 1, the
    driver's own settings do_syn.py:100,106,188-189), with the weights of the first decoder's path taken from
    tacotron2_amd.acoustic.synthetic_state -- the fixture pins MelProducer (the build's batched mel producer for
    BASELINE configs[4]) to the reference: the state_dict contract (names and shapes) and the decoded mel itself."""
    import yaml

    import model  # the reference's model.py
    from tacotron2_amd import acoustic

    with open(os.path.join(REF, "tc2_french.yaml")) as f:
        hps = yaml.safe_load(f)
    hps.update(n_symbols=acoustic.TC2_FRENCH["n_symbols"], dim_out_symbols=0, save_embeddings=[], stateful=False,
               code_PAR=0, mask_padding=False, batch_size=1, lgs_max=lgs_max)
    torch.manual_seed(0)
    ref = model.Tacotron2(hps)
    mine = acoustic.MelProducer(dict(acoustic.TC2_FRENCH, lgs_max=lgs_max))
    st = acoustic.synthetic_state(mine, 0)
    ref_sd = ref.state_dict()
    contract = {}
    for k, v in st.items():
        assert k in ref_sd and tuple(ref_sd[k].shape) == tuple(v.shape), k  # every tensor of the mel path exists there
        contract[k] = list(v.shape)
    missing = ref.load_state_dict(st, strict=False)
    assert not missing.unexpected_keys
    ref.eval()
    rng = np.random.default_rng(seed)
    text = torch.from_numpy(rng.integers(1, hps["n_symbols"], size=(1, n_text))).long()
    spk, style = torch.tensor([[3]]), torch.tensor([[5]])
    out = ref.inference((text, spk, style), seed)
    mel, gate = out[0][0], out[2][0]
    # second case, deterministic: the same model with the prenet's always-on dropout (model.py:136) at p = 0, so that
    # a decode on ANOTHER device (other random streams) can be held against it -- tests/test_gpu_e2e.py
    for d in ref.decoder:
        if hasattr(d, "prenet"):
            d.prenet.p_prenet_dropout = 0.0
    out0 = ref.inference((text, spk, style), seed)
    mel0, gate0 = out0[0][0], out0[2][0]
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), text=text.numpy(), spk=np.int64(3), style=np.int64(5),
                        seed=np.int64(seed), lgs_max=np.float64(lgs_max), mel=mel.numpy(), gate=gate.numpy(),
                        mel_nodrop=mel0.numpy(), gate_nodrop=gate0.numpy(), contract=json.dumps(contract))
    print(name, tuple(mel.shape), "mel rms %.4f" % float(mel.pow(2).mean().sqrt()), "| dropout off:", tuple(mel0.shape),
          "max |mel - mel_nodrop| %.3f" % float((mel - mel0).abs().max()))


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--only-tacotron2" in sys.argv:
        tacotron2_case("tacotron2_tc2_decode")
        return
    if "--only-waveglow-c512-T64" in sys.argv:
        waveglow_case("waveglow_c512_T64", synth.WAVEGLOW_512, 0, 64, 1, 1234, 1234, 0.666)
        return
    if "--only-postnet" in sys.argv:
        postnet_case("postnet_tc2_ragged", synth.POSTNET_TC2, 0, [40, 52, 64], 4321)
        return
    v1 = json.load(open(os.path.join(REF, "hifigan", "config.json")))
    hifigan_case("hifigan_v1_T64", v1, 0, 64, 1234)
    hifigan_case("hifigan_v1_T500", v1, 0, 500, 1234)
    small = copy.deepcopy(v1)
    small["upsample_initial_channel"] = 64
    hifigan_case("hifigan_c64_T48", small, 1, 48, 77, slices=False)
    hifigan_ragged("hifigan_v1_ragged", v1, 0, [40, 52, 64], 4321)
    fold_case("hifigan_v1_fold", v1, 0)
    wg64 = copy.deepcopy(synth.WAVEGLOW_512)
    wg64["WN_config"]["n_channels"] = 64
    waveglow_case("waveglow_c64_T16", wg64, 0, 16, 2, 1234, 1234, 0.666)
    waveglow_case("waveglow_c512_T16", synth.WAVEGLOW_512, 0, 16, 1, 1234, 1234, 0.666)
    # several 128- / 256-row tiles, dilation 128 crossing tile edges (L = 32 T = 2048 steps)
    waveglow_case("waveglow_c512_T64", synth.WAVEGLOW_512, 0, 64, 1, 1234, 1234, 0.666)
    wg256 = copy.deepcopy(synth.WAVEGLOW_512)
    wg256["WN_config"]["n_channels"] = 256  # the repo's own waveglow/config.json:37 width
    waveglow_case("waveglow_c256_T64", wg256, 0, 64, 1, 1234, 1234, 0.666)
    pcm_case("pcm16_v1_T64")
    waveglow_pickle_case("waveglow_tiny_ref")
    postnet_case("postnet_tc2_ragged", synth.POSTNET_TC2, 0, [40, 52, 64], 4321)
    tacotron2_case("tacotron2_tc2_decode")


if __name__ == "__main__":
    main()
