"""End-to-end path (BASELINE configs[4]): batched mel producer -> native Postnet -> native vocoder on the device
(tacotron2_amd/do_syn_e2e.py) against the same chain run one utterance at a time at batch 1, which is how the
reference drives it (do_syn.py:302-394).  The producer itself is pinned to the reference's Tacotron2.inference on
the CPU (tests/test_acoustic.py), the Postnet and the vocoders by their own golden tests; this test pins the
plumbing between them (lengths, padding, bucket order, PCM)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tacotron2_amd import do_syn_batch, do_syn_e2e  # noqa: E402


def test_bucketed_e2e_equals_one_utterance_at_a_time():
    dev = torch.device("cuda", 0)
    producer = do_syn_e2e.build_producer(dev)
    producer.decoder[0].prenet.p = 0.0  # dropout draws depend on the batch shape: off for the comparison
    postnet = do_syn_e2e.build_postnet(dev, "fp32")
    voc = do_syn_batch.build_vocoder("synthetic-hifigan", dev, "fp32")
    texts, steps = do_syn_e2e.synthetic_texts(7, seed=3, lo=5, hi=30, steps_per_symbol=1.0)
    res, st = do_syn_e2e.synthesize_shard(producer, postnet, voc, texts, steps, device=dev, max_batch=4, width=16,
                                          keep_audio=True)
    assert st["utterances"] == 7 and st["calls"] >= 2 and sorted(res) == list(range(7))
    assert st["t_tc2"] > 0 and st["t_voc"] > 0 and abs(st["audio_seconds"] - sum(steps) * 2 * 256 / 22050.0) < 1e-9
    for i in (0, 3, 6):
        text = torch.from_numpy(texts[i])[None].to(dev)
        mel, fr, _ = producer.infer(text, None, torch.tensor([i % 13], device=dev), None, seed=1234, n_steps=[steps[i]])
        assert int(fr[0]) == 2 * steps[i]
        post = postnet(mel, add_input=True)
        pcm = voc(post, out="pcm16")[0].cpu().numpy()
        assert pcm.shape == res[i].shape == (steps[i] * 2 * 256,)
        # the batched producer differs from batch 1 by fp32 summation order only (~1e-5 on the mel)
        assert np.abs(pcm.astype(np.int32) - res[i].astype(np.int32)).max() <= 33  # 1e-3 of full scale
    # two ranks, one after the other on this GPU: the shards cover the corpus exactly once
    got = {}
    for r in range(2):
        part, _ = do_syn_e2e.synthesize_shard(producer, postnet, voc, texts, steps, rank=r, world=2, device=dev,
                                              max_batch=4, width=16, keep_audio=True)
        assert not (set(part) & set(got))
        got.update(part)
    assert sorted(got) == list(range(7))
    for i in range(7):  # (the PyTorch side's GEMMs are not bit-reproducible from call to call)
        assert np.abs(got[i].astype(np.int32) - res[i].astype(np.int32)).max() <= 33


def test_graph_replayed_decode_equals_the_eager_loop():
    """The captured-step decode (one HIP graph launch per step) against the eager loop, dropout off."""
    from tacotron2_amd import acoustic

    dev = torch.device("cuda", 0)
    m = do_syn_e2e.build_producer(dev)
    m.decoder[0].prenet.p = 0.0
    rng = np.random.default_rng(9)
    lens = [31, 22, 9]
    text = torch.zeros((3, 31), dtype=torch.long)
    for b, n in enumerate(lens):
        text[b, :n] = torch.from_numpy(rng.integers(1, 96, size=n))
    args = (text.to(dev), torch.tensor(lens), torch.tensor([1, 2, 3], device=dev), None)
    a, fa, ga = m.infer(*args, n_steps=[40, 33, 12], graph=False)
    b, fb, gb = m.infer(*args, n_steps=[40, 33, 12], graph=True)
    assert fa.tolist() == fb.tolist() == [80, 66, 24]
    assert float((a - b).abs().max()) <= 1e-4 and float((ga - gb).abs().max()) <= 1e-3
    # gate-stopped decoding through the graph: every utterance ends at its own first firing step
    bias = m.decoder[0].gate_layer.linear_layer.bias
    bias.data -= float(torch.quantile(ga[:, :12].flatten(), 0.93))  # opens for a few (utterance, step) pairs only
    c, fc, gc = m.infer(*args, max_steps=40, poll=8, graph=False)
    d, fd, gd = m.infer(*args, max_steps=40, poll=8, graph=True)
    assert fc.tolist() == fd.tolist() and min(fc.tolist()) < 80
    n = min(c.shape[2], d.shape[2])
    assert float((c[:, :, :n] - d[:, :, :n]).abs().max()) <= 1e-4


def test_device_decode_equals_the_reference_without_dropout(golden_dir):
    """SURVEY 8f rank 3 pinned ON THE GPU: the fixture's deterministic case (tests/gen_golden.py ran the REFERENCE's
    Tacotron2.inference on the CPU with the prenet dropout at p = 0, model.py:136,532-569,725-761) against the device
    decode -- the eager loop and the HIP-graph replay, at batch 1 and as row 0 of a padded batch of 3."""
    import os

    from tacotron2_amd import acoustic

    d = np.load(os.path.join(golden_dir, "tacotron2_tc2_decode.npz"))
    dev = torch.device("cuda", 0)
    m = acoustic.MelProducer(dict(acoustic.TC2_FRENCH, lgs_max=float(d["lgs_max"])))
    m.load_state_dict(acoustic.synthetic_state(m, 0))
    m = m.eval().to(dev)
    m.decoder[0].prenet.p = 0.0
    ref, ref_gate = d["mel_nodrop"], d["gate_nodrop"]
    steps = ref.shape[2] // m.frames_per_step
    text = torch.from_numpy(d["text"]).long()
    spk, style = torch.tensor([int(d["spk"])], device=dev), torch.tensor([int(d["style"])], device=dev)
    for graph in (False, True):
        # gate-stopped like the reference's own call (it runs into the lgs_max cap), and with the length imposed
        mel, fr, gate = m.infer(text.to(dev), None, spk, style, seed=int(d["seed"]), graph=graph)
        assert int(fr[0]) == ref.shape[2]
        assert float(np.abs(mel.cpu().numpy() - ref).max()) <= 1e-4, graph
        assert float(np.abs(gate.cpu().numpy()[:, : ref_gate.shape[-1]] - ref_gate).max()) <= 1e-3
        mel, fr, _ = m.infer(text.to(dev), None, spk, style, seed=int(d["seed"]), n_steps=[steps], graph=graph)
        assert float(np.abs(mel.cpu().numpy() - ref).max()) <= 1e-4, graph
        # padded batch of 3: the golden text is the longest row; the others are shorter texts with their own masks
        rng = np.random.default_rng(17)
        n0 = text.shape[1]
        lens = [n0, n0 - 6, 5]
        tb = torch.zeros((3, n0), dtype=torch.long)
        tb[0] = text[0]
        for b in (1, 2):
            tb[b, : lens[b]] = torch.from_numpy(rng.integers(1, 96, size=lens[b]))
        spk3 = torch.tensor([int(d["spk"]), 1, 2], device=dev)
        sty3 = torch.tensor([int(d["style"]), 0, 6], device=dev)
        melb, frb, _ = m.infer(tb.to(dev), torch.tensor(lens), spk3, sty3, seed=int(d["seed"]),
                               n_steps=[steps, steps - 7, 4], graph=graph)
        assert frb.tolist() == [ref.shape[2], (steps - 7) * m.frames_per_step, 4 * m.frames_per_step]
        assert float(np.abs(melb[0].cpu().numpy() - ref[0]).max()) <= 1e-4, graph
        assert not melb[2, :, int(frb[2]):].any()


def test_e2e_chain_with_waveglow_against_the_oracle():
    """BASELINE configs[4] names both vocoders (do_syn.py:389: waveglow.infer on the postnet output).  The bucketed
    chain with a WaveGlow vocoder (64 channels, fp32) is replayed by hand with the same device seed -- same decode,
    same Postnet, the three noise tensors drawn explicitly in the reference's order -- and every utterance of that
    replay is held against the C oracle fed the same Postnet output and the same noise."""
    import copy

    from oracle import c_oracle
    from tacotron2_amd import batch, synth, vocoder
    from tacotron2_amd.waveglow import WaveGlow

    dev = torch.device("cuda", 0)
    producer = do_syn_e2e.build_producer(dev)
    producer.decoder[0].prenet.p = 0.0
    postnet = do_syn_e2e.build_postnet(dev, "fp32")
    cfg = copy.deepcopy(synth.WAVEGLOW_512)
    cfg["WN_config"]["n_channels"] = 64
    st = synth.waveglow_state(cfg, 2)
    wg = WaveGlow.remove_weightnorm(WaveGlow(**cfg))
    wg.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()})
    wg.eval().to(dev)
    wg.precision = "fp32"
    voc = vocoder.Vocoder("WAVEGLOW", wg, sigma=0.666)
    texts, steps = do_syn_e2e.synthetic_texts(3, seed=5, lo=6, hi=14, steps_per_symbol=1.0)
    torch.manual_seed(77)
    res, stt = do_syn_e2e.synthesize_shard(producer, postnet, voc, texts, steps, device=dev, max_batch=4, width=64,
                                           keep_audio=True)
    assert stt["utterances"] == 3 and stt["calls"] == 1 and sorted(res) == [0, 1, 2]
    # the same call sequence by hand (one bucket, longest first), noise drawn explicitly
    nfs = producer.frames_per_step
    frames = [s * nfs for s in steps]
    (bk,) = batch.make_buckets(frames, 4, 64, max_frames=4 * 1024)
    lin = [len(texts[i]) for i in bk.indices]
    text = torch.zeros((3, (max(lin) + 31) // 32 * 32), dtype=torch.long)
    for b, i in enumerate(bk.indices):
        text[b, : lin[b]] = torch.from_numpy(texts[i])
    torch.manual_seed(77)
    mel, fr, _ = producer.infer(text.to(dev), torch.tensor(lin), torch.tensor([i % 13 for i in bk.indices], device=dev),
                                None, seed=1234, n_steps=[steps[i] for i in bk.indices])
    T = bk.lengths[0]
    post = postnet(mel.float()[:, :, :T].contiguous(), lengths=fr, add_input=True)
    L = T * 256 // cfg["n_group"]
    noise = [torch.empty((3, c, L), device=dev).normal_() for c in (4, 2, 2)]
    y = wg.infer(post, sigma=0.666, lengths=fr, noise=noise)
    pcm = wg.infer(post, sigma=0.666, lengths=fr, noise=noise, out="pcm16").cpu().numpy()
    post_h, y_h = post.cpu().numpy(), y.cpu().numpy()
    for b, i in enumerate(bk.indices):
        n = frames[i]
        # the chain drew the same noise (a different draw would differ by the noise's own amplitude, ~0.7 of full scale)
        assert np.abs(pcm[b, : n * 256].astype(np.int32) - res[i].astype(np.int32)).max() <= 66, i
        ref = c_oracle.waveglow_infer(st, cfg, post_h[b, :, :n], [x[b, :, : n * 32].cpu().numpy() for x in noise], 0.666)
        assert np.abs(y_h[b, : n * 256] - ref).max() <= 1e-3, i
        assert not y_h[b, n * 256:].any()
