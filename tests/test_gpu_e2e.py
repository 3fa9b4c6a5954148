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
