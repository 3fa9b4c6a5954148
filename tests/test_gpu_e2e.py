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
    for i in range(7):
        assert np.array_equal(got[i], res[i])
