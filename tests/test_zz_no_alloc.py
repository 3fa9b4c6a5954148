"""mv_hifigan_forward_static (include/melvoc.h): the entry point that allocates nothing -- its scratch is a file-scope
device array inside libmelvoc.so -- against mv_hifigan_forward with a caller-owned workspace.  Same kernels, same
scratch layout, so the bar is bit-exact equality (and through it the parity bars of tests/test_gpu_hifigan.py), on two
calls in a row, the second one smaller (the shared array still holds the first call's contents), for both
precisions and both output formats.  Collected last (file name) so that a failure here hides no other test."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
from tacotron2_amd import _native, synth  # noqa: E402


def _call(h, mel, lengths, pcm, static):
    B, _, T = mel.shape
    wav = torch.full((B, T * h.hop), 3, dtype=torch.int16 if pcm else torch.float32, device="cuda")
    ld = torch.tensor(lengths, dtype=torch.int32, device="cuda")
    fmt = _native.OUT_PCM16 if pcm else _native.OUT_F32
    if static:
        rc = _native.lib().mv_hifigan_forward_static(h.ptr, mel.data_ptr(), ld.data_ptr(), B, T, wav.data_ptr(), fmt, None)
        _native.check(rc, "mv_hifigan_forward_static")
    else:
        n = h.workspace_bytes(B, T)
        ws = torch.empty(n, dtype=torch.uint8, device="cuda")
        rc = _native.lib().mv_hifigan_forward(h.ptr, mel.data_ptr(), ld.data_ptr(), B, T, wav.data_ptr(), fmt,
                                              ws.data_ptr(), n, None)
        _native.check(rc, "mv_hifigan_forward")
    torch.cuda.synchronize()
    return wav.cpu().numpy()


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_static_scratch_entry_point_equals_the_workspace_one(prec):
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    h = _native.create_hifigan(synth.HIFIGAN_V1, st, 0, _native.PRECISIONS[prec])
    big = torch.from_numpy(np.stack([synth.synthetic_mel(60, 21), synth.synthetic_mel(60, 22)])).cuda()
    small = torch.from_numpy(synth.synthetic_mel(9, 23)[None]).cuda()
    for mel, lengths in ((big, [60, 41]), (small, [9])):  # two calls in a row, the second with a smaller input
        for pcm in (False, True):
            want = _call(h, mel, lengths, pcm, static=False)
            got = _call(h, mel, lengths, pcm, static=True)
            assert np.array_equal(got, want), (prec, tuple(mel.shape), pcm, float(np.abs(got.astype(np.float64) - want).max()))
    # ... and through the workspace entry point, the reference itself (C oracle on the small input)
    y = _call(h, small, [9], False, static=True)[0]
    ref = c_oracle.hifigan_forward(st, synth.HIFIGAN_V1, synth.synthetic_mel(9, 23), bf16_operands=prec == "bf16")
    if prec == "fp32":
        assert np.abs(y - ref).max() <= 1e-3
    else:
        r = ref.astype(np.float64)
        assert 10 * np.log10((r ** 2).sum() / ((r - y) ** 2).sum()) >= 40.0


def test_static_scratch_rejects_what_does_not_fit():
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    h = _native.create_hifigan(synth.HIFIGAN_V1, st, 0, _native.PREC_BF16)
    assert h.workspace_bytes(4, 200) > _native.STATIC_SCRATCH_BYTES >= h.workspace_bytes(4, 100)
    mel = torch.zeros((4, 80, 200), device="cuda")
    wav = torch.full((4, 200 * 256), 5.0, device="cuda")
    rc = _native.lib().mv_hifigan_forward_static(h.ptr, mel.data_ptr(), None, 4, 200, wav.data_ptr(), _native.OUT_F32, None)
    assert rc == -4  # MV_ENOMEM
    assert b"static buffer" in _native.lib().mv_last_error()
    torch.cuda.synchronize()
    assert bool((wav == 5.0).all())  # nothing was launched
    # absurd sizes are refused by the same check, not by an overflowed product
    rc = _native.lib().mv_hifigan_forward_static(h.ptr, mel.data_ptr(), None, 2 ** 30, 2 ** 30, wav.data_ptr(),
                                                 _native.OUT_F32, None)
    assert rc == -4


def test_static_entry_point_allocates_nothing_and_is_legal_under_capture():
    """The property the entry point exists for: the call leaves the device's free memory where it was, and it can be
    captured into a HIP graph on a side stream (under capture hipMalloc is illegal) whose replay equals the eager
    call.  An out-of-range `lengths` entry is clamped on the device here as everywhere."""
    st = synth.hifigan_state(synth.HIFIGAN_V1, 0)
    h = _native.create_hifigan(synth.HIFIGAN_V1, st, 0, _native.PREC_BF16)
    lib = _native.lib()
    mel = torch.from_numpy(np.stack([synth.synthetic_mel(48, 31), synth.synthetic_mel(48, 32)])).cuda()
    ld = torch.tensor([48, 10 ** 6], dtype=torch.int32, device="cuda")  # the second entry is out of range: clamped to 48
    eager = torch.zeros((2, 48 * 256), dtype=torch.int16, device="cuda")
    replayed = torch.zeros_like(eager)
    want = _call(h, mel, [48, 48], True, static=False)

    def call(out, stream):
        rc = lib.mv_hifigan_forward_static(h.ptr, mel.data_ptr(), ld.data_ptr(), 2, 48, out.data_ptr(), _native.OUT_PCM16,
                                           stream)
        _native.check(rc, "mv_hifigan_forward_static")

    call(eager, None)  # (first call on this handle: anything one-time has happened before the measurement)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    call(eager, None)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    assert np.array_equal(eager.cpu().numpy(), want)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        call(replayed, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert not replayed.any()  # captured, not run
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(replayed.cpu().numpy(), want)
    replayed.zero_()
    g.replay()  # a graph is there to be replayed
    torch.cuda.synchronize()
    assert np.array_equal(replayed.cpu().numpy(), want)
