"""GPU parity of single convolutions through the C ABI test hooks (include/melvoc.h) against the
C oracle.  Tolerances: fp32 path 1e-4 absolute on O(1) outputs (exact-fp32 MFMA vs double
accumulation); bf16 path against the oracle's bf16-operand model 2e-3 (same rounding of operands,
fp32 vs double accumulation order) and SNR >= 40 dB against the fp32 oracle."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import c_oracle  # noqa: E402
from tacotron2_amd import _native  # noqa: E402


def _snr(ref, x):
    ref = ref.astype(np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((ref - x) ** 2).sum(), 1e-300))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _lrelu(x, s):
    return np.where(x >= 0, x, x * np.float32(s)).astype(np.float32)


CASES = [
    # Cin, Cout, K, dil, L, lengths
    (32, 32, 3, 1, 300, None),
    (32, 32, 11, 5, 700, [700, 333]),
    (64, 64, 7, 3, 515, [515, 100, 1]),
    (128, 128, 11, 5, 260, None),
    (256, 256, 7, 1, 130, [130, 77]),
    (96, 512, 7, 1, 70, [70, 64]),
    (64, 128, 1, 1, 200, None),
]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", CASES)
def test_conv1d_vs_oracle(case, prec):
    cin, cout, k, dil, L, lengths = case
    rng = np.random.default_rng(hash((cin, cout, k, dil)) & 0xFFFF)
    B = len(lengths) if lengths else 2
    x = rng.standard_normal((B, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k)) / np.sqrt(cin * k)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    slope = 0.1
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()  # channels-last
    yd = torch.full((B, L, cout), float("nan"), device="cuda")
    ld = torch.tensor(lengths, dtype=torch.int32, device="cuda") if lengths else None
    rc = _native.lib().mv_test_conv1d(xd.data_ptr(), ld.data_ptr() if ld is not None else None, B, L, cin, _fp(w),
                                      _fp(b), cout, k, dil, slope, _native.PRECISIONS[prec], yd.data_ptr(), None)
    _native.check(rc, "mv_test_conv1d")
    y = yd.cpu().numpy()
    for bi in range(B):
        n = lengths[bi] if lengths else L
        xin = _lrelu(x[bi, :, :n], slope)
        ref = c_oracle.conv1d(xin, w, b, dil, dil * (k - 1) // 2)
        got = y[bi, :n].T
        if prec == "fp32":
            assert np.abs(got - ref).max() < 1e-4
        else:
            refq = c_oracle.conv1d(xin, w, b, dil, dil * (k - 1) // 2, bf16_operands=True)
            assert np.abs(got - refq).max() < 2e-3
            assert _snr(ref, got) >= 40.0
        # rows past the sequence end are never written
        assert np.isnan(y[bi, n:]).all()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", [(64, 32, 16, 8, 4, 90, [90, 41]), (128, 64, 4, 2, 1, 333, None),
                                  (512, 256, 16, 8, 4, 40, [40, 17])])
def test_conv_transpose1d_vs_oracle(case, prec):
    cin, cout, k, stride, pad, L, lengths = case
    rng = np.random.default_rng(cin * 7 + k)
    B = len(lengths) if lengths else 2
    x = rng.standard_normal((B, cin, L)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, k)) / np.sqrt(cin * k / stride)).astype(np.float32)
    b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
    xd = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).cuda()
    yd = torch.full((B, L * stride, cout), float("nan"), device="cuda")
    ld = torch.tensor(lengths, dtype=torch.int32, device="cuda") if lengths else None
    rc = _native.lib().mv_test_conv_transpose1d(xd.data_ptr(), ld.data_ptr() if ld is not None else None, B, L, cin,
                                                _fp(w), _fp(b), cout, k, stride, pad, 0.1,
                                                _native.PRECISIONS[prec], yd.data_ptr(), None)
    _native.check(rc, "mv_test_conv_transpose1d")
    y = yd.cpu().numpy()
    for bi in range(B):
        n = lengths[bi] if lengths else L
        xin = _lrelu(x[bi, :, :n], 0.1)
        ref = c_oracle.conv_transpose1d(xin, w, b, stride, pad)
        got = y[bi, :n * stride].T
        assert ref.shape == got.shape
        if prec == "fp32":
            assert np.abs(got - ref).max() < 1e-4
        else:
            assert _snr(ref, got) >= 40.0
        assert np.isnan(y[bi, n * stride:]).all()


def test_bad_arguments_return_codes():
    l = _native.lib()
    assert l.mv_backend() == 1
    x = torch.zeros(1, 8, 30, device="cuda")
    w = np.zeros((32, 30, 3), np.float32)
    rc = l.mv_test_conv1d(x.data_ptr(), None, 1, 8, 30, _fp(w), None, 32, 3, 1, 1.0, 0, x.data_ptr(), None)
    assert rc == -1 and b"multiples of 32" in l.mv_last_error()
