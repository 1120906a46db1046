"""GPU: the HIP recurrent kernels (through the C-ABI) against the reference's golden vectors and
the NumPy oracle. Floating point: tolerance 1e-4 absolute on softmax probabilities (north_star),
2e-5 on the LSTM layer outputs."""
import os

import numpy as np
import pytest

from oracle import rnn_oracle
from pepper_thesis_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rnn_golden.npz")
TOL_PROBS = 1e-4
TOL_TAPS = 2e-5


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


@pytest.mark.parametrize("tag", ["p1", "p1sharp"])
def test_p1_matches_reference_golden(hip_ctx, gold, tag):
    w = synth.make_weights_p1(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    hip_ctx.load_p1(w)
    probs, enc, dec = hip_ctx.forward_p1(gold[tag + "/images"], taps=True)
    np.testing.assert_allclose(enc[0], gold[tag + "/enc0"], atol=TOL_TAPS, rtol=0)
    np.testing.assert_allclose(dec[0], gold[tag + "/dec0"], atol=TOL_TAPS, rtol=0)
    np.testing.assert_allclose(probs, gold[tag + "/probs"], atol=TOL_PROBS, rtol=0)


@pytest.mark.parametrize("B", [1, 31, 32, 33, 100, 512])
def test_p1_ragged_batches_vs_oracle(hip_ctx, B):
    """batch sizes around the 32-row tile edge; the oracle runs in float64"""
    w = synth.make_weights_p1(99, 2.5)
    hip_ctx.load_p1(w)
    x = synth.synth_windows(1000 + B, B)
    probs = hip_ctx.forward_p1(x)
    nb = min(B, 64)
    ref = rnn_oracle.p1_forward(w, x[:nb], np.float64)
    np.testing.assert_allclose(probs[:nb], ref, atol=TOL_PROBS, rtol=0)
    assert np.abs(probs.sum(1) - 1).max() < 1e-5
    if B > 64:  # rows are independent: the tail of a big batch equals the same windows run alone
        alone = hip_ctx.forward_p1(x[-40:])
        np.testing.assert_allclose(probs[-40:], alone, atol=1e-6, rtol=0)


def test_p1_extreme_inputs(hip_ctx):
    """int8 extremes (-128/127 after the wrap-around cast) and all-zero windows: saturating gates"""
    w = synth.make_weights_p1(7, 2.0)
    hip_ctx.load_p1(w)
    x = np.zeros((4, 33, 26), np.int8)
    x[1] = 127
    x[2] = -128
    x[3, ::2] = 127
    x[3, 1::2] = -128
    probs = hip_ctx.forward_p1(x)
    ref = rnn_oracle.p1_forward(w, x, np.float64)
    assert np.isfinite(probs).all()
    np.testing.assert_allclose(probs, ref, atol=TOL_PROBS, rtol=0)


def test_p1_empty_batch(hip_ctx):
    hip_ctx.load_p1(synth.make_weights_p1(7))
    assert hip_ctx.forward_p1(np.zeros((0, 33, 26), np.int8)).shape == (0, 3)


def test_forward_before_load_fails():
    from pepper_thesis_amd import _ffi, runtime
    c = runtime.Context(0)
    with pytest.raises(_ffi.PepperHipError) as e:
        c.forward_p1(np.zeros((1, 33, 26), np.int8))
    assert e.value.code == _ffi.PV_ERR_STATE
    c.close()
