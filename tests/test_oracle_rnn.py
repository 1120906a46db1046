"""CPU: the NumPy RNN oracle against the golden vectors produced by the REFERENCE's model classes
(tests/golden/make_rnn_golden.py). Floating point: tolerance 1e-5 absolute for the float64 oracle
against the reference's fp32 torch kernels (north_star bar for the product is 1e-4)."""
import os

import numpy as np
import pytest

from oracle import rnn_oracle
from pepper_thesis_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rnn_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


@pytest.mark.parametrize("tag", ["p1", "p1sharp"])
def test_p1_oracle_matches_reference(gold, tag):
    w = synth.make_weights_p1(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    probs, enc, dec, _ = rnn_oracle.p1_forward(w, gold[tag + "/images"], np.float64, taps=True)
    np.testing.assert_allclose(enc[0], gold[tag + "/enc0"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(dec[0], gold[tag + "/dec0"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(probs, gold[tag + "/probs"], atol=1e-5, rtol=0)
    assert np.abs(probs.sum(1) - 1).max() < 1e-12


@pytest.mark.parametrize("tag", ["p2", "p2sharp"])
def test_p2_oracle_matches_reference(gold, tag):
    w = synth.make_weights_p2(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    labels, acc = rnn_oracle.p2_forward(w, gold[tag + "/images"], np.float64)
    np.testing.assert_allclose(acc, gold[tag + "/acc"], atol=2e-5, rtol=0)
    # labels may differ only where the two best accumulated scores are within the tolerance
    diff = labels != gold[tag + "/labels"]
    if diff.any():
        top2 = np.sort(acc, axis=2)[..., -2:]
        assert ((top2[..., 1] - top2[..., 0])[diff] < 1e-4).all()
    assert diff.mean() < 1e-3


def test_p2_first_window_and_hidden(gold):
    tag = "p2"
    w = {k: v.astype(np.float64) for k, v in synth.make_weights_p2(int(gold[tag + "/seed"][0])).items()}
    x = gold[tag + "/images"].astype(np.float64)
    logits, h = rnn_oracle.p2_window(w, x[:, :100], np.zeros((2, x.shape[0], 128)))
    np.testing.assert_allclose(logits, gold[tag + "/first_logits"], atol=2e-5, rtol=0)


def test_fp32_oracle_close_to_fp64():
    w = synth.make_weights_p1(5, 2.0)
    x = synth.synth_windows(6, 4)
    a = rnn_oracle.p1_forward(w, x, np.float64)
    b = rnn_oracle.p1_forward(w, x, np.float32)
    assert np.abs(a - b).max() < 1e-4


def test_weight_generator_is_stable():
    """the fixture weights are regenerated from a seed: pin a few values"""
    w = synth.make_weights_p1(1234)
    assert w["encoder.weight_ih_l0"].shape == (1024, 26) and w["linear_1.weight"].shape == (512, 16896)
    assert sum(v.size for v in w.values()) == 11862019
    assert float(w["encoder.weight_ih_l0"][0, 0]) == np.float32(-0.04916325584053993)
    assert float(w["linear_1.weight"][511, 16895]) == np.float32(0.005830463487654924)
    assert float(w["output_layer_type.bias"][2]) == np.float32(0.017180591821670532)
    w2 = synth.make_weights_p2(4321)
    assert sum(v.size for v in w2.values()) == 405253
