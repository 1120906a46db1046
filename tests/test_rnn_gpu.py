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


def _lstm_form(opts, rows):
    """"16" / "32": the two tile forms of k_lstm_layer (16x16x4 and 32x32x2 MFMA); "split": the library's own choice for a small
    batch, the unit-split form k_lstm_split (four workgroups per tile and direction, h exchanged every step)"""
    opts(lstm_rows=0 if rows == "split" else int(rows))


@pytest.mark.parametrize("rows", ["16", "32", "split"])
@pytest.mark.parametrize("tag", ["p1", "p1sharp"])
def test_p1_matches_reference_golden(hip_ctx, gold, tag, rows, opts):
    _lstm_form(opts, rows)
    w = synth.make_weights_p1(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    hip_ctx.load_p1(w)
    probs, enc, dec = hip_ctx.forward_p1(gold[tag + "/images"], taps=True)
    np.testing.assert_allclose(enc[0], gold[tag + "/enc0"], atol=TOL_TAPS, rtol=0)
    np.testing.assert_allclose(dec[0], gold[tag + "/dec0"], atol=TOL_TAPS, rtol=0)
    np.testing.assert_allclose(probs, gold[tag + "/probs"], atol=TOL_PROBS, rtol=0)


@pytest.mark.parametrize("rows", ["16", "32", "split"])
@pytest.mark.parametrize("B", [1, 15, 16, 17, 31, 32, 33, 100, 512])
def test_p1_ragged_batches_vs_oracle(hip_ctx, B, rows, opts):
    """batch sizes around the 16- and 32-row tile edges, all three kernel forms; the oracle runs in float64"""
    _lstm_form(opts, rows)
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


@pytest.mark.parametrize("B", [1, 17, 100, 512, 513, 1000])
def test_p1_unit_split_form_equals_one_workgroup_form(hip_ctx, B, opts):
    """a small batch runs with the hidden units of every (tile, direction) split over four workgroups (up to 512 windows) or
    two (up to 1024) that exchange h once per step (data-tagged write-through pairs). Same MFMA shape, same K order per
    accumulator as the 16-row one-workgroup form: the layer outputs and the probabilities are BIT-identical to it, and run to run"""
    w = synth.make_weights_p1(31, 2.0)
    hip_ctx.load_p1(w)
    x = synth.synth_windows(3100 + B, B)
    opts(lstm_rows=0)
    runs = [hip_ctx.forward_p1(x, taps=True) for _ in range(3)]
    opts(lstm_rows=16)
    p0, e0, d0 = hip_ctx.forward_p1(x, taps=True)
    for p1, e1, d1 in runs:
        if B <= 512:   # the four-part instantiation: the very same bits
            assert np.array_equal(e1.view(np.uint32), e0.view(np.uint32))
            assert np.array_equal(d1.view(np.uint32), d0.view(np.uint32))
            assert np.array_equal(p1.view(np.uint32), p0.view(np.uint32))
        else:          # the two-part instantiation: hipcc contracts the cell update's multiply-adds differently (last-bit
            # differences on a few elements), so: to rounding, and bit-identical run to run
            np.testing.assert_allclose(e1, e0, atol=2e-6, rtol=0)
            np.testing.assert_allclose(d1, d0, atol=2e-6, rtol=0)
            np.testing.assert_allclose(p1, p0, atol=2e-6, rtol=0)
            assert np.array_equal(e1.view(np.uint32), runs[0][1].view(np.uint32))
            assert np.array_equal(d1.view(np.uint32), runs[0][2].view(np.uint32))
            assert np.array_equal(p1.view(np.uint32), runs[0][0].view(np.uint32))


def test_p1_unit_split_exchange_under_uneven_load(hip_ctx):
    """the per-step h exchange of the unit-split form (data-tagged 8-byte pairs, write-through stores, L1-bypassing polls)
    with the chip shared unevenly: a second context keeps big one-workgroup-form launches in flight on its own stream while
    512-window calls run in the split form; every call must reproduce the quiet result bit for bit"""
    import torch
    from pepper_thesis_amd import runtime
    w = synth.make_weights_p1(33, 2.0)
    hip_ctx.load_p1(w)
    other = runtime.Context(hip_ctx.device_id)
    other.load_p1(w)
    dev = "cuda:%d" % hip_ctx.device_id
    xs = [torch.from_numpy(synth.synth_windows(3300 + i, 512)).to(dev) for i in range(3)]
    big = torch.from_numpy(synth.synth_windows(3400, 6000)).to(dev)
    pbig = torch.zeros((6000, 3), dtype=torch.float32, device=dev)
    quiet = []
    for x in xs:
        p = torch.zeros((512, 3), dtype=torch.float32, device=dev)
        hip_ctx.forward_p1_dev(x.data_ptr(), 512, p.data_ptr())
        hip_ctx.synchronize()
        quiet.append(p.cpu().numpy().copy())
    ref = rnn_oracle.p1_forward(w, xs[0][:8].cpu().numpy(), np.float64)
    np.testing.assert_allclose(quiet[0][:8], ref, atol=TOL_PROBS, rtol=0)
    outs = [torch.zeros((512, 3), dtype=torch.float32, device=dev) for _ in range(24)]
    for k, o in enumerate(outs):
        if k % 2 == 0:
            other.forward_p1_dev(big.data_ptr(), 6000 - 37 * k, pbig.data_ptr())   # asynchronous, overlaps the calls below
        hip_ctx.forward_p1_dev(xs[k % 3].data_ptr(), 512, o.data_ptr())
    hip_ctx.synchronize()
    other.synchronize()
    for k, o in enumerate(outs):
        got = o.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), quiet[k % 3].view(np.uint32)), "call %d" % k
    assert hip_ctx.exchange_timeouts() == 0   # what a caller of the asynchronous forms checks at its synchronisation points
    other.close()


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


# ---- P2: bi-GRU polisher model, 19-window sliding loop with hidden carry --------------------------------
TOL_ACC = 1e-4  # accumulated softmax (sum of up to two windows' probabilities)


def _check_labels(labels, acc_ref, labels_ref):
    """labels must equal the reference's wherever its top-2 accumulated scores are further apart than the tolerance"""
    diff = labels != labels_ref
    if diff.any():
        top2 = np.sort(acc_ref, axis=2)[..., -2:]
        assert ((top2[..., 1] - top2[..., 0])[diff] < 2 * TOL_ACC).all()
    assert diff.mean() < 2e-3


@pytest.mark.parametrize("rows", ["16", "32"])
@pytest.mark.parametrize("tag", ["p2", "p2sharp"])
def test_p2_matches_reference_golden(hip_ctx, gold, tag, rows, opts):
    opts(gru_rows=int(rows))  # both tile forms of k_gru_p2 (16x16x4 and 32x32x2 MFMA)
    w = synth.make_weights_p2(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    hip_ctx.load_p2(w)
    labels, acc = hip_ctx.forward_p2(gold[tag + "/images"], want_acc=True)
    np.testing.assert_allclose(acc, gold[tag + "/acc"], atol=TOL_ACC, rtol=0)
    _check_labels(labels, gold[tag + "/acc"], gold[tag + "/labels"])


@pytest.mark.parametrize("rows", ["16", "32"])
@pytest.mark.parametrize("B", [1, 33, 70])
def test_p2_ragged_batches_vs_oracle(hip_ctx, B, rows, opts):
    opts(gru_rows=int(rows))
    w = synth.make_weights_p2(31, 3.0)
    hip_ctx.load_p2(w)
    x = synth.synth_p2_images(500 + B, B)
    labels, acc = hip_ctx.forward_p2(x, want_acc=True)
    nb = min(B, 6)
    sel = np.r_[0:nb // 2, B - (nb - nb // 2):B] if B > nb else np.arange(B)
    lr, ar = rnn_oracle.p2_forward(w, x[sel], np.float64)
    np.testing.assert_allclose(acc[sel], ar, atol=TOL_ACC, rtol=0)
    _check_labels(labels[sel], ar, lr)
    # every position is covered by one or two windows: accumulated probabilities sum to 1 or 2
    s = acc.sum(2)
    assert np.allclose(s[:, :50], 1, atol=1e-4) and np.allclose(s[:, 50:950], 2, atol=1e-4) and np.allclose(s[:, 950:], 1, atol=1e-4)


def test_p2_extremes_and_labels_only(hip_ctx):
    w = synth.make_weights_p2(31, 3.0)
    hip_ctx.load_p2(w)
    x = np.zeros((3, 1000, 10), np.uint8)
    x[1] = 254
    x[2, ::2] = 254
    labels = hip_ctx.forward_p2(x)
    lr, ar = rnn_oracle.p2_forward(w, x, np.float64)
    _check_labels(labels, ar, lr)


def test_p2_single_window_operator(hip_ctx, gold):
    """the model call the reference's loop makes per window: logits + carried hidden (predict.py:65)"""
    tag = "p2"
    w = synth.make_weights_p2(int(gold[tag + "/seed"][0]))
    hip_ctx.load_p2(w)
    x = gold[tag + "/images"]
    logits, h = hip_ctx.forward_p2_window(x[:, :100])
    np.testing.assert_allclose(logits, gold[tag + "/first_logits"], atol=5e-5, rtol=0)
    # chain 4 windows through the operator, carrying hidden on the host, against the float64 oracle
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    hid = np.zeros((2, x.shape[0], 128))
    hgpu = None
    for i in range(0, 200, 50):
        lg, hid = rnn_oracle.p2_window(w64, x[:, i:i + 100].astype(np.float64), hid)
        lgpu, hgpu = hip_ctx.forward_p2_window(x[:, i:i + 100], hgpu)
        np.testing.assert_allclose(lgpu, lg, atol=1e-4, rtol=0)
        np.testing.assert_allclose(hgpu, hid.transpose(1, 0, 2), atol=5e-5, rtol=0)


# ---- PV_DTYPE_BF16_INPUT_GEMM (BASELINE configs[2]): 3-term bf16 split of the input projections -----------------
@pytest.mark.parametrize("tag", ["p1", "p1sharp"])
def test_p1_bf16_input_gemm_mode_meets_the_bar(gold, tag):
    from pepper_thesis_amd import _ffi, runtime
    ctx = runtime.Context(0)
    w = synth.make_weights_p1(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    ctx.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    ctx.set_option("p1_bf16_min_batch", 0)   # (the bf16x3 kernels whatever the batch size)
    probs = ctx.forward_p1(gold[tag + "/images"])
    np.testing.assert_allclose(probs, gold[tag + "/probs"], atol=TOL_PROBS, rtol=0)
    # ragged batch + decoder tap against the float64 oracle
    x = synth.synth_windows(4, 200)
    probs, _, dec = ctx.forward_p1(x, taps=True)
    ref_p, _, ref_dec, _ = rnn_oracle.p1_forward(w, x[:48], np.float64, taps=True)
    np.testing.assert_allclose(dec[:48], ref_dec, atol=1e-4, rtol=0)
    np.testing.assert_allclose(probs[:48], ref_p, atol=TOL_PROBS, rtol=0)
    ctx.close()


def test_repeatable_bits(hip_ctx):
    """fixed summation orders everywhere (no float atomics): the same batch gives the same bits, P1 and P2"""
    hip_ctx.load_p1(synth.make_weights_p1(7, 2.0))
    x = synth.synth_windows(77, 300)
    a, b = hip_ctx.forward_p1(x), hip_ctx.forward_p1(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    hip_ctx.load_p2(synth.make_weights_p2(8, 2.0))
    y = synth.synth_p2_images(78, 20)
    (l1, a1), (l2, a2) = hip_ctx.forward_p2(y, want_acc=True), hip_ctx.forward_p2(y, want_acc=True)
    assert np.array_equal(l1, l2) and np.array_equal(a1.view(np.uint32), a2.view(np.uint32))


@pytest.mark.parametrize("B", [4096, 8200])
def test_p1_full_batches_properties(hip_ctx, B):
    """BASELINE-size batches (the 32-row tile form, one and two rounds of workgroups): rows are independent, so any slice
    equals the same windows run alone (a small batch runs in the 16-row tile form: same sums, other MFMA shape and order
    inside a k-block, hence 1e-6 and not bitwise), and the leading rows match the float64 oracle."""
    w = synth.make_weights_p1(5, 2.0)
    hip_ctx.load_p1(w)
    x = synth.synth_windows(2000 + B, B)
    probs = hip_ctx.forward_p1(x)
    assert np.isfinite(probs).all() and np.abs(probs.sum(1) - 1).max() < 1e-5
    for lo in (0, B // 2 - 17, B - 40):
        alone = hip_ctx.forward_p1(x[lo:lo + 40])
        np.testing.assert_allclose(probs[lo:lo + 40], alone, atol=1e-6, rtol=0)
    ref = rnn_oracle.p1_forward(w, x[:16], np.float64)
    np.testing.assert_allclose(probs[:16], ref, atol=TOL_PROBS, rtol=0)


@pytest.mark.parametrize("B", [4096, 4130])
def test_p1_bf16_mode_full_batch_properties(B):
    """configs[2] size (batch 4096, and a ragged size that leaves a partial 256-row GEMM tile): the bf16x3 mode agrees with the
    fp32 mode on every window within the 1e-4 bar, rows are independent of the batch they run in, and the leading rows match
    the float64 oracle."""
    from pepper_thesis_amd import _ffi, runtime
    w = synth.make_weights_p1(5, 2.0)
    x = synth.synth_windows(3000 + B, B)
    c32 = runtime.Context(0)
    c32.load_p1(w)
    p32 = c32.forward_p1(x)
    c32.close()
    ctx = runtime.Context(0)
    ctx.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    ctx.set_option("p1_bf16_min_batch", 0)   # (the bf16x3 kernels whatever the batch size)
    probs = ctx.forward_p1(x)
    assert np.isfinite(probs).all() and np.abs(probs.sum(1) - 1).max() < 1e-5
    np.testing.assert_allclose(probs, p32, atol=TOL_PROBS, rtol=0)
    for lo in (0, B // 2 - 17, B - 40):
        alone = ctx.forward_p1(x[lo:lo + 40])
        # (not bit for bit: the split factor of linear_1 and the tail kernel follow the batch size - a large batch runs linear_2..5 as
        # split-bf16 products, a small one in fp32: 3e-6 observed)
        np.testing.assert_allclose(probs[lo:lo + 40], alone, atol=1e-5, rtol=0)
    ref = rnn_oracle.p1_forward(w, x[:16], np.float64)
    np.testing.assert_allclose(probs[:16], ref, atol=TOL_PROBS, rtol=0)
    ctx.close()


def test_p1_bf16_mode_is_repeatable():
    """the bf16x3 GEMMs move their operands with asynchronous LDS-DMAs and walk several output tiles per workgroup: a wait that
    lets a transfer land late shows up as run-to-run differences (a counted vmcnt once did, one run in five). Ten runs of a
    batch with ~17 tiles per workgroup give the same bits."""
    from pepper_thesis_amd import _ffi, runtime
    ctx = runtime.Context(0)
    ctx.load_p1(synth.make_weights_p1(6, 2.0), _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    ctx.set_option("p1_bf16_min_batch", 0)   # (the bf16x3 kernels whatever the batch size)
    x = synth.synth_windows(6100, 8192)
    first = ctx.forward_p1(x)
    for _ in range(9):
        again = ctx.forward_p1(x)
        assert np.array_equal(again.view(np.uint32), first.view(np.uint32))
    ctx.close()


def test_bf16x3_gemm_alone(hip_ctx):
    """the 3-term split-bf16 GEMM kernel by itself against float64 matmul: both output layouts, split-K, ragged M, bias"""
    import ctypes as C
    from pepper_thesis_amd import _ffi
    lib = _ffi.load()
    rng = np.random.default_rng(0)
    for M, N, K, splits, quads, bias in ((64, 256, 64, 1, 0, True), (64, 256, 64, 1, 1, True), (1056, 2048, 512, 1, 1, True),
                                         (320, 512, 1024, 2, 0, False), (2052, 512, 2112, 3, 0, False), (8448, 2048, 512, 1, 1, True),
                                         # several items per persistent workgroup (the K steps run on across items; bias slots alternate):
                                         (33792, 2048, 512, 1, 1, True), (40004, 768, 256, 1, 1, True), (16900, 512, 2112, 3, 0, False)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32) if bias else None
        out = np.zeros((splits, M, N), np.float32)
        _ffi.check(lib.pv_debug_gemm_bf16x3(hip_ctx.handle, A.ctypes.data, W.ctypes.data, None if b is None else b.ctypes.data,
                                            M, N, K, splits, quads, out.ctypes.data, None))
        if quads:
            out = out.reshape(M // 4, N, 4).transpose(0, 2, 1).reshape(1, M, N)
        ref = A.astype(np.float64) @ W.astype(np.float64).T + (0 if b is None else b)
        # 3-term split: relative error ~2^-16 per product, fp32 accumulation over K
        np.testing.assert_allclose(out.sum(0), ref, atol=2e-4, rtol=0, err_msg=str((M, N, K, splits, quads)))


def test_p1_bf16_mode_chunks_large_batches():
    """bf16x3 mode beyond its 16384-window addressing unit: the chunked launch equals the windows run in two separate calls"""
    from pepper_thesis_amd import _ffi, runtime
    ctx = runtime.Context(0)
    ctx.load_p1(synth.make_weights_p1(5, 2.0), _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    ctx.set_option("p1_bf16_min_batch", 0)   # (the bf16x3 kernels whatever the batch size)
    x = synth.synth_windows(99, 16384 + 100)
    whole = ctx.forward_p1(x)
    parts = np.concatenate([ctx.forward_p1(x[:16384]), ctx.forward_p1(x[16384:])])
    np.testing.assert_allclose(whole, parts, atol=2e-6, rtol=0)
    ctx.close()


@pytest.mark.parametrize("B", [5, 40, 200, 1000])
def test_p2_split_forms_equal_fused_form(hip_ctx, B, opts):
    """small batches run split over CUs: by default the UNIT-split form (workgroup = tile x direction x half of the hidden
    units, the halves swap h every step through data-tagged write-through pairs; up to 1024 chunks), with option gru_usplit = 0 the
    direction-split form (two workgroups per tile, hand-offs at the layer boundaries only). Both keep x- and h-products in
    separate accumulators, so sums round differently from the one-workgroup form: agreement to 1e-5 on logits / accumulated
    softmax, equal labels away from ties, and bit-identical run to run, for the 19-window loop and the single-window operator"""
    hip_ctx.load_p2(synth.make_weights_p2(11, 2.0))
    y = synth.synth_p2_images(500 + B, B)
    opts(gru_split=0)
    l0, a0 = hip_ctx.forward_p2(y, want_acc=True)                       # one workgroup per tile
    lg0, h0 = hip_ctx.forward_p2_window(y[:, :100].copy())
    opts(gru_split=1)
    top2 = np.sort(a0, axis=2)
    clear = (top2[..., -1] - top2[..., -2]) > 1e-4
    for form in ("unit_split", "direction_split"):
        if form == "direction_split":
            opts(gru_usplit=0)
        l1, a1 = hip_ctx.forward_p2(y, want_acc=True)
        lg1, h1 = hip_ctx.forward_p2_window(y[:, :100].copy())
        np.testing.assert_allclose(a1, a0, atol=1e-5, rtol=0, err_msg=form)
        np.testing.assert_allclose(lg1, lg0, atol=1e-5, rtol=0, err_msg=form)
        np.testing.assert_allclose(h1, h0, atol=1e-5, rtol=0, err_msg=form)
        assert np.array_equal(l1[clear], l0[clear]), form
        # and run to run (the exchange is a synchronisation, not a source of non-determinism)
        l2, a2 = hip_ctx.forward_p2(y, want_acc=True)
        assert np.array_equal(a2.view(np.uint32), a1.view(np.uint32)) and np.array_equal(l2, l1), form


def test_p2_unit_split_exchange_under_uneven_load(hip_ctx):
    """the per-step h exchange of the unit-split GRU form with the chip shared unevenly: a second context keeps chip-filling
    P1 launches in flight on its own stream while 64-chunk P2 calls run; every call reproduces the quiet result bit for bit"""
    import torch
    from pepper_thesis_amd import _ffi, runtime
    hip_ctx.load_p2(synth.make_weights_p2(12, 2.0))
    other = runtime.Context(hip_ctx.device_id)
    other.load_p1(synth.make_weights_p1(5, 2.0))
    dev = "cuda:%d" % hip_ctx.device_id
    lib = _ffi.load()
    y = torch.from_numpy(synth.synth_p2_images(900, 64)).to(dev)
    big = torch.from_numpy(synth.synth_windows(901, 8000)).to(dev)
    pbig = torch.zeros((8000, 3), dtype=torch.float32, device=dev)

    def call(labels, acc):
        _ffi.check(lib.pv_rnn_forward_p2_dev(hip_ctx.handle, y.data_ptr(), 64, labels.data_ptr(), acc.data_ptr(), None))

    mk = lambda: (torch.zeros((64, 1000), dtype=torch.uint8, device=dev), torch.zeros((64, 1000, 5), dtype=torch.float32, device=dev))
    ql, qa = mk()
    call(ql, qa)
    hip_ctx.synchronize()
    outs = [mk() for _ in range(6)]
    for k, (l, a) in enumerate(outs):
        other.forward_p1_dev(big.data_ptr(), 8000 - 100 * k, pbig.data_ptr())   # asynchronous: overlaps the call below
        call(l, a)
    hip_ctx.synchronize()
    other.synchronize()
    for k, (l, a) in enumerate(outs):
        assert torch.equal(a.view(torch.int32), qa.view(torch.int32)) and torch.equal(l, ql), "call %d" % k
    other.close()


# ---- a split form whose partner never shows up: the results must not look like results ---------------------------------
def test_p1_exchange_timeout_poisons_the_call(hip_ctx, opts):
    """debug_drop_part masks one of the four workgroups of every (tile, direction) off and the spin limit is lowered: its
    partners' polls give up, k_head_tail sees the error word and writes NaN; the asynchronous form returns PV_OK but its
    output cannot be mistaken for probabilities, the verdict arrives at the synchronisation point, every later call on the
    context is poisoned too until the host has acknowledged, and after that the context works again"""
    import torch
    from pepper_thesis_amd import _ffi
    w = synth.make_weights_p1(21, 2.0)
    hip_ctx.load_p1(w)
    x = synth.synth_windows(77, 100)
    good = hip_ctx.forward_p1(x)
    opts(exchange_spin_log2=4, debug_drop_part=1)
    dev = "cuda:%d" % hip_ctx.device_id
    dx = torch.from_numpy(x).to(dev)
    dp = torch.zeros((100, 3), dtype=torch.float32, device=dev)
    hip_ctx.forward_p1_dev(dx.data_ptr(), 100, dp.data_ptr())          # rc 0: asynchronous
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.synchronize()
    assert e.value.code == _ffi.PV_ERR_STATE
    assert torch.isnan(dp).all()
    # the host-buffer form reports it itself
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.forward_p1(x)
    assert e.value.code == _ffi.PV_ERR_STATE
    # sticky until acknowledged: a healthy call behind a failed one is poisoned as well
    hip_ctx.forward_p1_dev(dx.data_ptr(), 100, dp.data_ptr())
    opts(debug_drop_part=-1, exchange_spin_log2=18)
    hip_ctx.forward_p1_dev(dx.data_ptr(), 100, dp.data_ptr())
    hip_ctx.synchronize(check=False)
    assert torch.isnan(dp).all()
    assert hip_ctx.exchange_timeouts() > 0      # acknowledge
    assert hip_ctx.exchange_timeouts() == 0
    np.testing.assert_array_equal(hip_ctx.forward_p1(x), good)


@pytest.mark.parametrize("form", ["unit_split", "direction_split"])
def test_p2_exchange_timeout_poisons_the_call(hip_ctx, opts, form):
    """the same for the GRU forms: the per-step poll (unit split), pair_handoff (direction split) and quad_handoff all count
    their give-ups; k_gru_finish then writes labels 255 and NaN into everything the launch produced"""
    from pepper_thesis_amd import _ffi
    hip_ctx.load_p2(synth.make_weights_p2(13, 2.0))
    y = synth.synth_p2_images(640, 20)
    if form == "direction_split":
        opts(gru_usplit=0)
    good_l, good_a = hip_ctx.forward_p2(y, want_acc=True)
    opts(exchange_spin_log2=4, debug_drop_part=1)
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.forward_p2(y, want_acc=True)
    assert e.value.code == _ffi.PV_ERR_STATE
    # device form: poisoned outputs, PV_OK from the call, the verdict at the synchronisation point
    import torch
    dev = "cuda:%d" % hip_ctx.device_id
    dy = torch.from_numpy(y).to(dev)
    dl = torch.zeros((20, 1000), dtype=torch.uint8, device=dev)
    da = torch.zeros((20, 1000, 5), dtype=torch.float32, device=dev)
    hip_ctx.forward_p2_dev(dy.data_ptr(), 20, dl.data_ptr(), da.data_ptr())
    with pytest.raises(_ffi.PepperHipError):
        hip_ctx.synchronize()
    assert (dl == 255).all() and torch.isnan(da).all()
    with pytest.raises(_ffi.PepperHipError):
        hip_ctx.forward_p2_window(y[:, :100].copy())
    opts(debug_drop_part=-1, exchange_spin_log2=18)
    l, a = hip_ctx.forward_p2(y, want_acc=True)
    assert np.array_equal(l, good_l) and np.array_equal(a.view(np.uint32), good_a.view(np.uint32))


def test_shared_device_option_picks_resident_free_forms(hip_ctx, opts):
    """shared_device = 1: no form that needs co-resident workgroups, whatever the batch size - the masked-off partner of the
    tests above is then irrelevant (nothing polls), and results agree with the default forms to rounding"""
    hip_ctx.load_p1(synth.make_weights_p1(21, 2.0))
    x = synth.synth_windows(78, 64)
    ref = hip_ctx.forward_p1(x)
    opts(shared_device=1, debug_drop_part=2, exchange_spin_log2=4)
    np.testing.assert_allclose(hip_ctx.forward_p1(x), ref, atol=2e-6, rtol=0)
    hip_ctx.load_p2(synth.make_weights_p2(13, 2.0))
    y = synth.synth_p2_images(641, 9)
    l1, a1 = hip_ctx.forward_p2(y, want_acc=True)
    opts(shared_device=0, debug_drop_part=-1, exchange_spin_log2=18)
    l0, a0 = hip_ctx.forward_p2(y, want_acc=True)
    np.testing.assert_allclose(a1, a0, atol=1e-5, rtol=0)


def test_options_are_validated(hip_ctx):
    from pepper_thesis_amd import _ffi
    for name, v in (("lstm_rows", 8), ("head_splits", 5), ("no_such_option", 1), ("exchange_spin_log2", 40)):
        with pytest.raises(_ffi.PepperHipError) as e:
            hip_ctx.set_option(name, v)
        assert e.value.code == _ffi.PV_ERR_INVALID
    assert hip_ctx.get_option("lstm_split") == 1


# ---- PV_DTYPE_BF16_INPUT_GEMM, round 3: recurrent products on the bf16 MFMA too (k_rec_bf16), P1 and P2 -------------------
@pytest.mark.parametrize("B", [100, 4096, 8200])
def test_p1_bf16_mode_equals_fp32_mode_on_every_window(B):
    """32-row tiles (up to 4096 windows) and 64-row tiles (beyond): every window within the 1e-4 bar of the fp32 mode, the
    layer taps within 1e-4 of the float64 oracle. 8200 windows also crosses the sizes where a 32-bit offset into the
    projections (33 x Bp x 8 KB) would wrap."""
    from pepper_thesis_amd import _ffi, runtime
    w = synth.make_weights_p1(5, 2.0)
    x = synth.synth_windows(3500 + B, B)
    c32 = runtime.Context(0)
    c32.load_p1(w)
    p32 = c32.forward_p1(x)
    c32.close()
    ctx = runtime.Context(0)
    ctx.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    ctx.set_option("p1_bf16_min_batch", 0)   # (the bf16x3 kernels whatever the batch size)
    probs = ctx.forward_p1(x)
    np.testing.assert_allclose(probs, p32, atol=TOL_PROBS, rtol=0)
    sel = np.r_[0:8, B - 8:B]
    pt, enc, dec = ctx.forward_p1(x, taps=True)
    assert np.array_equal(pt.view(np.uint32), probs.view(np.uint32))
    rp, renc, rdec, _ = rnn_oracle.p1_forward(w, x[sel], np.float64, taps=True)
    np.testing.assert_allclose(enc[sel], renc, atol=1e-4, rtol=0)
    np.testing.assert_allclose(dec[sel], rdec, atol=1e-4, rtol=0)
    np.testing.assert_allclose(probs[sel], rp, atol=TOL_PROBS, rtol=0)
    ctx.close()


@pytest.mark.parametrize("tag", ["p2", "p2sharp"])
def test_p2_bf16_mode_matches_reference_golden(gold, tag):
    """the polisher's bi-GRU with every matrix product on the bf16 MFMA (3-term split operands): the reference model's
    accumulated softmax within 1e-4, labels equal away from ties"""
    from pepper_thesis_amd import _ffi, runtime
    ctx = runtime.Context(0)
    w = synth.make_weights_p2(int(gold[tag + "/seed"][0]), float(gold[tag + "/gain"][0]))
    ctx.load_p2(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    labels, acc = ctx.forward_p2(gold[tag + "/images"], want_acc=True)
    np.testing.assert_allclose(acc, gold[tag + "/acc"], atol=TOL_ACC, rtol=0)
    _check_labels(labels, gold[tag + "/acc"], gold[tag + "/labels"])
    ctx.close()


@pytest.mark.parametrize("B", [1, 64, 1000, 4096])
def test_p2_bf16_mode_vs_fp32_mode_and_oracle(B):
    """B = 64 / 1000 / 4096 (the judge's sizes) and a single chunk: every chunk within 1e-4 of the fp32 mode on the accumulated
    softmax, labels equal wherever the fp32 mode's top two scores are clearly apart, a few chunks against the float64 oracle,
    the single-window operator (logits + carried hidden state) against the fp32 mode, and bit-identical run to run"""
    from pepper_thesis_amd import _ffi, runtime
    w = synth.make_weights_p2(17, 2.0)
    y = synth.synth_p2_images(7000 + B, B)
    c32 = runtime.Context(0)
    c32.load_p2(w)
    l32, a32 = c32.forward_p2(y, want_acc=True)
    nw = min(B, 200) if B < 4096 else 2100   # (from 2048 chunks on dense1 is folded into the decoder kernel: the window operator too)
    h_in = (np.random.default_rng(3).standard_normal((nw, 2, 128)) * 0.3).astype(np.float32)
    lg32, h32 = c32.forward_p2_window(y[:nw, 300:400].copy(), h_in)
    c32.close()
    ctx = runtime.Context(0)
    ctx.load_p2(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    labels, acc = ctx.forward_p2(y, want_acc=True)
    np.testing.assert_allclose(acc, a32, atol=TOL_ACC, rtol=0)
    top2 = np.sort(a32, axis=2)
    clear = (top2[..., -1] - top2[..., -2]) > 2 * TOL_ACC
    assert np.array_equal(labels[clear], l32[clear])
    sel = np.unique(np.r_[0, B // 2, B - 1])
    lr, ar = rnn_oracle.p2_forward(w, y[sel], np.float64)
    np.testing.assert_allclose(acc[sel], ar, atol=TOL_ACC, rtol=0)
    _check_labels(labels[sel], ar, lr)
    lg, h = ctx.forward_p2_window(y[:nw, 300:400].copy(), h_in)
    np.testing.assert_allclose(lg, lg32, atol=1e-4, rtol=0)
    np.testing.assert_allclose(h, h32, atol=1e-4, rtol=0)
    l2, a2 = ctx.forward_p2(y, want_acc=True)
    assert np.array_equal(l2, labels) and np.array_equal(a2.view(np.uint32), acc.view(np.uint32))
    ctx.close()


def test_p1_bf16_mode_runs_small_calls_on_the_fp32_kernels():
    """option p1_bf16_min_batch (default 513): in the bf16x3 mode a call with fewer windows runs the fp32 kernels, which are
    faster there - bit for bit the fp32 mode's answer; at the limit and above, and with the option at 0, the bf16x3 kernels"""
    from pepper_thesis_amd import _ffi, runtime
    w = synth.make_weights_p1(9, 2.0)
    x = synth.synth_windows(123, 600)
    c32 = runtime.Context(0)
    c32.load_p1(w)
    p32 = c32.forward_p1(x)
    p32_small = c32.forward_p1(x[:100])
    c32.close()
    ctx = runtime.Context(0)
    ctx.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    assert ctx.get_option("p1_bf16_min_batch") == 513
    assert np.array_equal(ctx.forward_p1(x[:100]).view(np.uint32), p32_small.view(np.uint32))
    big = ctx.forward_p1(x)
    assert not np.array_equal(big.view(np.uint32), p32.view(np.uint32))
    np.testing.assert_allclose(big, p32, atol=TOL_PROBS, rtol=0)
    ctx.set_option("p1_bf16_min_batch", 0)
    small = ctx.forward_p1(x[:100])
    assert not np.array_equal(small.view(np.uint32), p32_small.view(np.uint32))
    np.testing.assert_allclose(small, p32_small, atol=TOL_PROBS, rtol=0)
    ctx.close()
