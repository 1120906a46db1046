"""Diagnostic (GPU box): the bf16x3 mode against the fp32 mode over batch sizes around every tile / form boundary (P1: 32, 64, 256,
4096 windows; P2: 16, 32, 2048 chunks), every row compared (bar 1e-4)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from pepper_thesis_amd import runtime, synth, _ffi
t0 = time.time()
w = synth.make_weights_p1(5, 2.0)
c32 = runtime.Context(0); c32.load_p1(w)
cb = runtime.Context(0); cb.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM); cb.set_option("p1_bf16_min_batch", 0)
worst = 0.0
sizes = [1, 31, 32, 33, 63, 64, 65, 255, 256, 257, 511, 513, 1000, 2047, 2049, 4031, 4032, 4033, 4095, 4096, 4097, 4159, 4160, 4161, 6000, 8191, 8193]
xall = synth.synth_windows(77, max(sizes))
for B in sizes:
    x = xall[:B]
    a, b = c32.forward_p1(x), cb.forward_p1(x)
    e = float(np.abs(a - b).max())
    worst = max(worst, e)
    assert np.isfinite(b).all() and e < 1e-4, ("P1", B, e)
print("P1: %d sizes, worst |bf16x3 - fp32| %.3g" % (len(sizes), worst), flush=True)
c32.close(); cb.close()
w2 = synth.make_weights_p2(17, 2.0)
c32 = runtime.Context(0); c32.load_p2(w2)
cb = runtime.Context(0); cb.load_p2(w2, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
sizes = [1, 15, 16, 17, 31, 32, 33, 47, 100, 1023, 1025, 2033, 2047, 2048, 2049, 2063, 2064, 2065, 2100, 3000, 4095, 4097]
yall = synth.synth_p2_images(78, max(sizes))
worst = 0.0
for B in sizes:
    y = yall[:B]
    (l0, a0), (l1, a1) = c32.forward_p2(y, want_acc=True), cb.forward_p2(y, want_acc=True)
    e = float(np.abs(a0 - a1).max())
    worst = max(worst, e)
    top2 = np.sort(a0, axis=2)
    clear = (top2[..., -1] - top2[..., -2]) > 2e-4
    assert np.isfinite(a1).all() and e < 1e-4 and np.array_equal(l0[clear], l1[clear]), ("P2", B, e)
print("P2: %d sizes, worst |bf16x3 - fp32| on the accumulated softmax %.3g" % (len(sizes), worst), flush=True)
print("fuzz_sizes: ok in %.0f s" % (time.time() - t0))
