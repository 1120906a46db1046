"""Diagnostic: quick accuracy (vs fp32 mode and the float64 oracle) and per-kernel times of the bf16x3 mode, P1 and P2"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pepper_thesis_amd import runtime, synth, _ffi
from oracle import rnn_oracle
dev = "cuda:0"
def timeit(fn, sync, n=5):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3
# P1
w = synth.make_weights_p1(5, 2.0)
for B in (64, 8192):
    x = synth.synth_windows(10 + B, B)
    c32 = runtime.Context(0); c32.load_p1(w); p32 = c32.forward_p1(x)
    cb = runtime.Context(0); cb.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM); cb.set_option("p1_bf16_min_batch", 0)
    pb, enc, dec = cb.forward_p1(x, taps=True)
    rp, renc, rdec, _ = rnn_oracle.p1_forward(w, x[:8], np.float64, taps=True)
    print("P1 B=%d bf16 vs fp32 max err %.3g; vs f64 probs %.3g enc %.3g dec %.3g" % (B, np.abs(pb - p32).max(), np.abs(pb[:8] - rp).max(), np.abs(enc[:8] - renc).max(), np.abs(dec[:8] - rdec).max()), flush=True)
    dx = torch.from_numpy(x).to(dev); dp = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    for name, c in (("fp32", c32), ("bf16", cb)):
        ms = timeit(lambda: c.forward_p1_dev(dx.data_ptr(), B, dp.data_ptr()), lambda: c.synchronize())
        c.profile_begin(); c.forward_p1_dev(dx.data_ptr(), B, dp.data_ptr()); prof = c.profile_end()
        print("  %s: %.3f ms/call = %.0f windows/s; kernels %s" % (name, ms, B / ms * 1e3, {k: round(v[0], 3) for k, v in prof.items()}), flush=True)
    c32.close(); cb.close()
# P2
w2 = synth.make_weights_p2(17, 2.0)
for B in (64, 1000, 4096):
    y = synth.synth_p2_images(20 + B, B)
    c32 = runtime.Context(0); c32.load_p2(w2); l32, a32 = c32.forward_p2(y, want_acc=True)
    cb = runtime.Context(0); cb.load_p2(w2, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    lb, ab = cb.forward_p2(y, want_acc=True)
    lr, ar = rnn_oracle.p2_forward(w2, y[:2], np.float64)
    print("P2 B=%d bf16 vs fp32 acc err %.3g labels differ %.4g; vs f64 %.3g (fp32 mode vs f64 %.3g)" % (B, np.abs(ab - a32).max(), (lb != l32).mean(), np.abs(ab[:2] - ar).max(), np.abs(a32[:2] - ar).max()), flush=True)
    dy = torch.from_numpy(y).to(dev); dl = torch.zeros((B, 1000), dtype=torch.uint8, device=dev); da = torch.zeros((B, 1000, 5), dtype=torch.float32, device=dev)
    for name, c in (("fp32", c32), ("bf16", cb)):
        ms = timeit(lambda: c.forward_p2_dev(dy.data_ptr(), B, dl.data_ptr(), da.data_ptr()), lambda: c.synchronize(), n=3)
        c.profile_begin(); c.forward_p2_dev(dy.data_ptr(), B, dl.data_ptr(), da.data_ptr()); prof = c.profile_end()
        print("  %s: %.2f ms/call = %.0f windows/s; kernels %s" % (name, ms, 19 * B / ms * 1e3, {k: (round(v[0], 2), v[1]) for k, v in prof.items()}), flush=True)
    c32.close(); cb.close()
