import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pepper_thesis_amd import _ffi, runtime, synth
w = synth.make_weights_p1(1234, 2.0)
for B in (8, 40, 300):
    x = synth.synth_windows(4, B)
    c32 = runtime.Context(0); c32.load_p1(w); p32, e32, d32 = c32.forward_p1(x, taps=True); c32.close()
    c = runtime.Context(0); c.load_p1(w, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    p, e, d = c.forward_p1(x, taps=True)
    p2 = c.forward_p1(x)
    c.close()
    print("B", B, "enc", np.abs(e - e32).max(), "dec", np.abs(d - d32).max(), "probs", np.abs(p - p32).max(), "probs(no taps)", np.abs(p2 - p32).max())
    dd = np.abs(d - d32).max(axis=2)   # [B,33]
    bad = np.argwhere(dd > 1e-4)
    print(" bad (b,t) count", len(bad), bad[:12].tolist())
    if len(bad):
        b, t = bad[0]
        cols = np.flatnonzero(np.abs(d[b, t] - d32[b, t]) > 1e-4)
        print(" first bad cols", cols[:20], len(cols))
