import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pepper_thesis_amd import _ffi, runtime
ctx = runtime.Context(0); lib = _ffi.load()
def run(M, N, K, splits=1, quads=0, bias=True, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if bias else None
    Cc = np.zeros((splits, M, N), np.float32); ms = C.c_float(0)
    _ffi.check(lib.pv_debug_gemm_bf16x3(ctx.handle, A.ctypes.data, W.ctypes.data, None if b is None else b.ctypes.data, M, N, K, splits, quads, Cc.ctypes.data, C.byref(ms)))
    if quads:
        Cc = Cc.reshape(M // 4, N, 4).transpose(0, 2, 1).reshape(1, M, N)
    got = Cc.sum(0)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + (0 if b is None else b)
    err = np.abs(got - ref)
    bad = np.argwhere(err > 1e-3)
    print("M%d N%d K%d s%d q%d: max err %.3g, bad %d, %.3f ms, %.0f TF(3-term)" % (M, N, K, splits, quads, err.max(), len(bad), ms.value, 6.0 * M * N * K / ms.value / 1e9))
    if len(bad):
        rows = np.unique(bad[:, 0]); cols = np.unique(bad[:, 1])
        print("  bad rows", rows[:24], "...; bad cols", cols[:32])
    return err.max()
run(64, 256, 64, quads=0); run(64, 256, 64, quads=1); run(256, 256, 512, quads=1); run(1056, 2048, 512, quads=1)
run(320, 512, 1024, splits=2, quads=0, bias=False)
run(270336 // 8, 2048, 512, quads=1)
