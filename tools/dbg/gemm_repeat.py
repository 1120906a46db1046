"""Diagnostic: k_gemm_bf16x3 run many times on the same operands must give bit-identical results (a transfer consumed before
it has landed shows up as a sporadic difference)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pepper_thesis_amd import runtime, _ffi
lib = _ffi.load()
ctx = runtime.Context(0)
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for (M, N, K, quads) in ((33792, 2048, 512, 1), (40004, 768, 256, 1), (16900, 512, 2112, 0)):
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    splits = 3 if not quads else 1
    out = np.zeros((splits, M, N), np.float32); ref = None; bad = 0
    for k in range(n):
        out[:] = 0
        _ffi.check(lib.pv_debug_gemm_bf16x3(ctx.handle, A.ctypes.data, W.ctypes.data, b.ctypes.data if quads else None, M, N, K, splits, quads, out.ctypes.data, None))
        if ref is None:
            ref = out.copy()
        elif not np.array_equal(ref.view(np.uint32), out.view(np.uint32)):
            bad += 1
    print("M=%d N=%d K=%d: %d runs, %d differ from the first" % (M, N, K, n, bad), flush=True)
