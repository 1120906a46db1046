"""Diagnostic: phase cycle sums of k_gemm_bf16x3's K loop (needs the -DPV_GEMM_STAMPS variant)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pepper_thesis_amd import runtime, _ffi
lib = _ffi.load()
lib.pv_debug_gemm_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
ctx = runtime.Context(0)
rng = np.random.default_rng(0)
for (M, N, K) in ((270336, 2048, 512), (409600, 768, 256)):
    A = rng.standard_normal((M, K)).astype(np.float32); W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    out = np.zeros((1, M, N), np.float32); ms = C.c_float()
    o = (C.c_ulonglong * 8)(); 
    _ffi.check(lib.pv_debug_gemm_bf16x3(ctx.handle, A.ctypes.data, W.ctypes.data, None, M, N, K, 1, 1, out.ctypes.data, C.byref(ms)))
    lib.pv_debug_gemm_stamps(o)
    v = [int(x) for x in o]; steps = v[4] / 8.0 / 2   # 8 waves, two launches (warm + timed)
    tiles = v[4] / (K // 32)
    print("M=%d N=%d K=%d: %.3f ms; per K step and wave (cycles): wait %d barrier %d mfma-block %d between-steps/epilogue %d; K steps per wave %.0f; per TILE: end barrier %d, set-up + first DMAs %d, stores %d, accumulator init + rest %d" % (
        M, N, K, ms.value, v[0] / v[4], v[1] / v[4], v[2] / v[4], (v[3] + v[5] + v[6] + v[7]) / v[4], steps, v[5] / tiles, v[6] / tiles, v[7] / tiles, v[3] / tiles), flush=True)
