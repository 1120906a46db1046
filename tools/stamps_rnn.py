#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime sums of the staggered decoder (needs variants/lib_stamps.so built with -DPV_STAMPS)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import _ffi, runtime, synth  # noqa: E402

B = 4096
ctx = runtime.Context(0)
ctx.load_p1(synth.make_weights_p1(1234))
x = torch.from_numpy(synth.synth_windows(1, B)).cuda()
probs = torch.zeros((B, 3), dtype=torch.float32, device="cuda")
for _ in range(3):
    ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
ctx.synchronize()
n = 256 * 8 * 4
out = np.zeros(n, np.uint64)
lib = C.CDLL(_ffi.LIB_PATH)
lib.pv_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = lib.pv_debug_read_stamps(ctx.handle, out.ctypes.data, n)
st = out.reshape(256, 8, 4).astype(np.float64) / 33.0
print("rc", rc, "per-step s_memtime ticks (median over workgroups) [x-part, wait, h-part, cell+publish]")
for w in range(8):
    print("wave", w, np.median(st[:, w, :], axis=0).round(0), "sum", np.median(st[:, w, :].sum(1)).round(0))
