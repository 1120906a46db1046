#!/usr/bin/env python3
"""Micro-benchmark of the P2 (bi-GRU polisher) kernel. Usage: bench_gru.py [B] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import _ffi, runtime, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = runtime.Context(0)
ctx.load_p2(synth.make_weights_p2(4321))
x = torch.from_numpy(synth.synth_p2_images(1, B)).cuda()
labels = torch.zeros((B, 1000), dtype=torch.uint8, device="cuda")
lib = _ffi.load()


def fwd():
    _ffi.check(lib.pv_rnn_forward_p2_dev(ctx.handle, x.data_ptr(), B, labels.data_ptr(), None, None))


fwd()
ctx.synchronize()
ctx.profile_begin()
for _ in range(iters):
    fwd()
prof = ctx.profile_end()
key = "k_gru_us" if "k_gru_us" in prof else "k_gru_p2"   # unit-split form for small batches
ms = prof[key][0] / prof[key][1]
flop = 80_435_200 * 19 * B
print(key, "B=%d chunks: %.2f ms -> %.0f chunks/s, %.0f 100-col windows/s, %.2f TFLOP/s (fp32 peak 157.3)" %
      (B, ms, B / ms * 1e3, 19 * B / ms * 1e3, flop / ms / 1e9))

if os.environ.get("PV_GRU_STAMPS"):
    import ctypes as C
    from pepper_thesis_amd import _ffi
    lib = C.CDLL(_ffi.LIB_PATH)
    out = (C.c_ulonglong * 8)()
    lib.pv_debug_read_gru_stamps.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.pv_debug_read_gru_stamps(ctx.handle, out) == 0
    names = ["x_load issue", "h-part", "x-part + cell", "x_store + barrier"]
    for lay, off in (("encoder", 0), ("decoder", 4)):
        tot = sum(out[off + i] for i in range(4))
        print(lay, "cycles per step (last launch, wave 0 of workgroup 0):", {n: round(out[off + i] / 1900.0) for i, n in enumerate(names)}, "total", round(tot / 1900.0))
