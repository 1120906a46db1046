#!/usr/bin/env python3
"""Diagnostic: phase cycle sums of k_pileup_tiles. Needs a -DPV_PSTAMPS build:
    python -c "from pepper_thesis_amd import build; print(build.build_variant('pstamps', ['-DPV_PSTAMPS']))"
    PEPPER_HIP_LIB=$PWD/variants/libpepper_hip_pstamps.so python tools/pstamps.py        (on the GPU box)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import _ffi, runtime, synth  # noqa: E402
from pepper_thesis_amd.batch import PRESETS, pack_regions  # noqa: E402
from pepper_thesis_amd.device import DeviceBatch, DeviceOut  # noqa: E402

n = 8
regs = [synth.synth_region(1234 + 97 * i, site_every=260, ref_start=1_000_000 + i * 100_000) for i in range(n)]
b = pack_regions(regs)
ctx = runtime.Context(0)
db = DeviceBatch(b)
do = DeviceOut(512 * n, 16 * 512 * n)
P = PRESETS["ont_r9_guppy5_sup"]
for _ in range(3):
    ctx.summarize_dev(db, P, do)
ctx.synchronize()
lib = C.CDLL(_ffi.LIB_PATH)
out = (C.c_ulonglong * 6)()
lib.pv_debug_read_pstamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.pv_debug_read_pstamps(ctx.handle, out) == 0
names = ["pair batch", "op lookup + indel ops", "scan + staging", "expansion", "barrier after expansion", "flush"]
tot = sum(out)
n_tiles = (int(b.ref.shape[0]) + 511) // 512
for nm, v in zip(names, out):
    print("%-26s %6.1f %%   %8.0f cycles per tile" % (nm, 100.0 * v / tot, v / n_tiles))
print("total %.0f cycles per tile (wave 0 of each tile)" % (tot / n_tiles))
