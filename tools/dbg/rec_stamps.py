"""Diagnostic: phase cycle sums of k_rec_bf16 (needs the -DPV_REC_STAMPS variant: PEPPER_HIP_LIB=variants/libpepper_hip_recstamps.so)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pepper_thesis_amd import runtime, synth, _ffi
lib = _ffi.load()
lib.pv_debug_rec_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
def stamps():
    o = (C.c_ulonglong * 5)()
    lib.pv_debug_rec_stamps(o)
    return [int(v) for v in o]
names = ["x issue", "mfma", "cell", "init", "barrier"]
def show(tag, nwaves, steps):
    v = stamps(); tot = sum(v)
    print(tag, "cycles per step and wave:", {n: round(x / nwaves / steps) for n, x in zip(names, v)}, "total", round(tot / nwaves / steps), flush=True)
dev = "cuda:0"
ctx = runtime.Context(0)
ctx.load_p1(synth.make_weights_p1(5, 2.0), _ffi.PV_DTYPE_BF16_INPUT_GEMM)
for B in (2048, 8192):
    x = torch.from_numpy(synth.synth_windows(10, B)).to(dev); p = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    ctx.forward_p1_dev(x.data_ptr(), B, p.data_ptr()); ctx.synchronize(); stamps()
    ctx.forward_p1_dev(x.data_ptr(), B, p.data_ptr()); ctx.synchronize()
    show("P1 B=%d enc+dec (2 launches)" % B, 8, 66)
ctx.load_p2(synth.make_weights_p2(17, 2.0), _ffi.PV_DTYPE_BF16_INPUT_GEMM)
for B in (64, 4096):
    y = torch.from_numpy(synth.synth_p2_images(20, B)).to(dev); l = torch.zeros((B, 1000), dtype=torch.uint8, device=dev); a = torch.zeros((B, 1000, 5), dtype=torch.float32, device=dev)
    ctx.forward_p2_dev(y.data_ptr(), B, l.data_ptr(), a.data_ptr()); ctx.synchronize(); stamps()
    ctx.forward_p2_dev(y.data_ptr(), B, l.data_ptr(), a.data_ptr()); ctx.synchronize()
    show("P2 B=%d enc+dec (38 launches)" % B, 4, 3800)
