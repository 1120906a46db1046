#!/usr/bin/env python3
"""Micro-benchmark of the image-builder pipeline alone (8 regions per launch chain, HBM-resident inputs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import runtime, synth  # noqa: E402
from pepper_thesis_amd.batch import PRESETS, pack_regions  # noqa: E402
from pepper_thesis_amd.device import DeviceBatch, DeviceOut  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
regs = [synth.synth_region(1234 + 97 * i, site_every=260, ref_start=1_000_000 + i * 100_000) for i in range(n)]
b = pack_regions(regs)
ctx = runtime.Context(0)
db = DeviceBatch(b)
do = DeviceOut(512 * n, 16 * 512 * n)
P = PRESETS["ont_r9_guppy5_sup"]
for _ in range(3):
    ctx.summarize_dev(db, P, do)
ctx.synchronize()
ctx.profile_begin()
for _ in range(10):
    ctx.summarize_dev(db, P, do)
pr = ctx.profile_end()
ms = {k: v[0] / v[1] for k, v in pr.items()}
alg = b.algorithmic_bytes(do.n_out())
print("builder alone (%d regions, %d windows):" % (n, do.n_out()), {k: round(v, 4) for k, v in ms.items()})
# the same launch chain without the per-kernel events (which put a few microseconds between kernels)
import torch  # noqa: E402
stream = torch.cuda.Stream(device="cuda:0")
st = stream.cuda_stream
for _ in range(3):
    ctx.summarize_dev(db, P, do, stream=st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(20):
    ctx.summarize_dev(db, P, do, stream=st)
e1.record(stream)
e1.synchronize()
print("back to back, no per-kernel events: %.4f ms per launch chain" % (e0.elapsed_time(e1) / 20))
print("counts (windows, key bytes, status, sites):", do.counts.tolist())
print("algorithmic bytes %d -> %.1f GB/s (%.2f%% of 8 TB/s)" % (alg, alg / ms["summary_pipeline"] / 1e6, alg / ms["summary_pipeline"] / 1e6 / 80))
