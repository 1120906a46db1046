#!/usr/bin/env python3
"""Micro-benchmark of the polisher (P2) chain: summary-image builder (8 regions per launch chain, HBM-resident inputs)
and the bi-GRU over the chunks it produced, without leaving HBM. Usage: bench_polish.py [regions]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import runtime, synth  # noqa: E402
from pepper_thesis_amd.batch import pack_regions  # noqa: E402
from pepper_thesis_amd.device import DeviceBatch, DevicePolishOut  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
regs = [synth.synth_region(1234 + 97 * i, site_every=260, ref_start=1_000_000 + i * 100_000) for i in range(n)]
b = pack_regions(regs)
ctx = runtime.Context(0)
ctx.load_p2(synth.make_weights_p2(4321))
db = DeviceBatch(b)
cap = 300 * n
do = DevicePolishOut(cap)
labels = torch.zeros((cap, 1000), dtype=torch.uint8, device="cuda")
for _ in range(3):
    ctx.polish_summarize_dev(db, do)
ctx.synchronize()
nck = do.n_chunks()
assert do.status() == 0 and nck <= cap, (do.status(), nck)
ctx.profile_begin()
for _ in range(10):
    ctx.polish_summarize_dev(db, do)
    ctx.forward_p2_dev(do.images.data_ptr(), nck, labels.data_ptr())
pr = ctx.profile_end()
ms = {k: v[0] / v[1] for k, v in pr.items()}
cols = int((b.ref_end - b.ref_start + 1).sum())
alg = b.n_bases + 4 * b.n_cigar + 16 * b.n_reads + nck * 1000 * (10 + 12)  # bases (no qualities) + CIGAR + chunk rows out
print("polish chain (%d regions, %d columns, %d rows, %d chunks):" % (n, cols, int(do.counts[1].item()), nck),
      {k: round(v, 4) for k, v in ms.items()})
pipe = ms["polish_pipeline"]
print("builder: algorithmic bytes %d -> %.1f GB/s (%.2f%% of 8 TB/s), %.1f Mbp/s" % (alg, alg / pipe / 1e6, alg / pipe / 1e6 / 80, cols / pipe / 1e3))
gru = sum(v for k, v in ms.items() if "gru" in k)
if gru > 0:
    print("bi-GRU: %.3f ms for %d chunks -> %.0f 100-col windows/s; chain %.1f Mbp/s" % (gru, nck, nck * 19 / gru * 1e3, cols / (pipe + gru) / 1e3))
