#!/usr/bin/env python3
"""Micro-benchmark of the P1 RNN kernels alone (per-kernel HIP-event times). Usage: bench_rnn.py [B] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import runtime, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = runtime.Context(0)
ctx.load_p1(synth.make_weights_p1(1234), int(os.environ.get("PV_BENCH_DTYPE", "0")))
x = torch.from_numpy(synth.synth_windows(1, B)).cuda()
probs = torch.zeros((B, 3), dtype=torch.float32, device="cuda")
for _ in range(3):
    ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
ctx.synchronize()
ctx.profile_begin()
for _ in range(iters):
    ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
prof = ctx.profile_end()
tot = 0.0
for k, (ms, n) in prof.items():
    print("%-20s %8.3f ms" % (k, ms / n))
    tot += ms / n
print("B=%d total %.3f ms -> %.0f windows/s (PV_LSTM_ROWS=%s)" % (B, tot, B / tot * 1e3, os.environ.get("PV_LSTM_ROWS", "auto")) + " dtype=" + os.environ.get("PV_BENCH_DTYPE", "0"))
