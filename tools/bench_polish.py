#!/usr/bin/env python3
"""Micro-benchmark of the polisher (P2) chain: summary-image builder (8 regions per launch chain, HBM-resident inputs)
and the bi-GRU over the chunks it produced, without leaving HBM, in the fp32 form and with PV_DTYPE_BF16_INPUT_GEMM.
Usage: bench_polish.py [regions]; bench.py reports run() as p2_bigru.polish_chain."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(ctx=None, dev="cuda:0", n=8):
    import torch
    from pepper_thesis_amd import _ffi, runtime, synth
    from pepper_thesis_amd.batch import pack_regions
    from pepper_thesis_amd.device import DeviceBatch, DevicePolishOut
    regs = [synth.synth_region(1234 + 97 * i, site_every=260, ref_start=1_000_000 + i * 100_000) for i in range(n)]
    b = pack_regions(regs)
    cols = int((b.ref_end - b.ref_start + 1).sum())
    out = {}
    for mode, dtype in (("f32", _ffi.PV_DTYPE_F32), ("bf16x3", _ffi.PV_DTYPE_BF16_INPUT_GEMM)):
        c = runtime.Context(int(str(dev).split(":")[-1]) if ":" in str(dev) else 0)
        c.load_p2(synth.make_weights_p2(4321), dtype)
        db = DeviceBatch(b, dev)
        cap = 300 * n
        do = DevicePolishOut(cap, device=dev)
        labels = torch.zeros((cap, 1000), dtype=torch.uint8, device=dev)
        for _ in range(2):
            c.polish_summarize_dev(db, do)
        c.synchronize()
        nck = do.n_chunks()
        assert do.status() == 0 and nck <= cap, (do.status(), nck)
        c.forward_p2_dev(do.images.data_ptr(), nck, labels.data_ptr())
        c.synchronize()
        c.profile_begin()
        reps = 4
        for _ in range(reps):
            c.polish_summarize_dev(db, do)
            c.forward_p2_dev(do.images.data_ptr(), nck, labels.data_ptr())
        pr = c.profile_end()
        ms = {k: v[0] / reps for k, v in pr.items()}
        pipe = ms.get("polish_pipeline", 0.0)
        rnn = sum(v for k, v in ms.items() if k.startswith(("k_gru", "k_rec_bf16", "k_gemm_bf16x3", "k_p2_")))
        alg = b.n_bases + 4 * b.n_cigar + 16 * b.n_reads + nck * 1000 * (10 + 12)   # bases (no qualities) + CIGAR + chunk rows out
        out[mode] = {"regions": n, "columns": cols, "chunks": nck, "builder_ms": pipe, "builder_GBps_algorithmic": alg / pipe / 1e6 if pipe else None,
                     "rnn_ms": rnn, "windows_100col_per_s": nck * 19 / rnn * 1e3 if rnn else None, "chain_mbp_per_s": cols / (pipe + rnn) / 1e3 if pipe + rnn else None,
                     "kernel_ms": {k: round(v, 4) for k, v in ms.items()}}
        c.close()
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run(n=int(sys.argv[1]) if len(sys.argv) > 1 else 8), indent=1))
