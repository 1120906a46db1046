"""Secondary figure: the haplotag-aware image builder (pv_summarize_regions_hp_dev, 48 planes x 21 rows) on the headline
workload's regions with an HP tag drawn for every read (40 % untagged, 30 % / 30 % haplotype 1 / 2), inputs resident in HBM.
Used by bench.py (`hp_builder`) and runnable alone: python tools/bench_hp.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(ctx, dev, regions=None, reps=20):
    import torch
    from pepper_thesis_amd import _ffi, synth
    from pepper_thesis_amd.batch import PRESETS, hp_params, pack_regions
    from pepper_thesis_amd.device import DeviceBatch, DeviceOut
    if regions is None:
        regions = [synth.synth_region(1234 + 97 * i, region_len=100_200, depth=60, read_len=10_000, site_every=198,
                                      ref_start=1_000_000 + i * 100_000) for i in range(16)]   # bench.py's regions
    rng = np.random.default_rng(7)
    for r in regions:
        for rd in r.reads:
            rd.hp_tag = int(rng.choice((0, 0, 0, 0, 1, 1, 1, 2, 2, 2)))
    batch = pack_regions(regions)
    P = hp_params(PRESETS["ont_r9_guppy5_sup"])
    db = DeviceBatch(batch, dev)
    cap = 16384
    img = torch.zeros((cap, _ffi.PV_HP_WINDOW_ROWS, _ffi.PV_HP_FEATURES), dtype=torch.int8, device=dev)
    dout = DeviceOut(cap, cap * 16, dev, images=img)
    stream = torch.cuda.Stream(device=dev)   # the launches and the timing events share this stream
    st = stream.cuda_stream
    for _ in range(3):
        ctx.summarize_hp_dev(db, P, dout, stream=st)
    torch.cuda.synchronize()
    assert dout.status() == 0, dout.status()
    n_win = dout.n_out()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        ctx.summarize_hp_dev(db, P, dout, stream=st)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    cols = int((batch.ref_end - batch.ref_start + 1).sum())
    # algorithmic bytes: the 26-plane formula of SURVEY 8(d) with 4 more bytes per read (the tag) and 1008-byte windows
    alg = 2 * batch.n_bases + 4 * batch.n_cigar + 20 * batch.n_reads + cols + n_win * (_ffi.PV_HP_WINDOW_BYTES + 16)
    return {"workload": "16 regions x %d columns, %d reads, %.1f M bases, HP tag per read" % (cols // 16, batch.n_reads, batch.n_bases / 1e6),
            "ms_per_batch": round(ms, 4), "windows": n_win, "mbp_per_s": round(cols / ms / 1e3, 1),
            "algorithmic_gb_per_s": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / 8000.0, 4),
            "note": "same launch chain as the 26-plane builder with the haplotag forms of the tile / allele / window kernels"}


if __name__ == "__main__":
    import json
    import torch
    from pepper_thesis_amd import runtime
    ctx = runtime.Context(0)
    t = time.time()
    print(json.dumps(run(ctx, "cuda:0")))
    sys.stderr.write("[bench_hp] %.1f s\n" % (time.time() - t))
