"""TEST / MEASUREMENT INFRASTRUCTURE ONLY — the CPU baseline legs of bench.py's `cpu_baseline`.

Mirrors the reference's own CPU fan-out: one OS process per caller, `threads_per_caller = max(1,
threads / callers)` torch threads each (pepper_variant/modules/python/RunInference.py:101-116 with
callers = threads, i.e. ONE thread per process), every process running the image builder and the
eager predict loop on its own share of the work (ImageGenerationUI.py:211 assigns interval i to
worker i % threads). A worker times
  * builder leg: one full-size region through the REFERENCE's region_summary.cpp when oracle/_ref was
    built in the container that made this tree ("reference-c++"), else through the C restatement
    ("port-c"),
  * RNN leg: `n_windows` windows in batches of 512 through the stock torch.nn twin of the reference's
    eager predict loop (oracle/rnn_torch_twin.py; "torch-twin" — the reference's default onnxruntime
    path is not installed).
Never imported by the product path.
"""
import multiprocessing as mp
import os
import pickle
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(args):
    path, seed, n_windows, batch_size = args
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    torch.set_num_threads(1)
    from oracle import oracle, rnn_torch_twin
    from pepper_thesis_amd import synth
    from pepper_thesis_amd.batch import PRESETS
    with open(path, "rb") as f:
        batch = pickle.load(f)
    P = PRESETS["ont_r9_guppy5_sup"]
    fn = oracle.reference_summarize if oracle.have_reference() else oracle.summarize
    t0 = time.perf_counter()
    out = fn(batch, P)
    t_builder = time.perf_counter() - t0
    model = rnn_torch_twin.build_p1(synth.make_weights_p1(seed))
    x = synth.synth_windows(5, n_windows)
    rnn_torch_twin.predict_p1(model, x[:32])  # warm
    t0 = time.perf_counter()
    rnn_torch_twin.predict_p1(model, x, batch_size)
    t_rnn = time.perf_counter() - t0
    return len(out), t_builder, t_rnn


def measure(region_batch, n_procs, n_windows=512, batch_size=512, seed=1234):
    """-> dict with the 1-process and the n_procs-process figures (regions/s, windows/s per leg and combined)"""
    from oracle import oracle
    if not oracle.have_reference():
        oracle.build()
    kind_builder = "reference-c++" if oracle.have_reference() else "port-c"
    fd, path = tempfile.mkstemp(suffix=".pkl")
    with os.fdopen(fd, "wb") as f:
        pickle.dump(region_batch, f)
    ctx = mp.get_context("spawn")
    res = {}
    try:
        for n in sorted({1, max(1, int(n_procs))}):
            t0 = time.perf_counter()
            with ctx.Pool(n) as pool:
                rows = pool.map(_worker, [(path, seed, n_windows, batch_size)] * n)
            wall = time.perf_counter() - t0
            tb = max(r[1] for r in rows)   # all workers start together: the slowest one bounds the throughput
            tr = max(r[2] for r in rows)
            res[n] = {"procs": n, "windows_per_region": rows[0][0], "builder_s_per_region": tb, "rnn_s_per_%d" % n_windows: tr,
                      "builder_regions_per_s": n / tb, "rnn_windows_per_s": n * n_windows / tr,
                      # a step = 1 region + 512 windows, run back to back by every process
                      "windows_per_s": n * 512 / (tb + tr * 512.0 / n_windows), "wall_s_incl_spawn": wall}
    finally:
        os.unlink(path)
    return kind_builder, res
