#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native PEPPER hot path.

Metric (BASELINE.json): pileup windows/s (+ Mbp/s inferred) on the HG003-chr20-shaped ONT R9 workload,
configs[1]: batch = 512 windows per step, fp32, one MI355X per rank. No real BAM / checkpoint exists
offline, so inputs are the synthetic shapes SURVEY.md 8(d) fixes (R = 100 200 columns at 60x with
10 kb reads, planted sites for ~500 windows per region; random-init weights of the pepper_variant
architecture).

One STEP = one 100.2 kb region through the image builder + one 512-window batch through the RNN. The
planted sites give every region a few more than 512 windows (~527) and the builder's output capacity is
512 per step, so EVERY counted window was produced by the image builder in the same chain and inferred
(the builder does the work for the ~3 % it then drops); nothing is topped up. Like the
reference's `callers_per_gpu` (RunInferenceArguments.py:67-74) the host loop keeps CALLERS = 16 steps in
flight by fusing them into ONE launch chain (16 regions per builder call, 8192 windows per RNN call).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--min-seconds S]

Timed region: the K-step pattern is repeated R times back to back (R*K steps as full chains of 16; R is
the smallest count that makes the region last >= S seconds, default 1 s, and R*K a whole number of
chains), bracketed by barrier + synchronize on both sides, MAX over ranks. `steps` = K, `repeats` = R,
`ms_per_step` = time / (R*K), `value` = R*K*512*ranks / time. Every launch in the region has the same
size, so `roofline.launch_ms` x launches, `flop_per_launch` and `achieved` belong together.

N > 1: either launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the env), or — when those are
absent — this program starts the N ranks itself BEFORE touching the GPU (the fan-out the reference does in
RunInference.py:24-91) and relays rank 0's JSON line. One rank per GPU over RCCL (backend nccl); regions
shard across ranks (no data-path collective); one gather of the per-window predictions to rank 0 ends the
timed region. Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 512
CALLERS = int(os.environ.get("PV_BENCH_CALLERS", "16"))  # steps fused per launch chain (callers_per_gpu analogue)
REGION_LEN = 100_200             # 100 kb interval + 2 x 100 safe bases (AlignmentSummarizer.py:181-182)
DEPTH = 60
READ_LEN = 10_000
SITE_EVERY = int(os.environ.get("PV_BENCH_SITE_EVERY", "188"))  # planted sites: ~525-535 windows per region (~500): >= 512 each
FLOP_PER_WINDOW = 161_328_128
FLOP_DEC_PER_WINDOW = 103_809_024
FLOP_ENC_PER_WINDOW = 38_117_376
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_BF16_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


# ------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no torchrun environment
# ------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(args, argv):
    """Parent of a self-launched multi-rank run. Touches no GPU API (a process that has initialised the GPU must
    not be replaced or forked into ranks): it only counts devices, starts `torch.distributed.run` as a child with
    one rank per GPU, lets rank 0's JSON line through on stdout, and returns the child's exit code."""
    n = int(args.gpus)
    backend = os.environ.get("PV_BENCH_BACKEND", "nccl")
    if not args.selftest_launcher and backend == "nccl":
        import torch
        n_dev = torch.cuda.device_count()  # does not initialise the GPU
        if n > n_dev:
            sys.stderr.write("[bench] ERROR: --gpus %d but only %d HIP device(s) are visible; ranks never share a GPU "
                             "(one rank per GPU over RCCL). Nothing was run.\n" % (n, n_dev))
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    log("launching %d ranks: %s" % (n, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def selftest_launcher(args):
    """CPU rehearsal of the N-rank control path (tests/test_bench_launcher.py): join the process group, shard,
    gather to rank 0. No GPU, no measurement."""
    import torch
    import torch.distributed as dist
    from pepper_thesis_amd.dist import gather_predictions, shard_regions
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world == int(args.gpus), (dist.get_world_size(), world, args.gpus)
    mine = shard_regions(11, rank, world)
    local = torch.tensor([[float(i), float(rank), 1.0] for i in mine], dtype=torch.float32).reshape(-1, 3)
    res = gather_predictions(local, dst=0, keys=torch.tensor(mine, dtype=torch.int64))
    dist.barrier()
    if rank == 0:
        rows, keys, counts = res
        ok = sorted(keys.tolist()) == list(range(11)) and all(int(r[0]) == int(k) and int(r[1]) == int(k) % world
                                                                for r, k in zip(rows.tolist(), keys.tolist()))
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_joined": dist.get_world_size(),
                          "counts": counts, "ok": bool(ok)}))
        sys.stdout.flush()
    else:
        assert res is None
    dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------------------
# secondary figures
# ------------------------------------------------------------------------------------------------------
def cpu_baseline(region_batch, procs):
    """CPU baseline on a bounded sample of the same workload (oracle/cpu_baseline.py): one process per core, one
    torch thread each, as the reference's CPU fan-out does (RunInference.py:101-116); each process runs one
    full-size region through the image builder and 512 windows through the eager predict loop."""
    from oracle import cpu_baseline as cb
    kind_builder, res = cb.measure(region_batch, procs)
    one, many = res[1], res[max(res)]
    return {
        "value": many["windows_per_s"], "unit": "windows/s", "cores": many["procs"],
        "kind": "reference" if kind_builder == "reference-c++" else "port",
        "kind_detail": "image builder: %s; RNN: torch-twin (stock torch.nn modules = the reference's eager PyTorch predict "
                       "loop; its default onnxruntime path is not installed)" % kind_builder,
        "sample": "per process: 1 region (R=%d, %dx, %d windows) through the image builder + 512 windows through the eager "
                  "predict loop, 1 torch thread per process; %d processes side by side (slowest process bounds each leg)"
                  % (REGION_LEN, DEPTH, one["windows_per_region"], many["procs"]),
        "value_1core": one["windows_per_s"],
        "builder_regions_per_s": {"1": one["builder_regions_per_s"], str(many["procs"]): many["builder_regions_per_s"]},
        "rnn_windows_per_s": {"1": one["rnn_windows_per_s"], str(many["procs"]): many["rnn_windows_per_s"]},
        "mbp_per_s": many["windows_per_s"] / 512 * REGION_LEN / 1e6,
    }


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_per_launch.csv, produced
    by tools/profile.sh + tools/summarize_prof.py: separate --pmc runs for FETCH_SIZE and WRITE_SIZE; read
    bytes doubled for wide coalesced loads as MI355X_MICROARCH.md prescribes for gfx950). Counters cannot be
    collected inside this process, so the latest committed pass is reported."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_per_launch.csv")))
    if not files:
        return {}
    out = {"note": "HBM bytes/launch from %s (FETCH_SIZE x2 + WRITE_SIZE; the x2 of the guide's wide-load correction over-counts "
                   "where raw FETCH_SIZE already equals the input bytes)" % os.path.basename(files[-1])}
    builder = builder_raw = 0.0
    for row in csv.DictReader(open(files[-1])):
        k = row["kernel"]
        b = float(row["hbm_read_bytes_x2(gfx950 wide loads)"]) + float(row["hbm_write_bytes"])
        braw = float(row["hbm_read_bytes_raw"]) + float(row["hbm_write_bytes"])
        if "k_lstm_layer<512" in k:
            out["k_lstm_layer<512"] = b
        if any(t in k for t in ("k_cigar_scan", "k_tile_fill", "k_pileup", "k_site", "k_collect", "k_write_windows", "k_scan")):
            builder += b
            builder_raw += braw
    out["builder"] = builder
    out["builder_raw"] = builder_raw
    return out


def bf16_secondary(device_id, weights, dbatch, P, dev, pad, groups=6):
    """BASELINE configs[2] flavour: the same step (1 region + 512 windows, 16 fused per launch chain) with
    PV_DTYPE_BF16_INPUT_GEMM: EVERY matrix product of the two LSTM layers and linear_1 on the bf16 MFMA as 3-term hi/lo
    splits (softmax within 1e-4 of the reference: tests/test_rnn_gpu.py), cell updates and linear_2..5 fp32. Not part of `value`."""
    import torch
    from pepper_thesis_amd import _ffi, runtime
    from pepper_thesis_amd.device import DeviceOut
    c2 = runtime.Context(device_id)
    c2.load_p1(weights, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    win = torch.from_numpy(pad).to(dev)
    dout = DeviceOut(CALLERS * BATCH, CALLERS * BATCH * 16, dev, images=win)
    probs = torch.zeros((CALLERS * BATCH, 3), dtype=torch.float32, device=dev)

    def grp():
        c2.summarize_dev(dbatch, P, dout)
        c2.forward_p1_dev(win.data_ptr(), CALLERS * BATCH, probs.data_ptr())

    grp()
    c2.synchronize()
    c2.profile_begin()
    t0 = time.perf_counter()
    for _ in range(groups):
        grp()
    c2.synchronize()
    dt = time.perf_counter() - t0
    prof = c2.profile_end()
    c2.close()
    gname = next((k for k in prof if k.startswith("k_gemm_bf16x3_dec")), None)
    out = {"value": groups * CALLERS * BATCH / dt, "unit": "windows/s", "dtype": "bf16x3 (3-term split bf16 MFMA, fp32 accumulate) for the input "
                                                                                   "projections, the recurrent products and linear_1",
           "steps": groups * CALLERS, "ms_per_step": dt / (groups * CALLERS) * 1e3,
           "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items() if k.startswith("k_") and "summary" not in k}}
    if gname:
        gemm_ms = prof[gname][0] / prof[gname][1]
        flop = 3 * 2.0 * (CALLERS * BATCH * 33) * 2048 * 512
        out["input_gemm"] = {"kernel": "%s (decoder input projection, M=%d N=2048 K=512, 3 bf16 MFMA terms)" % (gname, CALLERS * BATCH * 33),
                             "launch_ms": gemm_ms, "achieved": flop / gemm_ms / 1e9, "peak": PEAK_BF16_TFLOPS,
                             "unit": "TFLOP/s (bf16 MFMA)", "frac": flop / gemm_ms / 1e9 / PEAK_BF16_TFLOPS}
        out["frac"] = out["input_gemm"]["frac"]
    rec = [k for k in prof if k.startswith("k_rec_bf16_lstm_dec")]
    if rec:
        ms = prof[rec[0]][0] / prof[rec[0]][1]
        flop = 3 * 2.0 * (CALLERS * BATCH * 33) * 2048 * 256   # h . W_hh^T, both directions, three bf16 terms
        out["recurrence"] = {"kernel": "k_rec_bf16<LSTM, decoder> (h . W_hh^T on the bf16 MFMA, 64-row tiles)", "launch_ms": ms,
                             "achieved": flop / ms / 1e9, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s (bf16 MFMA)",
                             "frac": flop / ms / 1e9 / PEAK_BF16_TFLOPS}
    return out


def p2_secondary(ctx, dev):
    """secondary figures for the bi-GRU polisher plan: north_star's literal '1000 windows x 100 columns x 10 features'
    shape as ONE single-window model call (pv_rnn_forward_p2_window, host buffers in and out), and the 19-window
    sliding loop over [B,1000,10] chunks at the SURVEY 8(d) batch (64 chunks), at 1000 chunks and at a chip-filling
    batch. Not part of `value`."""
    import torch
    from pepper_thesis_amd import _ffi, synth
    ctx.load_p2(synth.make_weights_p2(4321))
    lib = _ffi.load()
    out = {"flop_per_100col_window": 80_435_200, "windows_per_chunk": 19}
    # north_star literal: B = 1000 windows of [100, 10] through one TransducerGRU.forward call (host in/out)
    xw = synth.synth_p2_images(11, 1000, seq_len=100)
    ctx.forward_p2_window(xw)
    t0 = time.perf_counter()
    reps = 5
    ctx.profile_begin()
    for _ in range(reps):
        ctx.forward_p2_window(xw)
    prof = ctx.profile_end()
    dt = (time.perf_counter() - t0) / reps
    kms = sum(v[0] for k, v in prof.items()) / reps
    out["single_window_B1000"] = {"call": "pv_rnn_forward_p2_window (host buffers, PCIe-inclusive)", "ms_per_call": dt * 1e3,
                                  "windows_100col_per_s": 1000 / dt, "kernel_ms": kms,
                                  "kernel_tflops": 80_435_200 * 1000 / kms / 1e9 if kms > 0 else None,
                                  "frac_of_f32_peak": 80_435_200 * 1000 / kms / 1e9 / PEAK_F32_TFLOPS if kms > 0 else None}
    for B in (64, 1000, 2048, 4096, 8192):
        x = torch.from_numpy(synth.synth_p2_images(7, B)).to(dev)
        labels = torch.zeros((B, 1000), dtype=torch.uint8, device=dev)
        _ffi.check(lib.pv_rnn_forward_p2_dev(ctx.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
        ctx.synchronize()
        ctx.profile_begin()
        for _ in range(2):
            _ffi.check(lib.pv_rnn_forward_p2_dev(ctx.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
        prof = ctx.profile_end()
        ms = sum(v[0] for v in prof.values()) / 2
        out["B%d" % B] = {"ms": ms, "chunks_per_s": B / ms * 1e3, "windows_100col_per_s": 19 * B / ms * 1e3,
                          "tflops": 80_435_200 * 19 * B / ms / 1e9, "frac_of_f32_peak": 80_435_200 * 19 * B / ms / 1e9 / PEAK_F32_TFLOPS}
        del x, labels
    # the same model with every matrix product on the bf16 MFMA (PV_DTYPE_BF16_INPUT_GEMM: layer-wise path, per-window GEMM)
    try:
        from pepper_thesis_amd import runtime
        cb = runtime.Context(ctx.device_id)
        cb.load_p2(synth.make_weights_p2(4321), _ffi.PV_DTYPE_BF16_INPUT_GEMM)
        bf = {"dtype": "bf16x3 (3-term split bf16 MFMA, fp32 accumulate) for input projections and recurrent products"}
        cb.forward_p2_window(xw)
        t0 = time.perf_counter()
        for _ in range(reps):
            cb.forward_p2_window(xw)
        dtw = (time.perf_counter() - t0) / reps
        bf["single_window_B1000"] = {"ms_per_call": dtw * 1e3, "windows_100col_per_s": 1000 / dtw}
        for B in (64, 1000, 2048, 4096, 8192):
            x = torch.from_numpy(synth.synth_p2_images(7, B)).to(dev)
            labels = torch.zeros((B, 1000), dtype=torch.uint8, device=dev)
            _ffi.check(lib.pv_rnn_forward_p2_dev(cb.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
            cb.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                _ffi.check(lib.pv_rnn_forward_p2_dev(cb.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
            cb.synchronize()
            ms = (time.perf_counter() - t0) / 2 * 1e3
            cb.profile_begin()
            _ffi.check(lib.pv_rnn_forward_p2_dev(cb.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
            prof = cb.profile_end()
            bf["B%d" % B] = {"ms": ms, "windows_100col_per_s": 19 * B / ms * 1e3, "speedup_over_f32_form": out["B%d" % B]["ms"] / ms,
                             "kernel_ms_sum": {k: v[0] for k, v in prof.items()}}
            gk = [k for k in prof if k.startswith("k_gemm_bf16x3")]
            if gk:
                gms = prof[gk[0]][0] / prof[gk[0]][1]
                bp = (B + 31) // 32 * 32 if B * 2 <= 64 * 2 * 256 else (B + 63) // 64 * 64
                flop = 3 * 2.0 * (100 * bp) * 768 * 256
                bf["B%d" % B]["input_gemm"] = {"launch_ms": gms, "achieved": flop / gms / 1e9, "peak": PEAK_BF16_TFLOPS,
                                               "unit": "TFLOP/s (bf16 MFMA)", "frac": flop / gms / 1e9 / PEAK_BF16_TFLOPS}
            del x, labels
        cb.close()
        out["bf16_mode"] = bf
    except Exception as e:  # noqa: BLE001
        out["bf16_mode"] = {"error": repr(e)}
    # the polisher chain: summary images of 8 regions -> their chunks through the bi-GRU without leaving HBM (tools/bench_polish.py)
    try:
        from tools import bench_polish
        out["polish_chain"] = bench_polish.run(ctx, dev)
    except Exception as e:  # noqa: BLE001
        out["polish_chain"] = {"error": repr(e)}
    return out


def single_call_secondary(ctx, dev, weights):
    """secondary: ONE caller's batches through pv_rnn_forward_p1_dev alone (no caller fusion, no image builder), i.e. what
    a single `run_inference -bs 512` loop sees, and the same call at 2048 and 4096 windows. Not part of `value`."""
    import torch
    from pepper_thesis_amd import synth
    out = {}
    for B in (512, 1024, 2048, 4096):
        x = torch.from_numpy(synth.synth_windows(3, B)).to(dev)
        probs = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        for _ in range(2):
            ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
        ctx.synchronize()
        wall_ms = (time.perf_counter() - t0) / 20 * 1e3
        ctx.profile_begin()
        for _ in range(5):
            ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
        prof = ctx.profile_end()
        ms = sum(v[0] / v[1] for v in prof.values())
        out["B%d" % B] = {"kernel_ms": ms, "wall_ms_per_call_back_to_back": wall_ms, "windows_per_s": B / wall_ms * 1e3,
                          "tflops": FLOP_PER_WINDOW * B / wall_ms / 1e9,
                          "frac_of_f32_peak": FLOP_PER_WINDOW * B / wall_ms / 1e9 / PEAK_F32_TFLOPS}
        del x, probs
    # the reference's own way to fill a GPU from 512-window batches: callers_per_gpu independent predict loops on one device
    # (RunInferenceArguments.py:67-74, default 4). Here: N contexts = N HIP streams, each looping over its own 512-window
    # batches with no fusion across callers; the (tile, direction) workgroups of the callers share the CUs.
    from pepper_thesis_amd import runtime
    # (a 512-window call alone runs in the unit-split LSTM form, which takes the whole chip for one call; callers that share
    # the GPU without fusion do better in the one-workgroup form, option shared_device = 1: both are reported)
    for ncall, form in ((4, "default"), (8, "default"), (4, "one_workgroup_form"), (8, "one_workgroup_form")):
        key = "callers%d_x_B512" % ncall + ("" if form == "default" else "_" + form)
        try:
            ctxs = [ctx] + [runtime.Context(ctx.device_id) for _ in range(ncall - 1)]
            for c in ctxs[1:]:
                c.load_p1(weights)
            for c in ctxs:   # what run_inference sets for un-fused callers that share a GPU
                c.set_option("shared_device", 0 if form == "default" else 1)
            xs = [torch.from_numpy(synth.synth_windows(30 + i, 512)).to(dev) for i in range(ncall)]
            ps = [torch.zeros((512, 3), dtype=torch.float32, device=dev) for _ in range(ncall)]
            for c, x, p in zip(ctxs, xs, ps):
                c.forward_p1_dev(x.data_ptr(), 512, p.data_ptr())
            for c in ctxs:
                c.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                for c, x, p in zip(ctxs, xs, ps):
                    c.forward_p1_dev(x.data_ptr(), 512, p.data_ptr())
            for c in ctxs:
                c.synchronize()
            dt = time.perf_counter() - t0
            wps = reps * ncall * 512 / dt
            out[key] = {"windows_per_s": wps, "ms_per_round": dt / reps * 1e3, "tflops": FLOP_PER_WINDOW * wps / 1e12,
                                               "frac_of_f32_peak": FLOP_PER_WINDOW * wps / 1e12 / PEAK_F32_TFLOPS,
                                               "note": "%d independent callers (contexts / streams), 512 windows per call each, no fusion" % ncall}
            for c in ctxs[1:]:
                c.close()
        except Exception as e:
            out[key] = {"error": repr(e)}
    ctx.set_option("shared_device", 0)
    return out


def graph_secondary(ctx, dev, weights, dbatch, P, pad):
    """secondary: the same launch sequences replayed as hipGraphs (pv_graph_begin / _end / _launch; BASELINE configs[4] names the
    technique): one caller's 512-window P1 call (4 kernels), and the fused 16-caller chain (image builder + P1 over 8192
    windows, 18 launches). Eager and graph wall time per call, back to back on the context's stream."""
    import torch
    from pepper_thesis_amd import synth
    from pepper_thesis_amd.device import DeviceOut
    out = {}
    st = ctx.stream

    def timed(fn, reps):
        fn(); ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    ctx.load_p1(weights)
    x = torch.from_numpy(synth.synth_windows(3, 512)).to(dev)
    probs = torch.zeros((512, 3), dtype=torch.float32, device=dev)
    one = lambda: ctx.forward_p1_dev(x.data_ptr(), 512, probs.data_ptr(), stream=st)
    eager = timed(one, 50)
    with ctx.graph_capture(st) as g:
        one()
    graph = timed(g.launch, 50)
    g.close()
    out["p1_single_call_B512"] = {"eager_ms": eager, "graph_ms": graph, "frac_of_f32_peak_graph": FLOP_PER_WINDOW * 512 / graph / 1e9 / PEAK_F32_TFLOPS}
    wins = torch.from_numpy(pad).to(dev)
    dout = DeviceOut(CALLERS * BATCH, CALLERS * BATCH * 16, dev, images=wins)
    p2 = torch.zeros((CALLERS * BATCH, 3), dtype=torch.float32, device=dev)

    def chain():
        ctx.summarize_dev(dbatch, P, dout, stream=st)
        ctx.forward_p1_dev(wins.data_ptr(), CALLERS * BATCH, p2.data_ptr(), stream=st)

    eager = timed(chain, 20)
    with ctx.graph_capture(st) as g:
        chain()
    graph = timed(g.launch, 20)
    g.close()
    out["fused_chain_16_callers"] = {"eager_ms": eager, "graph_ms": graph, "windows_per_s_graph": CALLERS * BATCH / graph * 1e3,
                                     "note": "one batch of 16 regions replayed (the timed headline rotates four batches, eager)"}
    return out


# ------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--min-seconds", type=float, default=float(os.environ.get("PV_BENCH_MIN_SECONDS", "1.0")),
                    help="repeat the K-step pattern until the timed region lasts at least this long")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-p2", action="store_true", help="skip the secondary bi-GRU (P2) and single-call figures")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary configs[2] (bf16 input GEMM) figure")
    ap.add_argument("--no-h2d", action="store_true", help="skip the secondary PCIe-inclusive (overlapped H2D) figure")
    ap.add_argument("--no-filepath", action="store_true", help="skip the secondary BAM -> images -> predictions file-path figure")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU rehearsal of the N-rank launcher / gather path over gloo (no GPU, no measurement)")
    args = ap.parse_args(argv)

    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.gpus > 1 and not in_torchrun:
        return launch(args, argv)          # parent: starts the ranks, never touches the GPU
    if args.selftest_launcher:
        if not in_torchrun:
            sys.stderr.write("[bench] --selftest-launcher needs --gpus N > 1\n")
            return 2
        return selftest_launcher(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("PV_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only for single-GPU rehearsals
    n_dev = torch.cuda.device_count()
    if world != int(args.gpus):
        sys.stderr.write("[bench] ERROR: --gpus %d but WORLD_SIZE=%d: start exactly one rank per GPU\n" % (args.gpus, world))
        return 2
    if n_dev < 1:
        sys.stderr.write("[bench] ERROR: no HIP device visible (there is no CPU path)\n")
        return 2
    if world > 1 and backend == "nccl" and (world > n_dev or local_rank >= n_dev):
        sys.stderr.write("[bench] ERROR: %d ranks but %d HIP device(s): ranks never share a GPU over RCCL\n" % (world, n_dev))
        return 2
    dev_id = local_rank % n_dev   # == local_rank except in gloo rehearsals on one GPU (PV_BENCH_BACKEND=gloo)
    torch.cuda.set_device(dev_id)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
        # one rank per GPU, checked: every rank reports the device it will create its pv_ctx on
        ids = [None] * world
        dist.all_gather_object(ids, (os.uname().nodename, dev_id))
        if backend == "nccl" and len(set(ids)) != world:
            sys.stderr.write("[bench] ERROR: ranks share a device: %r\n" % (ids,))
            return 2
    else:
        dist = None
    dev = "cuda:%d" % dev_id

    from pepper_thesis_amd import runtime, synth
    from pepper_thesis_amd.batch import PRESETS, pack_regions
    from pepper_thesis_amd.device import DeviceBatch, DeviceOut, PinnedBatch
    from pepper_thesis_amd.dist import CabiGather, gather_predictions

    K, W = max(1, int(args.steps)), max(0, int(args.warmup))

    def chains_for(n):
        """exactly n steps as launch chains: as many chains of CALLERS steps as fit, then (if the rest allows) one chain of
        CALLERS/2 steps (still a whole round of workgroups), then the remainder"""
        out = [CALLERS] * (n // CALLERS)
        r = n % CALLERS
        half = CALLERS // 2
        if half and r >= half:
            out.append(half)
            r -= half
        if r:
            out.append(r)
        return out

    wchains = chains_for(W)

    # ---- synthetic workload (seeded; rank r gets its own regions: regions shard across GPUs) ---------
    t0 = time.time()
    # CALLERS distinct regions per rank; NBATCH batches = rotations of them at distinct HBM addresses, used round-robin by
    # the launch chains so that the image builder never finds its inputs (116 MB per batch) in the 256 MB Infinity Cache
    # from the previous chain (caches are address-based: a rotated copy is as cold as different data)
    NBATCH = max(1, int(os.environ.get("PV_BENCH_NBATCH", "4")))
    regions = [synth.synth_region(1234 + 97 * (rank * CALLERS + i), region_len=REGION_LEN, depth=DEPTH,
                                  read_len=READ_LEN, site_every=SITE_EVERY,
                                  ref_start=1_000_000 + (rank * CALLERS + i) * (REGION_LEN - 200))
               for i in range(CALLERS)]
    rot = max(1, CALLERS // NBATCH)
    batches = [pack_regions(regions[j * rot:] + regions[:j * rot]) for j in range(NBATCH)]
    batch = batches[0]
    weights = synth.make_weights_p1(1234)
    pad = synth.synth_windows(4242 + rank, CALLERS * BATCH)
    log("rank %d: generated %d regions (%d reads, %.1f M bases) in %.1f s" %
        (rank, CALLERS, batch.n_reads, batch.n_bases / 1e6, time.time() - t0))
    P = PRESETS["ont_r9_guppy5_sup"]

    ctx = runtime.Context(dev_id)
    ctx.load_p1(weights)
    dbatches = [DeviceBatch(b, dev) for b in batches]
    dbatch = dbatches[0]
    # Two window buffers, so that with PV_BENCH_OVERLAP=1 the image builder of group g+1 (stream s_build) may run beside the
    # RNN of group g (stream s_rnn); events order builder(g) -> rnn(g) and rnn(g) -> builder(g+2) (buffer reuse).
    wins = [torch.from_numpy(pad).to(dev) for _ in range(2)]          # [8192,33,26] int8, builder writes the front
    douts = [DeviceOut(CALLERS * BATCH, CALLERS * BATCH * 16, dev, images=w) for w in wins]
    s_build, s_rnn = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    s_copy = torch.cuda.Stream(device=dev)
    ev_built = [torch.cuda.Event() for _ in range(2)]
    ev_used = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    state = {"n": 0}

    # Default: ONE stream, builder and RNN of a group back to back. The LSTM workgroups take whole CUs (2 x 255 VGPRs per
    # SIMD, 132 KB LDS), so a second stream (PV_BENCH_OVERLAP=1) only lets builder kernels slip into the RNN's kernel
    # boundaries: +1 % throughput, but every kernel's duration is then stretched by its neighbours and the live per-kernel
    # times no longer agree with the rocprofv3 averages.
    overlap = os.environ.get("PV_BENCH_OVERLAP", "0") != "0"
    if not overlap:
        s_build = s_rnn

    # chains shorter than CALLERS steps (only the warm-up can have them): the first regions of batch 0
    rem_batches = {n: DeviceBatch(batches[0].select(list(range(n))), dev) for n in set(wchains) if n != CALLERS}
    rem_douts = {n: [DeviceOut(n * BATCH, n * BATCH * 16, dev, images=w) for w in wins] for n in rem_batches}
    probs_holder = {"t": torch.zeros((1, CALLERS * BATCH, 3), dtype=torch.float32, device=dev)}

    def group(g, ncall=CALLERS, h2d=None):
        k = state["n"] & 1
        probs = probs_holder["t"]
        if state["n"] >= 2:
            s_build.wait_event(ev_used[k])
        if h2d is not None:
            # PCIe-inclusive form: this chain's regions arrive from page-locked host memory on the copy stream (the previous
            # chain's kernels are still running meanwhile); the builder waits for them
            dsets, pinned, ev_copied, ev_consumed = h2d
            s_copy.wait_event(ev_consumed[k])
            dsets[k].upload_async(pinned[g % len(pinned)], s_copy)
            ev_copied[k].record(s_copy)
            s_build.wait_event(ev_copied[k])
            ctx.summarize_dev(dsets[k], P, douts[k], stream=s_build.cuda_stream)
            ev_consumed[k].record(s_build)
        elif ncall == CALLERS:
            ctx.summarize_dev(dbatches[g % NBATCH], P, douts[k], stream=s_build.cuda_stream)
        else:
            ctx.summarize_dev(rem_batches[ncall], P, rem_douts[ncall][k], stream=s_build.cuda_stream)
        ev_built[k].record(s_build)
        s_rnn.wait_event(ev_built[k])
        ctx.forward_p1_dev(wins[k].data_ptr(), ncall * BATCH, probs[g % probs.shape[0]].data_ptr(), stream=s_rnn.cuda_stream)
        ev_used[k].record(s_rnn)
        state["n"] += 1

    def drain():
        s_copy.synchronize()
        s_build.synchronize()
        s_rnn.synchronize()

    # ---- warm-up: workspace sizing, the isolated image-builder measurement, W steps, calibration ------------------------
    n_out_batch = []
    for db in dbatches:  # first calls size the workspace arena (every batch once)
        ctx.summarize_dev(db, P, douts[0], stream=s_build.cuda_stream)
        s_build.synchronize()
        assert douts[0].status() == 0, "device status %d" % douts[0].status()
        n_out_batch.append(douts[0].n_out())   # windows the regions yield; those beyond the capacity (512 per step) are dropped
        assert n_out_batch[-1] >= CALLERS * BATCH, "regions yield %d windows < %d: lower PV_BENCH_SITE_EVERY" % (n_out_batch[-1], CALLERS * BATCH)
    ctx.forward_p1_dev(wins[0].data_ptr(), CALLERS * BATCH, probs_holder["t"][0].data_ptr(), stream=s_rnn.cuda_stream)  # RNN workspace at full size
    s_rnn.synchronize()
    # image-builder roofline: measured in isolation (16 regions per launch chain), cold inputs (round-robin batches)
    ctx.profile_begin()
    for i in range(12):
        ctx.summarize_dev(dbatches[i % NBATCH], P, douts[0], stream=s_build.cuda_stream)
    prof_builder = ctx.profile_end()
    # the same 12 launch chains without the per-kernel events (two events per kernel put a few microseconds between kernels)
    eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eb0.record(s_build)
    for i in range(12):
        ctx.summarize_dev(dbatches[i % NBATCH], P, douts[0], stream=s_build.cuda_stream)
    eb1.record(s_build)
    eb1.synchronize()
    builder_chain_ms = eb0.elapsed_time(eb1) / 12
    n_windows_region = n_out_batch[0]                    # yielded by the 16 regions of batch 0 (written: at most 512 per step)
    for g, nc in enumerate(wchains):
        group(g, nc)
    drain()
    # calibration (untimed): two full chains -> how many repeats of the K-step pattern make >= min_seconds
    t0 = time.perf_counter()
    for g in range(2):
        group(g)
    drain()
    est_step = (time.perf_counter() - t0) / (2 * CALLERS)
    # per-kernel breakdown of the chain (untimed: every kernel bracketed by events; the timed region brackets the decoder only)
    ctx.profile_begin()
    for g in range(2):
        group(g)
    drain()
    prof_all = ctx.profile_end()
    if dist is not None:
        t_est = torch.tensor([est_step], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t_est, op=dist.ReduceOp.MAX)   # every rank must choose the same repeat count
        est_step = float(t_est.item())
    import math
    unit = CALLERS // math.gcd(K, CALLERS)             # repeats come in multiples of this: R*K is a whole number of chains
    R = max(1, math.ceil(args.min_seconds / max(est_step * K, 1e-9)))
    R = ((R + unit - 1) // unit) * unit
    total_steps = R * K
    n_chains = total_steps // CALLERS
    assert n_chains * CALLERS == total_steps
    probs_holder["t"] = torch.zeros((n_chains, CALLERS * BATCH, 3), dtype=torch.float32, device=dev)
    chain_ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_chains + 1)]
    for g in range(2):   # touch the new probs buffer, settle the clocks
        group(g)
    drain()

    # ---- timed region ----------------------------------------------------------------------------------
    cabi = None
    if dist is not None:
        if os.environ.get("PV_BENCH_GATHER", "torch") == "cabi" and backend == "nccl":
            cabi = CabiGather(ctx, rank, world)   # the exchange through the C-ABI (pv_gather) instead of torch.distributed

        def exchange(t):
            if cabi is None:
                return gather_predictions(t, dst=0)
            drain()
            rows, counts = cabi.gather(t.contiguous(), dst=0, capacity_rows=world * t.shape[0])
            return None if rows is None else (rows, None, counts)

        # untimed warm-up of the one exchange step as well: the first collective of a communicator sets up its channels
        # and loads its kernels (RCCL does that lazily), which must not land inside the timed region
        exchange(probs_holder["t"].view(-1, 3))
        dist.barrier()
    torch.cuda.synchronize()
    ctx.profile_begin(only="k_lstm_layer_dec")   # the roofline kernel's launch durations, live over the timed region
    t0 = time.perf_counter()
    chain_ev[0].record(s_rnn)
    for g in range(n_chains):
        group(g)
        chain_ev[g + 1].record(s_rnn)
    gathered = None
    if dist is not None:
        drain()
        gathered = exchange(probs_holder["t"].view(-1, 3))
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_end()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if rank == 0:
            assert gathered is not None and gathered[0].shape[0] == world * total_steps * BATCH, "gather lost rows"
    chain_ms = sorted(chain_ev[g].elapsed_time(chain_ev[g + 1]) for g in range(n_chains))
    median_chain_ms = chain_ms[len(chain_ms) // 2]

    rc = 0
    if rank == 0:
        total_windows = total_steps * BATCH * world
        value = total_windows / dt
        builder_windows = sum(min(n_out_batch[g % NBATCH], CALLERS * BATCH) for g in range(n_chains)) * world   # == total_windows
        dec_ms, dec_n = prof.get("k_lstm_layer_dec", (0.0, 0))
        dec_launch_ms = dec_ms / max(dec_n, 1)
        flop_per_launch = FLOP_DEC_PER_WINDOW * CALLERS * BATCH
        achieved_tf = flop_per_launch / (dec_launch_ms / 1e3) / 1e12 if dec_ms > 0 else 0.0
        sum_ms, sum_n = prof_builder.get("summary_pipeline", (0.0, 0))
        pile_ms, pile_n = prof_builder.get("k_pileup", (0.0, 0))
        traffic = pmc_traffic()
        alg_bytes = batch.algorithmic_bytes(min(n_windows_region, CALLERS * BATCH))
        out = {
            "metric": "pileup windows/sec (whole node) + Mbp/sec inferred, HG003 chr20 ONT R9",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / total_steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "repeats": R, "timed_steps": total_steps, "timed_seconds": dt,
            "median_chain_ms": median_chain_ms, "windows_per_s_from_median_chain": CALLERS * BATCH * world / (median_chain_ms / 1e3),
            "windows_from_builder_per_s": builder_windows / dt,
            "windows_note": "every counted window was written by the image builder of the same launch chain (capacity 512 per step; the "
                            "regions yield %.1f per step, the surplus is dropped after the builder's work for it) and inferred" % (n_windows_region / CALLERS),
            "config": {"workload": "configs[1]: HG003-chr20-shaped ONT R9 synthetic, batch=512 windows/step, fp32 bi-LSTM P1, "
                                   "1 region (R=100200, 60x, 10 kb reads) per step, %d steps fused per launch chain" % CALLERS,
                       "batch": BATCH, "callers": CALLERS, "region_len": REGION_LEN, "depth": DEPTH,
                       "windows_per_region": n_windows_region / CALLERS, "parallelism": "region-sharded x%d" % world,
                       "backend": backend if world > 1 else None,
                       "gather": ("pv_gather (C-ABI, RCCL)" if cabi is not None else "torch.distributed gather to rank 0") if world > 1 else None},
            "mbp_per_s": total_steps * world * REGION_LEN / 1e6 / dt,
            "roofline": {"bound": "mfma", "kernel": "k_lstm_layer<512> (decoder bi-LSTM, fused input projection + recurrence)",
                         "achieved": achieved_tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_F32_TFLOPS, "traffic": traffic.get("k_lstm_layer<512"),
                         "traffic_note": traffic.get("note"),
                         "launch_ms": dec_launch_ms, "launches": dec_n, "windows_per_launch": CALLERS * BATCH,
                         "flop_per_launch": flop_per_launch},
            "roofline_builder": {"bound": "hbm", "kernel": "summary pipeline (k_cigar_scan .. k_write_windows), %d regions/launch" % CALLERS,
                                 "achieved": alg_bytes / (builder_chain_ms / 1e3) / 1e9,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": alg_bytes / (builder_chain_ms / 1e3) / 1e9 / PEAK_HBM_GBS,
                                 "traffic": traffic.get("builder"), "traffic_raw": traffic.get("builder_raw"),
                                 "traffic_raw_note": "FETCH_SIZE as counted + WRITE_SIZE (the builder's loads are dwords and bytes, for which the guide's "
                                                     "x2 of wide coalesced loads does not apply)",
                                 "launch_ms": builder_chain_ms,
                                 "launch_ms_with_kernel_events": sum_ms / max(sum_n, 1),
                                 "k_pileup_ms": pile_ms / max(pile_n, 1),
                                 "measured": "in isolation, before the timed region: 12 launch chains back to back (launch_ms), and the "
                                             "same with every kernel bracketed by events (launch_ms_with_kernel_events, k_pileup_ms)",
                                 "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_ms": dict({k: v[0] / max(v[1], 1) for k, v in prof_all.items()},
                              **{k: v[0] / max(v[1], 1) for k, v in prof.items()}),
            "kernel_ms_note": "k_lstm_layer_dec: HIP events over the timed region; the others: two untimed chains with every kernel bracketed",
            "rnn_model_tflops": FLOP_PER_WINDOW * value / 1e12,
        }
        secondary = world == 1
        if secondary and not args.no_h2d:
            try:
                # PCIe-inclusive secondary: every chain's 16 regions are copied from page-locked host memory on a second
                # stream inside the timed loop (double-buffered device batches); never `value`
                pinned = [PinnedBatch(b) for b in batches]
                dsets = [DeviceBatch(batches[0], dev) for _ in range(2)]
                ev_copied = [torch.cuda.Event() for _ in range(2)]
                ev_consumed = [torch.cuda.Event() for _ in range(2)]
                h2d = (dsets, pinned, ev_copied, ev_consumed)
                for g in range(2):
                    group(g, h2d=h2d)
                drain()
                nch = max(4, n_chains // 4)
                t0 = time.perf_counter()
                for g in range(nch):
                    group(g, h2d=h2d)
                drain()
                dth = time.perf_counter() - t0
                out["h2d_overlapped"] = {"value": nch * CALLERS * BATCH / dth, "unit": "windows/s", "ms_per_step": dth / (nch * CALLERS) * 1e3,
                                         "bytes_per_step": pinned[0].nbytes / CALLERS, "pcie_GBps": nch * pinned[0].nbytes / dth / 1e9,
                                         "note": "per-chain H2D of the 16 regions from pinned memory on a copy stream, overlapped with the "
                                                 "previous chain's kernels", "ratio_to_value": nch * CALLERS * BATCH / dth / value}
                del pinned, dsets
            except Exception as e:
                out["h2d_overlapped"] = {"error": repr(e)}
        if secondary and not args.no_bf16:
            try:
                out["config2_bf16_input_gemm"] = bf16_secondary(dev_id, weights, dbatch, P, dev, pad)
            except Exception as e:
                out["config2_bf16_input_gemm"] = {"error": repr(e)}
        if secondary and not args.no_p2:
            try:
                out["p1_single_call"] = single_call_secondary(ctx, dev, weights)
            except Exception as e:
                out["p1_single_call"] = {"error": repr(e)}
            try:
                out["p2_bigru"] = p2_secondary(ctx, dev)
            except Exception as e:
                out["p2_bigru"] = {"error": repr(e)}
        if secondary and not args.no_p2:
            try:
                out["hipgraph"] = graph_secondary(ctx, dev, weights, dbatch, P, pad)
            except Exception as e:  # noqa: BLE001
                out["hipgraph"] = {"error": repr(e)}
        if secondary and not args.no_p2:
            try:
                from tools import bench_hp
                out["hp_builder"] = bench_hp.run(ctx, dev, regions=[batch.region(g) for g in range(batch.n_regions)])
            except Exception as e:  # noqa: BLE001
                out["hp_builder"] = {"error": repr(e)}
        if secondary and not args.no_filepath:
            try:
                from tools import bench_filepath
                out["file_path"] = bench_filepath.run(ctx, weights, dev)
            except Exception as e:
                out["file_path"] = {"error": repr(e)}
        if secondary and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(batches[0].select([0]), min(16, len(os.sched_getaffinity(0))))  # a 1-GPU box's CPU share is 16 cores
            except Exception as e:  # the baseline is reporting only; never fail the bench for it
                out["cpu_baseline"] = {"value": None, "unit": "windows/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()
    return rc


if __name__ == "__main__":
    sys.exit(main() or 0)
