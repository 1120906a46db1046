#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native PEPPER hot path.

Metric (BASELINE.json): pileup windows/s (+ Mbp/s inferred) on the HG003-chr20-shaped ONT R9 workload,
configs[1]: batch = 512 windows per step, fp32, one MI355X per rank. No real BAM / checkpoint exists
offline, so inputs are the synthetic shapes SURVEY.md 8(d) fixes (R = 100 200 columns at 60x with
10 kb reads; random-init weights of the pepper_variant architecture).

One STEP = one 100.2 kb region through the image builder + one 512-window batch through the RNN
(every window the region yields is inferred; the batch is topped up to exactly 512 with resident
synthetic windows). Like the reference's `callers_per_gpu` (RunInferenceArguments.py:67-74) the host
loop keeps CALLERS = 16 steps in flight (reference default 4, "up to 10 on an 11 GB GPU"), here by fusing them
into ONE launch chain per group (16 regions per builder call, 8192 windows per RNN call): a single stream then fills all
256 CUs for two back-to-back rounds of workgroups per kernel. --steps K times exactly K steps (K // 16 full chains, a chain of 8 and
one shorter chain for the remainder).

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: launched by torch.distributed.run, one rank per GPU; regions shard across ranks (no data-path
collective); one RCCL gather of the per-window predictions to rank 0 ends the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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
SITE_EVERY = 260                 # planted sites: ~430-480 windows per region (< 512 by construction)
FLOP_PER_WINDOW = 161_328_128
FLOP_DEC_PER_WINDOW = 103_809_024
FLOP_ENC_PER_WINDOW = 38_117_376
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32 MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def cpu_baseline(weights, region, threads):
    """CPU baseline on a bounded sample of the same workload: one region through the image builder
    (the REFERENCE's region_summary.cpp if oracle/_ref was built, else the C oracle) + 512-window
    batches through the stock-torch twin of the reference's eager predict loop."""
    import torch
    from oracle import oracle, rnn_torch_twin
    from pepper_thesis_amd import synth
    from pepper_thesis_amd.batch import PRESETS, pack_regions
    P = PRESETS["ont_r9_guppy5_sup"]
    b = pack_regions([region])
    kind_builder = "reference" if oracle.have_reference() else "port"
    fn = oracle.reference_summarize if oracle.have_reference() else oracle.summarize
    if not oracle.have_reference():
        oracle.build()
    t0 = time.perf_counter()
    out = fn(b, P)
    t_builder = time.perf_counter() - t0
    torch.set_num_threads(threads)
    model = rnn_torch_twin.build_p1(weights)
    x = synth.synth_windows(5, 2 * BATCH)
    rnn_torch_twin.predict_p1(model, x[:64])  # warm
    t0 = time.perf_counter()
    rnn_torch_twin.predict_p1(model, x, BATCH)
    t_rnn = (time.perf_counter() - t0) / 2.0
    per_step = t_builder + t_rnn
    return {
        "value": BATCH / per_step, "unit": "windows/s", "cores": threads,
        "kind": "port",
        "sample": "1 region (%d windows) through the %s image builder on 1 core: %.3f s; 2 x 512 windows through the "
                  "stock torch.nn twin of the reference's eager predict loop on %d threads: %.3f s per 512"
                  % (len(out), "reference C++ (oracle/_ref)" if kind_builder == "reference" else "C oracle",
                     t_builder, threads, t_rnn),
        "mbp_per_s": REGION_LEN / 1e6 / per_step,
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
    out = {"note": "HBM bytes/launch from %s (FETCH_SIZE x2 + WRITE_SIZE)" % os.path.basename(files[-1])}
    builder = 0.0
    for row in csv.DictReader(open(files[-1])):
        k = row["kernel"]
        b = float(row["hbm_read_bytes_x2(gfx950 wide loads)"]) + float(row["hbm_write_bytes"])
        if "k_lstm_layer<512" in k:
            out["k_lstm_layer<512"] = b
        if any(t in k for t in ("k_cigar_scan", "k_tile_fill", "k_pileup", "k_site", "k_collect", "k_write_windows", "k_scan")):
            builder += b
    out["builder"] = builder
    return out


def bf16_secondary(device_id, weights, dbatch, P, dev, pad, groups=6):
    """BASELINE configs[2] flavour: the same step (1 region + 512 windows, 8 fused per launch chain) with
    PV_DTYPE_BF16_INPUT_GEMM: decoder input projection and linear_1 on the bf16 MFMA as a 3-term hi/lo split
    (softmax still within 1e-4 of the reference), recurrence fp32. Not part of `value`."""
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
    gemm_ms = prof["k_gemm_bf16x3_dec"][0] / prof["k_gemm_bf16x3_dec"][1]
    flop = 3 * 2.0 * (CALLERS * BATCH * 33) * 2048 * 512
    return {"value": groups * CALLERS * BATCH / dt, "unit": "windows/s", "dtype": "bf16x3 input GEMMs + f32 recurrence",
            "steps": groups * CALLERS, "ms_per_step": dt / (groups * CALLERS) * 1e3,
            "input_gemm": {"kernel": "k_gemm_bf16x3 (decoder input projection, M=%d N=2048 K=512, 3 MFMA terms)" % (CALLERS * BATCH * 33),
                           "launch_ms": gemm_ms, "achieved": flop / gemm_ms / 1e9, "peak": 2500.0, "unit": "TFLOP/s (bf16 MFMA)",
                           "frac": flop / gemm_ms / 1e9 / 2500.0},
            "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items() if k.startswith("k_") and "summary" not in k}}


def p2_secondary(ctx, dev):
    """secondary figures for the bi-GRU polisher plan (north_star's '1000 x 100 x feature' shape): the 19-window
    sliding loop over [B,1000,10] chunks at the SURVEY 8(d) batch (64 chunks), at 1000 chunks and at a
    chip-filling batch. Not part of `value`."""
    import torch
    from pepper_thesis_amd import _ffi, synth
    ctx.load_p2(synth.make_weights_p2(4321))
    lib = _ffi.load()
    out = {"flop_per_100col_window": 80_435_200, "windows_per_chunk": 19}
    for B in (64, 1000, 8192):
        x = torch.from_numpy(synth.synth_p2_images(7, B)).to(dev)
        labels = torch.zeros((B, 1000), dtype=torch.uint8, device=dev)
        _ffi.check(lib.pv_rnn_forward_p2_dev(ctx.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
        ctx.synchronize()
        ctx.profile_begin()
        for _ in range(2):
            _ffi.check(lib.pv_rnn_forward_p2_dev(ctx.handle, x.data_ptr(), B, labels.data_ptr(), None, None))
        ms, n = ctx.profile_end()["k_gru_p2"]
        ms /= n
        out["B%d" % B] = {"ms": ms, "chunks_per_s": B / ms * 1e3, "windows_100col_per_s": 19 * B / ms * 1e3,
                          "tflops": 80_435_200 * 19 * B / ms / 1e9, "frac_of_f32_peak": 80_435_200 * 19 * B / ms / 1e9 / PEAK_F32_TFLOPS}
        del x, labels
    return out


def single_call_secondary(ctx, dev):
    """secondary: ONE caller's batches through pv_rnn_forward_p1_dev alone (no caller fusion, no image builder), i.e. what
    a single `run_inference -bs 512` loop sees, and the same call at 2048 and 4096 windows. Not part of `value`."""
    import torch
    from pepper_thesis_amd import synth
    out = {}
    for B in (512, 2048, 4096):
        x = torch.from_numpy(synth.synth_windows(3, B)).to(dev)
        probs = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        for _ in range(2):
            ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
        ctx.synchronize()
        ctx.profile_begin()
        for _ in range(5):
            ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
        prof = ctx.profile_end()
        ms = sum(v[0] / v[1] for v in prof.values())
        out["B%d" % B] = {"ms": ms, "windows_per_s": B / ms * 1e3, "tflops": FLOP_PER_WINDOW * B / ms / 1e9,
                          "frac_of_f32_peak": FLOP_PER_WINDOW * B / ms / 1e9 / PEAK_F32_TFLOPS}
        del x, probs
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-p2", action="store_true", help="skip the secondary bi-GRU (P2) figures")
    ap.add_argument("--no-bf16", action="store_true", help="skip the secondary configs[2] (bf16 input GEMM) figure")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_dev = torch.cuda.device_count()
    dev_id = local_rank % max(n_dev, 1)  # one rank per GPU; (rehearsals with more ranks than GPUs share devices)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_id)
        backend = os.environ.get("PV_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only for single-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_id))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(dev_id)
    dev = "cuda:%d" % dev_id

    from pepper_thesis_amd import runtime, synth
    from pepper_thesis_amd.batch import PRESETS, pack_regions
    from pepper_thesis_amd.device import DeviceBatch, DeviceOut
    from pepper_thesis_amd.dist import gather_predictions

    K, W = int(args.steps), int(args.warmup)
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

    chains, wchains = chains_for(K), chains_for(W)
    groups = max(len(chains), 1)

    # ---- synthetic workload (seeded; rank r gets its own regions: regions shard across GPUs) ---------
    t0 = time.time()
    # NBATCH distinct batches of CALLERS regions, used round-robin by the launch chains, so that the image builder never
    # finds its inputs (116 MB per batch) in the 256 MB Infinity Cache from the previous chain
    NBATCH = max(1, int(os.environ.get("PV_BENCH_NBATCH", "4")))
    batches = []
    for j in range(NBATCH):
        regs_j = [synth.synth_region(1234 + 97 * ((rank * NBATCH + j) * CALLERS + i), region_len=REGION_LEN, depth=DEPTH,
                                     read_len=READ_LEN, site_every=SITE_EVERY,
                                     ref_start=1_000_000 + ((rank * NBATCH + j) * CALLERS + i) * (REGION_LEN - 200))
                  for i in range(CALLERS)]
        if j == 0:
            regions = regs_j
        batches.append(pack_regions(regs_j))
    batch = batches[0]
    weights = synth.make_weights_p1(1234)
    pad = synth.synth_windows(4242 + rank, CALLERS * BATCH)
    log("rank %d: generated %d regions (%d reads, %.1f M bases) in %.1f s" %
        (rank, NBATCH * CALLERS, sum(b.n_reads for b in batches), sum(b.n_bases for b in batches) / 1e6, time.time() - t0))
    P = PRESETS["ont_r9_guppy5_sup"]

    ctx = runtime.Context(dev_id)
    ctx.load_p1(weights)
    dbatches = [DeviceBatch(b, dev) for b in batches]
    dbatch = dbatches[0]
    # Two window buffers: the image builder of group g+1 (stream s_build) overlaps the RNN of group g
    # (stream s_rnn); events order builder(g) -> rnn(g) and rnn(g) -> builder(g+2) (buffer reuse).
    wins = [torch.from_numpy(pad).to(dev) for _ in range(2)]          # [4096,33,26] int8, builder writes the front
    douts = [DeviceOut(CALLERS * BATCH, CALLERS * BATCH * 16, dev, images=w) for w in wins]
    probs = torch.zeros((groups, CALLERS * BATCH, 3), dtype=torch.float32, device=dev)
    s_build, s_rnn = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ev_built = [torch.cuda.Event() for _ in range(2)]
    ev_used = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    state = {"n": 0}

    # Default: ONE stream, builder and RNN of a group back to back. The LSTM workgroups take whole CUs (2 x 255 VGPRs per
    # SIMD, 132 KB LDS), so a second stream (PV_BENCH_OVERLAP=1: builder of group g+1 beside the RNN of group g) only lets
    # builder kernels slip into the RNN's kernel boundaries: +1 % throughput, but every kernel's duration is then stretched by
    # its neighbours and the live per-kernel times no longer agree with the rocprofv3 averages.
    overlap = os.environ.get("PV_BENCH_OVERLAP", "0") != "0"
    if not overlap:
        s_build = s_rnn

    # chains shorter than CALLERS steps (K or W not a multiple of CALLERS): the first regions of batch 0
    rem_batches = {n: DeviceBatch(batches[0].select(list(range(n))), dev) for n in set(chains + wchains) if n != CALLERS}
    rem_douts = {n: [DeviceOut(n * BATCH, n * BATCH * 16, dev, images=w) for w in wins] for n in rem_batches}

    def group(g, ncall=CALLERS):
        k = state["n"] & 1
        if state["n"] >= 2:
            s_build.wait_event(ev_used[k])
        if ncall == CALLERS:
            ctx.summarize_dev(dbatches[g % NBATCH], P, douts[k], stream=s_build.cuda_stream)
        else:
            ctx.summarize_dev(rem_batches[ncall], P, rem_douts[ncall][k], stream=s_build.cuda_stream)
        ev_built[k].record(s_build)
        s_rnn.wait_event(ev_built[k])
        ctx.forward_p1_dev(wins[k].data_ptr(), ncall * BATCH, probs[g % groups].data_ptr(), stream=s_rnn.cuda_stream)
        ev_used[k].record(s_rnn)
        state["n"] += 1

    def drain():
        s_build.synchronize()
        s_rnn.synchronize()

    # image-builder roofline: measured in isolation (its launches overlap the RNN in the timed region,
    # which stretches their event-bracketed durations)
    for db in dbatches + [dbatch]:  # first calls size the workspace arena (every batch once: their read / op counts differ)
        ctx.summarize_dev(db, P, douts[0], stream=s_build.cuda_stream)
    s_build.synchronize()
    ctx.forward_p1_dev(wins[0].data_ptr(), CALLERS * BATCH, probs[0].data_ptr(), stream=s_rnn.cuda_stream)  # RNN workspace at full size
    s_rnn.synchronize()
    ctx.profile_begin()
    for _ in range(10):
        ctx.summarize_dev(dbatch, P, douts[0], stream=s_build.cuda_stream)
    prof_builder = ctx.profile_end()
    dout = douts[0]
    n_windows_region = dout.n_out()   # windows of batch 0 (the batch the isolated builder measurement used)
    for g, nc in enumerate(wchains):
        group(g, nc)
    drain()
    for d_ in douts:
        assert d_.status() == 0, "device status %d" % d_.status()
        assert d_.n_out() <= CALLERS * BATCH, "regions yield %d windows > %d" % (d_.n_out(), CALLERS * BATCH)
    assert n_windows_region <= CALLERS * BATCH, "regions yield %d windows > %d" % (n_windows_region, CALLERS * BATCH)

    # ---- timed region ----------------------------------------------------------------------------------
    if dist is not None:
        # untimed warm-up of the one exchange step as well: the first all_gather of a communicator sets up its channels
        # and loads its kernels (RCCL does that lazily), which must not land inside the timed region
        gather_predictions(probs.view(-1, 3), dst=0)
        dist.barrier()
    torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for g, nc in enumerate(chains):
        group(g, nc)
    gathered = None
    if dist is not None:
        drain()
        gathered = gather_predictions(probs.view(-1, 3), dst=0)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_end()
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        total_windows = K * BATCH * world
        value = total_windows / dt
        dec_ms, dec_n = prof.get("k_lstm_layer_dec", (0.0, 0))
        dec_avg_s = dec_ms / max(dec_n, 1) / 1e3
        achieved_tf = FLOP_DEC_PER_WINDOW * K * BATCH / (dec_ms / 1e3) / 1e12 if dec_ms > 0 else 0.0  # all launches of the timed region
        sum_ms, sum_n = prof_builder.get("summary_pipeline", (0.0, 0))
        pile_ms, pile_n = prof_builder.get("k_pileup", (0.0, 0))
        traffic = pmc_traffic()
        alg_bytes = batch.algorithmic_bytes(n_windows_region)
        out = {
            "metric": "pileup windows/sec (whole node) + Mbp/sec inferred, HG003 chr20 ONT R9",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: HG003-chr20-shaped ONT R9 synthetic, batch=512 windows/step, fp32 bi-LSTM P1, "
                                   "1 region (R=100200, 60x, 10 kb reads) per step, %d steps fused per launch chain" % CALLERS,
                       "batch": BATCH, "callers": CALLERS, "region_len": REGION_LEN, "depth": DEPTH,
                       "windows_per_region": n_windows_region / CALLERS, "parallelism": "region-sharded x%d" % world},
            "mbp_per_s": K * world * REGION_LEN / 1e6 / dt,
            "roofline": {"bound": "mfma", "kernel": "k_lstm_layer<512> (decoder bi-LSTM, fused input projection + recurrence)",
                         "achieved": achieved_tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_F32_TFLOPS, "traffic": traffic.get("k_lstm_layer<512"),
                         "traffic_note": traffic.get("note"),
                         "launch_ms": dec_avg_s * 1e3, "flop_per_launch": FLOP_DEC_PER_WINDOW * CALLERS * BATCH},
            "roofline_builder": {"bound": "hbm", "kernel": "summary pipeline (k_cigar_scan .. k_write_windows), %d regions/launch" % CALLERS,
                                 "achieved": alg_bytes / (sum_ms / max(sum_n, 1) / 1e3) / 1e9 if sum_ms > 0 else 0.0,
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": (alg_bytes / (sum_ms / max(sum_n, 1) / 1e3) / 1e9 / PEAK_HBM_GBS) if sum_ms > 0 else 0.0,
                                 "traffic": traffic.get("builder"), "launch_ms": sum_ms / max(sum_n, 1),
                                 "k_pileup_ms": pile_ms / max(pile_n, 1), "measured": "in isolation, before the timed region",
                                 "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_ms": {k: v[0] / max(v[1], 1) for k, v in prof.items()},
            "rnn_model_tflops": FLOP_PER_WINDOW * value / 1e12,
        }
        if world == 1 and not args.no_bf16:
            try:
                out["config2_bf16_input_gemm"] = bf16_secondary(dev_id, weights, dbatch, P, dev, pad)
            except Exception as e:
                out["config2_bf16_input_gemm"] = {"error": repr(e)}
        if world == 1 and not args.no_p2:
            try:
                out["p1_single_call"] = single_call_secondary(ctx, dev)
            except Exception as e:
                out["p1_single_call"] = {"error": repr(e)}
            try:
                out["p2_bigru"] = p2_secondary(ctx, dev)
            except Exception as e:
                out["p2_bigru"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(weights, regions[0], min(len(os.sched_getaffinity(0)), 16))
            except Exception as e:  # the baseline is reporting only; never fail the bench for it
                out["cpu_baseline"] = {"value": None, "unit": "windows/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
