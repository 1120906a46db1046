"""One caller's small batches through pv_rnn_forward_p1_dev: the unit-split LSTM form (k_lstm_split) against the
one-workgroup 16-row form, per kernel; and N independent callers on their own streams in both forms.
python tools/bench_single.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLOP_PER_WINDOW = 161_328_128
PEAK = 157.3


def one(ctx, dev, B, reps=30):
    import torch
    from pepper_thesis_amd import synth
    x = torch.from_numpy(synth.synth_windows(3, B)).to(dev)
    probs = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    for _ in range(3):
        ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    ctx.profile_begin()
    for _ in range(5):
        ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
    prof = ctx.profile_end()
    return {"wall_ms": round(wall, 4), "frac_of_f32_peak": round(FLOP_PER_WINDOW * B / wall / 1e9 / PEAK, 4),
            "kernels_ms": {k: round(v[0] / v[1], 4) for k, v in prof.items()}}


def callers(ctx, dev, weights, ncall, reps=20):
    import torch
    from pepper_thesis_amd import runtime, synth
    ctxs = [ctx] + [runtime.Context(ctx.device_id) for _ in range(ncall - 1)]
    for c in ctxs[1:]:
        c.set_option("lstm_split", ctx.get_option("lstm_split"))
    for c in ctxs[1:]:
        c.load_p1(weights)
    xs = [torch.from_numpy(synth.synth_windows(30 + i, 512)).to(dev) for i in range(ncall)]
    ps = [torch.zeros((512, 3), dtype=torch.float32, device=dev) for _ in range(ncall)]
    for c, x, p in zip(ctxs, xs, ps):
        c.forward_p1_dev(x.data_ptr(), 512, p.data_ptr())
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for c, x, p in zip(ctxs, xs, ps):
            c.forward_p1_dev(x.data_ptr(), 512, p.data_ptr())
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    for c in ctxs[1:]:
        c.close()
    return round(reps * ncall * 512 / dt, 1)


if __name__ == "__main__":
    from pepper_thesis_amd import runtime, synth
    ctx = runtime.Context(0)
    w = synth.make_weights_p1(1234)
    ctx.load_p1(w)
    out = {}
    for form in ("split", "one_workgroup"):
        ctx.set_option("lstm_split", 1 if form == "split" else 0)
        out[form] = {"B%d" % B: one(ctx, "cuda:0", B) for B in (64, 256, 512, 1024)}
        out[form]["callers4_x_B512_windows_per_s"] = callers(ctx, "cuda:0", w, 4)
        sys.stderr.write("[bench_single] %s done\n" % form)
    print(json.dumps(out))
