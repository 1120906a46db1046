"""debug: where does the unit-split LSTM form differ from the 16-row one-workgroup form?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_thesis_amd import runtime, synth
ctx = runtime.Context(0)
B = 16
x = synth.synth_windows(3100 + B, B)
for mode in ("full", "hh_only_units_0_63"):
    w = synth.make_weights_p1(31, 2.0)
    for k in list(w):
        if mode == "no_hh" and "weight_hh" in k and k.startswith("encoder"):
            w[k] = np.zeros_like(w[k])
        if mode == "hh_only_units_0_63" and "weight_hh" in k and k.startswith("encoder"):
            w[k] = w[k].copy(); w[k][:, 64:] = 0
    ctx.load_p1(w)
    os.environ.pop("PV_LSTM_ROWS", None)
    p1, e1, d1 = ctx.forward_p1(x, taps=True)
    os.environ["PV_LSTM_ROWS"] = "16"
    p0, e0, d0 = ctx.forward_p1(x, taps=True)
    for t in (0, 1, 2):
        diff = np.abs(e1[:, t, :256] - e0[:, t, :256])   # forward direction
        bad = diff > 1e-6
        rows = np.unique(np.nonzero(bad)[0]).tolist(); cols = np.unique(np.nonzero(bad)[1])
        print(mode, "t", t, "bad", int(bad.sum()), "max", float(diff.max()), "rows", rows, "ncols", len(cols), "col range", (int(cols.min()), int(cols.max())) if len(cols) else None,
              "bad per 64-col part", [int(bad[:, 64 * q:64 * q + 64].sum()) for q in range(4)])
