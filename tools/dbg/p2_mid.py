import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pepper_thesis_amd import runtime, synth, _ffi
w2 = synth.make_weights_p2(17, 2.0)
for B in [int(a) for a in sys.argv[1:]] or [1024, 1500, 2048]:
    y = synth.synth_p2_images(20 + B, B)
    cb = runtime.Context(0); cb.load_p2(w2, _ffi.PV_DTYPE_BF16_INPUT_GEMM)
    dy = torch.from_numpy(y).to("cuda:0"); dl = torch.zeros((B, 1000), dtype=torch.uint8, device="cuda:0"); da = torch.zeros((B, 1000, 5), dtype=torch.float32, device="cuda:0")
    for _ in range(2): cb.forward_p2_dev(dy.data_ptr(), B, dl.data_ptr(), da.data_ptr())
    cb.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): cb.forward_p2_dev(dy.data_ptr(), B, dl.data_ptr(), da.data_ptr())
    cb.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    cb.profile_begin(); cb.forward_p2_dev(dy.data_ptr(), B, dl.data_ptr(), da.data_ptr()); prof = cb.profile_end()
    print("B=%d: %.2f ms; %s; checksum %.6f" % (B, ms, {k: round(v[0], 2) for k, v in prof.items()}, float(da.double().sum())), flush=True)
    cb.close()
