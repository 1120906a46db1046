"""configs[2] flavour, RNN only: per-kernel times of PV_DTYPE_BF16_INPUT_GEMM at a given batch (default 8192 windows)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pepper_thesis_amd import _ffi, runtime, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = runtime.Context(0)
ctx.load_p1(synth.make_weights_p1(1234), _ffi.PV_DTYPE_BF16_INPUT_GEMM)
ctx.set_option("p1_bf16_min_batch", 0)   # (the bf16x3 kernels at any size: this tool reports their times)
x = torch.from_numpy(synth.synth_windows(3, B)).to("cuda:0")
probs = torch.zeros((B, 3), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
ctx.synchronize()
wall = (time.perf_counter() - t0) / 10
ctx.profile_begin()
for _ in range(5):
    ctx.forward_p1_dev(x.data_ptr(), B, probs.data_ptr())
prof = ctx.profile_end()
ms = {k: v[0] / v[1] for k, v in prof.items()}
flop = 3 * 2.0 * (B * 33) * 2048 * 512
out = {"B": B, "wall_ms": wall * 1e3, "windows_per_s": B / wall, "kernel_ms": ms,
       "gemm_dec_tflops": flop / ms["k_gemm_bf16x3_dec"] / 1e9, "gemm_dec_frac": flop / ms["k_gemm_bf16x3_dec"] / 1e9 / 2500,
       "gemm_lin1_tflops": 3 * 2.0 * B * 512 * 16896 / ms["k_gemm_bf16x3_lin1"] / 1e9}
print(json.dumps(out, indent=1))
