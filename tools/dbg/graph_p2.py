import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pepper_thesis_amd import runtime, synth, _ffi
ctx = runtime.Context(0)
ctx.load_p2(synth.make_weights_p2(43, 2.0))
lib = _ffi.load()
dev = "cuda:0"
ys = [synth.synth_p2_images(4300 + i, 20) for i in range(2)]
ybuf = torch.from_numpy(ys[0]).to(dev)
lab = torch.zeros((20, 1000), dtype=torch.uint8, device=dev)
acc = torch.zeros((20, 1000, 5), dtype=torch.float32, device=dev)
def call():
    _ffi.check(lib.pv_rnn_forward_p2_dev(ctx.handle, ybuf.data_ptr(), 20, lab.data_ptr(), acc.data_ptr(), None))
eager = []
for y in ys:
    ybuf.copy_(torch.from_numpy(y)); torch.cuda.synchronize()
    call(); ctx.synchronize(check=False)
    eager.append((lab.cpu().numpy().copy(), acc.cpu().numpy().copy()))
ref = [ctx.forward_p2(y, want_acc=True) for y in ys]
for k in range(2):
    print("eager vs host form", k, np.array_equal(eager[k][0], ref[k][0]), np.abs(eager[k][1]-ref[k][1]).max())
with ctx.graph_capture() as g:
    call()
for k in (1, 0, 1):
    ybuf.copy_(torch.from_numpy(ys[k])); torch.cuda.synchronize()
    g.launch(); ctx.synchronize(check=False)
    print("graph", k, np.array_equal(lab.cpu().numpy(), eager[k][0]), np.abs(acc.cpu().numpy()-eager[k][1]).max(), "timeouts", -1 if "noexch" in sys.argv else ctx.exchange_timeouts())
print("final timeouts", ctx.exchange_timeouts())
