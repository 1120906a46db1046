"""GPU: hipGraph capture of *_dev call sequences (pv_graph_begin / pv_graph_end / pv_graph_launch). A replay must give what
the same eager calls give on the same buffers, for the small-batch split forms too (their exchange tags advance on the device)."""
import numpy as np
import pytest
import torch

from pepper_thesis_amd import _ffi, runtime, synth
from pepper_thesis_amd.batch import PRESETS, pack_regions
from pepper_thesis_amd.device import DeviceBatch, DeviceOut

pytestmark = pytest.mark.gpu


def test_graph_of_one_small_p1_call(hip_ctx):
    """a lone 512-window call (unit-split LSTM form): captured once, replayed on refilled buffers"""
    dev = "cuda:%d" % hip_ctx.device_id
    hip_ctx.load_p1(synth.make_weights_p1(41, 2.0))
    xs = [synth.synth_windows(4100 + i, 512) for i in range(3)]
    xbuf = torch.from_numpy(xs[0]).to(dev)
    pbuf = torch.zeros((512, 3), dtype=torch.float32, device=dev)
    eager = []
    for x in xs:
        xbuf.copy_(torch.from_numpy(x))
        torch.cuda.synchronize()
        hip_ctx.forward_p1_dev(xbuf.data_ptr(), 512, pbuf.data_ptr())
        hip_ctx.synchronize()
        eager.append(pbuf.cpu().numpy().copy())
    with hip_ctx.graph_capture() as g:
        hip_ctx.forward_p1_dev(xbuf.data_ptr(), 512, pbuf.data_ptr())
    for k in (1, 0, 2, 2, 1):
        xbuf.copy_(torch.from_numpy(xs[k]))
        pbuf.zero_()
        torch.cuda.synchronize()
        g.launch()
        hip_ctx.synchronize()
        assert np.array_equal(pbuf.cpu().numpy().view(np.uint32), eager[k].view(np.uint32)), k
    assert hip_ctx.exchange_timeouts() == 0
    g.close()


def test_graph_of_builder_plus_rnn_chain(hip_ctx):
    """image builder + P1 over its windows as ONE graph: replays reproduce the eager chain"""
    dev = "cuda:%d" % hip_ctx.device_id
    hip_ctx.load_p1(synth.make_weights_p1(42, 2.0))
    P = PRESETS["ont_r9_guppy5_sup"]
    regs = [synth.synth_region(4200 + k, region_len=3000, depth=30, read_len=900, site_every=40) for k in range(2)]
    db = DeviceBatch(pack_regions(regs), dev)
    cap = 1024
    images = torch.zeros((cap, 33, 26), dtype=torch.int8, device=dev)
    dout = DeviceOut(cap, cap * 16, dev, images=images)
    probs = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
    st = hip_ctx.stream

    def chain():
        hip_ctx.summarize_dev(db, P, dout, stream=st)
        hip_ctx.forward_p1_dev(images.data_ptr(), cap, probs.data_ptr(), stream=st)

    chain()
    hip_ctx.synchronize()
    n = dout.n_out()
    assert 50 < n <= cap and dout.status() == 0
    want_img, want_p = images.cpu().numpy().copy(), probs.cpu().numpy().copy()
    with hip_ctx.graph_capture(st) as g:
        chain()
    for _ in range(3):
        images.zero_(); probs.zero_(); dout.counts.zero_()
        torch.cuda.synchronize()
        g.launch()
        hip_ctx.synchronize()
        assert dout.n_out() == n and dout.status() == 0
        assert np.array_equal(images.cpu().numpy()[:n], want_img[:n])
        assert np.array_equal(probs.cpu().numpy()[:n].view(np.uint32), want_p[:n].view(np.uint32))
    g.close()


def test_graph_of_small_p2_call(hip_ctx):
    """the polisher's sliding loop over 20 chunks (unit-split GRU form) as a graph"""
    dev = "cuda:%d" % hip_ctx.device_id
    hip_ctx.load_p2(synth.make_weights_p2(43, 2.0))
    lib = _ffi.load()
    ys = [synth.synth_p2_images(4300 + i, 20) for i in range(2)]
    ybuf = torch.from_numpy(ys[0]).to(dev)
    lab = torch.zeros((20, 1000), dtype=torch.uint8, device=dev)
    acc = torch.zeros((20, 1000, 5), dtype=torch.float32, device=dev)

    def call():
        _ffi.check(lib.pv_rnn_forward_p2_dev(hip_ctx.handle, ybuf.data_ptr(), 20, lab.data_ptr(), acc.data_ptr(), None))

    eager = []
    for y in ys:
        ybuf.copy_(torch.from_numpy(y))
        torch.cuda.synchronize()
        call()
        hip_ctx.synchronize()
        eager.append((lab.cpu().numpy().copy(), acc.cpu().numpy().copy()))
    with hip_ctx.graph_capture() as g:
        call()
    for k in (1, 0, 1):
        ybuf.copy_(torch.from_numpy(ys[k]))
        torch.cuda.synchronize()
        g.launch()
        hip_ctx.synchronize()
        assert np.array_equal(lab.cpu().numpy(), eager[k][0])
        assert np.array_equal(acc.cpu().numpy().view(np.uint32), eager[k][1].view(np.uint32))
    assert hip_ctx.exchange_timeouts() == 0
    g.close()


def test_capture_refuses_a_cold_workspace():
    """inside a capture nothing may allocate: a first call of its size must have run eagerly before"""
    ctx = runtime.Context(0)
    ctx.load_p1(synth.make_weights_p1(44))
    x = torch.from_numpy(synth.synth_windows(4400, 64)).to("cuda:0")
    p = torch.zeros((64, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(_ffi.PepperHipError) as e:
        with ctx.graph_capture():
            ctx.forward_p1_dev(x.data_ptr(), 64, p.data_ptr())
    assert e.value.code == _ffi.PV_ERR_STATE
    ctx.forward_p1_dev(x.data_ptr(), 64, p.data_ptr())   # the context is usable afterwards
    ctx.synchronize()
    assert np.isfinite(p.cpu().numpy()).all()
    ctx.close()
