"""Thin Python handle on a pv_ctx (one HIP device + stream + workspace) of the C-ABI.

All compute happens in csrc/libpepper_hip.so; this module only marshals numpy arrays / device
pointers. There is no CPU fallback: constructing a Context without a HIP device raises.
"""
import ctypes as C
from typing import Optional

import numpy as np

from . import _ffi
from .batch import OutBuffers, Params, RegionBatch, SummaryOut


def _dir_struct(w, prefix, suffix, keep):
    d = _ffi.pv_rnn_dir()
    for field, key in (("w_ih", "weight_ih_l0"), ("w_hh", "weight_hh_l0"), ("b_ih", "bias_ih_l0"), ("b_hh", "bias_hh_l0")):
        a = np.ascontiguousarray(w["%s.%s%s" % (prefix, key, suffix)], dtype=np.float32)
        keep.append(a)
        setattr(d, field, a.ctypes.data)
    return d


class _GraphCapture:
    def __init__(self, ctx, stream):
        self.ctx, self.stream, self.handle = ctx, int(stream or ctx.stream), None

    def __enter__(self):
        _ffi.check(self.ctx.lib.pv_graph_begin(self.ctx.handle, self.stream))
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        rc = self.ctx.lib.pv_graph_end(self.ctx.handle, C.byref(h))
        if et is None:
            _ffi.check(rc)
            self.handle = h
        elif rc == 0:
            self.ctx.lib.pv_graph_destroy(h)
        return False

    def launch(self, stream: int = 0):
        _ffi.check(self.ctx.lib.pv_graph_launch(self.handle, int(stream or self.stream)))

    def close(self):
        if self.handle:
            self.ctx.lib.pv_graph_destroy(self.handle)
            self.handle = None


class Context:
    """pv_create / pv_destroy with the reference operators as methods."""

    def __init__(self, device_id: int = 0):
        self.lib = _ffi.load()
        self.handle = self.lib.pv_create(int(device_id))
        if not self.handle:
            msg = self.lib.pv_last_error()
            raise _ffi.PepperHipError(_ffi.PV_ERR_NO_DEVICE, msg.decode() if msg else "pv_create failed")
        self.device_id = int(device_id)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self.lib.pv_stream(self.handle) or 0)

    def synchronize(self, check: bool = True):
        """wait for the context's stream. check=True also asks whether a poll of the split RNN forms gave up since the last
        synchronisation point (the outputs of such calls are poisoned on the device: NaN / label 255) and raises
        PepperHipError(PV_ERR_STATE) if so - callers of the asynchronous *_dev forms get the verdict where they wait."""
        _ffi.check(self.lib.pv_synchronize(self.handle))
        if check:
            n = self.exchange_timeouts()
            if n:
                raise _ffi.PepperHipError(_ffi.PV_ERR_STATE, "%d exchange polls of a split RNN form timed out: the outputs of the "
                                          "calls since the last synchronisation are poisoned (set_option('shared_device', 1) "
                                          "on a GPU shared with other work)" % n)

    def set_option(self, name: str, value: int):
        """kernel-form options of this context (include/pepper_hip.h, pv_set_option)"""
        _ffi.check(self.lib.pv_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int()
        _ffi.check(self.lib.pv_get_option(self.handle, name.encode(), C.byref(v)))
        return int(v.value)

    def graph_capture(self, stream: int = 0):
        """context manager: the *_dev calls made inside (with this stream) are recorded into a hipGraph instead of run;
        `with ctx.graph_capture(s) as g: ...` then `g.launch()` per batch after refilling the same input buffers.
        Run the same calls once eagerly before (the workspace cannot grow inside a capture)."""
        return _GraphCapture(self, stream)

    def exchange_timeouts(self) -> int:
        """polls of the split kernel forms that gave up since the last call (synchronises; 0 = all results good)"""
        n = int(self.lib.pv_rnn_exchange_timeouts(self.handle))
        if n < 0:
            _ffi.check(n)
        return n

    def workspace_bytes(self) -> int:
        return int(self.lib.pv_workspace_bytes(self.handle))

    # ---- image builder ---------------------------------------------------------------------------
    def summarize(self, batch: RegionBatch, params: Params, want_i32: bool = False,
                  capacity: Optional[int] = None, str_capacity: Optional[int] = None) -> SummaryOut:
        """RegionalSummaryGenerator.generate_summary for every region of the batch (host buffers)."""
        cin, cp = batch.as_c(), params.as_c()
        cap = capacity or max(1024, batch.n_regions * 1024)
        scap = str_capacity or cap * 8
        for _ in range(3):
            ob = OutBuffers(cap, scap, want_i32)
            rc = self.lib.pv_summarize_regions(self.handle, C.byref(cin), C.byref(cp), C.byref(ob.c))
            if rc == _ffi.PV_ERR_CAPACITY:
                cap, scap = max(cap, int(ob.c.n_out)), max(scap, int(ob.c.str_bytes))
                continue
            _ffi.check(rc)
            return ob.result()
        _ffi.check(rc)

    def summarize_hp(self, batch: RegionBatch, params: Params, want_i32: bool = False,
                     capacity: Optional[int] = None, str_capacity: Optional[int] = None) -> SummaryOut:
        """RegionalSummaryGeneratorHP.generate_summary (haplotag-aware, 48 planes x 21 rows) for every region of the
        batch; batch.read_hp carries type_read::hp_tag (None = untagged reads); params must carry window 20 / 48 features
        (batch.hp_params)."""
        from .batch import hp_pointer
        cin, cp = batch.as_c(), params.as_c()
        cap = capacity or max(1024, batch.n_regions * 1024)
        scap = str_capacity or cap * 8
        for _ in range(3):
            ob = OutBuffers(cap, scap, want_i32, _ffi.PV_HP_WINDOW_ROWS, _ffi.PV_HP_FEATURES)
            rc = self.lib.pv_summarize_regions_hp(self.handle, C.byref(cin), hp_pointer(batch), C.byref(cp), C.byref(ob.c))
            if rc == _ffi.PV_ERR_CAPACITY:
                cap, scap = max(cap, int(ob.c.n_out)), max(scap, int(ob.c.str_bytes))
                continue
            _ffi.check(rc)
            return ob.result()
        _ffi.check(rc)

    # ---- RNN ------------------------------------------------------------------------------------------
    def load_p1(self, weights: dict, dtype: int = _ffi.PV_DTYPE_F32):
        """weights: state_dict of the pepper_variant TransducerGRU as numpy arrays (ModelHander.py:18-44)."""
        keep = []
        w = _ffi.pv_weights_p1()
        for d, suffix in enumerate(("", "_reverse")):
            w.encoder[d] = _dir_struct(weights, "encoder", suffix, keep)
            w.decoder[d] = _dir_struct(weights, "decoder", suffix, keep)
        for i in range(5):
            a = np.ascontiguousarray(weights["linear_%d.weight" % (i + 1)], dtype=np.float32)
            b = np.ascontiguousarray(weights["linear_%d.bias" % (i + 1)], dtype=np.float32)
            keep += [a, b]
            w.linear_w[i], w.linear_b[i] = a.ctypes.data, b.ctypes.data
        a = np.ascontiguousarray(weights["output_layer_type.weight"], dtype=np.float32)
        b = np.ascontiguousarray(weights["output_layer_type.bias"], dtype=np.float32)
        keep += [a, b]
        w.out_w, w.out_b = a.ctypes.data, b.ctypes.data
        _ffi.check(self.lib.pv_rnn_load_p1(self.handle, C.byref(w), int(dtype)))

    def forward_p1(self, images: np.ndarray, taps: bool = False):
        """images int8 [B,33,26] -> probs float32 [B,3] (+ encoder/decoder outputs when taps=True)."""
        x = np.ascontiguousarray(images, dtype=np.int8)
        assert x.ndim == 3 and x.shape[1:] == (33, 26), x.shape
        B = x.shape[0]
        probs = np.zeros((B, 3), np.float32)
        if not taps:
            _ffi.check(self.lib.pv_rnn_forward_p1(self.handle, x.ctypes.data, B, probs.ctypes.data))
            return probs
        enc = np.zeros((B, 33, 512), np.float32)
        dec = np.zeros((B, 33, 512), np.float32)
        _ffi.check(self.lib.pv_rnn_forward_p1_debug(self.handle, x.ctypes.data, B, probs.ctypes.data,
                                                    enc.ctypes.data, dec.ctypes.data))
        return probs, enc, dec

    def forward_p1_dev(self, d_images_ptr: int, B: int, d_probs_ptr: int, stream: int = 0):
        """device pointers in, asynchronous on `stream` (0 = the context's stream)."""
        _ffi.check(self.lib.pv_rnn_forward_p1_dev(self.handle, d_images_ptr, int(B), d_probs_ptr, stream or None))

    def load_p2(self, weights: dict, dtype: int = _ffi.PV_DTYPE_F32):
        keep = []
        w = _ffi.pv_weights_p2()
        for d, suffix in enumerate(("", "_reverse")):
            w.encoder[d] = _dir_struct(weights, "gru_encoder", suffix, keep)
            w.decoder[d] = _dir_struct(weights, "gru_decoder", suffix, keep)
        a = np.ascontiguousarray(weights["dense1.weight"], dtype=np.float32)
        b = np.ascontiguousarray(weights["dense1.bias"], dtype=np.float32)
        keep += [a, b]
        w.dense_w, w.dense_b = a.ctypes.data, b.ctypes.data
        _ffi.check(self.lib.pv_rnn_load_p2(self.handle, C.byref(w), int(dtype)))

    def forward_p2(self, images: np.ndarray, want_acc: bool = False):
        """images uint8 [B,1000,10] -> labels uint8 [B,1000] (+ accumulated softmax [B,1000,5])."""
        x = np.ascontiguousarray(images, dtype=np.uint8)
        assert x.ndim == 3 and x.shape[1:] == (1000, 10), x.shape
        B = x.shape[0]
        labels = np.zeros((B, 1000), np.uint8)
        acc = np.zeros((B, 1000, 5), np.float32) if want_acc else None
        _ffi.check(self.lib.pv_rnn_forward_p2(self.handle, x.ctypes.data, B, labels.ctypes.data,
                                              None if acc is None else acc.ctypes.data))
        return (labels, acc) if want_acc else labels

    def forward_p2_dev(self, d_images: int, B: int, d_labels: int, d_acc: int = 0, stream: int = 0):
        """asynchronous; device pointers: images uint8 [B,1000,10] -> labels uint8 [B,1000] (+ acc float [B,1000,5])."""
        _ffi.check(self.lib.pv_rnn_forward_p2_dev(self.handle, d_images, int(B), d_labels, d_acc or None, stream or None))

    def forward_p2_window(self, images: np.ndarray, hidden: Optional[np.ndarray] = None):
        """TransducerGRU.forward(x, hidden): images uint8 [B,100,10], hidden [B,2,128] or None -> (logits [B,100,5], hidden [B,2,128])"""
        x = np.ascontiguousarray(images, dtype=np.uint8)
        assert x.ndim == 3 and x.shape[1:] == (100, 10), x.shape
        B = x.shape[0]
        h_in = None if hidden is None else np.ascontiguousarray(hidden, dtype=np.float32)
        assert h_in is None or h_in.shape == (B, 2, 128)
        logits = np.zeros((B, 100, 5), np.float32)
        h_out = np.zeros((B, 2, 128), np.float32)
        _ffi.check(self.lib.pv_rnn_forward_p2_window(self.handle, x.ctypes.data, None if h_in is None else h_in.ctypes.data, B,
                                                     logits.ctypes.data, h_out.ctypes.data))
        return logits, h_out

    # ---- device-resident forms (the fused pipeline / benchmark) -----------------------------------------
    def summarize_dev(self, dbatch: "DeviceBatch", params: Params, dout: "DeviceOut", stream: int = 0):
        """asynchronous; every array lives in HBM (see device.py). Counters land in dout.counts."""
        cp = params.as_c()
        _ffi.check(self.lib.pv_summarize_regions_dev(
            self.handle, C.byref(dbatch.c), C.byref(cp), dbatch.n_reads, dbatch.n_bases, dbatch.n_cigar,
            dbatch.n_ref_bytes, dbatch.max_region_len, C.byref(dout.c), dout.counts.data_ptr(), stream or None))

    def upload_batch(self, batch: RegionBatch, stream: int = 0):
        """host batch -> the context's workspace (pv_upload_batch); returns (pv_batch_in with device pointers, totals) for
        summarize_uploaded. Valid until the next upload on this context."""
        cin = batch.as_c()
        dev = _ffi.pv_batch_in()
        totals = (C.c_int64 * 4)()
        _ffi.check(self.lib.pv_upload_batch(self.handle, C.byref(cin), C.byref(dev), totals, stream or None))
        return dev, [int(v) for v in totals], cin   # cin keeps the host arrays referenced until the copies are done

    def upload_batches(self, batches, stream: int = 0):
        """several host batches (e.g. one per interval) laid end to end on the device without a host-side concatenation
        (pv_upload_batches); same return value as upload_batch"""
        cins = [b.as_c() for b in batches if b.n_regions]
        arr = (C.POINTER(_ffi.pv_batch_in) * max(len(cins), 1))(*[C.pointer(c) for c in cins])
        dev = _ffi.pv_batch_in()
        totals = (C.c_int64 * 4)()
        _ffi.check(self.lib.pv_upload_batches(self.handle, len(cins), arr, C.byref(dev), totals, stream or None))
        return dev, [int(v) for v in totals], (cins, arr)

    def summarize_uploaded(self, uploaded, params: Params, dout: "DeviceOut", max_region_len: int = 0, stream: int = 0):
        """the device-resident builder on a batch staged with upload_batch; counters land in dout.counts"""
        dev, (n_reads, n_bases, n_cigar, n_ref), _ = uploaded
        cp = params.as_c()
        _ffi.check(self.lib.pv_summarize_regions_dev(self.handle, C.byref(dev), C.byref(cp), n_reads, n_bases, n_cigar, n_ref,
                                                     int(max_region_len), C.byref(dout.c), dout.counts.data_ptr(), stream or None))

    def summarize_hp_dev(self, dbatch: "DeviceBatch", params: Params, dout: "DeviceOut", stream: int = 0):
        """asynchronous, device-resident haplotag-aware builder: dout.images must be an int8 [capacity,21,48] tensor"""
        cp = params.as_c()
        assert tuple(dout.images.shape[1:]) == (_ffi.PV_HP_WINDOW_ROWS, _ffi.PV_HP_FEATURES), dout.images.shape
        _ffi.check(self.lib.pv_summarize_regions_hp_dev(
            self.handle, C.byref(dbatch.c), None if dbatch.read_hp is None else dbatch.read_hp.data_ptr(), C.byref(cp),
            dbatch.n_reads, dbatch.n_bases, dbatch.n_cigar, dbatch.n_ref_bytes, C.byref(dout.c), dout.counts.data_ptr(),
            stream or None))

    def polish_summarize(self, batch: RegionBatch, seq_length: int = 1000, seq_overlap: int = 50, want_flat: bool = False):
        """Polisher (P2) SummaryGenerator.generate_summary + chunk_images for a batch (host buffers), see polish_summary.py."""
        from .polish_summary import polish_summarize
        return polish_summarize(self, batch, seq_length, seq_overlap, want_flat)

    def polish_summarize_dev(self, dbatch: "DeviceBatch", dout: "DevicePolishOut", stream: int = 0):
        """asynchronous, device-resident: chunks land in dout.images ([capacity, seq_length, 10] uint8 in HBM), ready
        for forward_p2_dev; counters {n_chunks, n_rows, status, insert rows} in dout.counts."""
        _ffi.check(self.lib.pv_polish_summarize_regions_dev(
            self.handle, C.byref(dbatch.c), dbatch.n_reads, dbatch.n_bases, dbatch.n_cigar, dbatch.n_ref_bytes,
            dout.seq_length, dout.seq_overlap, C.byref(dout.c), dout.counts.data_ptr(), stream or None))

    def profile_begin(self, only: str = None):
        """bracket every kernel launch of this context with HIP events (only: just the kernels whose profile name starts
        with it - two events per launch put a few microseconds between kernels)"""
        if only:
            _ffi.check(self.lib.pv_profile_begin_only(self.handle, only.encode()))
        else:
            _ffi.check(self.lib.pv_profile_begin(self.handle))

    def profile_end(self) -> dict:
        """-> {kernel name: (total ms, launches)} measured with HIP events on the launch stream"""
        buf = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        cnt = (C.c_int * 64)()
        n = self.lib.pv_profile_end(self.handle, buf, 4096, ms, cnt, 64)
        if n < 0:
            _ffi.check(n)
        names = buf.value.decode().split("\n") if n else []
        return {names[i]: (float(ms[i]), int(cnt[i])) for i in range(n)}
