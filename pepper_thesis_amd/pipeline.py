"""The fused call_variant path: BAM -> pileup windows -> genotype probabilities with the windows never leaving HBM.

The reference chains make_images -> run_inference through image HDF5 files on disk (pepper_variant/modules/python/
CallVariant.py:84-104: every window is written as int8 [33,26], read back by dataloader_predict.py:46-78 and copied to the
device batch by batch, predict_distributed_gpu.py:58-69). Here a builder launch chain (`intervals_per_call` intervals) leaves
its windows in a device buffer (pv_summarize_regions_dev) and the P1 network reads that very buffer (pv_rnn_forward_p1_dev);
only the per-window records the consumer needs (contig, position, depth, allele key, allele frequency) and the [n,3]
probabilities come back to the host, where they are written as the reference's prediction file (DataStorePredict.py:49-66:
`predictions/batch_<k>` groups of `batch_size` windows) by a writer thread. Image files are written only on request
(`keep_images_dir`), from a copy of the same device windows.

Ranks: interval i belongs to rank i % world (ImageGenerationUI.py:211) and every rank writes its own
`pepper_prediction_<rank>.hdf` (RunInference.py:101-116); find_candidates globs the directory (FindCandidates.py:145-166).
"""
import os
import queue
import threading
import time
from typing import Optional

import numpy as np

from . import _ffi
from .batch import Params


def _writer_loop(q: "queue.Queue", pred_path: str, image_path: Optional[str], batch_size: int, T: dict, err: list, on_rows=None):
    import contextlib
    from .hdf5io import ImageStore, PredictionStore
    try:
        with (PredictionStore(pred_path, "w") if pred_path else contextlib.nullcontext()) as out:
            img = ImageStore(image_path, "w") if image_path else None
            try:
                batch_no = 0
                keys = ("contigs", "positions", "depths", "candidates", "candidate_frequency")
                carry = None   # windows not yet written: the groups hold exactly `batch_size` windows whatever the calls' sizes were

                def emit(rows, final):
                    nonlocal batch_no
                    n = len(rows["positions"])
                    i = 0
                    while n - i >= batch_size or (final and i < n):
                        sl = slice(i, i + batch_size)
                        out.write_prediction(batch_no, rows["contigs"][sl], rows["positions"][sl], rows["depths"][sl], rows["candidates"][sl],
                                             rows["candidate_frequency"][sl], rows["probs"][sl].astype(np.float64))
                        batch_no += 1
                        i += batch_size
                    return {k: v[i:] for k, v in rows.items()} if i < n else None

                while True:
                    item = q.get()
                    if item is None:
                        if carry is not None:
                            emit(carry, True)
                        break
                    t0 = time.perf_counter()
                    names, rec, probs, images = item
                    rows = {k: rec[k] for k in keys}
                    rows["probs"] = probs if probs is not None else np.zeros((0, 3), np.float32)
                    if on_rows is not None and probs is not None and len(probs):   # the consumer's view of these windows, as it would
                        tr = time.perf_counter()                                      # read them back from the prediction file
                        on_rows(dict(contigs=rec["contigs"], positions=rec["positions"], depths=rec["depths"], candidates=rec["candidates"],
                                     candidate_frequency=rec["candidate_frequency"], base_prediction=probs.astype(np.float64)))
                        T["on_rows_s"] = T.get("on_rows_s", 0.0) + time.perf_counter() - tr
                    if out is not None and len(probs):
                        if carry is not None:
                            w = max(carry["contigs"].dtype.itemsize, rows["contigs"].dtype.itemsize)
                            carry["contigs"], rows["contigs"] = carry["contigs"].astype("S%d" % w), rows["contigs"].astype("S%d" % w)
                            rows = {k: np.concatenate([carry[k], rows[k]]) for k in rows}
                        carry = emit(rows, False)
                    if img is not None:
                        for g, (contig, start, end) in enumerate(names):
                            sel = np.flatnonzero(rec["region"] == g)
                            img.write_summary("%s_%d_%d" % (contig, start, end), [contig] * sel.size, rec["positions"][sel], rec["depths"][sel],
                                              [[rec["candidates"][j, 0]] for j in sel], rec["candidate_frequency"][sel], images[sel])
                    T["hdf5_write_s"] += time.perf_counter() - t0
            finally:
                if img is not None:
                    img.close()
    except BaseException as e:   # surfaced by the producer
        err.append(e)
        while q.get() is not None:   # keep draining so that the producer never blocks on a full queue
            pass


def call_variant_fused(ctx, state_dict: dict, bam_path: str, fasta_path: str, pred_path: str, params: Params, region: str = None,
                       region_size: int = 100_000, min_mapq: int = 5, include_supplementary: bool = False,
                       downsample_rate: float = 1.0, batch_size: int = 512, intervals_per_call: int = 16, rank: int = 0,
                       world: int = 1, reader_threads: int = None, keep_images_path: Optional[str] = None, timers: dict = None,
                       dtype: int = _ffi.PV_DTYPE_F32, region_bed: str = None, inflate_helpers: int = None, on_rows=None) -> int:
    """-> number of windows predicted. One prediction file at `pred_path` (and one image file at `keep_images_path`, if given)
    for the intervals of this rank. state_dict None = images only (make_images): no model, no prediction file.
    on_rows (optional): called on the writer thread with every call's windows as the arrays of a prediction batch (what
    PredictionStore.batches() would read back) - call_variant selects its candidates there, while the device works on."""
    import torch
    from .device import DeviceOut
    from .make_images import region_batches
    from .predict import Predictor
    t_start = time.perf_counter()
    dev = "cuda:%d" % ctx.device_id
    T = dict(upload_s=0.0, device_call_s=0.0, readback_s=0.0, hdf5_write_s=0.0, builder_retries=0)
    # the readers start on the first intervals here; the model is loaded while they read
    batches = region_batches(bam_path, fasta_path, region, region_size, min_mapq, include_supplementary, downsample_rate,
                             intervals_per_call, rank, world, reader_threads, 1, T, region_bed, merge=False,
                             inflate_helpers=inflate_helpers)
    t0 = time.perf_counter()
    predict = state_dict is not None
    if predict:
        try:
            Predictor(ctx, state_dict, "p1", dtype)
        except BaseException:
            batches.close()
            raise
    else:
        assert keep_images_path and not pred_path
    T["load_weights_s"] = time.perf_counter() - t0
    q: "queue.Queue" = queue.Queue(maxsize=4)
    werr: list = []
    if pred_path:
        os.makedirs(os.path.dirname(os.path.abspath(pred_path)), exist_ok=True)
    if keep_images_path:
        os.makedirs(os.path.dirname(os.path.abspath(keep_images_path)), exist_ok=True)
    writer = threading.Thread(target=_writer_loop, args=(q, pred_path, keep_images_path, int(batch_size), T, werr, on_rows), daemon=True)
    writer.start()
    n_windows = 0
    cap, scap = 0, 0
    dout = probs = None
    # the writer thread runs Python (record selection, HDF5 bookkeeping) beside this one, which feeds the GPU between two ctypes
    # calls: with the interpreter's default 5 ms switch interval a launch could wait that long for the GIL
    import sys
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    try:
        for parts, names in batches:
            if werr:
                break
            t0 = time.perf_counter()
            up = ctx.upload_batches(parts)   # the readers' per-interval arrays go straight to their offsets on the device
            T["upload_s"] += time.perf_counter() - t0
            want = max(4096, 1024 * len(names))
            max_region_len = max(b.max_region_len for b in parts)
            while True:
                if dout is None or cap < want:
                    cap, scap = want, 16 * want
                    dout = DeviceOut(cap, scap, dev)
                    probs = torch.zeros((cap, 3), dtype=torch.float32, device=dev)
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.summarize_uploaded(up, params, dout, max_region_len)
                ctx.synchronize(check=False)
                n_out, str_bytes, status = (int(v) for v in dout.counts[:3].tolist())
                if status != _ffi.PV_OK:
                    raise _ffi.PepperHipError(status, "image builder reported status %d for intervals %s..%s" % (status, names[0], names[-1]))
                if n_out > cap or str_bytes > scap:   # more windows than the buffers hold: grow and run the chain again
                    want = max(n_out, (str_bytes + 15) // 16) * 5 // 4
                    T["builder_retries"] += 1
                    T["device_call_s"] += time.perf_counter() - t0
                    continue
                if n_out and predict:
                    ctx.forward_p1_dev(dout.images.data_ptr(), n_out, probs.data_ptr())
                    ctx.synchronize()   # raises if a split-form exchange timed out (the probabilities are then NaN)
                T["device_call_s"] += time.perf_counter() - t0
                break
            t0 = time.perf_counter()
            region_idx = dout.region[:n_out].cpu().numpy()
            offs = dout.cand_off[:n_out + 1].cpu().numpy()
            blob = dout.cand_str[:int(offs[-1]) if n_out else 0].cpu().numpy().tobytes()
            cands = np.empty((n_out, 1), dtype=object)
            for i in range(n_out):
                cands[i, 0] = blob[offs[i]:offs[i + 1]].decode()
            contigs = np.array([names[g][0] for g in region_idx], dtype="S") if n_out else np.zeros(0, dtype="S1")
            rec = dict(region=region_idx, contigs=contigs, positions=dout.position[:n_out].cpu().numpy().astype(np.int32),
                       depths=dout.depth[:n_out].cpu().numpy(), candidates=cands,
                       candidate_frequency=dout.cand_freq[:n_out].cpu().numpy().reshape(-1, 1))
            p = probs[:n_out].cpu().numpy() if predict else None
            imgs = dout.images[:n_out].cpu().numpy() if keep_images_path else None
            T["readback_s"] += time.perf_counter() - t0
            q.put((names, rec, p, imgs))
            n_windows += n_out
            del up
    finally:
        batches.close()   # (stops the readers if the loop was left early)
        q.put(None)
        writer.join()
        sys.setswitchinterval(old_switch)
    if werr:
        raise werr[0]
    T["wall_s"] = time.perf_counter() - t_start
    T["windows"] = n_windows
    if timers is not None:
        timers.update(T)
    return n_windows
