"""P2 (polisher) summary images on the GPU, behind the reference's own interface.

Host-side mirror of ``PEPPER.SummaryGenerator`` (pybind11, pepper/modules/src/pileup_summary/summary_generator.cpp:6-12,
371-392) and ``AlignmentSummarizer.chunk_images`` (pepper/modules/python/AlignmentSummarizer.py:19-56). All arithmetic
runs in the HIP kernels of csrc/summary_kernels.hip (``k_polish_*``) through ``pv_polish_summarize_regions``; nothing
here computes pixels.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .batch import Read, Region, RegionBatch, pack_regions

IMAGE_HEIGHT = 10    # ImageSizeOptions.IMAGE_HEIGHT (pepper/modules/python/Options.py:2)
SEQ_LENGTH = 1000    # ImageSizeOptions.SEQ_LENGTH
SEQ_OVERLAP = 50     # ImageSizeOptions.SEQ_OVERLAP


@dataclass
class PolishOut:
    """Chunked images of a batch (what chunk_images returns, as arrays) and, optionally, the un-chunked rows."""
    images: np.ndarray       # [n_chunks, seq_length, 10] uint8
    position: np.ndarray     # [n_chunks, seq_length] int64, -1 on padding rows
    index: np.ndarray        # [n_chunks, seq_length] int32, -1 on padding rows
    region: np.ndarray       # [n_chunks] int32
    chunk_id: np.ndarray     # [n_chunks] int32
    flat_images: Optional[np.ndarray] = None     # [n_rows, 10]
    flat_position: Optional[np.ndarray] = None
    flat_index: Optional[np.ndarray] = None
    region_row_off: Optional[np.ndarray] = None  # [n_regions + 1]


class PolishBuffers:
    """Caller-owned host buffers of one pv_polish_out."""

    def __init__(self, n_regions, chunk_capacity, row_capacity, seq_length, want_flat):
        self.seq_length = seq_length
        self.images = np.zeros((chunk_capacity, seq_length, IMAGE_HEIGHT), dtype=np.uint8)
        self.position = np.zeros((chunk_capacity, seq_length), dtype=np.int64)
        self.index = np.zeros((chunk_capacity, seq_length), dtype=np.int32)
        self.region = np.zeros(chunk_capacity, dtype=np.int32)
        self.chunk_id = np.zeros(chunk_capacity, dtype=np.int32)
        self.flat_images = self.flat_position = self.flat_index = self.region_row_off = None
        if want_flat:
            self.flat_images = np.zeros((row_capacity, IMAGE_HEIGHT), dtype=np.uint8)
            self.flat_position = np.zeros(row_capacity, dtype=np.int64)
            self.flat_index = np.zeros(row_capacity, dtype=np.int32)
            self.region_row_off = np.zeros(n_regions + 1, dtype=np.int64)
        c = _ffi.pv_polish_out()
        c.chunk_capacity = chunk_capacity
        c.row_capacity = row_capacity if want_flat else 0
        for f in ("images", "position", "index", "region", "chunk_id", "flat_images", "flat_position", "flat_index",
                  "region_row_off"):
            setattr(c, f, _ffi.ptr(getattr(self, f)))
        self.c = c

    def result(self) -> PolishOut:
        n, r = int(self.c.n_chunks), int(self.c.n_rows)
        out = PolishOut(self.images[:n].copy(), self.position[:n].copy(), self.index[:n].copy(), self.region[:n].copy(),
                        self.chunk_id[:n].copy())
        if self.flat_images is not None:
            out.flat_images = self.flat_images[:r].copy()
            out.flat_position = self.flat_position[:r].copy()
            out.flat_index = self.flat_index[:r].copy()
            out.region_row_off = self.region_row_off.copy()
        return out


def run_polish_summarizer(fn, batch: RegionBatch, seq_length=SEQ_LENGTH, seq_overlap=SEQ_OVERLAP, want_flat=False, ctx=None):
    """Call any function with pv_polish_summarize_regions' signature (minus the context when ctx is None), growing the
    caller-owned buffers on PV_ERR_CAPACITY. Returns (rc, PolishOut or None)."""
    cin = batch.as_c()
    cols = int((batch.ref_end - batch.ref_start + 1).sum()) if batch.n_regions else 0
    rows = cols + cols // 2 + 1024
    chunks = rows // max(1, seq_length - seq_overlap) + 2 * batch.n_regions + 2
    rc = 0
    for _ in range(3):
        pb = PolishBuffers(batch.n_regions, chunks, rows, seq_length, want_flat)
        args = (C.byref(cin), int(seq_length), int(seq_overlap), C.byref(pb.c))
        rc = fn(ctx, *args) if ctx is not None else fn(*args)
        if rc == _ffi.PV_ERR_CAPACITY:
            chunks, rows = max(chunks, int(pb.c.n_chunks)), max(rows, int(pb.c.n_rows))
            continue
        return rc, (pb.result() if rc == 0 else None)
    return rc, None


def polish_summarize(ctx, batch: RegionBatch, seq_length=SEQ_LENGTH, seq_overlap=SEQ_OVERLAP, want_flat=False) -> PolishOut:
    """generate_summary + chunk_images for every region of the batch on ctx's GPU (host buffers in and out)."""
    rc, out = run_polish_summarizer(ctx.lib.pv_polish_summarize_regions, batch, seq_length, seq_overlap, want_flat, ctx.handle)
    _ffi.check(rc)
    return out


class SummaryGenerator:
    """Same constructor, method and attributes as the pybind11 class (pybind_api.h of pepper/modules: image,
    genomic_pos; labels stay empty outside train mode)."""

    def __init__(self, reference_sequence: str, chromosome_name: str, ref_start: int, ref_end: int, ctx=None):
        from .runtime import Context
        self.reference_sequence = reference_sequence
        self.chromosome_name = chromosome_name
        self.ref_start = int(ref_start)
        self.ref_end = int(ref_end)
        self.ctx = ctx if ctx is not None else Context(0)
        self.image = np.zeros((0, IMAGE_HEIGHT), dtype=np.uint8)
        self.genomic_pos: List[tuple] = []
        self.labels: List[int] = []
        self.bad_label_positions: List[int] = []

    def generate_summary(self, reads: Sequence[Read], start_pos: int, end_pos: int):
        """summary_generator.cpp:371-392. start_pos/end_pos must be the constructor's ref_start/ref_end, which is how
        the reference's only caller uses it (AlignmentSummarizer.py:343-350)."""
        if int(start_pos) != self.ref_start or int(end_pos) != self.ref_end:
            raise ValueError("generate_summary is only defined for start_pos == ref_start and end_pos == ref_end")
        R = self.ref_end - self.ref_start + 1
        ref = self.reference_sequence.encode() if isinstance(self.reference_sequence, str) else bytes(self.reference_sequence)
        if len(ref) < R:
            ref = ref + b"N" * (R - len(ref))  # the polisher never reads the reference bytes
        batch = pack_regions([Region(self.ref_start, self.ref_end, ref, list(reads), contig=self.chromosome_name)])
        out = polish_summarize(self.ctx, batch, SEQ_LENGTH, SEQ_OVERLAP, want_flat=True)
        self.image = out.flat_images
        self.genomic_pos = list(zip(out.flat_position.tolist(), out.flat_index.tolist()))
        self._chunks = out
        return self


def chunk_images(summary: SummaryGenerator, chunk_size: int = SEQ_LENGTH, chunk_overlap: int = SEQ_OVERLAP):
    """AlignmentSummarizer.chunk_images (AlignmentSummarizer.py:19-56): (images, labels, positions, chunk_ids). The chunks
    were cut on the GPU by the same call that built the image when the sizes are the defaults."""
    out = getattr(summary, "_chunks", None)
    if out is None or out.images.shape[1] != chunk_size or chunk_overlap != SEQ_OVERLAP:
        raise ValueError("chunk_images: call generate_summary first; sizes other than %d/%d need polish_summarize()"
                         % (SEQ_LENGTH, SEQ_OVERLAP))
    images = [im for im in out.images]
    labels = [[0] * chunk_size for _ in images]
    positions = [list(zip(p.tolist(), i.tolist())) for p, i in zip(out.position, out.index)]
    return images, labels, positions, out.chunk_id.tolist()


MAX_READS_IN_REGION = 1500  # AlingerOptions.MAX_READS_IN_REGION (pepper/modules/python/Options.py:28)


def region_from_files(bam, fasta, contig: str, start: int, end: int) -> Optional[Region]:
    """The inference branch of the polisher's AlignmentSummarizer.create_summary up to the builder call
    (pepper/modules/python/AlignmentSummarizer.py:296-347): get_reads(contig, max(0, start), end, no supplementary, mapq 0,
    baseq 0), reservoir sampling to 1500 reads with RandomState(2719747673), reference fetch [start, end]. The optional SSW
    realignment (`realignment_flag`) is outside this path. None when the region has no reads (the reference returns empty lists)."""
    from .make_images import downsample_indices
    reads = bam.get_reads(contig, max(0, int(start)), int(end), False, 0, 0)
    if not reads:
        return None
    keep = downsample_indices(len(reads), 1.0, MAX_READS_IN_REGION)
    reads = [reads[i] for i in keep]
    R = int(end) - int(start) + 1
    ref = fasta.get_reference_sequence(contig, int(start), int(end) + 1).encode()
    if len(ref) < R:
        ref = ref + b"N" * (R - len(ref))  # past the contig end: columns without reads; the polisher never reads the bytes
    return Region(int(start), int(end), ref, reads, contig=contig)


def polish_regions(ctx, regions: Sequence[Region], want_acc: bool = False):
    """builder -> bi-GRU for a batch of regions: (PolishOut, labels uint8 [n_chunks, 1000][, accumulated softmax]).
    pv_rnn_load_p2 must have been called on ctx. positions/index of PolishOut name the reference position of every label,
    which is what the polisher's prediction file stores per chunk (DataStorePredict.write_prediction)."""
    out = polish_summarize(ctx, pack_regions(list(regions)), SEQ_LENGTH, SEQ_OVERLAP)
    res = ctx.forward_p2(out.images, want_acc=want_acc)
    return (out,) + (res if want_acc else (res,))
