"""Flat SoA batches of regions: the host-side data layout handed to the C-ABI.

Replaces the ``list[type_read]`` / ``list[CandidateImageSummary]`` objects that cross the pybind11
boundary in the reference (pepper_variant/modules/cpp/read.h:60-108, region_summary.h:88-111):
one contiguous array per field, offsets instead of nested vectors, so that a batch is uploaded to
HBM with a handful of copies and read by the kernels with coalesced loads.
"""
import ctypes as C
import re
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi

CIGAR_CODES = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6, "=": 7, "X": 8, "B": 9}
_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")


def parse_cigar(text: str) -> np.ndarray:
    """'31M2D47M' -> uint32 BAM-packed ops (len << 4 | op)."""
    ops = [(int(n) << 4) | CIGAR_CODES[c] for n, c in _CIGAR_RE.findall(text)]
    return np.asarray(ops, dtype=np.uint32)


def pack_cigar(pairs: Sequence) -> np.ndarray:
    """[(op, len), ...] (CigarOp order of the reference) -> uint32 BAM-packed ops."""
    return np.asarray([(int(l) << 4) | int(op) for op, l in pairs], dtype=np.uint32)


@dataclass
class Read:
    """One clipped read, the fields of type_read the image builder uses (read.h:60-71)."""
    pos: int
    cigar: np.ndarray            # uint32 packed
    bases: bytes
    quals: np.ndarray            # uint8
    is_reverse: bool = False
    mapq: int = 60
    hp_tag: int = 0              # type_read::hp_tag (the HP aux tag; only the haplotag-aware builder reads it)

    @staticmethod
    def make(pos, cigar, bases, quals=20, is_reverse=False, mapq=60, hp_tag=0):
        cg = parse_cigar(cigar) if isinstance(cigar, str) else np.asarray(cigar, dtype=np.uint32)
        bs = bases.encode() if isinstance(bases, str) else bytes(bases)
        if np.isscalar(quals):
            q = np.full(len(bs), quals, dtype=np.uint8)
        else:
            q = np.asarray(quals, dtype=np.uint8)
        assert len(q) == len(bs)
        return Read(int(pos), cg, bs, q, bool(is_reverse), int(mapq), int(hp_tag))


@dataclass
class Region:
    """Constructor + generate_summary arguments of one RegionalSummaryGenerator
    (region_summary.cpp:9-17, AlignmentSummarizer.py:220-238)."""
    ref_start: int
    ref_end: int                 # inclusive
    ref: bytes                   # len >= ref_end-ref_start+1
    reads: List[Read]
    cand_start: Optional[int] = None
    cand_end: Optional[int] = None
    contig: str = "contig"


@dataclass
class RegionBatch:
    n_regions: int
    ref_start: np.ndarray
    ref_end: np.ndarray
    cand_start: np.ndarray
    cand_end: np.ndarray
    ref_off: np.ndarray
    ref: np.ndarray
    read_off: np.ndarray
    read_pos: np.ndarray
    read_flags: np.ndarray
    read_mapq: np.ndarray
    base_off: np.ndarray
    bases: np.ndarray
    quals: np.ndarray
    cigar_off: np.ndarray
    cigar: np.ndarray
    contigs: List[str] = field(default_factory=list)
    read_hp: Optional[np.ndarray] = None   # int32 [n_reads] hp_tag per read, passed NEXT to pv_batch_in (pv_summarize_regions_hp)

    FIELDS = ("ref_start", "ref_end", "cand_start", "cand_end", "ref_off", "ref", "read_off", "read_pos",
              "read_flags", "read_mapq", "base_off", "bases", "quals", "cigar_off", "cigar")

    @property
    def n_reads(self):
        return int(self.read_pos.shape[0])

    @property
    def n_bases(self):
        return int(self.bases.shape[0])

    @property
    def n_cigar(self):
        return int(self.cigar.shape[0])

    @property
    def max_region_len(self):
        return int((self.ref_end - self.ref_start + 1).max()) if self.n_regions else 0

    def as_c(self) -> _ffi.pv_batch_in:
        s = _ffi.pv_batch_in()
        s.n_regions = self.n_regions
        for f in self.FIELDS:
            setattr(s, f, _ffi.ptr(getattr(self, f)))
        s._keep = self
        return s

    def algorithmic_bytes(self, n_windows: int) -> int:
        """sum_reads(2*len_bases + 4*n_cigar + 16) + R + N*(858+16)."""
        R = int((self.ref_end - self.ref_start + 1).sum())
        return 2 * self.n_bases + 4 * self.n_cigar + 16 * self.n_reads + R + n_windows * (858 + 16)

    def select(self, idx: Sequence[int]) -> "RegionBatch":
        """Sub-batch of the given regions (copies)."""
        regs = [self.region(i) for i in idx]
        return pack_regions(regs)

    def region(self, g: int) -> Region:
        r0, r1 = int(self.read_off[g]), int(self.read_off[g + 1])
        reads = []
        for r in range(r0, r1):
            b0, b1 = int(self.base_off[r]), int(self.base_off[r + 1])
            c0, c1 = int(self.cigar_off[r]), int(self.cigar_off[r + 1])
            reads.append(Read(int(self.read_pos[r]), self.cigar[c0:c1].copy(), self.bases[b0:b1].tobytes(),
                              self.quals[b0:b1].copy(), bool(self.read_flags[r] & 1), int(self.read_mapq[r]),
                              0 if self.read_hp is None else int(self.read_hp[r])))
        return Region(int(self.ref_start[g]), int(self.ref_end[g]),
                      self.ref[int(self.ref_off[g]):int(self.ref_off[g + 1])].tobytes(), reads,
                      int(self.cand_start[g]), int(self.cand_end[g]),
                      self.contigs[g] if self.contigs else "contig")


def pack_regions(regions: Sequence[Region]) -> RegionBatch:
    n = len(regions)
    i64 = np.int64
    ref_start = np.asarray([r.ref_start for r in regions], dtype=i64)
    ref_end = np.asarray([r.ref_end for r in regions], dtype=i64)
    cand_start = np.asarray([r.ref_start if r.cand_start is None else r.cand_start for r in regions], dtype=i64)
    cand_end = np.asarray([r.ref_end if r.cand_end is None else r.cand_end for r in regions], dtype=i64)
    ref_off = np.zeros(n + 1, dtype=i64)
    read_off = np.zeros(n + 1, dtype=i64)
    for g, r in enumerate(regions):
        ref_off[g + 1] = ref_off[g] + len(r.ref)
        read_off[g + 1] = read_off[g] + len(r.reads)
    ref = np.frombuffer(b"".join(r.ref for r in regions), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
    reads = [rd for r in regions for rd in r.reads]
    m = len(reads)
    read_pos = np.asarray([rd.pos for rd in reads], dtype=i64).reshape(m)
    read_flags = np.asarray([1 if rd.is_reverse else 0 for rd in reads], dtype=np.uint8).reshape(m)
    read_mapq = np.asarray([min(max(rd.mapq, 0), 255) for rd in reads], dtype=np.uint8).reshape(m)
    base_off = np.zeros(m + 1, dtype=i64)
    cigar_off = np.zeros(m + 1, dtype=i64)
    if m:
        base_off[1:] = np.cumsum([len(rd.bases) for rd in reads])
        cigar_off[1:] = np.cumsum([len(rd.cigar) for rd in reads])
    bases = np.frombuffer(b"".join(rd.bases for rd in reads), dtype=np.uint8).copy() if m else np.zeros(0, np.uint8)
    quals = np.concatenate([rd.quals for rd in reads]).astype(np.uint8) if m else np.zeros(0, np.uint8)
    cigar = np.concatenate([rd.cigar for rd in reads]).astype(np.uint32) if m else np.zeros(0, np.uint32)
    read_hp = np.asarray([rd.hp_tag for rd in reads], dtype=np.int32).reshape(m)
    return RegionBatch(n, ref_start, ref_end, cand_start, cand_end, ref_off, ref, read_off, read_pos, read_flags,
                       read_mapq, base_off, bases, quals, cigar_off, cigar, [r.contig for r in regions],
                       read_hp if read_hp.any() else None)


def merge_batches(batches: Sequence[RegionBatch]) -> RegionBatch:
    """concatenate region batches (regions keep their order): array concatenation plus offset fix-ups, no per-read work"""
    batches = [b for b in batches if b.n_regions]
    if len(batches) == 1:
        return batches[0]
    if not batches:
        return pack_regions([])

    def cat(f):
        return np.concatenate([getattr(b, f) for b in batches])

    def cat_off(f, totals):
        parts, base = [np.zeros(1, np.int64)], 0
        for b, t in zip(batches, totals):
            parts.append(getattr(b, f)[1:] + base)
            base += t
        return np.concatenate(parts)

    return RegionBatch(
        sum(b.n_regions for b in batches), cat("ref_start"), cat("ref_end"), cat("cand_start"), cat("cand_end"),
        cat_off("ref_off", [int(b.ref.shape[0]) for b in batches]), cat("ref"),
        cat_off("read_off", [b.n_reads for b in batches]), cat("read_pos"), cat("read_flags"), cat("read_mapq"),
        cat_off("base_off", [b.n_bases for b in batches]), cat("bases"), cat("quals"),
        cat_off("cigar_off", [b.n_cigar for b in batches]), cat("cigar"), [c for b in batches for c in b.contigs],
        None if all(b.read_hp is None for b in batches) else np.concatenate(
            [np.zeros(b.n_reads, np.int32) if b.read_hp is None else b.read_hp for b in batches]))


@dataclass
class Params:
    """The per-platform generate_summary scalars (SetParameters.py:12-283)."""
    min_snp_baseq: float = 1
    min_indel_baseq: float = 1
    snp_freq_threshold: float = 0.10
    insert_freq_threshold: float = 0.15
    delete_freq_threshold: float = 0.15
    min_coverage_threshold: float = 3
    snp_candidate_freq_threshold: float = 0.10
    indel_candidate_freq_threshold: float = 0.10
    candidate_support_threshold: float = 2
    skip_indels: bool = False
    candidate_window_size: int = 32
    feature_size: int = 26

    def as_c(self) -> _ffi.pv_params:
        p = _ffi.pv_params()
        for f in ("min_snp_baseq", "min_indel_baseq", "snp_freq_threshold", "insert_freq_threshold",
                  "delete_freq_threshold", "min_coverage_threshold", "snp_candidate_freq_threshold",
                  "indel_candidate_freq_threshold", "candidate_support_threshold"):
            setattr(p, f, float(getattr(self, f)))
        p.skip_indels = 1 if self.skip_indels else 0
        p.candidate_window_size = self.candidate_window_size
        p.feature_size = self.feature_size
        return p


# platform presets: min_snp_baseq, min_indel_baseq, snp_frequency, insert_frequency, delete_frequency,
# min_coverage_threshold, candidate_support_threshold, snp_candidate_frequency_threshold,
# indel_candidate_frequency_threshold, skip_indels  (SetParameters.py:15-37,179-201,...)
PRESETS = {
    "ont_r9_guppy5_sup": Params(1, 1, 0.10, 0.15, 0.15, 3, 0.10, 0.10, 2, False),
    "ont_r9_guppy4_hac": Params(1, 1, 0.10, 0.12, 0.12, 3, 0.10, 0.10, 2, False),
    "ont_r10_q20": Params(1, 1, 0.10, 0.10, 0.10, 3, 0.10, 0.10, 2, False),
    "hifi": Params(10, 10, 0.10, 0.12, 0.10, 2, 0.10, 0.10, 2, False),
    "clr": Params(0, 0, 0.10, 0.12, 0.12, 3, 0.10, 0.12, 2, True),
}


def hp_params(p: Params) -> Params:
    """the same platform scalars with the window geometry of the haplotag-aware builder (ImageSizeOptionsHP, Options.py:17-22)"""
    from dataclasses import replace
    return replace(p, candidate_window_size=_ffi.PV_HP_WINDOW_ROWS - 1, feature_size=_ffi.PV_HP_FEATURES)


# options.min_mapq per platform preset (SetParameters.py:16-17,71-72,126-127,180-181,234-235): BAM_handler.get_reads drops
# reads with MAPQ below it before the image builder sees them
PRESET_MIN_MAPQ = {
    "ont_r9_guppy5_sup": 5,
    "ont_r9_guppy4_hac": 5,
    "ont_r10_q20": 1,
    "hifi": 5,
    "clr": 5,
}


@dataclass
class SummaryOut:
    """Host-side view of pv_batch_out after a call."""
    region: np.ndarray
    position: np.ndarray
    depth: np.ndarray
    cand_freq: np.ndarray
    images: np.ndarray           # int8 [N,33,26] ([N,21,48] from the haplotag-aware builder)
    candidates: List[str]
    images_i32: Optional[np.ndarray] = None

    def __len__(self):
        return int(self.position.shape[0])


class OutBuffers:
    """Caller-owned output arrays + the C struct pointing at them."""

    def __init__(self, capacity: int, str_capacity: int, want_i32: bool = False, rows: int = _ffi.PV_WINDOW_ROWS,
                 features: int = _ffi.PV_FEATURES):
        capacity = max(int(capacity), 1)
        str_capacity = max(int(str_capacity), 1)
        self.capacity, self.str_capacity = capacity, str_capacity
        self.region = np.zeros(capacity, np.int32)
        self.position = np.zeros(capacity, np.int64)
        self.depth = np.zeros(capacity, np.uint8)
        self.cand_freq = np.zeros(capacity, np.uint8)
        self.images = np.zeros((capacity, rows, features), np.int8)
        self.images_i32 = np.zeros((capacity, rows, features), np.int32) if want_i32 else None
        self.cand_str = np.zeros(str_capacity, np.uint8)
        self.cand_off = np.zeros(capacity + 1, np.int64)
        s = _ffi.pv_batch_out()
        s.capacity, s.str_capacity = capacity, str_capacity
        for f in ("region", "position", "depth", "cand_freq", "images", "images_i32", "cand_str", "cand_off"):
            setattr(s, f, _ffi.ptr(getattr(self, f)))
        self.c = s

    def result(self) -> SummaryOut:
        n = int(self.c.n_out)
        raw = self.cand_str.tobytes()
        off = self.cand_off
        cands = [raw[int(off[i]):int(off[i + 1])].decode("latin-1") for i in range(n)]
        return SummaryOut(self.region[:n].copy(), self.position[:n].copy(), self.depth[:n].copy(),
                          self.cand_freq[:n].copy(), self.images[:n].copy(), cands,
                          None if self.images_i32 is None else self.images_i32[:n].copy())


def hp_pointer(batch: RegionBatch):
    """the `read_hp` argument of the *_hp entry points: int32 per read, NULL when no read carries a tag"""
    if batch.read_hp is None:
        return None
    assert batch.read_hp.dtype == np.int32 and batch.read_hp.shape[0] == batch.n_reads
    return batch.read_hp.ctypes.data_as(C.POINTER(C.c_int32))


def run_flat_summarizer(fn, batch: RegionBatch, params: Params, want_i32: bool = False,
                        capacity: int = 4096, str_capacity: int = 1 << 16, ctx=None, hp: bool = False) -> SummaryOut:
    """Call any function with the (pv_batch_in*, pv_params*, pv_batch_out*) signature — or, with hp, the
    (pv_batch_in*, const int32_t* read_hp, pv_params*, pv_batch_out*) one — growing the caller-owned buffers on
    PV_ERR_CAPACITY (the 2-call size query of SURVEY 8b)."""
    cin = batch.as_c()
    cp = params.as_c()
    for _ in range(3):
        if hp:
            ob = OutBuffers(capacity, str_capacity, want_i32, _ffi.PV_HP_WINDOW_ROWS, _ffi.PV_HP_FEATURES)
            args = (C.byref(cin), hp_pointer(batch), C.byref(cp), C.byref(ob.c))
        else:
            ob = OutBuffers(capacity, str_capacity, want_i32)
            args = (C.byref(cin), C.byref(cp), C.byref(ob.c))
        rc = fn(ctx, *args) if ctx is not None else fn(*args)
        if rc == _ffi.PV_ERR_CAPACITY:
            capacity = max(int(ob.c.n_out), capacity)
            str_capacity = max(int(ob.c.str_bytes), str_capacity)
            continue
        return rc, ob.result() if rc == 0 else None
    return rc, None
