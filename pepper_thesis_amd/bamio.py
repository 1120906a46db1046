"""BAM_handler / FASTA_handler look-alikes on the native readers of csrc/pv_io.cpp (include/pepper_io.h).

Reference bindings: pepper_variant/modules/cpp/pybind_api.h:224-235; call sites
AlignmentSummarizer.py:184-189 (get_reads) and :216-218 (get_reference_sequence).
"""
import ctypes as C
import os
from typing import List

import numpy as np

from .batch import Read, Region

_HERE = os.path.dirname(os.path.abspath(__file__))
IO_LIB_PATH = os.path.join(_HERE, "csrc", "libpepper_io.so")


class pvio_reads(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_bases", C.c_int64), ("n_cigar", C.c_int64),
                ("pos", C.POINTER(C.c_int64)), ("pos_end", C.POINTER(C.c_int64)), ("flag", C.POINTER(C.c_uint16)),
                ("is_reverse", C.POINTER(C.c_uint8)), ("mapq", C.POINTER(C.c_uint8)), ("hp_tag", C.POINTER(C.c_int32)),
                ("base_off", C.POINTER(C.c_int64)), ("bases", C.POINTER(C.c_uint8)), ("quals", C.POINTER(C.c_uint8)),
                ("cigar_off", C.POINTER(C.c_int64)), ("cigar", C.POINTER(C.c_uint32)),
                ("name_off", C.POINTER(C.c_int64)), ("names", C.POINTER(C.c_char))]


class pvio_batch(C.Structure):
    _fields_ = [("owner", C.c_void_p), ("n_regions", C.c_int32), ("reserved", C.c_int32),
                ("n_reads", C.c_int64), ("n_bases", C.c_int64), ("n_cigar", C.c_int64), ("n_ref_bytes", C.c_int64),
                ("max_region_len", C.c_int64),
                ("ref_start", C.POINTER(C.c_int64)), ("ref_end", C.POINTER(C.c_int64)), ("cand_start", C.POINTER(C.c_int64)),
                ("cand_end", C.POINTER(C.c_int64)), ("ref_off", C.POINTER(C.c_int64)), ("ref", C.POINTER(C.c_uint8)),
                ("read_off", C.POINTER(C.c_int64)), ("read_pos", C.POINTER(C.c_int64)),
                ("read_flags", C.POINTER(C.c_uint8)), ("read_mapq", C.POINTER(C.c_uint8)),
                ("base_off", C.POINTER(C.c_int64)), ("bases", C.POINTER(C.c_uint8)), ("quals", C.POINTER(C.c_uint8)),
                ("cigar_off", C.POINTER(C.c_int64)), ("cigar", C.POINTER(C.c_uint32)),
                ("interval_index", C.POINTER(C.c_int64)), ("reads_seen", C.POINTER(C.c_int64)),
                ("t_inflate", C.c_double), ("t_total", C.c_double), ("bytes_inflated", C.c_int64),
                ("read_hp", C.POINTER(C.c_int32)), ("t_helpers", C.c_double)]


IO_SYMBOLS = [
    ("pvio_last_error", C.c_char_p, []),
    ("pvio_inflate_backend", C.c_char_p, []),
    ("pvio_set_inflate_backend", C.c_int, [C.c_int]),
    ("pvio_bam_open", C.c_void_p, [C.c_char_p]),
    ("pvio_bam_close", None, [C.c_void_p]),
    ("pvio_bam_set_threads", C.c_int, [C.c_void_p, C.c_int]),
    ("pvio_bam_nref", C.c_int, [C.c_void_p]),
    ("pvio_bam_ref_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("pvio_bam_ref_len", C.c_int64, [C.c_void_p, C.c_int]),
    ("pvio_bam_get_reads", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(pvio_reads)]),
    ("pvio_fill_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.c_int, C.c_int, C.c_int, C.c_double, C.c_int64, C.c_uint32, C.POINTER(C.POINTER(pvio_batch))]),
    ("pvio_batch_free", None, [C.POINTER(pvio_batch)]),
    ("pvio_reservoir_indices", C.c_int64, [C.c_int64, C.c_double, C.c_int64, C.c_uint32, C.POINTER(C.c_int64)]),
    ("pvio_write_bam", C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("pvio_write_vcf_gz", C.c_int, [C.c_char_p, C.c_char_p, C.c_int64]),
    ("pvio_bgzf_read_all", C.c_int64, [C.c_char_p, C.c_char_p, C.c_int64]),
    ("pvio_fasta_open", C.c_void_p, [C.c_char_p]),
    ("pvio_fasta_close", None, [C.c_void_p]),
    ("pvio_fasta_nseq", C.c_int, [C.c_void_p]),
    ("pvio_fasta_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("pvio_fasta_len", C.c_int64, [C.c_void_p, C.c_char_p]),
    ("pvio_fasta_fetch", C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_char_p]),
]
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(IO_LIB_PATH):
            raise ImportError("libpepper_io.so is missing at %s - run __graft_entry__.build()" % IO_LIB_PATH)
        L = C.CDLL(IO_LIB_PATH)
        for name, res, args in IO_SYMBOLS:
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _err():
    m = load().pvio_last_error()
    return m.decode() if m else ""


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class BamHandler:
    """PEPPER_VARIANT.BAM_handler"""

    def __init__(self, path: str):
        self.h = load().pvio_bam_open(path.encode())
        if not self.h:
            raise IOError("BAM_handler: " + _err())

    def close(self):
        if getattr(self, "h", None):
            load().pvio_bam_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_threads(self, n_helpers: int) -> int:
        """n helper threads inflate BGZF blocks ahead of the thread that reads from this handle (0 = none); -> helpers running"""
        rc = load().pvio_bam_set_threads(self.h, int(n_helpers))
        if rc < 0:
            raise IOError("set_threads: " + _err())
        return rc

    def get_chromosome_sequence_names(self) -> List[str]:
        L = load()
        return [L.pvio_bam_ref_name(self.h, i).decode() for i in range(L.pvio_bam_nref(self.h))]

    def get_reads(self, chromosome: str, start: int, stop: int, include_supplementary: bool = False, min_mapq: int = 0,
                  min_baseq: int = 0) -> List[Read]:
        """region-clipped reads in BAM order (the list[type_read] of the reference)"""
        out = pvio_reads()
        rc = load().pvio_bam_get_reads(self.h, chromosome.encode(), int(start), int(stop), int(bool(include_supplementary)),
                                       int(min_mapq), int(min_baseq), C.byref(out))
        if rc:
            raise IOError("get_reads: " + _err())
        n = int(out.n_reads)
        pos = _np(out.pos, n, np.int64)
        pos_end = _np(out.pos_end, n, np.int64)
        rev = _np(out.is_reverse, n, np.uint8)
        mapq = _np(out.mapq, n, np.uint8)
        hp = _np(out.hp_tag, n, np.int32)
        boff = _np(out.base_off, n + 1, np.int64)
        coff = _np(out.cigar_off, n + 1, np.int64)
        noff = _np(out.name_off, n + 1, np.int64)
        bases = _np(out.bases, int(out.n_bases), np.uint8)
        quals = _np(out.quals, int(out.n_bases), np.uint8)
        cigar = _np(out.cigar, int(out.n_cigar), np.uint32)
        names = C.string_at(out.names, int(noff[n])) if n else b""
        reads = []
        for i in range(n):
            r = Read(int(pos[i]), cigar[coff[i]:coff[i + 1]].copy(), bases[boff[i]:boff[i + 1]].tobytes(),
                     quals[boff[i]:boff[i + 1]].copy(), bool(rev[i]), int(mapq[i]))
            r.pos_end = int(pos_end[i])
            r.hp_tag = int(hp[i])
            r.query_name = names[noff[i]:noff[i + 1]].decode()
            reads.append(r)
        return reads


class FastaHandler:
    """PEPPER_VARIANT.FASTA_handler"""

    def __init__(self, path: str):
        self.h = load().pvio_fasta_open(path.encode())
        if not self.h:
            raise IOError("FASTA_handler: " + _err())

    def close(self):
        if getattr(self, "h", None):
            load().pvio_fasta_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def get_chromosome_names(self) -> List[str]:
        L = load()
        return [L.pvio_fasta_name(self.h, i).decode() for i in range(L.pvio_fasta_nseq(self.h))]

    def get_chromosome_sequence_length(self, name: str) -> int:
        return int(load().pvio_fasta_len(self.h, name.encode()))

    def get_reference_sequence(self, contig: str, start: int, stop: int) -> str:
        n = max(int(stop) - int(start), 0)
        buf = C.create_string_buffer(n + 1)
        got = load().pvio_fasta_fetch(self.h, contig.encode(), int(start), int(stop), buf)
        if got < 0:
            raise IOError("get_reference_sequence: " + _err())
        return buf.raw[:got].decode()


def region_from_files(bam: BamHandler, fasta: FastaHandler, contig: str, start: int, end: int, min_mapq: int = 5,
                      include_supplementary: bool = False, downsample_rate: float = 1.0) -> Region:
    """the inference branch of AlignmentSummarizer.create_summary (AlignmentSummarizer.py:180-218) up to the
    builder call: safe bases, get_reads, reservoir down-sampling, reference fetch"""
    from .make_images import downsample_indices, interval_arithmetic
    rs, re_, cs, ce = interval_arithmetic(start, end)
    reads = bam.get_reads(contig, rs, re_, include_supplementary, min_mapq, 0)
    keep = downsample_indices(len(reads), downsample_rate)
    reads = [reads[i] for i in keep]
    ref = fasta.get_reference_sequence(contig, rs, re_ + 1)
    re_ = rs + len(ref) - 1  # the FASTA clamps at the contig end
    return Region(rs, re_, ref.encode(), reads, cs, min(ce, re_), contig)


# ---- whole batches straight into the flat layout (no per-read Python objects) -------------------------------------------

MAX_READS_IN_REGION = 5000       # AlingerOptions.MAX_READS_IN_REGION (Options.py)
RANDOM_SEED = 2719747673         # AlingerOptions.RANDOM_SEED


class FilledBatch:
    """A RegionBatch whose arrays are zero-copy views of a pvio_batch (freed when this object goes away) plus the
    bookkeeping of the fill: interval_index[g] = input interval of batch region g, stage timers."""

    def __init__(self, ptr, intervals):
        from .batch import RegionBatch
        self._ptr = ptr
        v = ptr.contents
        self.intervals = list(intervals)
        G, n, nb, nc = int(v.n_regions), int(v.n_reads), int(v.n_bases), int(v.n_cigar)

        def view(p, k, dtype):
            return np.ctypeslib.as_array(p, shape=(k,)) if k else np.zeros(0, dtype)

        self.interval_index = view(v.interval_index, G, np.int64).copy()
        self.reads_seen = view(v.reads_seen, G, np.int64).copy()
        self.t_inflate, self.t_total, self.bytes_inflated = float(v.t_inflate), float(v.t_total), int(v.bytes_inflated)
        self.t_helpers = float(v.t_helpers)
        self.batch = RegionBatch(
            G, view(v.ref_start, G, np.int64), view(v.ref_end, G, np.int64), view(v.cand_start, G, np.int64),
            view(v.cand_end, G, np.int64), view(v.ref_off, G + 1, np.int64), view(v.ref, int(v.n_ref_bytes), np.uint8),
            view(v.read_off, G + 1, np.int64), view(v.read_pos, n, np.int64), view(v.read_flags, n, np.uint8),
            view(v.read_mapq, n, np.uint8), view(v.base_off, n + 1, np.int64), view(v.bases, nb, np.uint8),
            view(v.quals, nb, np.uint8), view(v.cigar_off, n + 1, np.int64), view(v.cigar, nc, np.uint32),
            [self.intervals[int(i)][0] for i in self.interval_index])
        hp = view(v.read_hp, n, np.int32)
        self.batch.read_hp = hp if hp.any() else None   # type_read::hp_tag, read only by the haplotag-aware builder

    def close(self):
        if getattr(self, "_ptr", None) is not None:
            self.batch = None
            load().pvio_batch_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fill_batch(bam: BamHandler, fasta: FastaHandler, intervals, min_mapq: int = 5, include_supplementary: bool = False,
               downsample_rate: float = 1.0, safe_bases: int = 100) -> FilledBatch:
    """AlignmentSummarizer.create_summary's fetch (AlignmentSummarizer.py:180-218) for a list of (contig, start, end)
    intervals in ONE native call that releases the GIL (reader threads run it beside the GPU launches)."""
    n = len(intervals)
    names = (C.c_char_p * max(n, 1))(*[iv[0].encode() for iv in intervals])
    starts = (C.c_int64 * max(n, 1))(*[int(iv[1]) for iv in intervals])
    ends = (C.c_int64 * max(n, 1))(*[int(iv[2]) for iv in intervals])
    out = C.POINTER(pvio_batch)()
    rc = load().pvio_fill_batch(bam.h, fasta.h, n, names, starts, ends, int(safe_bases), int(bool(include_supplementary)),
                                int(min_mapq), float(downsample_rate), MAX_READS_IN_REGION, RANDOM_SEED, C.byref(out))
    if rc:
        raise IOError("fill_batch: " + _err())
    return FilledBatch(out, intervals)


def reservoir_indices(n_reads: int, downsample_rate: float = 1.0, max_reads: int = MAX_READS_IN_REGION,
                      seed: int = RANDOM_SEED) -> np.ndarray:
    out = np.zeros(max(int(n_reads), 1), np.int64)
    k = load().pvio_reservoir_indices(int(n_reads), float(downsample_rate), int(max_reads), int(seed),
                                      out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:k]


def write_bam(path: str, refs, read_tid, batch, level: int = 1):
    """coordinate-sorted BAM + BAI from the flat arrays of a RegionBatch-like object (read_pos, read_flags, read_mapq,
    base_off, bases, quals, cigar_off, cigar); refs = [(name, length)]; read_tid int32 per read"""
    names = (C.c_char_p * len(refs))(*[r[0].encode() for r in refs])
    lens = (C.c_int64 * len(refs))(*[int(r[1]) for r in refs])
    tid = np.ascontiguousarray(read_tid, dtype=np.int32)
    arrs = [np.ascontiguousarray(getattr(batch, f)) for f in ("read_pos", "read_flags", "read_mapq", "base_off", "bases", "quals",
                                                                "cigar_off", "cigar")]
    rc = load().pvio_write_bam(path.encode(), len(refs), names, lens, int(tid.shape[0]), tid.ctypes.data,
                               *[a.ctypes.data for a in arrs], int(level))
    if rc:
        raise IOError("write_bam: " + _err())


def write_vcf_gz(path: str, text: str):
    """bgzip + tabix (.tbi) of a VCF text, the reference's output form (VcfWriter.py:21-46)"""
    raw = text.encode()
    if load().pvio_write_vcf_gz(path.encode(), raw, len(raw)):
        raise IOError("write_vcf_gz: " + _err())


def bgzf_read_all(path: str) -> bytes:
    n = load().pvio_bgzf_read_all(path.encode(), None, 0)
    if n < 0:
        raise IOError("bgzf_read_all: " + _err())
    buf = C.create_string_buffer(max(int(n), 1))
    n = load().pvio_bgzf_read_all(path.encode(), buf, int(n))
    if n < 0:
        raise IOError("bgzf_read_all: " + _err())
    return buf.raw[:n]


def inflate_backend() -> str:
    """"libdeflate" or "zlib": what inflates BGZF blocks in this process (include/pepper_io.h)"""
    return load().pvio_inflate_backend().decode()


def set_inflate_backend(use_libdeflate: bool) -> bool:
    return bool(load().pvio_set_inflate_backend(1 if use_libdeflate else 0))
