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


IO_SYMBOLS = [
    ("pvio_last_error", C.c_char_p, []),
    ("pvio_bam_open", C.c_void_p, [C.c_char_p]),
    ("pvio_bam_close", None, [C.c_void_p]),
    ("pvio_bam_nref", C.c_int, [C.c_void_p]),
    ("pvio_bam_ref_name", C.c_char_p, [C.c_void_p, C.c_int]),
    ("pvio_bam_ref_len", C.c_int64, [C.c_void_p, C.c_int]),
    ("pvio_bam_get_reads", C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(pvio_reads)]),
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
