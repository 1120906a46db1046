"""TEST INFRASTRUCTURE ONLY — ctypes access to the CPU oracle (oracle/liboracle.so, built from
region_summary_oracle.c) and, where it has been built in this container, the reference's own image
builder (oracle/_ref/libref_region_summary.so, built by oracle/Makefile from the sources under
/root/reference). Neither is ever used by the product path."""
import ctypes as C
import os
import subprocess

from pepper_thesis_amd import _ffi
from pepper_thesis_amd.batch import Params, RegionBatch, run_flat_summarizer

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_region_summary.so")
REF_HP_SO = os.path.join(_HERE, "_ref", "libref_region_summary_hp.so")

_SIG = [C.POINTER(_ffi.pv_batch_in), C.POINTER(_ffi.pv_params), C.POINTER(_ffi.pv_batch_out)]
_libs = {}


_made = False


def build(force=False):
    """(re)build liboracle.so and, if /root/reference exists, oracle/_ref (make decides what is stale)."""
    global _made
    if force or not _made:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
        _made = True


_SIG_HP = [_SIG[0], C.POINTER(C.c_int32), _SIG[1], _SIG[2]]


def _load(path, sym, sig=None):
    key = (path, sym)
    if key not in _libs:
        lib = C.CDLL(path)
        fn = getattr(lib, sym)
        fn.restype = C.c_int
        fn.argtypes = sig or _SIG
        _libs[key] = fn
    return _libs[key]


def have_reference():
    return os.path.exists(REF_SO)


def summarize(batch: RegionBatch, params: Params, want_i32=False):
    """CPU restatement (oracle) of generate_summary over a batch."""
    build()
    rc, out = run_flat_summarizer(_load(ORACLE_SO, "oracle_summarize_regions"), batch, params, want_i32)
    if rc:
        raise RuntimeError("oracle_summarize_regions failed: %d" % rc)
    return out


def reference_summarize(batch: RegionBatch, params: Params, want_i32=False):
    """The reference's own region_summary.cpp (only where oracle/_ref was built)."""
    rc, out = run_flat_summarizer(_load(REF_SO, "ref_summarize_regions"), batch, params, want_i32)
    if rc:
        raise RuntimeError("ref_summarize_regions failed: %d" % rc)
    return out


def have_reference_hp():
    return os.path.exists(REF_HP_SO)


def summarize_hp(batch: RegionBatch, params: Params, want_i32=False):
    """CPU restatement of RegionalSummaryGeneratorHP.generate_summary (region_summary_hp_oracle.c)."""
    build()
    rc, out = run_flat_summarizer(_load(ORACLE_SO, "oracle_summarize_regions_hp", _SIG_HP), batch, params, want_i32, hp=True)
    if rc:
        raise RuntimeError("oracle_summarize_regions_hp failed: %d" % rc)
    return out


def reference_summarize_hp(batch: RegionBatch, params: Params, want_i32=False):
    """The reference's own region_summary_hp.cpp (only where oracle/_ref was built)."""
    rc, out = run_flat_summarizer(_load(REF_HP_SO, "ref_summarize_regions_hp", _SIG_HP), batch, params, want_i32, hp=True)
    if rc:
        raise RuntimeError("ref_summarize_regions_hp failed: %d" % rc)
    return out


def polish_summarize(batch: RegionBatch, seq_length=1000, seq_overlap=50, want_flat=True):
    """CPU restatement of the polisher's SummaryGenerator.generate_summary + chunk_images (PARITY UNPINNED, see
    polish_summary_oracle.c)."""
    from pepper_thesis_amd.polish_summary import run_polish_summarizer
    build()
    key = (ORACLE_SO, "oracle_polish_summarize_regions")
    if key not in _libs:
        fn = getattr(C.CDLL(ORACLE_SO), key[1])
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(_ffi.pv_batch_in), C.c_int, C.c_int, C.POINTER(_ffi.pv_polish_out)]
        _libs[key] = fn
    rc, out = run_polish_summarizer(_libs[key], batch, seq_length, seq_overlap, want_flat)
    if rc:
        raise RuntimeError("oracle_polish_summarize_regions failed: %d" % rc)
    return out
