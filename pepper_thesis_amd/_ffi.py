"""ctypes binding of the C-ABI declared in include/pepper_hip.h.

This is the only place the shared library is loaded. There is no CPU fallback: if
``libpepper_hip.so`` has not been built (``__graft_entry__.build()`` / ``pepper_thesis_amd.build``)
the import of the library fails loudly, and ``pv_create`` fails loudly when no HIP device exists.

Replaces the pybind11 module import ``from pepper_variant.build import PEPPER_VARIANT``
(reference: pepper_variant/modules/python/AlignmentSummarizer.py:1, pybind_api.h:24-279).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PEPPER_HIP_LIB") or os.path.join(_HERE, "csrc", "libpepper_hip.so")  # env override: A/B builds

c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)
c_i8_p = C.POINTER(C.c_int8)
c_f32_p = C.POINTER(C.c_float)

PV_OK = 0
PV_ERR_INVALID = -1
PV_ERR_NO_DEVICE = -2
PV_ERR_HIP = -3
PV_ERR_CAPACITY = -4
PV_ERR_LIMIT = -5
PV_ERR_STATE = -6

PV_WINDOW_ROWS = 33
PV_FEATURES = 26
PV_WINDOW_BYTES = PV_WINDOW_ROWS * PV_FEATURES
PV_HP_WINDOW_ROWS = 21
PV_HP_FEATURES = 48
PV_HP_WINDOW_BYTES = PV_HP_WINDOW_ROWS * PV_HP_FEATURES

PV_PLAN_P1_LSTM = 1
PV_PLAN_P2_GRU = 2
PV_DTYPE_F32 = 0
PV_DTYPE_BF16_INPUT_GEMM = 1


class pv_params(C.Structure):
    _fields_ = [
        ("min_snp_baseq", C.c_double),
        ("min_indel_baseq", C.c_double),
        ("snp_freq_threshold", C.c_double),
        ("insert_freq_threshold", C.c_double),
        ("delete_freq_threshold", C.c_double),
        ("min_coverage_threshold", C.c_double),
        ("snp_candidate_freq_threshold", C.c_double),
        ("indel_candidate_freq_threshold", C.c_double),
        ("candidate_support_threshold", C.c_double),
        ("skip_indels", C.c_int32),
        ("candidate_window_size", C.c_int32),
        ("feature_size", C.c_int32),
        ("reserved", C.c_int32),
    ]


class pv_batch_in(C.Structure):
    _fields_ = [
        ("n_regions", C.c_int32),
        ("reserved", C.c_int32),
        ("ref_start", C.c_void_p),
        ("ref_end", C.c_void_p),
        ("cand_start", C.c_void_p),
        ("cand_end", C.c_void_p),
        ("ref_off", C.c_void_p),
        ("ref", C.c_void_p),
        ("read_off", C.c_void_p),
        ("read_pos", C.c_void_p),
        ("read_flags", C.c_void_p),
        ("read_mapq", C.c_void_p),
        ("base_off", C.c_void_p),
        ("bases", C.c_void_p),
        ("quals", C.c_void_p),
        ("cigar_off", C.c_void_p),
        ("cigar", C.c_void_p),
    ]


class pv_batch_out(C.Structure):
    _fields_ = [
        ("capacity", C.c_int64),
        ("str_capacity", C.c_int64),
        ("region", C.c_void_p),
        ("position", C.c_void_p),
        ("depth", C.c_void_p),
        ("cand_freq", C.c_void_p),
        ("images", C.c_void_p),
        ("images_i32", C.c_void_p),
        ("cand_str", C.c_void_p),
        ("cand_off", C.c_void_p),
        ("n_out", C.c_int64),
        ("str_bytes", C.c_int64),
    ]


class pv_polish_out(C.Structure):
    _fields_ = [
        ("chunk_capacity", C.c_int64),
        ("row_capacity", C.c_int64),
        ("images", C.c_void_p),
        ("position", C.c_void_p),
        ("index", C.c_void_p),
        ("region", C.c_void_p),
        ("chunk_id", C.c_void_p),
        ("flat_images", C.c_void_p),
        ("flat_position", C.c_void_p),
        ("flat_index", C.c_void_p),
        ("region_row_off", C.c_void_p),
        ("n_chunks", C.c_int64),
        ("n_rows", C.c_int64),
    ]


class pv_rnn_dir(C.Structure):
    _fields_ = [("w_ih", C.c_void_p), ("w_hh", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p)]


class pv_weights_p1(C.Structure):
    _fields_ = [
        ("encoder", pv_rnn_dir * 2),
        ("decoder", pv_rnn_dir * 2),
        ("linear_w", C.c_void_p * 5),
        ("linear_b", C.c_void_p * 5),
        ("out_w", C.c_void_p),
        ("out_b", C.c_void_p),
    ]


class pv_weights_p2(C.Structure):
    _fields_ = [
        ("encoder", pv_rnn_dir * 2),
        ("decoder", pv_rnn_dir * 2),
        ("dense_w", C.c_void_p),
        ("dense_b", C.c_void_p),
    ]


# every symbol include/pepper_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("pv_create", C.c_void_p, [C.c_int]),
    ("pv_destroy", None, [C.c_void_p]),
    ("pv_last_error", C.c_char_p, []),
    ("pv_stream", C.c_void_p, [C.c_void_p]),
    ("pv_synchronize", C.c_int, [C.c_void_p]),
    ("pv_summarize_regions", C.c_int, [C.c_void_p, C.POINTER(pv_batch_in), C.POINTER(pv_params), C.POINTER(pv_batch_out)]),
    ("pv_summarize_regions_dev", C.c_int,
     [C.c_void_p, C.POINTER(pv_batch_in), C.POINTER(pv_params), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
      C.POINTER(pv_batch_out), C.c_void_p, C.c_void_p]),
    ("pv_upload_batch", C.c_int, [C.c_void_p, C.POINTER(pv_batch_in), C.POINTER(pv_batch_in), C.POINTER(C.c_int64), C.c_void_p]),
    ("pv_upload_batches", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.POINTER(pv_batch_in)), C.POINTER(pv_batch_in), C.POINTER(C.c_int64), C.c_void_p]),
    ("pv_summarize_regions_hp", C.c_int,
     [C.c_void_p, C.POINTER(pv_batch_in), C.POINTER(C.c_int32), C.POINTER(pv_params), C.POINTER(pv_batch_out)]),
    ("pv_summarize_regions_hp_dev", C.c_int,
     [C.c_void_p, C.POINTER(pv_batch_in), C.c_void_p, C.POINTER(pv_params), C.c_int64, C.c_int64, C.c_int64, C.c_int64,
      C.POINTER(pv_batch_out), C.c_void_p, C.c_void_p]),
    ("pv_polish_summarize_regions", C.c_int, [C.c_void_p, C.POINTER(pv_batch_in), C.c_int, C.c_int, C.POINTER(pv_polish_out)]),
    ("pv_polish_summarize_regions_dev", C.c_int,
     [C.c_void_p, C.POINTER(pv_batch_in), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
      C.POINTER(pv_polish_out), C.c_void_p, C.c_void_p]),
    ("pv_rnn_load_p1", C.c_int, [C.c_void_p, C.POINTER(pv_weights_p1), C.c_int]),
    ("pv_rnn_load_p2", C.c_int, [C.c_void_p, C.POINTER(pv_weights_p2), C.c_int]),
    ("pv_rnn_forward_p1", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("pv_rnn_forward_p1_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("pv_rnn_forward_p1_debug", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pv_rnn_forward_p2", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("pv_rnn_forward_p2_dev", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("pv_rnn_forward_p2_window", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    ("pv_comm_unique_id", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pv_comm_create", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("pv_comm_destroy", None, [C.c_void_p]),
    ("pv_gather", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_int, C.c_void_p]),
    ("pv_gather_counts", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    ("pv_debug_gemm_bf16x3", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.POINTER(C.c_float)]),
    ("pv_profile_begin", C.c_int, [C.c_void_p]),
    ("pv_profile_begin_only", C.c_int, [C.c_void_p, C.c_char_p]),
    ("pv_profile_end", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
    ("pv_rnn_exchange_timeouts", C.c_int, [C.c_void_p]),
    ("pv_set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("pv_get_option", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    ("pv_graph_begin", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pv_graph_end", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    ("pv_graph_launch", C.c_int, [C.c_void_p, C.c_void_p]),
    ("pv_graph_destroy", None, [C.c_void_p]),
    ("pv_workspace_bytes", C.c_int64, [C.c_void_p]),
    ("pv_version", C.c_int, []),
]

_lib = None


class PepperHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pepper_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libpepper_hip.so (once) and set prototypes. Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libpepper_hip.so is missing at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        # PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64. Load them FIRST so that this library's
        # DT_NEEDED entries resolve to the copies already in the process: two HIP runtimes in one process cannot both
        # open the device (torch then reports "No HIP GPUs are available" when it initialises second).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != PV_OK:
        msg = load().pv_last_error()
        raise PepperHipError(code, msg.decode() if msg else "")
    return code


def ptr(a):
    """address of a numpy array (must be C-contiguous) or 0 for None"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data
