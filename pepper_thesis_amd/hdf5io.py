"""HDF5 image / prediction files in the reference's on-disk format, without h5py.

The reference writes its files with h5py (pepper_variant/modules/python/DataStore.py:54-71,
DataStorePredict.py:49-66) and reads them back in dataloader_predict.py:46-78 and
FindCandidates.py:157-166. h5py is not installed for this interpreter, so this module drives the HDF5
C library itself (libhdf5, present in the image under /opt/conda/lib) through ctypes; files are
therefore real HDF5 written by the same library h5py wraps: contiguous datasets, fixed-length byte
strings for `contigs`, variable-length UTF-8 strings of shape (N,1) for `candidates`
(h5py.special_dtype(vlen=str)), int32/uint8/int8/float64 numerics.

Fails loudly (ImportError) if no libhdf5 can be found; set PEPPER_HDF5_LIB to point at one.
"""
import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Iterator, List, Sequence, Tuple

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_ASCII, H5T_CSET_UTF8 = 0, 1
H5T_STR_NULLTERM, H5T_STR_NULLPAD = 0, 1
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_SGN_NONE = 0

_lib = None


class _H5G_info(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]


def _candidates() -> List[str]:
    c = []
    if os.environ.get("PEPPER_HDF5_LIB"):
        c.append(os.environ["PEPPER_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5")
    if found:
        c.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        c += sorted(p for p in glob.glob(pat) if "_cpp" not in p and "_hl" not in p and "_fortran" not in p)
    return c


def lib():
    global _lib
    if _lib is not None:
        return _lib
    err = None
    for path in _candidates():
        try:
            L = C.CDLL(path)
            L.H5open()
            break
        except OSError as e:  # try the next candidate
            err = e
    else:
        raise ImportError("libhdf5 not found (looked in PEPPER_HDF5_LIB, the linker path, /opt/conda/lib): %r" % (err,))
    hid_fns = ["H5Fcreate", "H5Fopen", "H5Pcreate", "H5Gcreate2", "H5Gopen2", "H5Screate_simple", "H5Tcopy",
               "H5Dcreate2", "H5Dopen2", "H5Dget_space", "H5Dget_type", "H5Tget_native_type"]
    for f in hid_fns:
        getattr(L, f).restype = hid_t
    L.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid_t, hid_t]
    L.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid_t]
    L.H5Fclose.argtypes = [hid_t]
    L.H5Pcreate.argtypes = [hid_t]
    L.H5Pclose.argtypes = [hid_t]
    L.H5Pset_create_intermediate_group.argtypes = [hid_t, C.c_uint]
    L.H5Gcreate2.argtypes = [hid_t, C.c_char_p, hid_t, hid_t, hid_t]
    L.H5Gopen2.argtypes = [hid_t, C.c_char_p, hid_t]
    L.H5Gclose.argtypes = [hid_t]
    L.H5Gget_info.argtypes = [hid_t, C.POINTER(_H5G_info)]
    L.H5Lexists.argtypes = [hid_t, C.c_char_p, hid_t]
    L.H5Lget_name_by_idx.argtypes = [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]
    L.H5Lget_name_by_idx.restype = C.c_ssize_t
    L.H5Screate_simple.argtypes = [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]
    L.H5Sclose.argtypes = [hid_t]
    L.H5Sget_simple_extent_ndims.argtypes = [hid_t]
    L.H5Sget_simple_extent_dims.argtypes = [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]
    L.H5Tcopy.argtypes = [hid_t]
    L.H5Tclose.argtypes = [hid_t]
    L.H5Tset_size.argtypes = [hid_t, C.c_size_t]
    L.H5Tset_cset.argtypes = [hid_t, C.c_int]
    L.H5Tset_strpad.argtypes = [hid_t, C.c_int]
    L.H5Tget_class.argtypes = [hid_t]
    L.H5Tget_size.argtypes = [hid_t]
    L.H5Tget_size.restype = C.c_size_t
    L.H5Tget_sign.argtypes = [hid_t]
    L.H5Tis_variable_str.argtypes = [hid_t]
    L.H5Tget_native_type.argtypes = [hid_t, C.c_int]
    L.H5Dcreate2.argtypes = [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]
    L.H5Dopen2.argtypes = [hid_t, C.c_char_p, hid_t]
    L.H5Dclose.argtypes = [hid_t]
    L.H5Dget_space.argtypes = [hid_t]
    L.H5Dget_type.argtypes = [hid_t]
    L.H5Dwrite.argtypes = [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]
    L.H5Dread.argtypes = [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]
    L.H5Dvlen_reclaim.argtypes = [hid_t, hid_t, hid_t, C.c_void_p]
    _lib = L
    return L


def _g(name: str) -> int:
    return hid_t.in_dll(lib(), name).value


_NATIVE = {np.dtype(np.int32): "H5T_NATIVE_INT32_g", np.dtype(np.uint8): "H5T_NATIVE_UINT8_g",
           np.dtype(np.int8): "H5T_NATIVE_INT8_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
           np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g",
           np.dtype(np.int16): "H5T_NATIVE_INT16_g", np.dtype(np.uint16): "H5T_NATIVE_UINT16_g",
           np.dtype(np.uint32): "H5T_NATIVE_UINT32_g", np.dtype(np.uint64): "H5T_NATIVE_UINT64_g"}


def _chk(rc, what):
    if rc < 0:
        raise IOError("HDF5 call failed: %s" % what)
    return rc


class H5File:
    """minimal h5py.File look-alike: f[path] = array; f.read(path); f.keys(group); 'x' in f"""

    def __init__(self, path: str, mode: str = "r"):
        L = lib()
        self.path, self.mode = path, mode
        if mode == "w":
            self.fid = L.H5Fcreate(path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "r":
            self.fid = L.H5Fopen(path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.fid < 0:
            raise IOError("cannot open %s (mode %s)" % (path, mode))
        self.lcpl = L.H5Pcreate(_g("H5P_CLS_LINK_CREATE_ID_g"))
        L.H5Pset_create_intermediate_group(self.lcpl, 1)

    def close(self):
        if getattr(self, "fid", -1) >= 0:
            lib().H5Pclose(self.lcpl)
            lib().H5Fclose(self.fid)
            self.fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __contains__(self, path: str) -> bool:
        L = lib()
        cur = ""
        for part in path.strip("/").split("/"):
            cur = cur + "/" + part
            if L.H5Lexists(self.fid, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def keys(self, group: str = "/") -> List[str]:
        L = lib()
        gid = _chk(L.H5Gopen2(self.fid, group.encode(), H5P_DEFAULT), "H5Gopen2 " + group)
        info = _H5G_info()
        _chk(L.H5Gget_info(gid, C.byref(info)), "H5Gget_info")
        names = []
        buf = C.create_string_buffer(4096)
        for i in range(int(info.nlinks)):
            n = L.H5Lget_name_by_idx(gid, b".", 0, 0, i, buf, 4096, H5P_DEFAULT)
            _chk(n, "H5Lget_name_by_idx")
            names.append(buf.value.decode())
        L.H5Gclose(gid)
        return names

    # ---- write ---------------------------------------------------------------------------------------
    def __setitem__(self, path: str, value):
        self.write(path, value)

    def write(self, path: str, value, vlen_str: bool = False):
        """numeric / fixed 'S' numpy arrays, or (vlen_str=True) a nested list / object array of str"""
        L = lib()
        if vlen_str:
            arr = np.asarray(value, dtype=object)
            flat = [s if isinstance(s, bytes) else str(s).encode("utf-8") for s in arr.ravel()]
            ptrs = (C.c_char_p * len(flat))(*flat)
            tid = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(tid, H5T_VARIABLE)
            L.H5Tset_cset(tid, H5T_CSET_UTF8)
            L.H5Tset_strpad(tid, H5T_STR_NULLTERM)
            shape, buf, ftid, mtid, own = arr.shape, C.cast(ptrs, C.c_void_p), tid, tid, True
        else:
            arr = np.ascontiguousarray(value)
            if arr.dtype.kind == "S":
                tid = L.H5Tcopy(_g("H5T_C_S1_g"))
                L.H5Tset_size(tid, max(arr.dtype.itemsize, 1))
                L.H5Tset_strpad(tid, H5T_STR_NULLPAD)
                ftid, mtid, own = tid, tid, True
            elif arr.dtype in _NATIVE:
                ftid = mtid = _g(_NATIVE[arr.dtype])
                own = False
            else:
                raise TypeError("unsupported dtype %s" % arr.dtype)
            shape, buf = arr.shape, C.c_void_p(arr.ctypes.data)
        dims = (hsize_t * max(len(shape), 1))(*shape)
        sid = _chk(L.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")
        did = _chk(L.H5Dcreate2(self.fid, path.encode(), ftid, sid, self.lcpl, H5P_DEFAULT, H5P_DEFAULT), "H5Dcreate2 " + path)
        if int(np.prod(shape)) > 0:
            _chk(L.H5Dwrite(did, mtid, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), "H5Dwrite " + path)
        L.H5Dclose(did)
        L.H5Sclose(sid)
        if own:
            L.H5Tclose(ftid)

    # ---- read ------------------------------------------------------------------------------------------
    def read(self, path: str):
        """-> numpy array; variable-length strings come back as an object array of str (decoded UTF-8)"""
        L = lib()
        did = _chk(L.H5Dopen2(self.fid, path.encode(), H5P_DEFAULT), "H5Dopen2 " + path)
        sid = L.H5Dget_space(did)
        nd = L.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(sid, dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        n = int(np.prod(shape)) if nd > 0 else 1
        tid = L.H5Dget_type(did)
        cls = L.H5Tget_class(tid)
        try:
            if cls == H5T_STRING and L.H5Tis_variable_str(tid) > 0:
                ptrs = (C.c_void_p * max(n, 1))()
                if n:
                    _chk(L.H5Dread(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs), "H5Dread " + path)
                out = np.empty(n, dtype=object)
                for i in range(n):
                    out[i] = C.string_at(ptrs[i]).decode("utf-8") if ptrs[i] else ""
                if n:
                    L.H5Dvlen_reclaim(tid, sid, H5P_DEFAULT, ptrs)
                return out.reshape(shape)
            if cls == H5T_STRING:
                size = int(L.H5Tget_size(tid))
                out = np.zeros(shape, dtype="S%d" % size)
                if n:
                    _chk(L.H5Dread(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), "H5Dread " + path)
                return out
            size = int(L.H5Tget_size(tid))
            if cls == H5T_FLOAT:
                dt = {4: np.float32, 8: np.float64}[size]
            elif cls == H5T_INTEGER:
                unsigned = L.H5Tget_sign(tid) == H5T_SGN_NONE
                dt = {(1, False): np.int8, (1, True): np.uint8, (2, False): np.int16, (2, True): np.uint16,
                      (4, False): np.int32, (4, True): np.uint32, (8, False): np.int64, (8, True): np.uint64}[(size, unsigned)]
            else:
                raise TypeError("unsupported HDF5 class %d in %s" % (cls, path))
            out = np.zeros(shape, dtype=dt)
            if n:
                _chk(L.H5Dread(did, _g(_NATIVE[np.dtype(dt)]), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), "H5Dread " + path)
            return out
        finally:
            L.H5Tclose(tid)
            L.H5Sclose(sid)
            L.H5Dclose(did)


# ---- the reference's two stores ------------------------------------------------------------------------

class ImageStore:
    """DataStore (pepper_variant/modules/python/DataStore.py:7-71), inference mode"""
    _summary_path_ = "summaries"

    def __init__(self, filename: str, mode: str = "r"):
        self.f = H5File(filename, mode)
        self._written = set()

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def write_summary(self, summary_name: str, contigs: Sequence, positions, depths, all_candidates, all_candidate_frequency,
                      all_images):
        """shapes/dtypes of DataStore.write_summary: contigs S (N,), positions int32 (N,), depths uint8 (N,),
        candidates vlen-str (N,1), candidate_frequency uint8 (N,1), images int8 (N,33,26)"""
        if summary_name in self._written:
            return
        self._written.add(summary_name)
        base = "%s/%s/" % (self._summary_path_, summary_name)
        if len(positions) == 0:
            # np.array([], dtype=...) of the reference's empty lists: one-dimensional empty datasets
            self.f.write(base + "contigs", np.zeros(0, dtype="S1"))
            self.f.write(base + "positions", np.zeros(0, dtype=np.int32))
            self.f.write(base + "depths", np.zeros(0, dtype=np.uint8))
            self.f.write(base + "candidates", np.zeros(0, dtype=object), vlen_str=True)
            self.f.write(base + "candidate_frequency", np.zeros(0, dtype=np.uint8))
            self.f.write(base + "images", np.zeros(0, dtype=np.int8))
            return
        self.f.write(base + "contigs", np.array([c.encode() if isinstance(c, str) else c for c in contigs], dtype="S"))
        self.f.write(base + "positions", np.asarray(positions, dtype=np.int32))
        self.f.write(base + "depths", np.asarray(depths, dtype=np.uint8))
        cands = np.asarray(all_candidates, dtype=object)
        self.f.write(base + "candidates", cands.reshape(len(cands), -1) if cands.ndim == 1 else cands, vlen_str=True)
        freq = np.asarray(all_candidate_frequency, dtype=np.uint8)
        self.f.write(base + "candidate_frequency", freq.reshape(len(freq), -1) if freq.ndim == 1 else freq)
        self.f.write(base + "images", np.asarray(all_images, dtype=np.int8))

    def summaries(self) -> List[str]:
        return self.f.keys("/" + self._summary_path_) if self._summary_path_ in self.f else []

    def read_summary(self, name: str) -> Dict[str, np.ndarray]:
        base = "%s/%s/" % (self._summary_path_, name)
        return {k: self.f.read(base + k) for k in ("contigs", "positions", "depths", "candidates", "candidate_frequency", "images")}


class PredictionStore:
    """DataStore of DataStorePredict.py:49-66: groups predictions/batch_<k>"""
    _prediction_path_ = "predictions"

    def __init__(self, filename: str, mode: str = "r"):
        self.f = H5File(filename, mode)
        self._written = set()

    def close(self):
        self.f.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def write_prediction(self, batch_no, contigs, positions, depths, candidates, candidate_frequencies, base_predictions):
        name = "batch_" + str(batch_no)
        if name in self._written:
            return
        self._written.add(name)
        base = "%s/%s/" % (self._prediction_path_, name)
        self.f.write(base + "contigs", np.array([c.encode() if isinstance(c, str) else c for c in contigs], dtype="S"))
        self.f.write(base + "positions", np.asarray(positions, dtype=np.int32))
        self.f.write(base + "depths", np.asarray(depths, dtype=np.uint8))
        cands = np.asarray(candidates, dtype=object)
        self.f.write(base + "candidates", cands.reshape(len(cands), -1) if cands.ndim == 1 else cands, vlen_str=True)
        freq = np.asarray(candidate_frequencies, dtype=np.uint8)
        self.f.write(base + "candidate_frequency", freq.reshape(len(freq), -1) if freq.ndim == 1 else freq)
        self.f.write(base + "base_prediction", np.asarray(base_predictions, dtype=np.float64))  # np.float == float64

    def batches(self) -> Iterator[Tuple[str, Dict[str, np.ndarray]]]:
        """FindCandidates.py:157-166 iterates every batch key of every file"""
        if self._prediction_path_ not in self.f:
            return
        for name in self.f.keys("/" + self._prediction_path_):
            base = "%s/%s/" % (self._prediction_path_, name)
            yield name, {k: self.f.read(base + k) for k in
                         ("contigs", "positions", "depths", "candidates", "candidate_frequency", "base_prediction")}
