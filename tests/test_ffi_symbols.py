"""CPU: the C-ABI library loads and exports every symbol include/pepper_hip.h declares
(no compute calls without a GPU); and the product fails loudly when no device exists."""
import os
import re

import pytest

from pepper_thesis_amd import _ffi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build()
    return _ffi.load()


def test_header_symbols_are_all_bound_and_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "pepper_hip.h")).read()
    declared = set(re.findall(r"\b(pv_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in _ffi.SYMBOLS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(lib, name)


def test_version(lib):
    assert lib.pv_version() >= 100


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pepper_thesis_amd import runtime
    with pytest.raises(_ffi.PepperHipError) as e:
        runtime.Context(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under pepper_thesis_amd/ may import, load or call it"""
    pkg = os.path.join(ROOT, "pepper_thesis_amd")
    pat = re.compile(r"(from\s+oracle|import\s+oracle|liboracle|oracle_summarize|rnn_oracle|_ref/)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(text), f


def test_io_library_exports_its_header():
    """include/pepper_io.h <-> libpepper_io.so"""
    from pepper_thesis_amd import bamio
    build.build_io()
    lib_ = bamio.load()
    hdr = open(os.path.join(ROOT, "include", "pepper_io.h")).read()
    declared = set(re.findall(r"\b(pvio_[a-z0-9_]+)\s*\(", hdr))
    assert declared == {n for n, _, _ in bamio.IO_SYMBOLS}
    for n in declared:
        assert hasattr(lib_, n)
