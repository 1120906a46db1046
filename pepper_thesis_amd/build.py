"""Builds csrc/libpepper_hip.so for gfx950 with hipcc (in-tree, so that it travels to the GPU box).

hipcc cross-compiles without a GPU; one object per translation unit so that an edit recompiles
only its file.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libpepper_hip.so")
SOURCES = ["pv_api.hip", "summary_kernels.hip", "rnn_kernels.hip", "rnn_gru.hip", "rnn_rec_bf16.hip", "pv_comm.hip"]
HEADERS = ["pv_common.hpp", "mfma_tiles.hpp", "rnn_bf16.hpp", os.path.join("..", "..", "include", "pepper_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


IO_LIB = os.path.join(CSRC, "libpepper_io.so")


def build_io(force=False, verbose=False):
    """libpepper_io.so: native BAM/BAI + FASTA/FAI readers (host C++, zlib only)"""
    src = os.path.join(CSRC, "pv_io.cpp")
    hdr = os.path.normpath(os.path.join(CSRC, "..", "..", "include", "pepper_io.h"))
    if force or _stale(IO_LIB, [src, hdr]):
        cmd = [os.environ.get("CXX", "g++"), "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-o", IO_LIB, src, "-lz", "-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return IO_LIB


def build(force=False, verbose=False):
    build_io(force, verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


def build_variant(name, extra_flags):
    """Diagnostic / A-B builds: variants/libpepper_hip_<name>.so with extra compile flags (e.g. ["-DPV_PSTAMPS"]); select it
    at run time with PEPPER_HIP_LIB=<path>. variants/ is git-ignored but travels to the GPU box with gpurun."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    vdir = os.path.normpath(os.path.join(HERE, "..", "variants"))
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        o = os.path.join(vdir, "%s_%s" % (name, src.replace(".hip", ".o")))
        subprocess.check_call([hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    lib = os.path.join(vdir, "libpepper_hip_%s.so" % name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
