"""CPU: the image / prediction HDF5 files match the reference's format: round trip through
this module, and - where a real h5py exists (/opt/conda python) - both directions against h5py itself, with
the writer script restating DataStore.write_summary's dtypes (DataStore.py:63-68)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pepper_thesis_amd import hdf5io

CONDA_PY = "/opt/conda/bin/python3.9"


def _have_h5py():
    if not os.path.exists(CONDA_PY):
        return False
    return subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True).returncode == 0


def _sample(n=7):
    rng = np.random.default_rng(0)
    return dict(contigs=["chr20"] * n, positions=rng.integers(1, 1 << 30, n), depths=rng.integers(0, 126, n),
                candidates=[["1T"], ["2AGG"], ["3CAA"], ["1N"], ["2" + "A" * 60], ["3CT"], ["1G"]][:n],
                freq=rng.integers(0, 126, (n, 1)), images=rng.integers(-128, 128, (n, 33, 26)).astype(np.int8),
                probs=rng.random((n, 3)))


def test_roundtrip(tmp_path):
    s = _sample()
    p = str(tmp_path / "img.hdf5")
    with hdf5io.ImageStore(p, "w") as st:
        st.write_summary("chr20_100_200", s["contigs"], s["positions"], s["depths"], s["candidates"], s["freq"], s["images"])
        st.write_summary("chr20_200_300", s["contigs"][:2], s["positions"][:2], s["depths"][:2], s["candidates"][:2], s["freq"][:2], s["images"][:2])
    with hdf5io.ImageStore(p, "r") as st:
        assert sorted(st.summaries()) == ["chr20_100_200", "chr20_200_300"]
        r = st.read_summary("chr20_100_200")
    assert r["contigs"].dtype == np.dtype("S5") and r["contigs"].tolist() == [b"chr20"] * 7
    assert r["positions"].dtype == np.int32 and r["positions"].tolist() == s["positions"].tolist()
    assert r["depths"].dtype == np.uint8 and r["candidate_frequency"].shape == (7, 1)
    assert r["candidates"].shape == (7, 1) and r["candidates"][:, 0].tolist() == [c[0] for c in s["candidates"]]
    assert r["images"].dtype == np.int8 and np.array_equal(r["images"], s["images"])
    q = str(tmp_path / "pred.hdf")
    with hdf5io.PredictionStore(q, "w") as st:
        st.write_prediction(0, s["contigs"], s["positions"], s["depths"], s["candidates"], s["freq"], s["probs"])
    with hdf5io.PredictionStore(q, "r") as st:
        (name, b), = list(st.batches())
    assert name == "batch_0" and b["base_prediction"].dtype == np.float64 and np.array_equal(b["base_prediction"], s["probs"])


def test_interval_rules():
    from pepper_thesis_amd.make_images import downsample_indices, interval_arithmetic, split_intervals
    assert interval_arithmetic(1000, 101000) == (900, 101100, 1000, 101000)
    assert interval_arithmetic(50, 100) == (0, 200, 50, 100)
    assert split_intervals("chr20", 1, 250001) == [("chr20", 1, 100001), ("chr20", 100001, 200001), ("chr20", 200001, 250001)]
    assert downsample_indices(100).tolist() == list(range(100))
    k = downsample_indices(6000)
    assert len(k) == 5000 and len(set(k.tolist())) == 5000 and k.max() >= 5000


@pytest.mark.skipif(not _have_h5py(), reason="no h5py interpreter in this image")
def test_against_real_h5py(tmp_path):
    s = _sample()
    # (1) our file, read by h5py exactly as dataloader_predict.py:56-61 does
    p = str(tmp_path / "ours.hdf5")
    with hdf5io.ImageStore(p, "w") as st:
        st.write_summary("chr20_100_200", s["contigs"], s["positions"], s["depths"], s["candidates"], s["freq"], s["images"])
    code = (
        "import h5py, json, numpy as np\n"
        "f = h5py.File(%r, 'r'); g = f['summaries']['chr20_100_200']\n"
        "c = g['candidates'][()]\n"
        "print(json.dumps(dict(keys=sorted(g.keys()), contigs=[x.decode() for x in g['contigs'][()]], positions=g['positions'][()].tolist(),\n"
        "  cand=[x[0].decode() if isinstance(x[0], bytes) else x[0] for x in c], cshape=list(c.shape), vlen=str(h5py.check_string_dtype(g['candidates'].dtype)),\n"
        "  img_dtype=str(g['images'].dtype), img_sum=int(g['images'][()].astype(np.int64).sum()), chunks=str(g['images'].chunks),\n"
        "  dt=[str(g[k].dtype) for k in ('positions','depths','candidate_frequency')])))\n" % p)
    res = json.loads(subprocess.check_output([CONDA_PY, "-c", code]).decode())
    assert res["keys"] == ["candidate_frequency", "candidates", "contigs", "depths", "images", "positions"]
    assert res["contigs"] == ["chr20"] * 7 and res["positions"] == s["positions"].tolist()
    assert res["cand"] == [c[0] for c in s["candidates"]] and res["cshape"] == [7, 1]
    assert "utf-8" in res["vlen"] and "None" in res["vlen"]  # variable length UTF-8
    assert res["img_dtype"] == "int8" and res["img_sum"] == int(s["images"].astype(np.int64).sum()) and res["chunks"] == "None"
    assert res["dt"] == ["int32", "uint8", "uint8"]
    # (2) a file written by h5py with the dtypes of DataStore.write_summary, read by this module
    q = str(tmp_path / "theirs.hdf5")
    np.savez(str(tmp_path / "s.npz"), positions=s["positions"], depths=s["depths"], freq=s["freq"], images=s["images"])
    code = (
        "import h5py, numpy as np\n"
        "z = np.load(%r)\n"
        "f = h5py.File(%r, 'w'); dt = h5py.special_dtype(vlen=str); b = 'summaries/chr20_100_200/'\n"
        "f[b + 'contigs'] = np.array(['chr20'] * 7, dtype='S')\n"
        "f[b + 'positions'] = np.array(z['positions'], dtype=np.int32)\n"
        "f[b + 'depths'] = np.array(z['depths'], dtype=np.uint8)\n"
        "f[b + 'candidates'] = np.array(%r, dtype=dt)\n"
        "f[b + 'candidate_frequency'] = np.array(z['freq'], dtype=np.uint8)\n"
        "f[b + 'images'] = np.array(z['images'], dtype=np.int8)\n"
        "f.close()\n" % (str(tmp_path / "s.npz"), q, s["candidates"]))
    subprocess.check_call([CONDA_PY, "-c", code])
    with hdf5io.ImageStore(q, "r") as st:
        r = st.read_summary("chr20_100_200")
    assert r["candidates"][:, 0].tolist() == [c[0] for c in s["candidates"]]
    assert np.array_equal(r["images"], s["images"]) and r["positions"].tolist() == s["positions"].tolist()
    assert r["contigs"].tolist() == [b"chr20"] * 7 and r["candidate_frequency"].shape == (7, 1)
