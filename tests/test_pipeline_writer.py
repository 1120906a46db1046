"""CPU: the writer thread of the fused call_variant pipeline (pipeline._writer_loop): prediction groups hold exactly `batch_size`
windows (DataStorePredict.py:49-66 writes one group per DataLoader batch) whatever sizes the device calls had - a call takes the
intervals that have been read by then, so its size depends on timing - and records keep their order."""
import queue

import numpy as np
import pytest

from pepper_thesis_amd import hdf5io, pipeline


def _item(rng, start, n, contig):
    rec = dict(region=np.zeros(n, np.int64), contigs=np.array([contig] * n, dtype="S") if n else np.zeros(0, "S1"),
               positions=np.arange(start, start + n, dtype=np.int32), depths=rng.integers(1, 90, n).astype(np.int32),
               candidates=np.array([["%d_A_%d" % (start + i, i % 3)] for i in range(n)], dtype=object).reshape(n, 1),
               candidate_frequency=rng.integers(1, 50, (n, 1)).astype(np.int32))
    return ([(contig.decode(), start, start + n)], rec, rng.random((n, 3)).astype(np.float32), None)


@pytest.mark.parametrize("sizes,batch", [((3, 700, 0, 511, 1, 40), 512), ((5, 5, 5), 4), ((0, 0), 8), ((1024,), 512)])
def test_groups_of_exactly_batch_size(tmp_path, sizes, batch):
    try:
        hdf5io.lib()
    except Exception as e:   # no libhdf5 on this host
        pytest.skip(str(e))
    rng = np.random.default_rng(3)
    items, start = [], 100
    for k, n in enumerate(sizes):
        items.append(_item(rng, start, n, b"chr20" if k % 2 == 0 else b"chr20_long_name"))   # contig widths differ between calls
        start += n
    q = queue.Queue()
    for it in items:
        q.put(it)
    q.put(None)
    T, err = dict(hdf5_write_s=0.0), []
    path = str(tmp_path / "p.hdf")
    pipeline._writer_loop(q, path, None, batch, T, err)
    assert not err, err
    with hdf5io.PredictionStore(path, "r") as st:
        got = [bt for _, bt in sorted(st.batches(), key=lambda kv: int(kv[0].split("_")[-1]))]
    total = sum(sizes)
    assert [len(g["positions"]) for g in got] == [batch] * (total // batch) + ([total % batch] if total % batch else [])
    if total:
        cat = {k: np.concatenate([g[k] for g in got]) for k in ("positions", "depths", "candidate_frequency", "base_prediction")}
        exp_pos = np.concatenate([it[1]["positions"] for it in items])
        assert cat["positions"].tolist() == exp_pos.tolist()
        assert cat["depths"].tolist() == np.concatenate([it[1]["depths"] for it in items]).tolist()
        assert np.array_equal(cat["base_prediction"], np.concatenate([it[2] for it in items]).astype(np.float64))
        names = [bytes(c) for g in got for c in g["contigs"]]
        assert names == [bytes(c) for it in items for c in it[1]["contigs"]]
        cands = [str(c[0]) for g in got for c in g["candidates"]]
        assert cands == [str(c[0]) for it in items for c in it[1]["candidates"]]
