"""GPU: the polisher (P2) summary-image builder (k_polish_* through pv_polish_summarize_regions) against the CPU oracle.
Integer / byte work: bit-exact. The oracle itself is PARITY UNPINNED (see tests/test_oracle_polish.py)."""
import numpy as np
import pytest

import cases
from oracle import rnn_oracle
from test_oracle_polish import POLISH_EDGE_REGIONS, _reads_of, dict_summary, py_chunks
from pepper_thesis_amd import polish_summary, synth
from pepper_thesis_amd.batch import Read, Region, pack_regions

pytestmark = pytest.mark.gpu


def assert_polish_equal(got, exp, tag=""):
    assert len(got.images) == len(exp.images), tag
    for f in ("images", "position", "index", "region", "chunk_id", "flat_images", "flat_position", "flat_index", "region_row_off"):
        a, b = getattr(got, f), getattr(exp, f)
        assert (a is None) == (b is None), (tag, f)
        if a is not None:
            assert a.shape == b.shape and np.array_equal(a, b), (tag, f)


def test_known_answers_and_edges(hip_ctx, oracle_lib):
    b = pack_regions(POLISH_EDGE_REGIONS)
    for L, O in ((1000, 50), (16, 3), (7, 0)):
        assert_polish_equal(hip_ctx.polish_summarize(b, L, O, want_flat=True), oracle_lib.polish_summarize(b, L, O), (L, O))
    reads = [Read.make(100, "5M", "ACGTA"), Read.make(100, "2M1I3M", "ACTGTA"), Read.make(100, "2M2D1M", "ACA"),
             Read.make(101, "4M", "CGTN", is_reverse=True)]
    o = hip_ctx.polish_summarize(pack_regions([Region(100, 104, b"ACGTA", reads)]), want_flat=True)
    assert o.flat_images.tolist() == [[0, 0, 0, 0, 254, 0, 0, 0, 0, 0], [0, 63, 0, 0, 0, 190, 0, 0, 0, 0],
                                      [0, 0, 0, 0, 0, 0, 0, 63, 0, 0], [0, 0, 50, 0, 0, 0, 101, 0, 0, 50],
                                      [0, 0, 0, 84, 0, 0, 0, 169, 0, 84], [0, 0, 0, 0, 190, 0, 0, 0, 63, 0]]
    # deletion-only column: 2/max(1,0)*254 = 508 -> low byte
    reads = [Read.make(9, "1M3D1M", "AC"), Read.make(9, "1M3D1M", "AC")]
    o = hip_ctx.polish_summarize(pack_regions([Region(9, 13, b"NNNNN", reads)]), want_flat=True)
    assert o.flat_images[2].tolist() == [0] * 9 + [252] and o.flat_images[1].tolist() == [0] * 9 + [84]


def test_p1_edge_inputs(hip_ctx, oracle_lib):
    b = cases.all_edges_batch()
    regs = [Region(int(b.ref_start[g]), int(b.ref_end[g]), b"N" * int(b.ref_end[g] - b.ref_start[g] + 1), _reads_of(b, g))
            for g in range(b.n_regions)]
    bb = pack_regions(regs)
    assert_polish_equal(hip_ctx.polish_summarize(bb, 64, 8, want_flat=True), oracle_lib.polish_summarize(bb, 64, 8), "edges")


@pytest.mark.parametrize("seed", range(4))
def test_random_regions_vs_oracle(hip_ctx, oracle_lib, seed):
    rng = np.random.default_rng(40 + seed)
    regs = [synth.synth_region(700 + 10 * seed + k, region_len=int(rng.integers(300, 9000)), depth=int(rng.integers(5, 90)),
                               read_len=int(rng.integers(200, 3000)), site_every=int(rng.integers(15, 200)),
                               n_rate=0.002 * (seed % 2)) for k in range(5)]
    b = pack_regions(regs)
    L, O = ((1000, 50), (100, 10), (333, 0), (1000, 999))[seed]
    got = hip_ctx.polish_summarize(b, L, O, want_flat=True)
    assert_polish_equal(got, oracle_lib.polish_summarize(b, L, O), seed)
    # chunk-only call (no flat arrays) gives the same chunks
    g2 = hip_ctx.polish_summarize(b, L, O, want_flat=False)
    assert np.array_equal(g2.images, got.images) and np.array_equal(g2.position, got.position)


def test_mirror_class_and_chunk_images(hip_ctx):
    reg = synth.synth_region(77, region_len=2500, depth=20, read_len=600, site_every=40)
    sg = polish_summary.SummaryGenerator(reg.ref.decode(), "chr20", reg.ref_start, reg.ref_end, ctx=hip_ctx)
    sg.generate_summary(reg.reads, reg.ref_start, reg.ref_end)
    img, gpos = dict_summary(reg)
    assert np.array_equal(sg.image, img) and sg.genomic_pos == gpos
    images, labels, positions, chunk_ids = polish_summary.chunk_images(sg, 1000, 50)
    spans = py_chunks(len(gpos), 1000, 50)
    assert chunk_ids == list(range(len(spans))) and len(images) == len(spans)
    for (s, e), im, ps, lb in zip(spans, images, positions, labels):
        assert np.array_equal(im[:e - s], img[s:e]) and not im[e - s:].any()
        assert ps[:e - s] == gpos[s:e] and all(p == (-1, -1) for p in ps[e - s:]) and lb == [0] * 1000
    with pytest.raises(ValueError):
        sg.generate_summary(reg.reads, reg.ref_start + 1, reg.ref_end)


def test_full_size_region_properties_and_retry(hip_ctx, oracle_lib):
    # BASELINE-size region (100 200 columns, 60x): oracle comparison on the whole region plus size-independent properties
    reg = synth.synth_region(1234, region_len=100_200, depth=60, read_len=10_000, site_every=260)
    # the second region is insert-dense (20 % insert rate at 200x)
    b = pack_regions([reg, synth.synth_region(5, region_len=3000, depth=200, read_len=400, site_every=9, ins_rate=0.2)])
    got = hip_ctx.polish_summarize(b, want_flat=True)
    assert_polish_equal(got, oracle_lib.polish_summarize(b), "full")
    for g in range(2):
        r0, r1 = int(got.region_row_off[g]), int(got.region_row_off[g + 1])
        base_rows = got.flat_index[r0:r1] == 0
        assert int(base_rows.sum()) == int(b.ref_end[g] - b.ref_start[g] + 1)
        assert np.array_equal(got.flat_position[r0:r1][base_rows], np.arange(b.ref_start[g], b.ref_end[g] + 1))
        # insert rows count up from 1 behind their anchor
        idx = got.flat_index[r0:r1]
        assert ((idx[1:] == 0) | (idx[1:] == idx[:-1] + 1)).all()
    # consecutive chunks of a region share their overlap rows
    same = (got.region[1:] == got.region[:-1])
    assert np.array_equal(got.images[1:][same][:, :50], got.images[:-1][same][:, 950:])


def test_long_insert_exceeds_workspace_heuristic(hip_ctx, oracle_lib):
    # one 9000-base insert in a 300-column region: more insert rows than the workspace heuristic (2 per column + 4096)
    # allows for, so the host API takes the device's count and runs again
    rng = np.random.default_rng(3)
    ins = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), 9000))
    reads = [Read.make(10, "100M", "A" * 100), Read.make(20, "5M9000I60M", b"C" * 5 + ins + b"G" * 60, is_reverse=True),
             Read.make(20, "5M2I60M", "T" * 67)]
    b = pack_regions([Region(0, 299, b"A" * 300, reads)])
    got = hip_ctx.polish_summarize(b, want_flat=True)
    assert len(got.flat_images) == 300 + 9000
    assert_polish_equal(got, oracle_lib.polish_summarize(b), "long insert")


def test_malformed_read_is_an_error(hip_ctx):
    from pepper_thesis_amd import _ffi
    bad = Region(0, 20, b"A" * 21, [Read.make(0, "20M", "ACGT")])
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.polish_summarize(pack_regions([bad]))
    assert e.value.code == _ffi.PV_ERR_INVALID
    with pytest.raises(_ffi.PepperHipError):
        hip_ctx.polish_summarize(pack_regions([Region(0, 20, b"A" * 21, [])]), 100, 100)


def test_device_resident_builder_into_gru(hip_ctx, oracle_lib):
    """builder -> GRU without leaving HBM: pv_polish_summarize_regions_dev writes the chunk batch that
    pv_rnn_forward_p2_dev reads; labels equal the oracle chain's (oracle builder -> float64 GRU restatement)."""
    import torch
    from pepper_thesis_amd.device import DeviceBatch, DevicePolishOut
    regs = [synth.synth_region(300 + k, region_len=2600, depth=30, read_len=900, site_every=60) for k in range(2)]
    b = pack_regions(regs)
    w = synth.make_weights_p2(31, 3.0)
    hip_ctx.load_p2(w)
    db = DeviceBatch(b)
    dout = DevicePolishOut(16)
    hip_ctx.polish_summarize_dev(db, dout)
    hip_ctx.synchronize()
    n = dout.n_chunks()
    exp = oracle_lib.polish_summarize(b)
    assert dout.status() == 0 and n == len(exp.images)
    assert np.array_equal(dout.images[:n].cpu().numpy(), exp.images)
    labels = torch.zeros((n, 1000), dtype=torch.uint8, device="cuda")
    acc = torch.zeros((n, 1000, 5), dtype=torch.float32, device="cuda")
    hip_ctx.forward_p2_dev(dout.images.data_ptr(), n, labels.data_ptr(), acc.data_ptr())
    hip_ctx.synchronize()
    lr, ar = rnn_oracle.p2_forward(w, exp.images[:3], np.float64)
    np.testing.assert_allclose(acc[:3].cpu().numpy(), ar, atol=1e-4, rtol=0)
    diff = labels[:3].cpu().numpy() != lr
    assert diff.mean() < 2e-3


def test_bam_to_polish_labels(hip_ctx, oracle_lib, tmp_path):
    """BAM + FASTA -> native readers -> polisher builder -> bi-GRU labels; equal to the oracle chain on the same clipped reads"""
    import bam_writer as bw
    from pepper_thesis_amd import bamio, build
    build.build_io()
    rng = np.random.default_rng(21)
    ref = "".join(rng.choice(list("ACGT"), size=20_000))
    bw.write_fasta(str(tmp_path / "ref.fa"), [("ctg1", ref)])
    recs = bw.random_records(rng, 300, 20_000, tid=0, mean_len=1500)   # includes N / P ops: deletions for the polisher
    for r in recs:
        r["mapq"] = int(rng.integers(0, 61))
    bw.write_bam(str(tmp_path / "reads.bam"), [("ctg1", len(ref))], recs)
    b, f = bamio.BamHandler(str(tmp_path / "reads.bam")), bamio.FastaHandler(str(tmp_path / "ref.fa"))
    ivs = [(1000, 4999), (5000, 8200), (19_000, 19_999)]
    regs = [polish_summary.region_from_files(b, f, "ctg1", s, e) for s, e in ivs]
    assert all(r is not None and len(r.reads) > 5 for r in regs)
    w = synth.make_weights_p2(31, 3.0)
    hip_ctx.load_p2(w)
    out, labels, acc = polish_summary.polish_regions(hip_ctx, regs, want_acc=True)
    exp = oracle_lib.polish_summarize(pack_regions(regs), want_flat=False)
    assert np.array_equal(out.images, exp.images) and np.array_equal(out.position, exp.position)
    assert out.region.tolist() == exp.region.tolist() and len(out.images) >= 8
    lr, ar = rnn_oracle.p2_forward(w, exp.images[:2], np.float64)
    np.testing.assert_allclose(acc[:2], ar, atol=1e-4, rtol=0)
    assert (labels[:2] != lr).mean() < 2e-3
