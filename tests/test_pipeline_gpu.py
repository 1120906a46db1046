"""GPU: make_images -> image HDF5 -> run_inference -> prediction HDF5, end to end against the oracles
(the file formats are the reference's; the consumer find_candidates would read these files as is)."""
import os

import numpy as np
import pytest

from pepper_thesis_amd import hdf5io, make_images, run_inference, synth
from pepper_thesis_amd.batch import PRESETS, pack_regions

pytestmark = pytest.mark.gpu


def test_make_images_then_run_inference(hip_ctx, oracle_lib, tmp_path):
    from oracle import rnn_oracle
    P = PRESETS["ont_r9_guppy5_sup"]
    intervals = [("chr20", 1_000_000 + 2800 * k, 1_000_000 + 2800 * (k + 1)) for k in range(3)]
    regs = []
    for k, (contig, start, end) in enumerate(intervals):
        rs, re_, cs, ce = make_images.interval_arithmetic(start, end)
        r = synth.synth_region(40 + k, region_len=re_ - rs + 1, depth=30, read_len=900, site_every=45, ref_start=rs, safe=0)
        r.cand_start, r.cand_end, r.contig = cs, ce, contig
        regs.append(r)
    batch = pack_regions(regs)
    img_dir, out_dir = tmp_path / "images", tmp_path / "predictions"
    os.makedirs(img_dir)
    n = make_images.write_image_file(hip_ctx, str(img_dir / "pepper_variants_images_thread_0.hdf5"), batch, intervals, P)
    exp = oracle_lib.summarize(batch, P)
    assert n == len(exp) > 50
    w = synth.make_weights_p1(3, 2.0)
    np.savez(str(tmp_path / "model.npz"), **w)
    run_inference.main(["-i", str(img_dir), "-m", str(tmp_path / "model.npz"), "-o", str(out_dir), "-bs", "64", "-per_gpu", "2"])
    with hdf5io.PredictionStore(str(out_dir / "pepper_prediction.hdf"), "r") as st:
        batches = dict(st.batches())
    keys = sorted(batches, key=lambda s: int(s.split("_")[1]))
    pos = np.concatenate([batches[k]["positions"] for k in keys])
    cand = np.concatenate([batches[k]["candidates"] for k in keys])[:, 0].tolist()
    probs = np.concatenate([batches[k]["base_prediction"] for k in keys])
    assert all(batches[k]["base_prediction"].shape[0] <= 64 for k in keys)
    # image files group by interval name; h5 lists groups alphabetically = numeric order here
    np.testing.assert_array_equal(pos, exp.position.astype(np.int32))
    assert cand == exp.candidates
    ref = rnn_oracle.p1_forward(w, exp.images, np.float64)
    np.testing.assert_allclose(probs, ref, atol=1e-4, rtol=0)
    assert probs.dtype == np.float64


def test_identical_call_decisions(hip_ctx):
    """north_star: 'identical candidate-variant calls'. The consumer's hard decisions (genotype argmax, p-value
    thresholds, integer phred cut-offs) computed from the GPU probabilities equal those computed from the
    float64 oracle, except possibly for windows within the 1e-4 tolerance of a decision edge."""
    from oracle import rnn_oracle
    from pepper_thesis_amd import calls
    w = synth.make_weights_p1(21, 3.0)  # sharp head: probabilities spread over (0,1)
    hip_ctx.load_p1(w)
    x = synth.synth_windows(22, 1024)
    cands = ["%d%s" % (1 + i % 3, "ACGT"[i % 4]) for i in range(len(x))]
    got = hip_ctx.forward_p1(x)
    ref = rnn_oracle.p1_forward(w, x, np.float64)
    r = calls.compare(got, ref, cands, tol=1e-4)
    assert r["mismatches_off_edge"] == 0, r
    assert r["mismatches"] <= 0.01 * r["n"], r
    sig = calls.decision_signature(ref, cands)
    assert len(set(sig[:, 0].tolist())) >= 2  # the test is not degenerate: more than one genotype occurs


def test_bam_to_predictions_end_to_end(hip_ctx, oracle_lib, tmp_path):
    """BAM + FASTA -> make_images (native readers + HIP builder) -> run_inference (HIP RNN) -> prediction HDF5;
    the windows equal the oracle's on the same clipped reads"""
    import bam_writer as bw
    from pepper_thesis_amd import bamio, build
    build.build_io()
    rng = np.random.default_rng(11)
    ref = "".join(rng.choice(list("ACGT"), size=60_000))
    bw.write_fasta(str(tmp_path / "ref.fa"), [("chr20", ref)])
    recs = bw.random_records(rng, 600, 60_000, tid=0, mean_len=2500, allow_skip=False)
    # read bases follow the reference with a few mismatches so that real candidate sites appear
    for r in recs:
        seq, qi, rp = list(r["seq"]), 0, r["pos"]
        for op, ln in r["cigar"]:
            if op in (0, 7, 8):
                for i in range(ln):
                    if rp + i < len(ref) and rng.random() > 0.04:
                        seq[qi + i] = ref[rp + i]
                qi += ln; rp += ln
            elif op in (1, 4):
                qi += ln
            elif op in (2, 3):
                rp += ln
        r["seq"] = "".join(seq)
        r["mapq"] = 60
        r["flag"] &= 0x10
    bw.write_bam(str(tmp_path / "reads.bam"), [("chr20", len(ref))], recs)
    P = PRESETS["ont_r9_guppy5_sup"]
    n = make_images.generate_images(hip_ctx, str(tmp_path / "reads.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "images"), P,
                                    region="chr20:5000-45000", region_size=10_000, intervals_per_call=3)
    # expected: the same intervals through the oracle
    b, f = bamio.BamHandler(str(tmp_path / "reads.bam")), bamio.FastaHandler(str(tmp_path / "ref.fa"))
    ivs = make_images.split_intervals("chr20", 5000, 45000, 10_000)
    assert ivs == [("chr20", 5000, 15000), ("chr20", 15000, 25000), ("chr20", 25000, 35000), ("chr20", 35000, 45000)]
    exp = oracle_lib.summarize(pack_regions([bamio.region_from_files(b, f, c, a, e) for c, a, e in ivs]), P)
    assert n == len(exp) > 100
    with hdf5io.ImageStore(str(tmp_path / "images" / "pepper_variants_images_thread_0.hdf5"), "r") as st:
        names = st.summaries()
        parts = {nm: st.read_summary(nm) for nm in names}
    assert sorted(names) == sorted("%s_%d_%d" % iv for iv in ivs)
    got_imgs = np.concatenate([parts["%s_%d_%d" % iv]["images"] for iv in ivs])
    np.testing.assert_array_equal(got_imgs, exp.images)
    w = synth.make_weights_p1(3, 2.0)
    np.savez(str(tmp_path / "model.npz"), **w)
    run_inference.main(["-i", str(tmp_path / "images"), "-m", str(tmp_path / "model.npz"), "-o", str(tmp_path / "pred")])
    with hdf5io.PredictionStore(str(tmp_path / "pred" / "pepper_prediction.hdf"), "r") as st:
        total = sum(bt["base_prediction"].shape[0] for _, bt in st.batches())
    assert total == n
    # make_images -hp: the same BAM (half of its reads carry an HP tag) through the haplotag-aware builder
    from pepper_thesis_amd.batch import hp_params
    make_images.main(["-b", str(tmp_path / "reads.bam"), "-f", str(tmp_path / "ref.fa"), "-o", str(tmp_path / "images_hp"),
                      "-r", "chr20:5000-45000", "--region_size", "10000", "--ont_r9_guppy5_sup", "-hp"])
    tagged = pack_regions([bamio.region_from_files(b, f, c, a, e) for c, a, e in ivs])
    assert tagged.read_hp is not None and set(np.unique(tagged.read_hp).tolist()) == {0, 1, 2}
    exp_hp = oracle_lib.summarize_hp(tagged, hp_params(P))
    with hdf5io.ImageStore(str(tmp_path / "images_hp" / "pepper_variants_images_thread_0_hp.hdf5"), "r") as st:
        parts = {nm: st.read_summary(nm) for nm in st.summaries()}
    got_hp = np.concatenate([parts["%s_%d_%d" % iv]["images"] for iv in ivs])
    assert got_hp.shape == (len(exp_hp), 21, 48) and len(exp_hp) > 100
    np.testing.assert_array_equal(got_hp, exp_hp.images)


def test_call_variant_bam_to_vcf(hip_ctx, oracle_lib, tmp_path):
    """the whole pipeline on this code base: BAM + FASTA -> images -> predictions -> VCFs; and the VCF records
    computed from the GPU probabilities equal those computed from the float64 oracle probabilities
    (north_star: identical candidate-variant calls)."""
    import bam_writer as bw
    from oracle import rnn_oracle
    from pepper_thesis_amd import bamio, build, call_variant, find_candidates as fc
    build.build_io()
    rng = np.random.default_rng(17)
    ref = "".join(rng.choice(list("ACGT"), size=40_000))
    bw.write_fasta(str(tmp_path / "ref.fa"), [("chr20", ref)])
    recs = bw.random_records(rng, 500, 40_000, tid=0, mean_len=2500, allow_skip=False)
    for r in recs:
        seq, qi, rp = list(r["seq"]), 0, r["pos"]
        for op, ln in r["cigar"]:
            if op in (0, 7, 8):
                for i in range(ln):
                    if rp + i < len(ref) and rng.random() > 0.04:
                        seq[qi + i] = ref[rp + i]
                qi += ln; rp += ln
            elif op in (1, 4):
                qi += ln
            elif op in (2, 3):
                rp += ln
        r["seq"], r["mapq"] = "".join(seq), 60
        r["flag"] &= 0x10
    bw.write_bam(str(tmp_path / "reads.bam"), [("chr20", len(ref))], recs)
    w = synth.make_weights_p1(3, 3.0)
    np.savez(str(tmp_path / "model.npz"), **w)
    out = tmp_path / "out"
    base = ["-b", str(tmp_path / "reads.bam"), "-f", str(tmp_path / "ref.fa"), "-m", str(tmp_path / "model.npz"), "-s", "HG003",
            "--ont_r9_guppy5_sup", "-r", "chr20:2000-38000", "--region_size", "12000"]
    counts = call_variant.main(base + ["-o", str(out), "--keep_images"])   # the default, fused form (+ the image file for the checks below)
    # the reference's three steps through image HDF5 files give the same files bit for bit: predictions, images, VCF text
    out2 = tmp_path / "out_steps"
    counts2 = call_variant.main(base + ["-o", str(out2), "--no_fused"])
    assert counts2 == counts

    def only(d, prefix):
        (name,) = [p for p in os.listdir(d) if p.startswith(prefix)]
        return d / name
    with hdf5io.PredictionStore(str(only(out, "predictions_") / "pepper_prediction.hdf"), "r") as a, \
            hdf5io.PredictionStore(str(only(out2, "predictions_") / "pepper_prediction.hdf"), "r") as b:
        ba, bb = list(a.batches()), list(b.batches())
    # (the two-step form reads the image groups back in HDF5 name order, the fused form writes them in interval order: the
    # same records, bit for bit, under their keys)
    def keyed(batches):
        d = {}
        for _, bt in batches:
            for i in range(len(bt["positions"])):
                key = (bytes(bt["contigs"][i]), int(bt["positions"][i]), str(bt["candidates"][i][0]))
                assert key not in d
                d[key] = (int(bt["depths"][i]), int(bt["candidate_frequency"][i][0]), bt["base_prediction"][i].tobytes())
            assert bt["base_prediction"].dtype == np.float64 and bt["positions"].dtype == np.int32
        return d
    assert len(ba) == len(bb) >= 1 and keyed(ba) == keyed(bb) and len(keyed(ba)) > 100
    with hdf5io.ImageStore(str(only(out, "images_") / "pepper_variants_images_thread_0.hdf5"), "r") as a, \
            hdf5io.ImageStore(str(only(out2, "images_") / "pepper_variants_images_thread_0.hdf5"), "r") as b:
        assert a.summaries() == b.summaries()
        for nm in a.summaries():
            x, y = a.read_summary(nm), b.read_summary(nm)
            for key in x:
                assert x[key].tolist() == y[key].tolist(), (nm, key)
    import gzip as _gz
    for fn in ("PEPPER_VARIANT_FULL.vcf.gz", "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING.vcf.gz"):
        assert _gz.open(out / fn, "rt").read() == _gz.open(out2 / fn, "rt").read()
    # without --keep_images the fused form writes no image directory at all
    out3 = tmp_path / "out_noimg"
    assert call_variant.main(base + ["-o", str(out3)]) == counts
    assert not [p for p in os.listdir(out3) if p.startswith("images_")]
    assert counts["total"] > 20 and counts["total"] == counts["pepper"] + counts["variant_calling"]
    import gzip
    vcf_text = gzip.open(out / "PEPPER_VARIANT_FULL.vcf.gz", "rt").read()
    assert os.path.exists(out / "PEPPER_VARIANT_FULL.vcf.gz.tbi") and os.path.exists(out / "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_SNPs.vcf.gz.tbi")
    full = [l for l in vcf_text.splitlines(True) if not l.startswith("#")]
    assert len(full) == counts["total"]
    pos = [int(l.split("\t")[1]) for l in full]
    assert pos == sorted(pos) and len(set(pos)) == len(pos)
    assert vcf_text.startswith("##fileformat=VCF")
    # the same prediction records with oracle probabilities -> identical VCF lines
    pred_dir = [p for p in os.listdir(out) if p.startswith("predictions_")][0]
    records = list(fc.read_prediction_records(str(out / pred_dir)))
    img_dir = [p for p in os.listdir(out) if p.startswith("images_")][0]
    with hdf5io.ImageStore(str(out / img_dir / "pepper_variants_images_thread_0.hdf5"), "r") as st:
        parts = [st.read_summary(n) for n in st.summaries()]
    parts = [p for p in parts if len(p["positions"])]
    images = np.concatenate([p["images"] for p in parts])
    keys = [(p["positions"][i], p["candidates"][i, 0]) for p in parts for i in range(len(p["positions"]))]
    ref_probs = dict(zip(keys, rnn_oracle.p1_forward(w, images, np.float64)))
    recs_oracle = [dict(r, prediction=ref_probs[(r["position"], r["candidates"][0])]) for r in records]
    fasta = bamio.FastaHandler(str(tmp_path / "ref.fa"))
    opt = fc.CandidateOptions()
    v_gpu = fc.dedupe_by_position(fc.select_candidates(records, fasta.get_reference_sequence, opt))
    v_ora = fc.dedupe_by_position(fc.select_candidates(recs_oracle, fasta.get_reference_sequence, opt))

    def key_fields(v):  # everything except the float-valued AP field
        out_ = []
        for line, sel, snp in fc.variant_records(v, opt):
            f = line.split("\t")
            s = f[9].split(":")
            out_.append((f[0], f[1], f[3], f[4], f[5], f[6], s[0], s[2], s[3], s[4], s[5], s[6], sel, snp))
        return out_
    assert key_fields(v_gpu) == key_fields(v_ora)
