"""CPU: the native BAM/BAI + FASTA/FAI readers (csrc/pv_io.cpp) against files produced by the test-side writer
and the Python restatement of the reference's clipping rules (tests/bam_writer.py). htslib is not available
offline, so these tests pin self-consistency with the format specification and bam_handler.cpp:115-451."""
import numpy as np
import pytest

import bam_writer as bw
from pepper_thesis_amd import bamio, build


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    build.build_io()
    d = tmp_path_factory.mktemp("bam")
    rng = np.random.default_rng(5)
    seqs = [("chr20", "".join(rng.choice(list("ACGTacgtN"), size=130_000, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]))),
            ("chrM", "".join(rng.choice(list("ACGT"), size=16_500)))]
    bw.write_fasta(str(d / "ref.fa"), seqs, width=70)
    recs = bw.random_records(rng, 900, 130_000, tid=0) + bw.random_records(rng, 60, 16_500, tid=1, mean_len=800)
    bw.write_bam(str(d / "reads.bam"), [(n, len(s)) for n, s in seqs], recs)
    # DNA-like records (no N/P ops: the reference's builder advances the read index on REF_SKIP, region_summary.cpp:556-561,
    # which walks such reads past their end) for the builder hand-off test
    dna = bw.random_records(rng, 700, 130_000, tid=0, allow_skip=False)
    bw.write_bam(str(d / "dna.bam"), [(n, len(s)) for n, s in seqs], dna)
    return dict(bam=str(d / "reads.bam"), dna_bam=str(d / "dna.bam"), dna=dna, fa=str(d / "ref.fa"), seqs=dict(seqs), recs=recs)


def test_header(files):
    b = bamio.BamHandler(files["bam"])
    assert b.get_chromosome_sequence_names() == ["chr20", "chrM"]
    f = bamio.FastaHandler(files["fa"])
    assert f.get_chromosome_names() == ["chr20", "chrM"]
    assert f.get_chromosome_sequence_length("chr20") == 130_000 and f.get_chromosome_sequence_length("nope") == -2


def test_fasta_fetch(files):
    f = bamio.FastaHandler(files["fa"])
    s = files["seqs"]["chr20"]
    for a, b_ in [(0, 10), (65, 75), (69, 71), (1000, 1000 + 4321), (129_990, 130_050), (130_000, 130_010), (50, 50)]:
        assert f.get_reference_sequence("chr20", a, b_) == s[a:b_].upper(), (a, b_)
    with pytest.raises(IOError):
        f.get_reference_sequence("chr7", 0, 10)


@pytest.mark.parametrize("region", [("chr20", 0, 1000), ("chr20", 16_300, 16_500), ("chr20", 49_900, 60_100),
                                    ("chr20", 100_000, 129_999), ("chr20", 65_535, 65_537), ("chrM", 100, 16_400)])
@pytest.mark.parametrize("supp,min_mapq", [(False, 5), (True, 0)])
def test_get_reads_matches_restated_clipping(files, region, supp, min_mapq):
    contig, start, stop = region
    tid = 0 if contig == "chr20" else 1
    b = bamio.BamHandler(files["bam"])
    got = b.get_reads(contig, start, stop, supp, min_mapq, 0)
    exp = bw.expected_reads(files["recs"], tid, start, stop, supp, min_mapq)
    assert len(got) == len(exp) and len(exp) > 0
    for g, e in zip(got, exp):
        assert (g.pos, g.pos_end, g.is_reverse, g.mapq, g.hp_tag, g.query_name) == (e["pos"], e["pos_end"], e["rev"], e["mapq"], e["hp"], e["name"])
        assert g.bases.decode() == e["seq"]
        assert g.quals.tolist() == e["qual"]
        assert [(int(c) & 0xF, int(c) >> 4) for c in g.cigar] == e["cigar"]


def test_region_from_files_feeds_the_builder(files, oracle_lib):
    """BAM + FASTA -> Region -> (oracle) image builder: the file path produces the same windows as handing the
    builder the clipped reads directly"""
    from pepper_thesis_amd.batch import PRESETS, Read, Region, pack_regions
    from pepper_thesis_amd.make_images import interval_arithmetic
    b, f = bamio.BamHandler(files["dna_bam"]), bamio.FastaHandler(files["fa"])
    reg = bamio.region_from_files(b, f, "chr20", 20_000, 30_000, min_mapq=5)
    rs, re_, cs, ce = interval_arithmetic(20_000, 30_000)
    assert (reg.ref_start, reg.ref_end, reg.cand_start, reg.cand_end) == (rs, re_, cs, ce)
    assert reg.ref.decode() == files["seqs"]["chr20"][rs:re_ + 1].upper()
    exp = bw.expected_reads(files["dna"], 0, rs, re_, False, 5)
    reads = [Read.make(e["pos"], bw_pack(e["cigar"]), e["seq"], e["qual"], e["rev"], e["mapq"]) for e in exp]
    direct = Region(rs, re_, reg.ref, reads, cs, ce, "chr20")
    P = PRESETS["ont_r9_guppy5_sup"]
    o1 = oracle_lib.summarize(pack_regions([reg]), P)
    o2 = oracle_lib.summarize(pack_regions([direct]), P)
    assert len(o1) == len(o2) > 0 and o1.candidates == o2.candidates
    np.testing.assert_array_equal(o1.images, o2.images)


def bw_pack(cig):
    return np.asarray([(l << 4) | op for op, l in cig], dtype=np.uint32)
