"""CPU: platform presets. Every scalar SetParameters.py sets per preset (10 image-generation scalars + min_mapq, 13
candidate-finding scalars) is typed here straight from pepper_variant/modules/argparse/SetParameters.py:12-283 and compared
with the tables the CLIs read; a MAPQ-3 read survives --ont_r10_q20 (min_mapq 1) and is dropped by the other presets."""
import argparse
import dataclasses

import numpy as np
import pytest

import bam_writer as bw
from pepper_thesis_amd import bamio, build, find_candidates as fc, make_images
from pepper_thesis_amd.batch import PRESET_MIN_MAPQ, PRESETS

# min_mapq, min_snp_baseq, min_indel_baseq, snp_frequency, insert_frequency, delete_frequency, min_coverage_threshold,
# candidate_support_threshold, snp_candidate_frequency_threshold, indel_candidate_frequency_threshold, skip_indels
IMAGE = {
    "ont_r9_guppy5_sup": (5, 1, 1, 0.10, 0.15, 0.15, 3, 2, 0.10, 0.10, False),     # SetParameters.py:15-37
    "ont_r9_guppy4_hac": (5, 1, 1, 0.10, 0.12, 0.12, 3, 2, 0.10, 0.10, False),     # :70-92
    "ont_r10_q20": (1, 1, 1, 0.1, 0.1, 0.1, 3, 2, 0.10, 0.10, False),              # :125-147
    "hifi": (5, 10, 10, 0.10, 0.12, 0.10, 2, 2, 0.10, 0.10, False),                # :179-201
    "clr": (5, 0, 0, 0.10, 0.12, 0.12, 3, 2, 0.10, 0.12, True),                    # :233-255
}
# allowed_multiallelics, snp/insert/delete p, snp/indel q, report snp/indel freq, snp/insert/delete p in lc, snp/indel q in lc
CAND = {
    "ont_r9_guppy5_sup": (4, 0.1, 0.1, 0.1, 20, 15, 0, 0, 0.1, 0.15, 0.1, 20, 10),          # :39-66
    "ont_r9_guppy4_hac": (4, 0.10, 0.25, 0.25, 20, 15, 0, 0, 0.05, 0.01, 0.01, 20, 10),     # :93-121
    "ont_r10_q20": (4, 0.00001, 0.001, 0.001, 15, 30, 0, 0, 0.000001, 0.001, 0.001, 20, 35),  # :148-176
    "hifi": (4, 0, 0, 0, 15, 20, 0, 0, 0, 0, 0, 15, 20),                                     # :202-230
    "clr": (4, 0.1, 0.2, 0.2, 20, 20, 0, 0, 0.05, 0.05, 0.05, 20, 20),                       # :256-283
}


@pytest.mark.parametrize("preset", sorted(IMAGE))
def test_image_scalars(preset):
    p = PRESETS[preset]
    got = (PRESET_MIN_MAPQ[preset], p.min_snp_baseq, p.min_indel_baseq, p.snp_freq_threshold, p.insert_freq_threshold,
           p.delete_freq_threshold, p.min_coverage_threshold, p.candidate_support_threshold, p.snp_candidate_freq_threshold,
           p.indel_candidate_freq_threshold, p.skip_indels)
    assert got == IMAGE[preset]


@pytest.mark.parametrize("preset", sorted(CAND))
def test_candidate_scalars(preset):
    assert dataclasses.astuple(fc.CANDIDATE_PRESETS[preset]) == CAND[preset]
    assert len(dataclasses.fields(fc.CandidateOptions)) == 13


def test_cli_overrides_take_precedence():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min_mapq", type=int, default=None)
    make_images.add_image_arguments(ap)
    fc.add_candidate_arguments(ap)
    a = ap.parse_args([])
    params, mq = make_images.image_options_from_args(a, "ont_r10_q20")
    assert mq == 1 and params == PRESETS["ont_r10_q20"]
    assert fc.candidate_options_from_args(a, "hifi") == fc.CANDIDATE_PRESETS["hifi"]
    a = ap.parse_args(["--min_mapq", "7", "--insert_frequency", "0.3", "--snp_q_cutoff", "11", "--delete_p_value", "0.4", "--skip_indels"])
    params, mq = make_images.image_options_from_args(a, "ont_r10_q20")
    assert mq == 7 and params.insert_freq_threshold == 0.3 and params.skip_indels is True and params.delete_freq_threshold == 0.1
    o = fc.candidate_options_from_args(a, "ont_r10_q20")
    assert o.snp_q_cutoff == 11 and o.delete_p_value == 0.4 and o.indel_q_cutoff == 30


def test_mapq3_read_kept_only_by_r10_preset(tmp_path):
    """--ont_r10_q20 sets min_mapq = 1 (SetParameters.py:126-127); the other presets 5"""
    build.build_io()
    rng = np.random.default_rng(3)
    seq = "".join(rng.choice(list("ACGT"), size=5000))
    bw.write_fasta(str(tmp_path / "r.fa"), [("c1", seq)])
    recs = []
    for i, mq in enumerate((60, 3, 0)):
        pos = 1000 + 10 * i
        recs.append(dict(tid=0, pos=pos, mapq=mq, flag=0, cigar=[(0, 200)], seq=seq[pos:pos + 200], qual=[30] * 200,
                         name="read%d" % i, hp=0))
    bw.write_bam(str(tmp_path / "r.bam"), [("c1", 5000)], recs)
    b, f = bamio.BamHandler(str(tmp_path / "r.bam")), bamio.FastaHandler(str(tmp_path / "r.fa"))
    kept = {p: sorted(r.mapq for r in bamio.region_from_files(b, f, "c1", 900, 1500, min_mapq=PRESET_MIN_MAPQ[p]).reads)
            for p in PRESET_MIN_MAPQ}
    assert kept["ont_r10_q20"] == [3, 60]
    for p in ("ont_r9_guppy5_sup", "ont_r9_guppy4_hac", "hifi", "clr"):
        assert kept[p] == [60]


def test_interval_list_rules(tmp_path):
    """whole contig = [0, len-1]; user regions clamped to len-1; pieces share their boundary (ImageGenerationUI.py:289-316)"""
    build.build_io()
    bw.write_fasta(str(tmp_path / "r.fa"), [("c1", "ACGT" * 625)])     # 2500 bp
    bw.write_bam(str(tmp_path / "r.bam"), [("c1", 2500)], [])
    b, f = bamio.BamHandler(str(tmp_path / "r.bam")), bamio.FastaHandler(str(tmp_path / "r.fa"))
    assert make_images.list_intervals(f, b, None, 1000) == [("c1", 0, 1000), ("c1", 1000, 2000), ("c1", 2000, 2499)]
    assert make_images.list_intervals(f, b, "c1:500-9000", 1000) == [("c1", 500, 1500), ("c1", 1500, 2499)]
    assert make_images.list_intervals(f, b, "c1", 5000) == [("c1", 0, 2499)]
