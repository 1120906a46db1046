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


# ---- the reference's command line (pepper_variant.py:34-91): every option name of its argparse definitions parses here ----------
# (long / short names typed from pepper_variant/modules/argparse/{CallVariants,MakeImages,RunInference,FindCandidates}Arguments.py)
_IMG = [("-d", "0.5"), ("--downsample_rate", "0.5"), ("-r", "chr20:1-1000"), ("--region", "chr1-22"), ("--region_size", "50000"),
        ("--region_bed", "x.bed"), ("-rb", "x.bed"), ("-hp", None), ("--use_hp_info", None), ("--include_supplementary", None),
        ("--min_mapq", "3"), ("--min_snp_baseq", "2"), ("--min_indel_baseq", "2"), ("--snp_frequency", "0.2"), ("--insert_frequency", "0.2"),
        ("--delete_frequency", "0.2"), ("--min_coverage_threshold", "4"), ("--candidate_support_threshold", "3"),
        ("--snp_candidate_frequency_threshold", "0.2"), ("--indel_candidate_frequency_threshold", "0.2"), ("--skip_indels", None)]
_INF = [("-bs", "256"), ("--batch_size", "256"), ("-g", None), ("--gpu", None), ("-per_gpu", "2"), ("--callers_per_gpu", "2"),
        ("-d_ids", "0,1"), ("--device_ids", "0"), ("--quantized", None), ("--no_quantized", None), ("-w", "4"), ("--num_workers", "4")]
_CAND = [("--allowed_multiallelics", "2")] + [("--" + n, "0.5") for n in (
    "snp_p_value", "insert_p_value", "delete_p_value", "snp_p_value_in_lc", "insert_p_value_in_lc", "delete_p_value_in_lc", "snp_q_cutoff",
    "indel_q_cutoff", "snp_q_cutoff_in_lc", "indel_q_cutoff_in_lc", "report_snp_above_freq", "report_indel_above_freq")]
_PLATFORMS = ["--ont_r9_guppy5_sup", "--ont_r9_guppy4_hac", "--ont_r10_q20", "--hifi", "--clr"]


def _each(parser_fn, required, options):
    for opt, val in options:
        argv = list(required) + [opt] + ([val] if val is not None else [])
        a = parser_fn().parse_args(argv)
        assert a is not None, opt


def test_call_variant_accepts_every_reference_option():
    from pepper_thesis_amd import cli
    req = ["-b", "x.bam", "-f", "x.fa", "-m", "m.pkl", "-o", "out", "-s", "S", "-t", "8", "--ont_r9_guppy5_sup"]
    _each(cli.call_variant_parser, req, _IMG + _INF + _CAND)
    for pf in _PLATFORMS:
        cli.call_variant_parser().parse_args(["--bam", "x.bam", "--fasta", "x.fa", "--model_path", "m", "--output_dir", "o", "--sample_name", "S",
                                              "--threads", "4", pf])
    with pytest.raises(SystemExit):   # exactly one platform flag is required (CallVariantsArguments.py: mutually exclusive group)
        cli.call_variant_parser().parse_args(req[:-1])
    with pytest.raises(SystemExit):
        cli.call_variant_parser().parse_args(req + ["--hifi"])
    a = cli.call_variant_parser().parse_args(req)
    assert (a.batch_size, a.callers_per_gpu, a.region_size, a.downsample_rate, a.num_workers, a.quantized) == (512, 4, 100000, 1.0, 0, False)
    assert a.fused and not a.keep_images


def test_make_images_run_inference_find_candidates_accept_every_reference_option():
    from pepper_thesis_amd import cli
    _each(cli.make_images_parser, ["-b", "x.bam", "-f", "x.fa", "-o", "out", "-t", "4", "--hifi"], _IMG)
    _each(cli.run_inference_parser, ["-i", "img", "-m", "m.pkl", "-o", "out"],
          _INF + [("-t", "8"), ("--threads", "8"), ("-hp", None), ("--use_hp_info", None), ("--dry", None), ("--ont_r9_guppy5_sup", None)])
    _each(cli.find_candidates_parser, ["-i", "pred", "-b", "x.bam", "-f", "x.fa", "-s", "S", "-o", "out", "-t", "4", "--clr"],
          _CAND + [("-hp", None), ("--freq_based", None), ("--freq", "0.2")])


def test_dispatcher_and_refusals(capsys):
    """`python -m pepper_thesis_amd <sub-command>` (pepper_variant.py:34-91): --version, unknown / missing sub-command, --dry refused"""
    from pepper_thesis_amd import cli
    assert cli.main(["--version"]) == 0
    assert "PEPPER VERSION" in capsys.readouterr().out
    assert cli.main([]) == 2
    assert cli.main(["merge_variants"]) == 2
    assert cli.main(["run_inference", "-i", "img", "-m", "m", "-o", "out", "--dry"]) == 2
    assert "--dry" in capsys.readouterr().err
    assert cli.main(["call_variant", "-b", "x", "-f", "x", "-m", "m", "-o", "o", "--hifi", "-hp"]) == 2


def test_region_grammar():
    """-r: comma lists, name:start-end, ranges chr1-22 (ImageGenerationUI.py:131-169); --region_bed is parsed (train-mode only use)"""
    e = make_images.expand_region_names
    assert e("chr20") == ["chr20"] and e("chr20:1-1000000") == ["chr20:1-1000000"]
    assert e("chr1-3") == ["chr1", "chr2", "chr3"] and e("3-1") == ["1", "2", "3"]
    assert e("chr1-2:5-10, chrX") == ["chr1:5-10", "chr2:5-10", "chrX"]
    assert len(e("chr1-22")) == 22


def test_region_bed_is_validated_not_applied(tmp_path):
    build.build_io()
    bw.write_fasta(str(tmp_path / "r.fa"), [("c1", "ACGT" * 625)])
    bw.write_bam(str(tmp_path / "r.bam"), [("c1", 2500)], [])
    b, f = bamio.BamHandler(str(tmp_path / "r.bam")), bamio.FastaHandler(str(tmp_path / "r.fa"))
    bed = tmp_path / "hc.bed"
    bed.write_text("c1\t100\t200\nc1\t900\t300\n")
    assert make_images.read_bed(str(bed)) == {"c1": [[100, 200], [300, 900]]}
    assert make_images.list_intervals(f, b, "c1", 1000, str(bed)) == make_images.list_intervals(f, b, "c1", 1000)
    bad = tmp_path / "bad.bed"
    bad.write_text("c1\tabc\t5\n")
    with pytest.raises(ValueError):
        make_images.list_intervals(f, b, "c1", 1000, str(bad))
