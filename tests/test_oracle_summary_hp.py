"""CPU: the haplotag-aware image-builder oracle (oracle/region_summary_hp_oracle.c) against
 (1) the committed golden vectors produced by the REFERENCE's region_summary_hp.cpp, and
 (2) where oracle/_ref exists (build container), that reference build itself on fresh random regions.
Bit-exact (integer work)."""
import numpy as np
import pytest

import cases
from golden_io import assert_summary_equal, golden_names, hp_golden_case, summary_as_expected
from pepper_thesis_amd import synth
from pepper_thesis_amd.batch import PRESETS, hp_params, pack_regions

# column 40 of cases.hp_known_answer(), worked by hand from region_summary_hp.cpp:393-463 and :970-974:
# ref 'A'; T in (hp 0, fwd), (hp 1, fwd), (hp 2, rev); A in (0, rev), (1, fwd), (2, fwd), (1, rev), (2, rev)
HP_KNOWN_ROW = [1, 4, 0, 0, -3, 2, 0, 0, -1, 0, 0, 0, 0, 0, 0, -2, 0, 0, 0, -2, 0, 0, 0, 0, 0, 0,
                -2, 1, 0, 0, -1, 0, 0, 0, 0, 0, 0, -3, 1, 0, 0, -2, 0, 0, 0, 0, 0, 0]


def test_golden_has_all_cases(summary_hp_golden):
    assert len(golden_names(summary_hp_golden)) == 1 + 2 * len(cases.EDGE_CASES) + len(cases.GOLDEN_RANDOM)


def test_oracle_matches_reference_golden(oracle_lib, summary_hp_golden):
    n = 0
    for entry in golden_names(summary_hp_golden):
        batch, params, exp = hp_golden_case(summary_hp_golden, entry)
        out = oracle_lib.summarize_hp(batch, params, want_i32=True)
        assert_summary_equal(out, exp, entry)
        n += len(out)
    assert n > 400


def test_known_answer(oracle_lib):
    P = hp_params(PRESETS["ont_r9_guppy5_sup"])
    o = oracle_lib.summarize_hp(pack_regions([cases.hp_known_answer()]), P, True)
    assert (len(o), int(o.position[0]), int(o.depth[0]), o.candidates, int(o.cand_freq[0])) == (1, 40, 8, ["1T"], 3)
    assert o.images_i32.shape == (1, 21, 48)
    assert o.images_i32[0, 10].tolist() == HP_KNOWN_ROW
    # a neighbouring row: every read matches; column 41 is 'C' -> planes 9 / 20 / 31 / 42, REF planes as above
    row = o.images_i32[0, 11]
    assert row[0] == 2 and row[[4, 15, 26, 37]].tolist() == [-3, -2, -2, -3]
    assert row[[9, 20, 31, 42]].tolist() == [-3, -2, -2, -3] and int(np.abs(row).sum()) == 2 + 20


def test_untagged_reads_count_in_both_haplotypes(oracle_lib):
    """hp_tag 0 everywhere: the two haplotype halves of every window are equal (planes 4..25 == 26..47)"""
    b = cases.random_batch(11, cases.GOLDEN_RANDOM[0][1])
    assert b.read_hp is None
    o = oracle_lib.summarize_hp(b, hp_params(PRESETS["ont_r9_guppy5_sup"]), True)
    assert len(o) > 20
    np.testing.assert_array_equal(o.images_i32[:, :, 4:26], o.images_i32[:, :, 26:48])


@pytest.mark.parametrize("seed", range(6))
def test_oracle_vs_live_reference(oracle_lib, seed):
    if not oracle_lib.have_reference_hp():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(seed)
    preset = list(PRESETS)[seed % len(PRESETS)]
    regs = [synth.synth_region(1000 + 10 * seed + k, region_len=int(rng.integers(300, 4000)),
                               depth=int(rng.integers(5, 90)), read_len=int(rng.integers(200, 3000)),
                               site_every=int(rng.integers(15, 200)), n_rate=0.001 * (seed % 2),
                               mismatch=0.03 * (1 + seed % 3), ins_rate=0.02, del_rate=0.03)
            for k in range(3)]
    batch = pack_regions(cases.tag_reads(regs, seed))
    P = hp_params(PRESETS[preset])
    o = oracle_lib.summarize_hp(batch, P, True)
    r = oracle_lib.reference_summarize_hp(batch, P, True)
    assert_summary_equal(o, summary_as_expected(r), "seed %d" % seed)
    assert len(o) > 0


def test_wrong_geometry_is_refused(oracle_lib):
    with pytest.raises(RuntimeError):
        oracle_lib.summarize_hp(cases.hp_edge_batch("kat1_snp"), PRESETS["ont_r9_guppy5_sup"])  # 32 / 26: the other builder
