"""CPU: the image-builder oracle (oracle/region_summary_oracle.c) against
 (1) the committed golden vectors produced by the REFERENCE's region_summary.cpp, and
 (2) where oracle/_ref exists (build container), the reference itself on fresh random regions.
Bit-exact (integer work)."""
import numpy as np
import pytest

import cases
from golden_io import assert_summary_equal, golden_case, golden_names, summary_as_expected
from pepper_thesis_amd import synth
from pepper_thesis_amd.batch import PRESETS, pack_regions


def test_golden_has_all_cases(summary_golden):
    names = golden_names(summary_golden)
    assert len(names) == 2 * len(cases.EDGE_CASES) + len(cases.GOLDEN_RANDOM)


def test_oracle_matches_reference_golden(oracle_lib, summary_golden):
    for entry in golden_names(summary_golden):
        batch, params, exp = golden_case(summary_golden, entry)
        out = oracle_lib.summarize(batch, params, want_i32=True)
        assert_summary_equal(out, exp, entry)


def test_known_answers(oracle_lib):
    """SURVEY Appendix D.3 known answers, typed in by hand"""
    P = PRESETS["ont_r9_guppy5_sup"]
    o = oracle_lib.summarize(cases.edge_batch("kat1_snp"), P, True)
    assert (len(o), int(o.position[0]), int(o.depth[0]), o.candidates, int(o.cand_freq[0])) == (1, 40, 6, ["1T"], 3)
    assert o.images_i32[0, 16].tolist() == [1, 4, 0, 0, -3, 2, 0, 0, -1, 0, 0, 2, 0, 0, 0, -3, 1, 0, 0, -2, 0, 0, 1, 0, 0, 0]
    o = oracle_lib.summarize(cases.edge_batch("kat23_indel"), P, True)
    assert o.candidates == ["3CAA", "2GGG"] and o.position.tolist() == [1030, 1050] and o.cand_freq.tolist() == [4, 3]
    assert o.images_i32[0, 16].tolist() == [2, 0, 0, 3, -2, 0, 0, 2, 0, -4, 0, 0, 0, 2, 0, -2, 0, 0, 2, 0, -4, 0, 0, 0, 2, 0]
    assert o.images_i32[0, 17].tolist() == [1, 0, 0, 3, -2, 0, 0, 2, -2, 0, 0, 0, 0, 0, 2, -2, 0, 0, 2, -2, 0, 0, 0, 0, 0, 2]
    assert o.images_i32[1, 16].tolist() == [3, 0, 3, 0, -2, 0, 2, 0, 0, 0, -4, 0, 2, 0, 0, -3, 0, 1, 0, 0, 0, -4, 0, 1, 0, 0]
    o = oracle_lib.summarize(cases.edge_batch("kat4_clamp"), P, True)
    assert (int(o.depth[0]), o.candidates, int(o.cand_freq[0])) == (125, ["1G"], 60)
    assert o.images_i32[0, 16, :11].tolist() == [1, 3, 0, 0, -300, 60, 0, 0, -240, 0, 60]
    assert o.images_i32[0, 15, 11] == -125 and o.images[0, 16, 4] == -44 and o.images[0, 16, 8] == 16


def test_capacity_query(oracle_lib):
    """two-call size query: PV_ERR_CAPACITY reports the need"""
    from pepper_thesis_amd import _ffi
    from pepper_thesis_amd.batch import run_flat_summarizer
    from oracle.oracle import _load, ORACLE_SO
    b = cases.random_batch(11, cases.GOLDEN_RANDOM[0][1])
    rc, out = run_flat_summarizer(_load(ORACLE_SO, "oracle_summarize_regions"), b, PRESETS["ont_r9_guppy5_sup"],
                                  capacity=2, str_capacity=4)
    assert rc == 0 and len(out) == 74


@pytest.mark.parametrize("seed", range(6))
def test_oracle_vs_live_reference(oracle_lib, seed):
    if not oracle_lib.have_reference():
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    rng = np.random.default_rng(seed)
    preset = list(PRESETS)[seed % len(PRESETS)]
    regs = [synth.synth_region(1000 + 10 * seed + k, region_len=int(rng.integers(300, 4000)),
                               depth=int(rng.integers(5, 90)), read_len=int(rng.integers(200, 3000)),
                               site_every=int(rng.integers(15, 200)), n_rate=0.001 * (seed % 2),
                               mismatch=0.03 * (1 + seed % 3), ins_rate=0.02, del_rate=0.03)
            for k in range(3)]
    batch = pack_regions(regs)
    o = oracle_lib.summarize(batch, PRESETS[preset], True)
    r = oracle_lib.reference_summarize(batch, PRESETS[preset], True)
    assert_summary_equal(o, summary_as_expected(r), "seed %d" % seed)
    assert len(o) > 0


def test_all_edges_in_one_batch(oracle_lib):
    """regions are independent: a batch equals the concatenation of its regions"""
    P = PRESETS["ont_r9_guppy5_sup"]
    whole = oracle_lib.summarize(cases.all_edges_batch(), P, True)
    n = 0
    for name in cases.EDGE_CASES:
        part = oracle_lib.summarize(cases.edge_batch(name), P, True)
        k = len(part)
        assert whole.candidates[n:n + k] == part.candidates
        np.testing.assert_array_equal(whole.images[n:n + k], part.images)
        n += k
    assert n == len(whole)
