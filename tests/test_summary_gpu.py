"""GPU: the HIP image builder (through the C-ABI) against the reference's golden vectors and the
CPU oracle. Integer work: bit-exact."""
import numpy as np
import pytest

import cases
from golden_io import assert_summary_equal, golden_case, golden_names, summary_as_expected
from pepper_thesis_amd import synth
from pepper_thesis_amd.batch import PRESETS, pack_regions

pytestmark = pytest.mark.gpu


def test_golden_vectors(hip_ctx, summary_golden):
    for entry in golden_names(summary_golden):
        batch, params, exp = golden_case(summary_golden, entry)
        out = hip_ctx.summarize(batch, params, want_i32=True)
        assert_summary_equal(out, exp, entry)


def test_known_answer_kat1(hip_ctx):
    o = hip_ctx.summarize(cases.edge_batch("kat1_snp"), PRESETS["ont_r9_guppy5_sup"], True)
    assert (len(o), int(o.position[0]), int(o.depth[0]), o.candidates, int(o.cand_freq[0])) == (1, 40, 6, ["1T"], 3)
    assert o.images_i32[0, 16].tolist() == [1, 4, 0, 0, -3, 2, 0, 0, -1, 0, 0, 2, 0, 0, 0, -3, 1, 0, 0, -2, 0, 0, 1, 0, 0, 0]


def test_all_edges_one_batch(hip_ctx, oracle_lib):
    for preset in PRESETS:
        b = cases.all_edges_batch()
        o = hip_ctx.summarize(b, PRESETS[preset], True)
        assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, PRESETS[preset], True)), preset)


@pytest.mark.parametrize("seed", range(4))
def test_random_regions_vs_oracle(hip_ctx, oracle_lib, seed):
    rng = np.random.default_rng(seed)
    preset = list(PRESETS)[seed % len(PRESETS)]
    regs = [synth.synth_region(500 + 10 * seed + k, region_len=int(rng.integers(300, 6000)),
                               depth=int(rng.integers(5, 90)), read_len=int(rng.integers(200, 3000)),
                               site_every=int(rng.integers(15, 200)), n_rate=0.002 * (seed % 2),
                               ref_n_rate=0.0, mismatch=0.03 * (1 + seed % 3))
            for k in range(5)]
    batch = pack_regions(regs)
    o = hip_ctx.summarize(batch, PRESETS[preset], True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(batch, PRESETS[preset], True)), "seed %d" % seed)
    assert len(o) > 0


def test_lowercase_and_foreign_bytes(hip_ctx, oracle_lib):
    """bytes bam_handler never emits: lower-case reads / reference, arbitrary symbols - still exact"""
    from pepper_thesis_amd.batch import Read, Region
    ref = b"ACGTacgtACGTNNACGTacgtACGTACGTAAAACCCCGGGGTTTTACGT"
    reads = []
    rng = np.random.default_rng(3)
    syms = np.frombuffer(b"ACGTacgtNnRYKM*-=", dtype=np.uint8)
    for i in range(40):
        seq = bytearray(ref)
        for j in rng.integers(0, len(ref), size=6):
            seq[j] = int(rng.choice(syms))
        reads.append(Read.make(10, "%dM" % len(ref), bytes(seq), 20, i % 2 == 0))
    P = PRESETS["ont_r9_guppy5_sup"]
    b = pack_regions([Region(10, 10 + len(ref) - 1, ref, reads, 10, 10 + 11)])  # candidates away from the N columns
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "foreign")


def test_capacity_growth(hip_ctx, oracle_lib):
    b = cases.random_batch(11, cases.GOLDEN_RANDOM[0][1])
    o = hip_ctx.summarize(b, PRESETS["ont_r9_guppy5_sup"], capacity=3, str_capacity=5)
    assert len(o) == 74


def test_malformed_read_is_reported(hip_ctx):
    from pepper_thesis_amd import _ffi
    from pepper_thesis_amd.batch import Read, Region
    ref = b"ACGTACGTACGTACGTACGT"
    bad = Read.make(0, "20M", b"ACGTACGTAC", 20)  # CIGAR consumes 20 bases, read has 10
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.summarize(pack_regions([Region(0, 19, ref, [bad])]), PRESETS["ont_r9_guppy5_sup"])
    assert e.value.code == _ffi.PV_ERR_INVALID


def test_full_size_region_properties(hip_ctx, oracle_lib):
    """BASELINE-size region (R = 100 200, 60x, 10 kb reads): compare with the oracle and check
    size-independent properties: order, determinism, independence of batch composition."""
    reg = synth.synth_region(2024)
    b1 = pack_regions([reg])
    P = PRESETS["ont_r9_guppy5_sup"]
    o1 = hip_ctx.summarize(b1, P, True)
    assert_summary_equal(o1, summary_as_expected(oracle_lib.summarize(b1, P, True)), "full size")
    assert len(o1) > 300
    assert (np.diff(o1.position) >= 0).all()
    # same region twice in one batch == the single-region result, twice (regions are independent)
    reg2 = synth.synth_region(2024, ref_start=reg.ref_start)
    o2 = hip_ctx.summarize(pack_regions([reg, reg2]), P, False)
    n = len(o1)
    assert len(o2) == 2 * n
    np.testing.assert_array_equal(o2.images[:n], o1.images)
    np.testing.assert_array_equal(o2.images[n:], o1.images)
    assert o2.candidates[:n] == o1.candidates and o2.candidates[n:] == o1.candidates
    # run-to-run determinism (atomics only add integers)
    o3 = hip_ctx.summarize(b1, P, False)
    np.testing.assert_array_equal(o3.images, o1.images)


def test_depth_5000_max_reads(hip_ctx, oracle_lib):
    """MAX_READS_IN_REGION = 5000 (Options.py:98): several pair batches per tile, depth clipped to 125"""
    reg = synth.synth_region(77, region_len=700, depth=5000, read_len=600, site_every=40)
    assert len(reg.reads) > 4000
    b = pack_regions([reg])
    P = PRESETS["ont_r9_guppy5_sup"]
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "depth 5000")
    assert int(o.depth.max()) == 125


def test_region_read_limit_of_the_16_bit_planes(hip_ctx, oracle_lib):
    """the per-column counters are 16-bit: a region with 32767 reads (every one over the same columns: coverage 32767, planes
    at -32767) is still exact, one more read is refused with PV_ERR_LIMIT by the host form and reported as status by the
    device-resident form"""
    import torch
    from pepper_thesis_amd import _ffi
    from pepper_thesis_amd.batch import Read, Region
    from pepper_thesis_amd.device import DeviceBatch, DeviceOut
    rng = np.random.default_rng(12)
    R = 120
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R).astype(np.uint8)
    alt = ref.copy()
    alt[60] = ord("A") if ref[60] != ord("A") else ord("C")

    def region(n):
        reads = [Read.make(1000, "%dM" % R, (alt if i % 3 == 0 else ref).tobytes(), 20, is_reverse=bool(i & 1)) for i in range(n)]
        return Region(1000, 1000 + R - 1, ref.tobytes(), reads, 1000, 1000 + R - 1)
    P = PRESETS["ont_r9_guppy5_sup"]
    b = pack_regions([region(32767)])
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "32767 reads")
    assert len(o) == 1 and int(o.images_i32[0, 16, 4]) == -16384   # REFF: the 16384 forward reads (un-clamped plane), exact in 16 bits
    big = pack_regions([region(32768)])
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.summarize(big, P)
    assert e.value.code == _ffi.PV_ERR_LIMIT and "32767" in str(e.value)
    dout = DeviceOut(64, 1024)
    hip_ctx.summarize_dev(DeviceBatch(big), P, dout)
    hip_ctx.synchronize()
    assert dout.status() == _ffi.PV_ERR_LIMIT


def test_every_column_a_site_triggers_workspace_retry(hip_ctx, oracle_lib):
    """far more sites than the default workspace heuristic expects: the host entry point retries with exact bounds"""
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(9)
    R = 40_000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R).astype(np.uint8)
    reads = []
    for i in range(6):
        seq = ref.copy()
        flip = rng.random(R) < 0.5
        seq[flip] = np.frombuffer(b"ACGT", np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", np.uint8), seq[flip]) + 1) % 4]
        reads.append(Read.make(0, "%dM" % R, seq.tobytes(), 20, i % 2 == 0))
    b = pack_regions([Region(0, R - 1, ref.tobytes(), reads)])
    P = PRESETS["ont_r9_guppy5_sup"]
    o = hip_ctx.summarize(b, P, False)
    e = oracle_lib.summarize(b, P, False)
    assert len(o) == len(e) > 30_000
    np.testing.assert_array_equal(o.images, e.images)
    assert o.candidates == e.candidates


def test_more_than_1024_alleles_at_one_site_is_reported(hip_ctx):
    """documented limit of the per-site LDS allele table: PV_ERR_LIMIT, not a crash or a wrong answer"""
    from pepper_thesis_amd import _ffi
    from pepper_thesis_amd.batch import Read, Region
    ref = b"ACGTACGTACGTACGTACGTACGTACGTAC"
    reads = []
    for i in range(1100):
        ins = "".join("ACGT"[(i >> (2 * k)) & 3] for k in range(6))  # 1100 distinct 6-mers
        reads.append(Read.make(0, "15M6I15M", ref[:15] + ins.encode() + ref[15:], 30, i % 2 == 0))
    with pytest.raises(_ffi.PepperHipError) as e:
        hip_ctx.summarize(pack_regions([Region(0, 29, ref, reads)]), PRESETS["ont_r9_guppy5_sup"])
    assert e.value.code == _ffi.PV_ERR_LIMIT


def test_long_indels_and_padded_reference(hip_ctx, oracle_lib):
    """a 3000-base insertion (quality sum over the whole insert, :448-450), a deletion longer than the region
    remainder, and a reference buffer longer than the region (ref_len > R)"""
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(4)
    ref = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=400).astype(np.uint8))
    reads = []
    for i in range(8):
        ins = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=3000).astype(np.uint8))
        reads.append(Read.make(100, "50M3000I50M", ref[100:150] + ins + ref[150:200], 12 + i, i % 2 == 0))
    for i in range(6):
        reads.append(Read.make(100, "120M500D10M", ref[100:220] + b"ACGTACGTAC", 20, i % 2 == 0))
    for i in range(6):
        reads.append(Read.make(90, "200M", ref[90:290], 20, i % 2 == 1))
    b = pack_regions([Region(100, 299, ref[100:], reads)])  # 300 reference bytes for a 200-column region
    for preset in ("ont_r9_guppy5_sup", "hifi"):
        P = PRESETS[preset]
        o = hip_ctx.summarize(b, P, True)
        assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), preset)


@pytest.mark.parametrize("min_q", [0.0, 0.5, 1.0, 17.0, 126.5, 127.0, 128.0, 128.5, 200.0, 255.0, 255.5, 300.0])
def test_quality_bar_anywhere_in_the_byte_range(hip_ctx, oracle_lib, min_q):
    """k_pileup_tiles compares four quality bytes at a time (high bit decides, else the low seven bits): every position of
    the bar relative to 128 and to the ends of the byte range, qualities over all of 0..255, odd bytes among the bases"""
    from dataclasses import replace
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(int(min_q * 2) + 7)
    R = 700
    ref = rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), size=R, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]).astype(np.uint8)
    reads = []
    for i in range(50):
        start = int(rng.integers(0, 200))
        n = int(rng.integers(200, R - start))
        seq = ref[start:start + n].copy()
        seq[(seq >= 97)] -= 32                                       # reads mostly upper case
        flip = rng.random(n) < 0.08
        seq[flip] = rng.choice(np.frombuffer(b"ACGTacgtN*RY", np.uint8), size=int(flip.sum()))
        quals = rng.integers(0, 256, size=n).astype(np.uint8)
        reads.append(Read.make(start, "%dM" % n, seq.tobytes(), quals, i % 2 == 0, 60))
    b = pack_regions([Region(0, R - 1, ref.tobytes(), reads)])
    P = replace(PRESETS["ont_r9_guppy5_sup"], min_snp_baseq=min_q)
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "min_snp_baseq %g" % min_q)


def _mutated(ref, rng, rate):
    seq = ref.copy()
    flip = rng.random(len(seq)) < rate
    seq[flip] = np.frombuffer(b"ACGT", np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", np.uint8), seq[flip]) + 1) % 4]
    return seq


def test_read_over_more_than_256_tiles_takes_the_search_path(hip_ctx, oracle_lib):
    """k_tile_fill keeps a per-wave table of a read's tile boundaries for up to 256 tiles; a read over more of them (a region
    beyond 131 kb) finds its op ranges by binary search and carries no sub-tile index (k_collect then searches the whole range)"""
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(41)
    R = 140_000
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R).astype(np.uint8)
    reads = []
    for i in range(8):
        if i < 4:   # one op over 270 tiles
            reads.append(Read.make(200, "%dM" % (R - 1000), _mutated(ref[200:R - 800], rng, 0.002).tobytes(), 25, i % 2 == 0))
        else:       # many ops over 270 tiles: 900M 1I 900M 2D ...
            cig, seq, pos = [], [], 300
            while pos + 2000 < R - 500:
                cig.append("900M1I900M2D")
                seq.append(_mutated(ref[pos:pos + 900], rng, 0.002).tobytes() + b"A" + _mutated(ref[pos + 900:pos + 1800], rng, 0.002).tobytes())
                pos += 1802
            reads.append(Read.make(300, "".join(cig), b"".join(seq), 25, i % 2 == 0))
    b = pack_regions([Region(0, R - 1, ref.tobytes(), reads)])
    P = PRESETS["ont_r9_guppy5_sup"]
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "270 tiles")
    assert len(o) > 100


def test_read_with_more_than_65535_ops(hip_ctx, oracle_lib):
    """the tile-boundary table holds 16-bit op offsets: a read with more ops than that takes the search path as well"""
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(43)
    n_units = 34_000                       # 1M1D x 34 000 = 68 000 ops over 68 000 columns
    R = 2 * n_units + 400
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R).astype(np.uint8)
    reads = []
    for i in range(5):
        seq = ref[100:100 + 2 * n_units:2].copy()      # the bases under the M ops
        seq = _mutated(seq, rng, 0.01)
        reads.append(Read.make(100, "1M1D" * n_units, seq.tobytes(), 25, i % 2 == 0))
    for i in range(5):                                  # ordinary reads over the same columns
        reads.append(Read.make(50, "%dM" % (R - 100), _mutated(ref[50:R - 50], rng, 0.01).tobytes(), 25, i % 2 == 1))
    b = pack_regions([Region(0, R - 1, ref.tobytes(), reads)])
    P = PRESETS["ont_r9_guppy5_sup"]
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "68 k ops")
    assert len(o) > 100


def test_op_batches_beyond_the_lookup_tables(hip_ctx, oracle_lib):
    """k_pileup_tiles finds an op's pair / a slot's op through owner tables of fixed size (24 k ops per pair batch, 32 k base slots
    per op batch) and searches beyond them: (a) 140 reads that are one long match each (512 slots per op), (b) 140 reads with
    an op on nearly every column (over 100 k ops in a pair batch)"""
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(47)
    R = 1500
    ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R).astype(np.uint8)
    P = PRESETS["ont_r9_guppy5_sup"]
    alt = _mutated(ref, rng, 0.02)          # shared differences -> sites
    reads = [Read.make(0, "%dM" % R, _mutated(alt, rng, 0.01).tobytes(), 25, i % 2 == 0) for i in range(140)]
    b = pack_regions([Region(0, R - 1, ref.tobytes(), reads)])
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "long matches")
    assert len(o) > 10
    reads = []
    for i in range(140):
        n_units = (R - 20) // 2
        seq = bytearray()
        for u in range(n_units):            # 1M1I1M: two reference bases, three read bases
            seq += bytes([int(alt[10 + 2 * u]), ord("ACGT"[(u + (i & 1)) % 4]), int(alt[11 + 2 * u])])
        reads.append(Read.make(10, "1M1I1M" * n_units, bytes(seq), 25, i % 2 == 0))
    # ... and reads with 300 one-base inserts in a row (more than 255 ops inside 64 columns: the sub-tile index of their pair
    # records saturates)
    for i in range(6):
        seq = alt[40:340].tobytes() + b"ACGT" * 75 + alt[340:900].tobytes()
        reads.append(Read.make(40, "300M" + "1I" * 300 + "560M", seq, 25, i % 2 == 0))
    b = pack_regions([Region(0, R - 1, ref.tobytes(), reads)])
    o = hip_ctx.summarize(b, P, True)
    assert_summary_equal(o, summary_as_expected(oracle_lib.summarize(b, P, True)), "dense ops")
    assert len(o) > 10


def test_more_than_8192_tiles_in_a_batch(hip_ctx, oracle_lib):
    """the tile scan is one workgroup looping over 8192-entry passes: 4.4 M columns are 8 594 tiles (two passes)"""
    from pepper_thesis_amd.batch import Read, Region
    rng = np.random.default_rng(53)
    regs = []
    for g in range(44):
        R = 100_000
        ref = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=R).astype(np.uint8)
        start = int(rng.integers(1000, R - 3000))
        alt = _mutated(ref[start:start + 1500], rng, 0.02)
        reads = [Read.make(start, "1500M", _mutated(alt, rng, 0.005).tobytes(), 25, i % 2 == 0) for i in range(6)]
        regs.append(Region(g * 1_000_000, g * 1_000_000 + R - 1, ref.tobytes(),
                           [Read.make(g * 1_000_000 + r.pos, r.cigar, r.bases, r.quals, r.is_reverse, r.mapq) for r in reads]))
    b = pack_regions(regs)
    assert (b.ref.shape[0] + 511) // 512 > 8192
    P = PRESETS["ont_r9_guppy5_sup"]
    o = hip_ctx.summarize(b, P, False)
    e = oracle_lib.summarize(b, P, False)
    assert len(o) == len(e) > 100
    np.testing.assert_array_equal(o.images, e.images)
    np.testing.assert_array_equal(o.position, e.position)
    assert o.candidates == e.candidates
