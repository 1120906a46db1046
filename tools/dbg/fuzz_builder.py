"""Diagnostic (GPU box, uses the test-side oracle): the image builder against the C oracle on many random region batches - sizes,
depths, read lengths, error rates, N rates, presets, lower-case reference - until the time budget is spent. Prints the first
mismatch and stops."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pepper_thesis_amd import runtime, synth
from pepper_thesis_amd.batch import PRESETS, pack_regions
from oracle import oracle
from golden_io import assert_summary_equal, summary_as_expected
oracle.build()
ctx = runtime.Context(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
t0 = time.time(); n = 0; windows = 0; last_note = t0
while time.time() - t0 < budget:
    seed = seed0 + n
    rng = np.random.default_rng(seed)
    preset = list(PRESETS)[int(rng.integers(len(PRESETS)))]
    big = len(sys.argv) > 3 and sys.argv[3] == "big"   # longer regions (many tiles), sparser / denser sites, longer reads
    regs = [synth.synth_region(7000 + 31 * seed + k, region_len=int(rng.integers(40, 60000 if big else 9000)), depth=int(rng.integers(1, 60 if big else 140)),
                               read_len=int(rng.integers(30, 20000 if big else 4000)), site_every=int(rng.choice([8, 40, 300, 3000, 50000]) if big else rng.integers(8, 300)),
                               n_rate=float(rng.choice([0.0, 0.002, 0.02])), ref_n_rate=float(rng.choice([0.0, 0.0, 0.01])),
                               mismatch=float(rng.choice([0.0, 0.03, 0.1])), ins_rate=float(rng.choice([0.0, 0.02, 0.08])),
                               del_rate=float(rng.choice([0.0, 0.03, 0.08])))
            for k in range(int(rng.integers(1, 7)))]
    if max(len(r.reads) for r in regs) > 32767:   # beyond the 16-bit counters: refused by the builder (tested elsewhere)
        seed0 += 1
        continue
    batch = pack_regions(regs)
    form = n % 3   # the 26-plane builder, the haplotag-aware one, the polisher's
    what = "seed %d preset %s form %d" % (seed, preset, form)
    if form == 0:
        o = ctx.summarize(batch, PRESETS[preset], True)
        assert_summary_equal(o, summary_as_expected(oracle.summarize(batch, PRESETS[preset], True)), what)
        windows += len(o)
    elif form == 1:
        import cases
        from pepper_thesis_amd.batch import hp_params
        bh = pack_regions(cases.tag_reads(regs, seed))
        P = hp_params(PRESETS[preset])
        o = ctx.summarize_hp(bh, P, True)
        assert_summary_equal(o, summary_as_expected(oracle.summarize_hp(bh, P, True)), what)
        windows += len(o)
    else:
        L, O = [(1000, 50), (64, 7), (16, 3)][int(rng.integers(3))]
        got, exp = ctx.polish_summarize(batch, L, O, want_flat=True), oracle.polish_summarize(batch, L, O)
        for f in ("images", "position", "index", "region", "chunk_id", "flat_images", "flat_position", "flat_index", "region_row_off"):
            a, b = getattr(got, f), getattr(exp, f)
            assert (a is None) == (b is None) and (a is None or (a.shape == b.shape and np.array_equal(a, b))), (what, f)
        windows += len(got.images)
    n += 1
    if time.time() - last_note > 50:   # (a silent run is taken for a hung one on the GPU box)
        last_note = time.time()
        print("  ... %d batches, %d windows / chunks so far" % (n, windows), flush=True)
print("fuzz_builder: %d random batches (%d windows / chunks; image builder, haplotag-aware builder, polisher builder in turn) identical "
      "to the oracle in %.0f s (seeds %d..%d)" % (n, windows, time.time() - t0, seed0, seed0 + n - 1))
