"""CPU: the restated consumer (find_candidates / VCF records). The reference module is not importable here and
has no fixtures (parity unpinned): these tests pin the restatement's own invariants against the source's rules."""
import numpy as np

from pepper_thesis_amd import find_candidates as fc


def test_repeat_annotation():
    assert fc.repeat_annotation("ACGT", 1) == [1, 1, 1, 1]
    assert fc.repeat_annotation("AAAAAC", 1) == [5, 5, 5, 5, 5, 1]
    assert fc.repeat_annotation("CAAAT", 1) == [1, 3, 3, 3, 1]


def _ref(seq):
    return lambda contig, a, b: seq[max(a, 0):b]


def test_selection_rules():
    seq = "ACGTACGTACGTACGTACGTAAAAAAAGCT" + "ACGT" * 10
    opt = fc.CandidateOptions()
    recs = [
        dict(contig="c", position=8, depth=20, candidates=["1T"], candidate_frequency=[10], prediction=[0.05, 0.9, 0.05]),   # het SNP
        dict(contig="c", position=9, depth=20, candidates=["1G"], candidate_frequency=[3], prediction=[0.95, 0.03, 0.02]),   # below p-value: dropped
        dict(contig="c", position=12, depth=20, candidates=["3ACG"], candidate_frequency=[9], prediction=[0.1, 0.2, 0.7]),   # hom deletion
        dict(contig="c", position=13, depth=20, candidates=["2CTT"], candidate_frequency=[8], prediction=[0.2, 0.7, 0.1]),   # insertion
        dict(contig="c", position=14, depth=20, candidates=["1N"], candidate_frequency=[8], prediction=[0.0, 1.0, 0.0]),     # invalid allele
        dict(contig="c", position=22, depth=20, candidates=["2AA"], candidate_frequency=[8], prediction=[0.86, 0.12, 0.02]),  # in homopolymer: lc threshold 0.15
    ]
    sel = fc.select_candidates(recs, _ref(seq), opt)
    by_pos = {s[1]: s for s in sel}
    assert sorted(by_pos) == [8, 12, 13]
    assert by_pos[8][3:6] == ("A", ["T"], [0, 1])
    assert by_pos[12][3:6] == ("ACG", ["A"], [1, 1]) and by_pos[12][2] == 15      # REF = anchor + deleted, ALT = anchor
    assert by_pos[13][3:5] == ("C", ["CTT"])
    v = fc.dedupe_by_position(sel)
    lines = [l for l, _, _ in fc.variant_records(v, opt)]
    assert lines[0].split("\t")[:7] == ["c", "9", ".", "A", "T", "10", "PASS"]
    f = lines[0].split("\t")
    assert f[8] == "GT:AP:GQ:DP:AD:VAF:REP" and f[9].startswith("0/1:0.9:10:20:10:0.5:0")
    # qual = int(-10 log10(1 - p)): 0.9 -> 10 <= snp_q_cutoff 20 -> selected for re-genotyping
    flags = [(sel_, snp) for _, sel_, snp in fc.variant_records(v, opt)]
    assert flags[0] == (True, True) and flags[1][1] is False


def test_multiallelic_normalisation():
    opt = fc.CandidateOptions()
    a = ("c", 100, 101, "A", ["T"], [0, 1], 30, [12], 0.99, np.array([0.005, 0.99, 0.005]), [0.99], False)
    b = ("c", 100, 103, "ACG", ["A"], [0, 1], 28, [9], 0.98, np.array([0.01, 0.98, 0.01]), [0.98], False)
    contig, rs, re_, ref, alts, gt, depth, sup, gq, naps, rep = fc.candidate_list_to_variant([a, b], opt)
    assert (ref, alts, gt, depth, sup) == ("ACG", ["TCG", "A"], [1, 2], 28, [12, 9])
    assert abs(gq - 0.98) < 1e-12 and re_ == 103


def test_batch_selection_equals_the_per_record_rules():
    """select_candidates_batch (one reference fetch per contig, the homopolymer test on a matrix) gives the tuples of
    select_candidates, record for record: random sites incl. the first and last bases of a contig, homopolymer runs, N and
    lower-case reference, invalid alleles, all three allele types, the frequency rules switched on, bytes and str candidates"""
    import dataclasses
    rng = np.random.default_rng(11)
    seqs = {}
    for name, L in (("c1", 5000), ("c2", 37), ("tiny", 12)):
        s = rng.choice(list("ACGT"), size=L)
        for _ in range(L // 40):                       # plant homopolymer runs of 3..9
            a = int(rng.integers(0, L)); s[a:a + int(rng.integers(3, 10))] = s[a]
        for _ in range(L // 100 + 1):
            s[int(rng.integers(0, L))] = "N"
        s = "".join(s)
        seqs[name] = "".join(ch.lower() if rng.random() < 0.1 else ch for ch in s)

    def get_ref(contig, a, b):   # FASTA_handler.get_reference_sequence: clamped, upper case
        return seqs[contig][max(a, 0):b].upper()
    n = 4000
    contigs = rng.choice(["c1", "c1", "c1", "c2", "tiny"], size=n)
    pos = np.array([int(rng.integers(0, len(seqs[c]) + 3)) for c in contigs])      # a few beyond the end
    pos[:40] = np.concatenate([np.arange(20), len(seqs["c1"]) - 1 - np.arange(20)]); contigs[:40] = "c1"
    types = rng.choice(["1", "2", "3", "4"], size=n, p=[0.4, 0.28, 0.28, 0.04])
    alleles = ["".join(rng.choice(list("ACGTN"), size=int(rng.integers(0, 5)), p=[.24, .24, .24, .24, .04])) for _ in range(n)]
    cand = np.array([[t + a] for t, a in zip(types, alleles)], dtype=object)
    pred = rng.random((n, 3)) ** 3
    pred /= pred.sum(1, keepdims=True)
    pred[::17] = [0.5, 0.25, 0.25]                     # ties and threshold edges
    pred[::19] = [0.9, 0.1, 0.0]
    batch = dict(contigs=np.array([c.encode() for c in contigs]), positions=pos.astype(np.int32),
                 depths=rng.integers(1, 90, n).astype(np.uint8), candidates=cand,
                 candidate_frequency=rng.integers(0, 60, (n, 1)).astype(np.uint8), base_prediction=pred)
    recs = [dict(contig=str(contigs[i]), position=int(pos[i]), depth=int(batch["depths"][i]), candidates=[str(cand[i, 0])],
                 candidate_frequency=[int(batch["candidate_frequency"][i, 0])], prediction=pred[i]) for i in range(n)]

    def same(a, b):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x[:8] == y[:8] and x[11] == y[11], (x, y)
            assert float(x[8]) == float(y[8]) and np.array_equal(x[9], y[9]) and [float(v) for v in x[10]] == [float(v) for v in y[10]]
    for opt in (fc.CandidateOptions(), dataclasses.replace(fc.CANDIDATE_PRESETS["ont_r9_guppy4_hac"], report_snp_above_freq=0.3, report_indel_above_freq=0.4),
                fc.CANDIDATE_PRESETS["hifi"]):
        exp = fc.select_candidates(recs, get_ref, opt)
        got = fc.select_candidates_batch(batch, get_ref, opt)
        assert len(exp) > 500
        same(got, exp)
        bb = dict(batch, candidates=np.array([[c[0].encode()] for c in cand], dtype=object))     # vlen strings read back as bytes
        same(fc.select_candidates_batch(bb, get_ref, opt), exp)
        # and the records that follow are the same text
        la = [l for l, _, _ in fc.variant_records(fc.dedupe_by_position(got), opt)]
        lb = [l for l, _, _ in fc.variant_records(fc.dedupe_by_position(exp), opt)]
        assert la == lb and len(la) > 300
    # several candidates per window, or a zero depth: the per-record path, same answers
    two = dict(batch, candidates=np.concatenate([cand, cand], axis=1), candidate_frequency=np.concatenate([batch["candidate_frequency"]] * 2, axis=1))
    recs2 = [dict(r, candidates=r["candidates"] * 2, candidate_frequency=r["candidate_frequency"] * 2) for r in recs]
    same(fc.select_candidates_batch(two, get_ref, fc.CandidateOptions()), fc.select_candidates(recs2, get_ref, fc.CandidateOptions()))


def test_single_record_sites_take_the_short_path_to_the_same_variant():
    opt = fc.CandidateOptions()
    rng = np.random.default_rng(12)
    for _ in range(200):
        p = rng.random(3); p /= p.sum()
        g = int(np.argmax(p))
        c = ("c", 10, 11, "A", ["T"], ([0, 0], [0, 1], [1, 1])[g], 30, [12], p[g], p, [max(p[1], p[2])], bool(rng.integers(2)))
        import dataclasses
        slow = fc.candidate_list_to_variant([c, c][:1] * 1, dataclasses.replace(opt, allowed_multiallelics=4))
        # the general path, reached by handing the same record over as a two-element list whose second entry is cut off
        general = fc.candidate_list_to_variant([c, c], dataclasses.replace(opt, allowed_multiallelics=1))
        assert slow[:8] == general[:8] and float(slow[8]) == float(general[8]) and slow[10] == general[10]
        assert [float(v) for v in slow[9]] == [float(v) for v in general[9]]
