"""Hand-built and seeded regions shared by the golden-fixture generator and the parity tests.

The edge cases are the ones SURVEY.md section 8(c) lists for the image builder; each names the
reference behaviour (region_summary.cpp line) it exercises.
"""
import numpy as np

from pepper_thesis_amd import synth
from pepper_thesis_amd.batch import PRESETS, Params, Read, Region, pack_regions


def _flip(seq: bytes, i: int, b: str) -> bytes:
    s = bytearray(seq)
    s[i] = ord(b)
    return bytes(s)


def kat1_snp():
    """SURVEY Appendix D KAT-1"""
    ref = ("ACGT" * 19).encode()
    reads = [Read.make(0, "76M", _flip(ref, 40, "T") if i < 3 else ref, 20, is_reverse=(i % 2 == 1)) for i in range(6)]
    return Region(0, 75, ref, reads)


def kat23_indel():
    """SURVEY Appendix D KAT-2 / KAT-3"""
    ref = ("ACGTTGCA" * 10).encode()
    reads = []
    for i in range(8):
        if i < 4:
            seq, cg = ref[:31] + ref[33:], "31M2D47M"
        elif i < 7:
            seq, cg = ref[:51] + b"GG" + ref[51:], "51M2I29M"
        else:
            seq, cg = ref, "80M"
        reads.append(Read.make(1000, cg, seq, 20, is_reverse=(i % 2 == 1)))
    return Region(1000, 1079, ref, reads)


def kat4_clamp():
    """SURVEY KAT-4: depth>125 (clip before the frequency test, :682-689), clamp of planes 11-24 only
    (:648-653) and int8 wrap of the unclamped planes (DataStore.py:68)."""
    ref = ("ACGT" * 20).encode()
    reads = [Read.make(0, "80M", _flip(ref, 40, "G") if i < 60 else ref, 20) for i in range(300)]
    return Region(0, 79, ref, reads)


def multiallelic():
    """three SNP alleles + an insert + a delete on one site; set<string> order (:670)"""
    ref = ("GATTACAGATTACAGGCCTTAA" * 5).encode()
    reads = []
    p = 50
    for i in range(24):
        rv = i % 3 == 0
        if i < 5:
            reads.append(Read.make(200, "110M", _flip(ref, p, "C" if ref[p:p + 1] != b"C" else "A"), 30, rv))
        elif i < 9:
            reads.append(Read.make(200, "110M", _flip(ref, p, "T" if ref[p:p + 1] != b"T" else "G"), 30, rv))
        elif i < 12:
            reads.append(Read.make(200, "110M", _flip(ref, p, "N"), 30, rv))
        elif i < 16:
            reads.append(Read.make(200, "51M3I59M", ref[:51] + b"TTT" + ref[51:], 30, rv))
        elif i < 19:
            reads.append(Read.make(200, "51M1I59M", ref[:51] + b"T" + ref[51:], 30, rv))
        elif i < 22:
            reads.append(Read.make(200, "51M4D55M", ref[:51] + ref[55:], 30, rv))
        else:
            reads.append(Read.make(200, "110M", ref, 30, rv))
    return Region(200, 309, ref, reads)


def softclip_refskip():
    """S inside the read, N (REF_SKIP falls through to SOFT_CLIP: read_index += len, :556-561),
    P, H, = and X ops, unknown op 9"""
    ref = ("ACGTACGTTTGACCA" * 8).encode()
    reads = []
    for i in range(10):
        rv = i % 2 == 0
        if i < 3:
            # 5S kept inside the read, then aligned
            reads.append(Read.make(500, "20M5S30M", ref[:20] + b"GGGGG" + ref[20:50], 25, rv))
        elif i < 6:
            # REF_SKIP of 10: reference jumps 10, AND 10 read bases are skipped (fall-through)
            reads.append(Read.make(500, "20M10N40M", ref[:20] + b"A" * 10 + ref[30:70], 25, rv))
        elif i < 8:
            reads.append(Read.make(500, "10=1X9=2H3P20M", ref[:10] + b"N" + ref[11:20] + b"CCC" + ref[20:40], 25, rv))
        else:
            cg = np.asarray([(30 << 4) | 0, (4 << 4) | 9, (30 << 4) | 0], dtype=np.uint32)
            reads.append(Read.make(500, cg, ref[:60], 25, rv))
    # a few mismatching reads so that sites exist
    for i in range(4):
        reads.append(Read.make(500, "120M", _flip(_flip(ref, 25, "T"), 60, "G"), 25, i % 2 == 0))
    return Region(500, 619, ref, reads)


def edges_and_clipping():
    """reads starting before the region (:361-365), ending after it (break only at op boundaries,
    :355), delete running past the region end (substr truncation, :500), delete longer than the
    window (rows stop at 31, :885), candidates in the first/last 16 columns (zero rows, :835),
    mapq==0 read (:619), low-quality bases and the insert anchor coverage rule (:452-454)."""
    ref = ("TTGACCAGTAGGCATCA" * 6).encode()  # 102
    R = len(ref)
    reads = []
    long_ref = b"ACGTA" * 4 + ref + b"GGCAT" * 4
    for i in range(8):
        rv = i % 2 == 1
        # starts 20 before the region and ends 20 after it
        seq = _flip(long_ref, 20 + 3, "A" if ref[3:4] != b"A" else "C")
        seq = _flip(seq, 20 + R - 2, "A" if ref[R - 2:R - 1] != b"A" else "C")
        reads.append(Read.make(3000 - 20, "%dM" % len(long_ref), seq, 22, rv))
    for i in range(5):
        # 40-base deletion anchored at column 30 (longer than the 16 rows right of the centre)
        reads.append(Read.make(3000, "31M40D%dM" % (R - 71), ref[:31] + ref[71:], 22, i % 2 == 0))
    for i in range(4):
        # deletion that runs past the region end: anchor at R-3, length 10
        reads.append(Read.make(3000, "%dM10D5M" % (R - 2), ref[:R - 2] + b"ACGTA", 22, i % 2 == 0))
    for i in range(4):
        # 70-base deletion: key longer than 61 -> counted on plane 13/24 but never an allele (:511)
        reads.append(Read.make(3000 + 10, "5M70D10M", ref[10:15] + ref[85:95], 22, i % 2 == 0))
    for i in range(3):
        # insert of 60 bases: 1+61 = 62 > 61 -> rejected; insert of 59 -> accepted (:461)
        reads.append(Read.make(3000, "20M60I20M", ref[:20] + b"A" * 60 + ref[20:40], 22, False))
        reads.append(Read.make(3000, "20M59I20M", ref[:20] + b"C" * 59 + ref[20:40], 22, True))
    # mapq 0 read carrying a private SNP: must be ignored
    reads.append(Read.make(3000, "%dM" % R, _flip(ref, 50, "A" if ref[50:51] != b"A" else "C"), 22, False, mapq=0))
    reads.append(Read.make(3000, "%dM" % R, _flip(ref, 50, "A" if ref[50:51] != b"A" else "C"), 22, True, mapq=0))
    # insert whose anchor base has quality 0 (< min_snp_baseq) but whose mean quality passes
    for i in range(4):
        q = np.full(45, 22, np.uint8)
        q[19] = 0
        reads.append(Read.make(3000 + 40, "20M5I20M", ref[40:60] + b"GATTA" + ref[60:80], q, i % 2 == 0))
    # low-quality insert (rejected by the quality rule) and low-quality mismatches (not counted)
    for i in range(3):
        q = np.full(44, 22, np.uint8)
        q[20:24] = 0
        q[19] = 0
        reads.append(Read.make(3000 + 40, "20M4I20M", ref[40:60] + b"TTTT" + ref[60:80], q, i % 2 == 0))
    return Region(3000, 3000 + R - 1, ref, reads, 3000, 3000 + R - 1)


def nonacgt_reference():
    """reference N/lower-case columns: get_feature_index returns -1 (:201-229); SNP alleles still
    counted (:394-421). Candidates ON such a column are UB in the reference, so the
    candidate range excludes them; neighbours still copy those rows."""
    ref = bytearray(("CATGGTACCA" * 9).encode())
    ref[44] = ord("N")
    ref[45] = ord("N")
    ref = bytes(ref)
    reads = []
    for i in range(12):
        seq = ref.replace(b"N", b"A")
        if i < 6:
            seq = _flip(seq, 40, "T" if ref[40:41] != b"T" else "G")
            seq = _flip(seq, 48, "T" if ref[48:49] != b"T" else "G")
        reads.append(Read.make(7000, "90M", seq, 15, i % 2 == 0))
    for i in range(4):
        reads.append(Read.make(7000, "44M3D43M", ref[:44].replace(b"N", b"A") + ref[47:], 15, i % 2 == 0))
    return Region(7000, 7089, ref, reads, 7000, 7043)


def deep_wrap():
    """>128 same-strand reads: unclamped planes 4, 8-10 drop below -128 and wrap in int8 (Q18);
    plane 25 (*REV) is NOT clamped while 14 (*FRW) is (:648)."""
    ref = ("ACGGT" * 16).encode()
    reads = []
    for i in range(150):
        reads.append(Read.make(100, "80M", _flip(ref, 30, "T") if i % 3 == 0 else ref, 30, False))
    for i in range(140):
        # reverse reads with a 1-base deletion at column 50 -> plane 25 (*REV) = -140 unclamped... and 24 clamped
        reads.append(Read.make(100, "50M1D29M", ref[:50] + ref[51:], 30, True))
    return Region(100, 179, ref, reads)


def empty_and_tiny():
    """a region with no reads, a region whose reads are all mapq 0, a 1-column region"""
    ref = b"ACGTACGTAC"
    r0 = Region(10, 19, ref, [])
    r1 = Region(10, 19, ref, [Read.make(10, "10M", b"ACGTTCGTAC", 20, False, mapq=0) for _ in range(5)])
    r2 = Region(14, 14, b"T", [Read.make(12, "5M", b"GTCCG", 20, i % 2 == 0) for i in range(4)])
    return [r0, r1, r2]


EDGE_CASES = {
    "kat1_snp": lambda: [kat1_snp()],
    "kat23_indel": lambda: [kat23_indel()],
    "kat4_clamp": lambda: [kat4_clamp()],
    "multiallelic": lambda: [multiallelic()],
    "softclip_refskip": lambda: [softclip_refskip()],
    "edges_and_clipping": lambda: [edges_and_clipping()],
    "nonacgt_reference": lambda: [nonacgt_reference()],
    "deep_wrap": lambda: [deep_wrap()],
    "empty_and_tiny": empty_and_tiny,
}


def edge_batch(name):
    return pack_regions(EDGE_CASES[name]())


def all_edges_batch():
    regs = []
    for name in EDGE_CASES:
        regs.extend(EDGE_CASES[name]())
    return pack_regions(regs)


# seeded random regions committed as golden fixtures: (seed, kwargs, preset)
GOLDEN_RANDOM = [
    (11, dict(region_len=2000, depth=25, read_len=700, site_every=60), "ont_r9_guppy5_sup"),
    (12, dict(region_len=2000, depth=25, read_len=700, site_every=60, n_rate=0.002), "ont_r9_guppy4_hac"),
    (13, dict(region_len=1500, depth=30, read_len=1500, site_every=80, mismatch=0.002, ins_rate=0.002, del_rate=0.002), "hifi"),
    (14, dict(region_len=1500, depth=20, read_len=400, site_every=50), "ont_r10_q20"),
    (15, dict(region_len=1200, depth=35, read_len=600, site_every=40, mismatch=0.08, ins_rate=0.05, del_rate=0.05), "clr"),
    (16, dict(region_len=1800, depth=150, read_len=900, site_every=70), "ont_r9_guppy5_sup"),
]


def random_batch(seed, kw):
    return pack_regions([synth.synth_region(seed, **kw)])


# ---- haplotag-aware builder (region_summary_hp.cpp): the same regions with an HP tag on every read -------------------
# tags 0 (untagged), 1, 2 and the values a real BAM never carries but the reference still has a defined answer for (3, -1)
HP_TAG_CHOICES = (0, 0, 1, 2, 1, 2, 3, -1)


def tag_reads(regions, seed, choices=HP_TAG_CHOICES):
    rng = np.random.default_rng(seed)
    for r in regions:
        for rd in r.reads:
            rd.hp_tag = int(rng.choice(choices))
    return regions


def hp_edge_batch(name, seed=5):
    return pack_regions(tag_reads(EDGE_CASES[name](), seed))


def hp_all_edges_batch(seed=5):
    regs = []
    for name in EDGE_CASES:
        regs.extend(EDGE_CASES[name]())
    return pack_regions(tag_reads(regs, seed))


def hp_known_answer():
    """hand-checkable: 8 reads over ACGTACGTAC...; column 40 (ref 'A') carries a T in three reads (one per tag 0, 1, 2)"""
    ref = (b"ACGT" * 30)[:100]
    reads = []
    for i, (hp, rev, alt) in enumerate([(0, False, True), (1, False, True), (2, True, True), (0, True, False),
                                        (1, False, False), (2, False, False), (1, True, False), (2, True, False)]):
        seq = _flip(ref, 40, "T") if alt else ref
        reads.append(Read.make(0, "100M", seq, 30, rev, hp_tag=hp))
    return Region(0, 99, ref, reads)


def hp_random_batch(seed, kw):
    return pack_regions(tag_reads([synth.synth_region(seed, **kw)], seed))
