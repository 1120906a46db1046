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


# ---- hardening: sizes taken from the file are validated; long CIGARs come from the CG tag ----------------------------------

def _small_files(tmp_path, recs, n=4000, name="h"):
    rng = np.random.default_rng(1)
    seq = "".join(rng.choice(list("ACGT"), size=n))
    fa, bam = str(tmp_path / (name + ".fa")), str(tmp_path / (name + ".bam"))
    bw.write_fasta(fa, [("c1", seq)])
    bw.write_bam(bam, [("c1", n)], recs)
    return fa, bam, seq


def test_cg_tag_long_cigar(tmp_path):
    """a read whose real CIGAR sits in CG:B,I behind the <l_seq>S<rlen>N placeholder is returned exactly like the same read
    with an inline CIGAR (htslib resolves the tag in bam_read1, so the reference never sees the placeholder)"""
    build.build_io()
    rng = np.random.default_rng(2)
    recs = bw.random_records(rng, 30, 4000, tid=0, mean_len=600, allow_skip=False)
    for r in recs:
        r["flag"], r["mapq"] = r["flag"] & 0x10, 60
    import copy
    tagged = copy.deepcopy(recs)
    for r in tagged[::2]:
        r["cg"] = True
    fa, bam1, _ = _small_files(tmp_path, recs, name="inline")
    _, bam2, _ = _small_files(tmp_path, tagged, name="cgtag")
    a = bamio.BamHandler(bam1).get_reads("c1", 500, 3500, False, 0, 0)
    b = bamio.BamHandler(bam2).get_reads("c1", 500, 3500, False, 0, 0)
    assert len(a) == len(b) > 10
    for x, y in zip(a, b):
        assert (x.pos, x.pos_end, x.bases, x.quals.tolist(), x.cigar.tolist()) == (y.pos, y.pos_end, y.bases, y.quals.tolist(), y.cigar.tolist())
    exp = bw.expected_reads(recs, 0, 500, 3500, False, 0)
    assert [e["seq"] for e in exp] == [y.bases.decode() for y in b]


def test_truncated_and_corrupt_inputs_fail_cleanly(tmp_path):
    build.build_io()
    rng = np.random.default_rng(4)
    recs = bw.random_records(rng, 200, 4000, tid=0, mean_len=500, allow_skip=False)
    fa, bam, _ = _small_files(tmp_path, recs)
    raw = open(bam, "rb").read()
    bai = open(bam + ".bai", "rb").read()
    # truncated BAI: every prefix either loads or is refused with a message, never crashes
    for cut in (6, 9, 13, 21, 40, len(bai) // 2, len(bai) - 3):
        p = str(tmp_path / ("t%d.bam" % cut))
        open(p, "wb").write(raw)
        open(p + ".bai", "wb").write(bai[:cut])
        with pytest.raises(IOError):
            bamio.BamHandler(p)
    # truncated BAM body: reads up to the cut come back or a clean error is raised
    p = str(tmp_path / "cut.bam")
    open(p, "wb").write(raw[: len(raw) * 2 // 3])
    open(p + ".bai", "wb").write(bai)
    try:
        bamio.BamHandler(p).get_reads("c1", 0, 4000, True, 0, 0)
    except IOError as e:
        assert "BGZF" in str(e) or "BAM" in str(e) or "truncated" in str(e)
    # a record whose l_seq / n_cigar claim more bytes than block_size holds is refused (no read past the record)
    bad = [dict(r) for r in recs[:5]]
    p2 = str(tmp_path / "bad.bam")
    bw.write_bam(p2, [("c1", 4000)], bad)
    body = bytearray(bgzf_plain(open(p2, "rb").read()))
    # first alignment record starts after the header: find it through the BAI-independent layout
    import struct
    l_text = struct.unpack_from("<I", body, 4)[0]
    off = 8 + l_text
    n_ref = struct.unpack_from("<I", body, off)[0]
    off += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<I", body, off)[0]
        off += 4 + ln + 4
    struct.pack_into("<I", body, off + 4 + 16, 0x00FFFFFF)   # l_seq far beyond block_size
    p3 = str(tmp_path / "bad2.bam")
    # same block layout as the writer's (header block, then the records), so that the index offsets still apply
    open(p3, "wb").write(bw._bgzf_block(bytes(body[:off])) + bw._bgzf_block(bytes(body[off:])) + bw.BGZF_EOF)
    open(p3 + ".bai", "wb").write(open(p2 + ".bai", "rb").read())
    with pytest.raises(IOError) as ei:
        bamio.BamHandler(p3).get_reads("c1", 0, 4000, True, 0, 0)
    assert "corrupt BAM record" in str(ei.value)


def bgzf_plain(raw: bytes) -> bytes:
    import zlib
    out, p = bytearray(), 0
    while p < len(raw):
        bsize = int.from_bytes(raw[p + 16:p + 18], "little") + 1
        out += zlib.decompress(raw[p + 18:p + bsize - 8], -15)
        p += bsize
    return bytes(out)


def test_full_64k_blocks_and_native_writer_roundtrip(tmp_path):
    """the native writer's BAM (blocks filled to the limit) read back by the native reader equals the restated clipping"""
    build.build_io()
    from pepper_thesis_amd.batch import Read, Region, pack_regions
    rng = np.random.default_rng(8)
    recs = bw.random_records(rng, 400, 50_000, tid=0, mean_len=3000, allow_skip=False)
    for r in recs:
        r["flag"], r["mapq"], r["hp"] = r["flag"] & 0x10, 60, None
        r["seq"] = r["seq"].replace("N", "A")
    reads = [Read.make(r["pos"], bw_pack(r["cigar"]), r["seq"], r["qual"], bool(r["flag"] & 0x10), r["mapq"]) for r in recs]
    b = pack_regions([Region(0, 49_999, b"A" * 50_000, reads)])
    path = str(tmp_path / "native.bam")
    bamio.write_bam(path, [("c1", 50_000)], np.zeros(len(reads), np.int32), b, level=1)
    got = bamio.BamHandler(path).get_reads("c1", 10_000, 40_000, False, 0, 0)
    exp = bw.expected_reads(recs, 0, 10_000, 40_000, False, 0)
    assert len(got) == len(exp) > 50
    for g, e in zip(got, exp):
        assert (g.pos, g.pos_end, g.bases.decode(), g.quals.tolist()) == (e["pos"], e["pos_end"], e["seq"], e["qual"])


def test_fill_batch_equals_per_read_path_and_reservoir(files):
    """pvio_fill_batch (flat arrays for a list of intervals in one native call) == region_from_files per interval; the native
    reservoir indices equal NumPy's legacy RandomState stream"""
    from pepper_thesis_amd.batch import RegionBatch, pack_regions
    from pepper_thesis_amd.make_images import downsample_indices
    for n, rate in ((6000, 1.0), (100, 0.5), (5001, 1.0), (7, 0.0), (12000, 0.8), (10, 1.0)):
        np.testing.assert_array_equal(bamio.reservoir_indices(n, rate), downsample_indices(n, rate))
    b, f = bamio.BamHandler(files["dna_bam"]), bamio.FastaHandler(files["fa"])
    ivs = [("chr20", 20_000, 30_000), ("chr20", 30_000, 40_000), ("chrM", 0, 5000), ("chr20", 120_000, 129_999)]
    for rate in (1.0, 0.3):
        fb = bamio.fill_batch(b, f, ivs, 5, False, rate)
        regs = [bamio.region_from_files(b, f, c, a, e, 5, False, rate) for c, a, e in ivs]
        keep = [i for i, r in enumerate(regs) if r.reads]
        ref = pack_regions([regs[i] for i in keep])
        assert fb.interval_index.tolist() == keep
        for fld in RegionBatch.FIELDS:
            np.testing.assert_array_equal(getattr(ref, fld), getattr(fb.batch, fld), err_msg=fld)
        # the HP tags travel next to the batch (input of the haplotag-aware builder)
        assert ref.read_hp is not None and set(np.unique(ref.read_hp).tolist()) == {0, 1, 2}
        np.testing.assert_array_equal(ref.read_hp, fb.batch.read_hp)
        fb.close()


def test_vcf_gz_and_tabix_roundtrip(tmp_path):
    build.build_io()
    lines = ["##fileformat=VCFv4.2", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    for c in ("chr1", "chr2"):
        for p in range(100, 400_000, 37):
            lines.append("%s\t%d\t.\tAC\tA\t30\tPASS\t." % (c, p))
    text = "\n".join(lines) + "\n"
    path = str(tmp_path / "x.vcf.gz")
    bamio.write_vcf_gz(path, text)
    import gzip
    assert gzip.open(path, "rb").read().decode() == text          # a BGZF file is a valid multi-member gzip file
    assert bamio.bgzf_read_all(path).decode() == text
    tbi = bamio.bgzf_read_all(path + ".tbi")
    import struct
    assert tbi[:4] == b"TBI\x01"
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", tbi, 4)
    assert (n_ref, fmt, col_seq, col_beg, col_end, meta, skip) == (2, 2, 1, 2, 0, ord("#"), 0)
    assert tbi[36:36 + l_nm] == b"chr1\x00chr2\x00"
    # walk the index: every chunk's virtual offsets point into the file and chunk starts are record starts
    off = 36 + l_nm
    raw = open(path, "rb").read()
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", tbi, off)[0]; off += 4
        for _ in range(n_bin):
            _bin, n_chunk = struct.unpack_from("<Ii", tbi, off); off += 8
            for _ in range(n_chunk):
                beg, end = struct.unpack_from("<QQ", tbi, off); off += 16
                assert (beg >> 16) < len(raw) and beg < end
        n_intv = struct.unpack_from("<i", tbi, off)[0]; off += 4 + 8 * n_intv
    assert off == len(tbi)


def test_libdeflate_and_zlib_backends_are_byte_identical(tmp_path):
    """BGZF blocks inflate through libdeflate (dlopen) when the host has it, zlib otherwise: the same reads, byte for byte, and
    the same inflated stream; PEPPER_INFLATE / pvio_set_inflate_backend switch"""
    build.build_io()
    rng = np.random.default_rng(9)
    recs = bw.random_records(rng, 400, 8000, tid=0, mean_len=700, allow_skip=False)
    fa, bam, _ = _small_files(tmp_path, recs)
    got = {}
    have = bamio.set_inflate_backend(True)
    for use in ([True] if have else []) + [False]:
        assert bamio.set_inflate_backend(use) == (use and have)
        assert bamio.inflate_backend() == ("libdeflate" if use and have else "zlib")
        reads = bamio.BamHandler(bam).get_reads("c1", 0, 8000, True, 0, 0)
        got[use] = ([(r.pos, r.pos_end, r.bases, r.quals.tolist(), r.cigar.tolist(), r.mapq) for r in reads], bamio.bgzf_read_all(bam))
    assert len(got[False][0]) > 100
    if have:
        assert got[True] == got[False]
    bamio.set_inflate_backend(True)


def test_corrupt_block_in_the_middle_of_a_query_is_an_error(tmp_path):
    """a flipped byte inside a BGZF block's payload: the CRC32 trailer (or the inflate) catches it and the region query FAILS - it
    must not come back with the reads before the damage as if the file ended there (both inflate backends)"""
    build.build_io()
    rng = np.random.default_rng(10)
    recs = bw.random_records(rng, 3000, 60000, tid=0, mean_len=900, allow_skip=False)
    fa, bam, _ = _small_files(tmp_path, recs)
    raw = bytearray(open(bam, "rb").read())
    # walk the blocks; damage one in the middle of the file
    offs, p = [], 0
    while p < len(raw):
        offs.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    assert len(offs) >= 6
    victim = offs[len(offs) // 2]
    n_good = len(bamio.BamHandler(bam).get_reads("c1", 0, 60000, True, 0, 0))
    for kind in ("payload", "crc"):
        bad = bytearray(raw)
        if kind == "payload":
            bad[victim + 18 + 40] ^= 0x5A
        else:
            nxt = victim + int.from_bytes(raw[victim + 16:victim + 18], "little") + 1
            bad[nxt - 8] ^= 0xFF      # first byte of the CRC32 trailer
        pth = str(tmp_path / ("bad_%s.bam" % kind))
        open(pth, "wb").write(bytes(bad))
        open(pth + ".bai", "wb").write(open(bam + ".bai", "rb").read())
        for use in (True, False):
            bamio.set_inflate_backend(use)
            with pytest.raises(IOError) as ei:
                bamio.BamHandler(pth).get_reads("c1", 0, 60000, True, 0, 0)
            assert "BGZF" in str(ei.value) or "inflate" in str(ei.value) or "corrupt" in str(ei.value)
    bamio.set_inflate_backend(True)
    assert n_good > 1000


def test_inflate_helper_threads_give_the_same_reads_and_the_same_errors(tmp_path):
    """pvio_bam_set_threads: helper threads inflate blocks ahead of the reading thread. The reads of many queries (forward,
    backward, overlapping: the read-ahead ring is restarted, skipped into and run to the end of the file) are identical to the
    handle without helpers; a damaged block and a truncated file are errors with helpers as without."""
    build.build_io()
    rng = np.random.default_rng(12)
    recs = bw.random_records(rng, 3000, 60000, tid=0, mean_len=900, allow_skip=False)
    fa, bam, _ = _small_files(tmp_path, recs)
    queries = [(0, 60000), (30000, 31000), (1000, 20000), (59000, 60000), (0, 500), (15000, 45000), (44000, 60000), (100, 200)]

    def run(handle):
        out = []
        for a, b_ in queries:
            out.append([(r.pos, r.pos_end, r.bases, r.quals.tolist(), r.cigar.tolist(), r.mapq, r.query_name) for r in handle.get_reads("c1", a, b_, True, 0, 0)])
        return out
    base = bamio.BamHandler(bam)
    exp = run(base)
    assert len(exp[0]) > 1000
    for n in (1, 3, 6):
        h = bamio.BamHandler(bam)
        assert h.set_threads(n) == n
        assert run(h) == exp
        assert h.set_threads(0) == 0 and run(h) == exp      # back to the plain reader on the same handle
        assert h.set_threads(2) == 2 and run(h)[3] == exp[3]
        h.close()
    f = bamio.FastaHandler(fa)
    a = bamio.fill_batch(base, f, [("c1", 1000, 30000), ("c1", 30000, 60000)], 0, True, 1.0, 100)
    h = bamio.BamHandler(bam)
    h.set_threads(3)
    b = bamio.fill_batch(h, f, [("c1", 1000, 30000), ("c1", 30000, 60000)], 0, True, 1.0, 100)
    for k in ("bases", "quals", "cigar", "read_pos", "base_off", "cigar_off", "read_off", "ref"):
        assert np.array_equal(getattr(a.batch, k), getattr(b.batch, k)), k
    assert b.t_helpers > 0 and a.t_helpers == 0
    a.close(); b.close(); h.close()
    # a damaged block in the middle, and a file cut inside a block
    raw = bytearray(open(bam, "rb").read())
    offs, p = [], 0
    while p < len(raw):
        offs.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    bad = bytearray(raw)
    bad[offs[len(offs) // 2] + 18 + 40] ^= 0x5A
    cut = raw[:offs[len(offs) // 2] + 100]
    for name, data, words in (("bad.bam", bad, ("BGZF", "inflate", "corrupt")), ("cut.bam", cut, ("truncated",))):
        pth = str(tmp_path / name)
        open(pth, "wb").write(bytes(data))
        open(pth + ".bai", "wb").write(open(bam + ".bai", "rb").read())
        for n in (0, 3):
            h = bamio.BamHandler(pth)
            h.set_threads(n)
            with pytest.raises(IOError) as ei:
                h.get_reads("c1", 0, 60000, True, 0, 0)
            assert any(w in str(ei.value) for w in words), str(ei.value)
            assert [r.pos for r in h.get_reads("c1", 0, 500, True, 0, 0)] == [e[0] for e in exp[4]]   # the handle stays usable before the damage
            h.close()


def test_fasta_fetch_of_a_long_span(tmp_path):
    """pvio_fasta_fetch reads a span with one seek and sequential pieces of 4 MB: a 9.5 Mbp fetch across piece boundaries, line
    width not a divisor of the piece size, start and end inside lines"""
    build.build_io()
    rng = np.random.default_rng(13)
    seq = "".join(rng.choice(list("ACGTacgt"), size=10_000_000))
    bw.write_fasta(str(tmp_path / "big.fa"), [("c0", "ACGT" * 10), ("big", seq)], width=61)
    f = bamio.FastaHandler(str(tmp_path / "big.fa"))
    for a, b_ in [(123_456, 9_654_321), (0, 10_000_000), (4_194_300, 4_194_310), (60, 62), (9_999_990, 10_000_050)]:
        assert f.get_reference_sequence("big", a, b_) == seq[a:b_].upper(), (a, b_)
