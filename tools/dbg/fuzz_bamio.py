"""Diagnostic (CPU): the native BAM reader's region-clipped reads against the Python restatement of the reference's clipping rules
(tests/bam_writer.py) on many random record sets and regions, with and without read-ahead helper threads."""
import sys, os, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bam_writer as bw
from pepper_thesis_amd import bamio, build
build.build_io()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time(); n = 0; nreads = 0
d = tempfile.mkdtemp()
try:
    while time.time() - t0 < budget:
        rng = np.random.default_rng(900 + n)
        L = int(rng.integers(300, 40000))
        seq = "".join(rng.choice(list("ACGTacgtN"), size=L, p=[.22, .22, .22, .22, .02, .02, .02, .02, .04]))
        bw.write_fasta(os.path.join(d, "r.fa"), [("c1", seq)], width=int(rng.integers(20, 90)))
        recs = bw.random_records(rng, int(rng.integers(5, 600)), L, tid=0, mean_len=int(rng.integers(30, 2500)), allow_skip=bool(rng.integers(2)))
        bw.write_bam(os.path.join(d, "r.bam"), [("c1", L)], recs)
        h = bamio.BamHandler(os.path.join(d, "r.bam"))
        h.set_threads(int(rng.integers(0, 4)))
        for _ in range(6):
            a = int(rng.integers(0, L)); b = int(min(L, a + rng.integers(1, 6000)))
            supp, mq = bool(rng.integers(2)), int(rng.integers(0, 30))
            got = h.get_reads("c1", a, b, supp, mq, 0)
            exp = bw.expected_reads(recs, 0, a, b, supp, mq)
            assert len(got) == len(exp), (n, a, b, len(got), len(exp))
            for g, e in zip(got, exp):
                assert (g.pos, g.pos_end, g.is_reverse, g.mapq, g.hp_tag, g.query_name) == (e["pos"], e["pos_end"], e["rev"], e["mapq"], e["hp"], e["name"]), (n, a, b)
                assert g.bases.decode() == e["seq"] and g.quals.tolist() == e["qual"], (n, a, b)
                assert [(int(c) & 0xF, int(c) >> 4) for c in g.cigar] == e["cigar"], (n, a, b)
            nreads += len(got)
        h.close()
        n += 1
finally:
    shutil.rmtree(d)
print("fuzz_bamio: %d random files, %d clipped reads identical to the restated rules in %.0f s" % (n, nreads, time.time() - t0))
