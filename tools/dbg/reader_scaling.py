"""Diagnostic: the native reader alone (no GPU): wall time of reading a synthetic 3.2 Mbp / 60x BAM with 1..N reader threads"""
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_filepath as bf
from pepper_thesis_amd import bamio, make_images
d = tempfile.mkdtemp()
bam, fa, info = bf.make_files(d, 3_200_000)
print("cpus", len(os.sched_getaffinity(0)), "backend", bamio.inflate_backend(), flush=True)
b, f = bamio.BamHandler(bam), bamio.FastaHandler(fa)
t0 = time.perf_counter()
fb = bamio.fill_batch(b, f, [("chr20", 100000, 200000)], 5, False, 1.0, 100)
print("one interval, one thread: %.1f ms total, inflate %.1f ms (%.0f MB/s), decode %.1f ms, %d reads %.1f M bases" % (
    (time.perf_counter() - t0) * 1e3, fb.t_inflate * 1e3, fb.bytes_inflated / fb.t_inflate / 1e6, (fb.t_total - fb.t_inflate) * 1e3, fb.batch.n_reads, fb.batch.n_bases / 1e6), flush=True)
for backend in (True, False):
    bamio.set_inflate_backend(backend)
    for thr, helpers in ((4, 0), (8, 0), (16, 0), (32, 0), (8, 3), (16, 3), (16, 1), (16, 7)):
        T = {}
        t0 = time.perf_counter()
        n = 0
        first = None
        for parts, names in make_images.region_batches(bam, fa, None, 100000, 5, False, 1.0, 16, 0, 1, thr, 1, T, None, merge=False, inflate_helpers=helpers):
            if first is None:
                first = time.perf_counter() - t0
            n += sum(p.n_reads for p in parts)
        w = time.perf_counter() - t0
        print("%s budget %2d (readers %d x helpers %d): wall %.3f s = %.1f Mbp/s, first interval after %.0f ms; summed thread time: reader in load_block %.2f, decode %.2f, helpers %.2f" % (
            bamio.inflate_backend(), thr, T["reader_threads"], T["inflate_helpers"], w, 3.2 / w, first * 1e3, T["read_inflate_cpu_s"], T["read_decode_cpu_s"], T["read_helper_cpu_s"]), flush=True)
shutil.rmtree(d)
