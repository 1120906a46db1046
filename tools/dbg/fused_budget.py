"""Diagnostic: the fused file path at several host-thread budgets (walls of 5 runs each)"""
import sys, os, time, tempfile, shutil, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_filepath as bf
from pepper_thesis_amd import pipeline, runtime, synth, make_images
from pepper_thesis_amd.batch import PRESETS
d = tempfile.mkdtemp()
bam, fa, info = bf.make_files(d, 3_200_000)
ctx = runtime.Context(0)
w = synth.make_weights_p1(1234)
P = PRESETS["ont_r9_guppy5_sup"]
print("cpu_share", make_images.cpu_share(), flush=True)
pipeline.call_variant_fused(ctx, w, bam, fa, os.path.join(d, "warm", "p.hdf"), P, region="chr20:0-50000", min_mapq=5)
for arg in sys.argv[1:] or ["16", "14", "12"]:
    budget, helpers = (int(v) for v in arg.split("x")) if "x" in arg else (int(arg), None)
    walls, last = [], None
    for k in range(5):
        T = {}
        pipeline.call_variant_fused(ctx, w, bam, fa, os.path.join(d, "pf", "p.hdf"), P, min_mapq=5, timers=T, reader_threads=budget, inflate_helpers=helpers)
        walls.append(T["wall_s"])
        last = T
    print("budget %2d (readers %d x helpers %d): walls ms %s -> median %.1f Mbp/s; last run: stall %.0f upload %.0f device %.0f readback %.0f weights %.0f write %.0f ms" % (
        budget, last["reader_threads"], last["inflate_helpers"], " ".join("%.0f" % (x * 1e3) for x in walls), 3.2 / sorted(walls)[2],
        last["reader_stall_s"] * 1e3, last["upload_s"] * 1e3, last["device_call_s"] * 1e3, last["readback_s"] * 1e3, last["load_weights_s"] * 1e3, last["hdf5_write_s"] * 1e3), "reader cpu: bgzf %.2f helpers %.2f decode %.2f" % (last["read_inflate_cpu_s"], last["read_helper_cpu_s"], last["read_decode_cpu_s"]), flush=True)
shutil.rmtree(d)
