"""Diagnostic: call_variant end to end (BAM -> VCFs) in the fp32 and the bf16x3 mode on one synthetic file: same sites, genotypes, QUALs?"""
import sys, os, tempfile, gzip, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_filepath as bf
from pepper_thesis_amd import call_variant, synth
d = tempfile.mkdtemp()
bam, fa, info = bf.make_files(d, 600_000)
w = synth.make_weights_p1(1234)
np.savez(os.path.join(d, "model.npz"), **{k: np.asarray(v) for k, v in w.items()})
base = ["-b", bam, "-f", fa, "-m", os.path.join(d, "model.npz"), "-s", "S", "-t", "8", "--ont_r9_guppy5_sup"]
res = {}
for name, extra in (("fp32", []), ("bf16", ["--bf16"])):
    t0 = time.time()
    counts = call_variant.main(base + ["-o", os.path.join(d, name)] + extra)
    res[name] = (counts, gzip.open(os.path.join(d, name, "PEPPER_VARIANT_FULL.vcf.gz"), "rt").read(), time.time() - t0)
    print(name, counts, "%.2f s" % res[name][2], flush=True)
a = [l.split("\t") for l in res["fp32"][1].splitlines() if not l.startswith("#")]
b = [l.split("\t") for l in res["bf16"][1].splitlines() if not l.startswith("#")]
same_sites = [x[:5] for x in a] == [x[:5] for x in b]
gt_diff = sum(1 for x, y in zip(a, b) if x[9].split(":")[0] != y[9].split(":")[0]) if same_sites else -1
q_diff = sum(1 for x, y in zip(a, b) if x[5] != y[5]) if same_sites else -1
print("records", len(a), len(b), "same sites/alleles:", same_sites, "genotype differences:", gt_diff, "QUAL differences:", q_diff)
