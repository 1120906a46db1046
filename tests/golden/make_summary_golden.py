"""Generates tests/golden/summary_golden.npz from the REFERENCE's own image builder.

Run in the build container only (needs /root/reference): `python tests/golden/make_summary_golden.py`.
It compiles /root/reference/pepper_variant/modules/cpp/region_summary.cpp in place through
oracle/Makefile (-> oracle/_ref/libref_region_summary.so), feeds it the hand-built edge cases and the
seeded random regions of tests/cases.py and stores INPUTS and EXPECTED OUTPUTS (data only).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import oracle  # noqa: E402
from pepper_thesis_amd.batch import PRESETS, RegionBatch  # noqa: E402


def main():
    oracle.build(force=True)
    assert oracle.have_reference(), "needs /root/reference to build oracle/_ref"
    blob = {}
    names = []

    def add(name, batch, preset):
        out = oracle.reference_summarize(batch, PRESETS[preset], want_i32=True)
        names.append(name + "|" + preset)
        for f in RegionBatch.FIELDS:
            blob["%s/in/%s" % (name, f)] = getattr(batch, f)
        blob["%s/out/region" % name] = out.region
        blob["%s/out/position" % name] = out.position
        blob["%s/out/depth" % name] = out.depth
        blob["%s/out/cand_freq" % name] = out.cand_freq
        blob["%s/out/images_i32" % name] = out.images_i32
        blob["%s/out/images" % name] = out.images
        blob["%s/out/candidates" % name] = np.asarray(out.candidates, dtype="S")
        print("%-28s %-18s regions=%d reads=%d bases=%d windows=%d" % (
            name, preset, batch.n_regions, batch.n_reads, batch.n_bases, len(out)))

    for name in cases.EDGE_CASES:
        for preset in ("ont_r9_guppy5_sup", "hifi"):
            add("%s@%s" % (name, preset), cases.edge_batch(name), preset)
    for seed, kw, preset in cases.GOLDEN_RANDOM:
        add("random%d" % seed, cases.random_batch(seed, kw), preset)
    blob["names"] = np.asarray(names, dtype="S")
    path = os.path.join(ROOT, "tests", "golden", "summary_golden.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
