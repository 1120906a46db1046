"""Generates tests/golden/summary_hp_golden.npz from the REFERENCE's own haplotag-aware image builder.

Run in the build container only (needs /root/reference): `python tests/golden/make_summary_hp_golden.py`.
It compiles /root/reference/pepper_variant/modules/cpp/region_summary_hp.cpp in place through oracle/Makefile
(-> oracle/_ref/libref_region_summary_hp.so), feeds it the edge cases and seeded random regions of tests/cases.py with
an HP tag drawn for every read, and stores INPUTS and EXPECTED OUTPUTS (data only; int8 images, the un-cast values
only for the edge cases to keep the file small).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import oracle  # noqa: E402
from pepper_thesis_amd.batch import PRESETS, RegionBatch, hp_params, pack_regions  # noqa: E402


def main():
    oracle.build(force=True)
    assert oracle.have_reference_hp(), "needs /root/reference to build oracle/_ref"
    blob = {}
    names = []

    def add(name, batch, preset, keep_i32):
        out = oracle.reference_summarize_hp(batch, hp_params(PRESETS[preset]), want_i32=True)
        names.append(name + "|" + preset)
        for f in RegionBatch.FIELDS:
            blob["%s/in/%s" % (name, f)] = getattr(batch, f)
        blob["%s/in/read_hp" % name] = batch.read_hp if batch.read_hp is not None else np.zeros(batch.n_reads, np.int32)
        blob["%s/out/region" % name] = out.region
        blob["%s/out/position" % name] = out.position
        blob["%s/out/depth" % name] = out.depth
        blob["%s/out/cand_freq" % name] = out.cand_freq
        if keep_i32:
            blob["%s/out/images_i32" % name] = out.images_i32
        blob["%s/out/images" % name] = out.images
        blob["%s/out/candidates" % name] = np.asarray(out.candidates, dtype="S")
        print("%-32s %-18s regions=%d reads=%d bases=%d windows=%d" % (
            name, preset, batch.n_regions, batch.n_reads, batch.n_bases, len(out)))

    add("hp_known_answer", pack_regions([cases.hp_known_answer()]), "ont_r9_guppy5_sup", True)
    for name in cases.EDGE_CASES:
        for preset in ("ont_r9_guppy5_sup", "hifi"):
            add("%s@%s" % (name, preset), cases.hp_edge_batch(name), preset, True)
    for seed, kw, preset in cases.GOLDEN_RANDOM:
        add("random%d" % seed, cases.hp_random_batch(seed, kw), preset, False)
    blob["names"] = np.asarray(names, dtype="S")
    path = os.path.join(ROOT, "tests", "golden", "summary_hp_golden.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
