"""helpers to read tests/golden/summary_golden.npz back into batches / expected outputs"""
import numpy as np

from pepper_thesis_amd.batch import PRESETS, RegionBatch


def golden_names(g):
    return [n.decode() for n in g["names"]]


def golden_case(g, entry):
    name, preset = entry.split("|")
    arrs = {f: np.ascontiguousarray(g["%s/in/%s" % (name, f)]) for f in RegionBatch.FIELDS}
    batch = RegionBatch(n_regions=int(arrs["ref_start"].shape[0]), **arrs)
    exp = {k: g["%s/out/%s" % (name, k)] for k in
           ("region", "position", "depth", "cand_freq", "images_i32", "images", "candidates")}
    exp["candidates"] = [c.decode("latin-1") for c in exp["candidates"]]
    return batch, PRESETS[preset], exp


def hp_golden_case(g, entry):
    """tests/golden/summary_hp_golden.npz: the same layout plus in/read_hp; parameters carry the 20 / 48 window geometry"""
    from pepper_thesis_amd.batch import hp_params
    name, preset = entry.split("|")
    arrs = {f: np.ascontiguousarray(g["%s/in/%s" % (name, f)]) for f in RegionBatch.FIELDS}
    batch = RegionBatch(n_regions=int(arrs["ref_start"].shape[0]), **arrs)
    hp = np.ascontiguousarray(g["%s/in/read_hp" % name]).astype(np.int32)
    batch.read_hp = hp if hp.any() else None
    exp = {k: g["%s/out/%s" % (name, k)] for k in ("region", "position", "depth", "cand_freq", "images", "candidates")}
    key = "%s/out/images_i32" % name
    exp["images_i32"] = g[key] if key in g.files else None
    exp["candidates"] = [c.decode("latin-1") for c in exp["candidates"]]
    return batch, hp_params(PRESETS[preset]), exp


def assert_summary_equal(out, exp, what=""):
    assert len(out) == len(exp["position"]), "%s: window count %d != %d" % (what, len(out), len(exp["position"]))
    assert out.candidates == list(exp["candidates"]), what
    np.testing.assert_array_equal(out.region, exp["region"], err_msg=what)
    np.testing.assert_array_equal(out.position, exp["position"], err_msg=what)
    np.testing.assert_array_equal(out.depth, exp["depth"], err_msg=what)
    np.testing.assert_array_equal(out.cand_freq, exp["cand_freq"], err_msg=what)
    np.testing.assert_array_equal(out.images, exp["images"], err_msg=what)
    if out.images_i32 is not None and exp.get("images_i32") is not None:
        np.testing.assert_array_equal(out.images_i32, exp["images_i32"], err_msg=what)


def summary_as_expected(out):
    return dict(region=out.region, position=out.position, depth=out.depth, cand_freq=out.cand_freq,
                images=out.images, images_i32=out.images_i32, candidates=out.candidates)
