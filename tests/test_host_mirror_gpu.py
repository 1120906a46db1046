"""GPU: the host-side mirrors read like the reference's call sites
(AlignmentSummarizer.py:220-238, predict_distributed_gpu.py:58-69)."""
import numpy as np
import pytest

import cases
from pepper_thesis_amd import synth
from pepper_thesis_amd.batch import PRESETS, pack_regions
from pepper_thesis_amd.predict import Predictor
from pepper_thesis_amd.region_summary import RegionalSummaryGenerator

pytestmark = pytest.mark.gpu


class _Flags:
    def __init__(self, rev):
        self.is_reverse = rev


class _TypeRead:  # the fields of read.h:60-71 as pybind exposes them
    def __init__(self, rd):
        self.pos = rd.pos
        self.sequence = rd.bases.decode("latin-1")
        self.base_qualities = [int(q) for q in rd.quals]
        self.cigar_tuples = [(int(c) & 0xF, int(c) >> 4) for c in rd.cigar]
        self.mapping_quality = rd.mapq
        self.flags = _Flags(rd.is_reverse)


def test_regional_summary_generator_like_the_reference(hip_ctx, oracle_lib):
    region = cases.kat23_indel()
    all_reads = [_TypeRead(r) for r in region.reads]
    ref_seq = region.ref.decode()
    # AlignmentSummarizer.py:220-238, ONT preset
    regional_summary = RegionalSummaryGenerator("chr20", region.ref_start, region.ref_end, ref_seq, ctx=hip_ctx)
    regional_summary.generate_max_insert_summary(all_reads)
    candidate_image_summary = regional_summary.generate_summary(
        all_reads, 1, 1, 0.10, 0.15, 0.15, 3, 0.10, 0.10, 2, False, region.ref_start, region.ref_end, 32, 26, False)
    exp = oracle_lib.summarize(pack_regions([region]), PRESETS["ont_r9_guppy5_sup"])
    assert [c.candidates[0] for c in candidate_image_summary] == exp.candidates == ["3CAA", "2GGG"]
    for i, c in enumerate(candidate_image_summary):
        assert (c.contig, c.position, c.depth, c.candidate_frequency) == ("chr20", int(exp.position[i]), int(exp.depth[i]), [int(exp.cand_freq[i])])
        np.testing.assert_array_equal(np.array(c.image_matrix, dtype=np.int8), exp.images[i])
        assert c.base_label == 0 and c.type_label == 0


def test_predictor_loop(hip_ctx):
    from oracle import rnn_oracle
    w = synth.make_weights_p1(5, 2.0)
    sd = {"module." + k: v for k, v in w.items()}  # checkpoints saved from DataParallel carry the prefix
    p = Predictor(hip_ctx, sd, "p1")
    x = synth.synth_windows(9, 70)
    probs = p.predict(x, batch_size=16, callers=2)
    np.testing.assert_allclose(probs, rnn_oracle.p1_forward(w, x, np.float64), atol=1e-4, rtol=0)
    got = dict(p.predict_batches([x[:10], x[10:30]]))
    assert got[1].dtype == np.float64 and got[1].shape == (20, 3)
    np.testing.assert_allclose(got[1], probs[10:30], atol=1e-6)


@pytest.mark.gpu
def test_pv_gather_single_rank(hip_ctx):
    """pv_comm_* / pv_gather through RCCL with a one-rank communicator (the multi-rank form needs one GPU per rank: it is
    exercised by bench.py --gpus N on a multi-GPU node; here: library loading, id, communicator, count exchange, own-rows copy)"""
    import torch
    from pepper_thesis_amd.dist import CabiGather
    g = CabiGather(hip_ctx, 0, 1)
    x = torch.arange(21, dtype=torch.float32, device="cuda:0").reshape(7, 3)
    rows, counts = g.gather(x, dst=0)
    assert counts == [7] and torch.equal(rows, x)
    rows, counts = g.gather(x[:0].contiguous(), dst=0)
    assert counts == [0] and rows.shape[0] == 0
    # an explicit bound that is too small: PV_ERR_CAPACITY (the verdict every rank would reach), and the communicator is
    # still usable afterwards (no group left open)
    from pepper_thesis_amd import _ffi
    with pytest.raises(_ffi.PepperHipError) as e:
        g.gather(x, dst=0, capacity_rows=3)
    assert e.value.code == _ffi.PV_ERR_CAPACITY
    rows, counts = g.gather(x, dst=0, capacity_rows=7)
    assert counts == [7] and torch.equal(rows, x)
    g.close()
