"""CPU, world_size 2 over gloo: the multi-rank plumbing of call_variant (call_variant.run): the ranks agree on ONE time-stamped
directory, interval i goes to rank i % world (ImageGenerationUI.py:211), every rank writes pepper_prediction_<rank>.hdf
(RunInference.py:101-116), rank 0 runs find_candidates over the union after the barrier. The device step is replaced by a stub
that writes prediction records for this rank's intervals (no GPU here); the GPU test of the fused step is in test_pipeline_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_predict_rank(args, rank, world, device, image_dir, pred_dir, params, min_mapq):
    """one confident heterozygous SNP candidate in the middle of every interval this rank owns"""
    from pepper_thesis_amd import bamio, hdf5io, make_images
    bam, fasta = bamio.BamHandler(args.bam), bamio.FastaHandler(args.fasta)
    todo = make_images.list_intervals(fasta, bam, args.region, args.region_size)
    mine = [iv for i, iv in enumerate(todo) if i % world == rank]
    name = "pepper_prediction.hdf" if world == 1 else "pepper_prediction_%d.hdf" % rank
    with hdf5io.PredictionStore(os.path.join(pred_dir, name), "w") as out:
        for k, (contig, a, b) in enumerate(mine):
            pos = (a + b) // 2
            ref = fasta.get_reference_sequence(contig, pos, pos + 1)
            alt = "A" if ref != "A" else "C"
            out.write_prediction(k, [contig], [pos], [30], [["1" + alt]], [[15]], np.array([[0.01, 0.98, 0.01]]))
    with open(os.path.join(pred_dir, "intervals_%d.txt" % rank), "w") as fh:
        fh.write("\n".join("%s %d %d" % iv for iv in mine))
    return len(mine)


def _worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from pepper_thesis_amd import call_variant, cli
    args = cli.call_variant_parser().parse_args(["-b", os.path.join(tmp, "r.bam"), "-f", os.path.join(tmp, "r.fa"), "-m", "unused", "-o",
                                                 os.path.join(tmp, "out"), "-s", "S", "-t", "1", "--ont_r9_guppy5_sup", "-r", "c1:100-6900",
                                                 "--region_size", "1000", "-d_ids", "3,5"])
    assert cli.rank_world_device(args) == (rank, world, (3, 5)[rank])
    counts = call_variant.run(args, predict_rank=_stub_predict_rank)
    q.put((rank, counts))


@pytest.mark.timeout(180)
def test_call_variant_world2_gloo(tmp_path):
    import bam_writer as bw
    from pepper_thesis_amd import build
    build.build_io()
    rng = np.random.default_rng(5)
    bw.write_fasta(str(tmp_path / "r.fa"), [("c1", "".join(rng.choice(list("ACGT"), size=7000)))])
    bw.write_bam(str(tmp_path / "r.bam"), [("c1", 7000)], [])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out = tmp_path / "out"
    pred_dirs = [d for d in os.listdir(out) if d.startswith("predictions_")]
    assert len(pred_dirs) == 1                                        # both ranks used rank 0's time stamp
    files = sorted(os.listdir(out / pred_dirs[0]))
    assert [f for f in files if f.endswith(".hdf")] == ["pepper_prediction_0.hdf", "pepper_prediction_1.hdf"]
    iv = [open(out / pred_dirs[0] / ("intervals_%d.txt" % r)).read().split("\n") for r in range(2)]
    expect = ["c1 %d %d" % (a, min(a + 1000, 6900)) for a in range(100, 6900, 1000)]
    assert iv[0] == expect[0::2] and iv[1] == expect[1::2]            # interval i -> rank i % 2, every interval exactly once
    assert got[1] is None and got[0]["total"] == len(expect)          # rank 0 alone wrote the VCFs, from BOTH ranks' predictions
    assert os.path.exists(out / "PEPPER_VARIANT_FULL.vcf.gz") and os.path.exists(out / "PEPPER_VARIANT_FULL.vcf.gz.tbi")
