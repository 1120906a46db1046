"""`call_variant`: make_images -> run_inference -> find_candidates on the MI355X path
(pepper_variant/modules/python/CallVariant.py:12-109: same three steps, same output directories and file names).

  python -m pepper_thesis_amd call_variant -b reads.bam -f ref.fa -m model.pkl -o out/ -s SAMPLE -t 16 --ont_r9_guppy5_sup [-r chr20:1-1000000]

Default is the FUSED form (pipeline.py): the windows of a builder launch chain stay in HBM and go straight through the network;
only the prediction file is written (`--keep_images` adds the image file, `--no_fused` runs the reference's three steps through
image HDF5 files). Several ranks (torchrun: RANK / WORLD_SIZE / LOCAL_RANK; `-d_ids` to map ranks to devices): intervals are
dealt i % world (ImageGenerationUI.py:211), every rank writes `pepper_prediction_<rank>.hdf` into the same `predictions_<ts>`
directory (RunInference.py:101-116), and after a barrier rank 0 runs find_candidates over the directory. The ranks agree on
the time stamp and meet at the barrier through torch.distributed (gloo: two tiny collectives, no data path).
"""
import os
import sys
import time
from datetime import datetime


class _Comm:
    """the two host-side collectives of a multi-rank run: one string broadcast, one barrier (torch.distributed / gloo)"""

    def __init__(self, rank: int, world: int):
        self.rank, self.world, self.own = rank, world, False
        if world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=rank, world_size=world)
                self.own = True
            self.dist = dist

    def broadcast_str(self, s: str) -> str:
        if self.world == 1:
            return s
        box = [s]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def close(self):
        if self.world > 1 and self.own:
            self.dist.destroy_process_group()


def _predict_rank(args, rank, world, device, image_dir, pred_dir, params, min_mapq, on_rows=None):
    """this rank's share of make_images + run_inference; returns the number of windows. (Replaced by a stub in the CPU test of
    the multi-rank plumbing.)"""
    from . import _ffi, make_images, pipeline, run_inference
    from .runtime import Context
    ctx = Context(device)
    dtype = _ffi.PV_DTYPE_BF16_INPUT_GEMM if getattr(args, "bf16", False) else _ffi.PV_DTYPE_F32
    pred_name = "pepper_prediction.hdf" if world == 1 else "pepper_prediction_%d.hdf" % rank
    state = run_inference.load_state_dict(args.model_path)
    try:
        if args.fused:
            keep = os.path.join(image_dir, "pepper_variants_images_thread_%d.hdf5" % rank) if args.keep_images else None
            return pipeline.call_variant_fused(ctx, state, args.bam, args.fasta, os.path.join(pred_dir, pred_name), params, args.region,
                                               args.region_size, min_mapq, args.include_supplementary, args.downsample_rate,
                                               args.batch_size, max(1, int(args.callers_per_gpu)) * 4, rank, world, args.threads, keep,
                                               dtype=dtype, region_bed=args.region_bed, on_rows=on_rows)
        n = make_images.generate_images(ctx, args.bam, args.fasta, image_dir, params, args.region, args.region_size, min_mapq,
                                        args.include_supplementary, args.downsample_rate, rank=rank, world=world,
                                        reader_threads=args.threads, region_bed=args.region_bed)
        os.makedirs(pred_dir, exist_ok=True)
        mine = [os.path.join(image_dir, "pepper_variants_images_thread_%d.hdf5" % rank)]
        run_inference.predict_files(ctx, state, [p for p in mine if os.path.exists(p)], os.path.join(pred_dir, pred_name),
                                    args.batch_size, max(1, int(args.callers_per_gpu)) * 4, dtype=dtype)
        return n
    finally:
        ctx.close()


def run(args, predict_rank=_predict_rank):
    from . import cli, find_candidates, make_images
    preset = cli.preset_of(args)
    # platform preset -> image-generation scalars + min_mapq, and candidate-finding scalars (SetParameters.py:12-283)
    params, min_mapq = make_images.image_options_from_args(args, preset)
    cand_opt = find_candidates.candidate_options_from_args(args, preset)
    rank, world, device = cli.rank_world_device(args)
    if args.use_hp_info:
        # CallVariant.py passes -hp on to all three steps, but the reference's network hard-codes 33-row images while -hp images
        # have 21 rows (simple_model.py:35 vs Options.py:22): its own call_variant -hp cannot run. make_images -hp works here.
        sys.stderr.write("ERROR: call_variant -hp: the reference's model cannot consume haplotag-aware images (21 x 48); "
                         "use `make_images -hp` for the images alone.\n")
        return 2
    comm = _Comm(rank, world)
    try:
        ts = comm.broadcast_str(datetime.now().strftime("%m%d%Y_%H%M%S"))   # one directory name for all ranks
        image_dir = os.path.join(args.output_dir, "images_" + ts)
        pred_dir = os.path.join(args.output_dir, "predictions_" + ts)
        os.makedirs(pred_dir, exist_ok=True)
        t0 = time.time()
        # one rank, fused: the candidates are selected from every call's windows while the pipeline runs (on its writer thread);
        # with several ranks rank 0 reads all the prediction files afterwards, as the reference does
        collector = None
        if world == 1 and args.fused and predict_rank is _predict_rank:
            collector = find_candidates.CandidateCollector(args.fasta, cand_opt)
            n = predict_rank(args, rank, world, device, image_dir, pred_dir, params, min_mapq, on_rows=collector)
        else:
            n = predict_rank(args, rank, world, device, image_dir, pred_dir, params, min_mapq)
        sys.stderr.write("INFO: [RANK %d/%d] [1-2/3] %s: %d WINDOWS PREDICTED (%.1f s)\n" %
                         (rank, world, "FUSED IMAGES + INFERENCE" if args.fused else "IMAGES, INFERENCE", n, time.time() - t0))
        comm.barrier()   # every rank's prediction file is complete
        counts = None
        if rank == 0:
            counts = find_candidates.process_candidates(pred_dir, args.fasta, args.sample_name, args.output_dir, cand_opt,
                                                        selected=collector.selected if collector is not None else None)
            sys.stderr.write("INFO: [3/3] CANDIDATES: %s (%.1f s)\n" % (counts, time.time() - t0))
        comm.barrier()
        return counts
    finally:
        comm.close()


def main(argv=None):
    from . import cli
    return run(cli.call_variant_parser().parse_args(argv))


if __name__ == "__main__":
    main()
