"""`call_variant`: make_images -> run_inference -> find_candidates on the MI355X path
(pepper_variant/modules/python/CallVariant.py:12-109, same three steps, same intermediate directories).

  python -m pepper_thesis_amd.call_variant -b reads.bam -f ref.fa -m model.pkl -o out/ -s SAMPLE --ont_r9_guppy5_sup [-r chr20:1-1000000]
"""
import argparse
import os
import sys
import time
from datetime import datetime


def main(argv=None):
    from . import find_candidates, make_images, run_inference
    from .batch import PRESETS
    from .runtime import Context
    ap = argparse.ArgumentParser(prog="call_variant")
    ap.add_argument("-b", "--bam", required=True)
    ap.add_argument("-f", "--fasta", required=True)
    ap.add_argument("-m", "--model_path", required=True)
    ap.add_argument("-o", "--output_dir", required=True)
    ap.add_argument("-s", "--sample_name", default="SAMPLE")
    ap.add_argument("-t", "--threads", type=int, default=1)
    ap.add_argument("-r", "--region", default=None)
    ap.add_argument("--region_size", type=int, default=100_000)
    ap.add_argument("-bs", "--batch_size", type=int, default=512)
    ap.add_argument("-per_gpu", "--callers_per_gpu", type=int, default=16)
    ap.add_argument("-g", "--gpu", action="store_true", default=True)
    ap.add_argument("-d", "--downsample_rate", type=float, default=1.0)
    ap.add_argument("--include_supplementary", action="store_true")
    ap.add_argument("--min_mapq", type=int, default=None, help="default: the platform preset's value (SetParameters.py)")
    g = ap.add_mutually_exclusive_group(required=True)
    for name in PRESETS:
        g.add_argument("--" + name, action="store_true")
    make_images.add_image_arguments(ap)
    find_candidates.add_candidate_arguments(ap)
    args = ap.parse_args(argv)
    preset = next(n for n in PRESETS if getattr(args, n))
    # platform preset -> image-generation scalars + min_mapq, and candidate-finding scalars (SetParameters.py:12-283)
    params, min_mapq = make_images.image_options_from_args(args, preset)
    cand_opt = find_candidates.candidate_options_from_args(args, preset)
    ts = datetime.now().strftime("%m%d%Y_%H%M%S")
    image_dir = os.path.join(args.output_dir, "images_" + ts)
    pred_dir = os.path.join(args.output_dir, "predictions_" + ts)
    t0 = time.time()
    ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    n = make_images.generate_images(ctx, args.bam, args.fasta, image_dir, params, args.region, args.region_size, min_mapq,
                                    args.include_supplementary, args.downsample_rate)
    sys.stderr.write("INFO: [1/3] IMAGES: %d WINDOWS (%.1f s)\n" % (n, time.time() - t0))
    import glob
    files = sorted(glob.glob(os.path.join(image_dir, "*.hdf5")))
    os.makedirs(pred_dir, exist_ok=True)
    run_inference.predict_files(ctx, run_inference.load_state_dict(args.model_path), files,
                                os.path.join(pred_dir, "pepper_prediction.hdf"), args.batch_size, args.callers_per_gpu)
    ctx.close()
    sys.stderr.write("INFO: [2/3] INFERENCE DONE (%.1f s)\n" % (time.time() - t0))
    counts = find_candidates.process_candidates(pred_dir, args.fasta, args.sample_name, args.output_dir, cand_opt)
    sys.stderr.write("INFO: [3/3] CANDIDATES: %s (%.1f s)\n" % (counts, time.time() - t0))
    return counts


if __name__ == "__main__":
    main()
