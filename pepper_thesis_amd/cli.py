"""The pepper_variant command line on this code base: the option names of the reference's argparse definitions
(pepper_variant/modules/argparse/CallVariantsArguments.py:7-336, MakeImagesArguments.py, RunInferenceArguments.py:7-124,
FindCandidatesArguments.py) and its sub-command dispatcher (pepper_variant/pepper_variant.py:34-91).

Options that have no meaning on the MI355X path are still ACCEPTED, so that the command lines of the existing
PEPPER-Margin-DeepVariant pipeline scripts run unchanged:
  --quantized / --no_quantized  (the reference's CPU ONNX quantisation), -w/--num_workers (torch DataLoader workers),
  -g/--gpu (always on: there is no CPU path), -t/--threads (CPU fan-out; here the reader thread count),
  -per_gpu/--callers_per_gpu (caller processes sharing a GPU; here the number of `batch_size` batches fused per launch).
`--dry` (the reference's fake predictor that turns training labels into predictions, predict_distributed_cpu_fake.py:12-52)
is refused: it only exists for labelled training images.
"""
import argparse

from .batch import PRESETS

__version__ = "0.8.0-mi355x"


def _platform_group(ap, required=True):
    g = ap.add_mutually_exclusive_group(required=required)
    for name in PRESETS:   # --ont_r9_guppy5_sup | --ont_r9_guppy4_hac | --ont_r10_q20 | --hifi | --clr
        g.add_argument("--" + name, action="store_true", default=False)


def add_image_options(ap):
    """MakeImagesArguments.py / the image block of CallVariantsArguments.py:52-180"""
    ap.add_argument("-d", "--downsample_rate", type=float, default=1.0)
    ap.add_argument("-r", "--region", type=str, default=None, help="contig[:start-end], comma list, ranges chr1-22")
    ap.add_argument("--region_size", type=int, default=100000)
    ap.add_argument("--region_bed", "-rb", type=str, default=None, help="parsed; consulted in train mode only, as in the reference")
    ap.add_argument("-hp", "--use_hp_info", action="store_true", default=False)
    ap.add_argument("--include_supplementary", action="store_true", default=False)
    ap.add_argument("--min_mapq", type=int, default=None)
    ap.add_argument("--min_snp_baseq", type=int, default=None)
    ap.add_argument("--min_indel_baseq", type=int, default=None)
    ap.add_argument("--snp_frequency", type=float, default=None)
    ap.add_argument("--insert_frequency", type=float, default=None)
    ap.add_argument("--delete_frequency", type=float, default=None)
    ap.add_argument("--min_coverage_threshold", type=int, default=None)
    ap.add_argument("--candidate_support_threshold", type=int, default=None)
    ap.add_argument("--snp_candidate_frequency_threshold", type=float, default=None)
    ap.add_argument("--indel_candidate_frequency_threshold", type=float, default=None)
    ap.add_argument("--skip_indels", action="store_true", default=False)


def add_inference_options(ap, per_gpu_default=4):
    """RunInferenceArguments.py:29-110 / CallVariantsArguments.py:182-236"""
    ap.add_argument("-bs", "--batch_size", type=int, default=512)
    ap.add_argument("-g", "--gpu", action="store_true", default=False, help="accepted; this build has no CPU path")
    ap.add_argument("-per_gpu", "--callers_per_gpu", type=int, default=per_gpu_default)
    ap.add_argument("-d_ids", "--device_ids", type=str, default=None, help="comma list: rank r uses device_ids[r %% len]")
    ap.add_argument("--quantized", dest="quantized", action="store_true", default=False, help="accepted and ignored")
    ap.add_argument("--no_quantized", dest="quantized", action="store_false", help="accepted and ignored")
    ap.add_argument("-w", "--num_workers", type=int, default=0, help="accepted and ignored")
    ap.add_argument("--bf16", action="store_true", default=False,
                    help="PV_DTYPE_BF16_INPUT_GEMM: matrix products on the bf16 MFMA with 3-term split operands (within 1e-4 of fp32)")


def add_candidate_options(ap):
    """FindCandidatesArguments.py / CallVariantsArguments.py:238-336 (None = the platform preset's value)"""
    ap.add_argument("--allowed_multiallelics", type=int, default=None)
    for name in ("snp_p_value", "insert_p_value", "delete_p_value", "snp_p_value_in_lc", "insert_p_value_in_lc", "delete_p_value_in_lc",
                 "snp_q_cutoff", "indel_q_cutoff", "snp_q_cutoff_in_lc", "indel_q_cutoff_in_lc", "report_snp_above_freq",
                 "report_indel_above_freq"):
        ap.add_argument("--" + name, type=float, default=None)


def call_variant_parser(ap=None):
    ap = ap or argparse.ArgumentParser(prog="call_variant")
    ap.add_argument("-b", "--bam", type=str, required=True)
    ap.add_argument("-f", "--fasta", type=str, required=True)
    ap.add_argument("-m", "--model_path", type=str, required=True)
    ap.add_argument("-o", "--output_dir", type=str, required=True)
    ap.add_argument("-s", "--sample_name", type=str, required=False, default="SAMPLE")
    ap.add_argument("-t", "--threads", type=int, required=False, default=None, help="reader threads (default: the CPU share)")
    add_image_options(ap)
    add_inference_options(ap)
    add_candidate_options(ap)
    _platform_group(ap)
    # this build's own switches
    ap.add_argument("--fused", dest="fused", action="store_true", default=True,
                    help="(default) windows stay in HBM between the image builder and the network; no image files")
    ap.add_argument("--no_fused", dest="fused", action="store_false", help="the reference's three steps through image HDF5 files")
    ap.add_argument("--keep_images", action="store_true", default=False, help="with --fused: also write the image HDF5 files")
    return ap


def make_images_parser(ap=None):
    ap = ap or argparse.ArgumentParser(prog="make_images")
    ap.add_argument("-b", "--bam", type=str, required=True)
    ap.add_argument("-f", "--fasta", type=str, required=True)
    ap.add_argument("-o", "--output_dir", type=str, required=True)
    ap.add_argument("-t", "--threads", type=int, required=False, default=None)
    add_image_options(ap)
    _platform_group(ap)
    return ap


def run_inference_parser(ap=None):
    ap = ap or argparse.ArgumentParser(prog="run_inference")
    ap.add_argument("-i", "--image_dir", type=str, required=True)
    ap.add_argument("-m", "--model_path", type=str, required=True)
    ap.add_argument("-o", "--output_dir", type=str, required=True)
    add_inference_options(ap)
    ap.add_argument("-t", "--threads", type=int, default=8)
    ap.add_argument("-hp", "--use_hp_info", action="store_true", default=False)
    ap.add_argument("--dry", action="store_true", default=False)
    _platform_group(ap, required=False)   # parsed, unused by inference (RunInferenceArguments.py:112-124)
    return ap


def find_candidates_parser(ap=None):
    ap = ap or argparse.ArgumentParser(prog="find_candidates")
    ap.add_argument("-i", "--input_dir", type=str, required=True)
    ap.add_argument("-b", "--bam", type=str, required=False, default=None, help="accepted (the reference opens it for the contig list only)")
    ap.add_argument("-f", "--fasta", type=str, required=True)
    ap.add_argument("-s", "--sample_name", type=str, required=False, default="SAMPLE")
    ap.add_argument("-o", "--output_dir", type=str, required=True)
    ap.add_argument("-t", "--threads", type=int, required=False, default=1)
    ap.add_argument("-hp", "--use_hp_info", action="store_true", default=False)
    add_candidate_options(ap)
    ap.add_argument("--freq_based", action="store_true", default=False)
    ap.add_argument("--freq", type=float, default=0.10)
    _platform_group(ap)
    return ap


def preset_of(args) -> str:
    return next(n for n in PRESETS if getattr(args, n, False))


def rank_world_device(args=None):
    """(rank, world, device): RANK / WORLD_SIZE / LOCAL_RANK as torchrun sets them; -d_ids maps rank r to device_ids[r % len]"""
    import os
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    ids = getattr(args, "device_ids", None)
    if ids:
        devs = [int(d) for d in str(ids).split(",") if d.strip() != ""]
        return rank, world, devs[rank % len(devs)]
    return rank, world, int(os.environ.get("LOCAL_RANK", "0"))


def main(argv=None):
    """`python -m pepper_thesis_amd <sub-command> ...` = pepper_variant.py:18-91"""
    import sys
    ap = argparse.ArgumentParser(prog="pepper_variant", description="PEPPER variant calling on the MI355X-native hot path")
    ap.add_argument("--version", action="store_true", default=False)
    sub = ap.add_subparsers(dest="sub_command")
    call_variant_parser(sub.add_parser("call_variant", help="make_images -> run_inference -> find_candidates (fused on the device by default)"))
    make_images_parser(sub.add_parser("make_images", help="pileup summary images of the reads aligned to the reference"))
    run_inference_parser(sub.add_parser("run_inference", help="genotype probabilities for generated images"))
    find_candidates_parser(sub.add_parser("find_candidates", help="candidate variants (VCF) from the predictions"))
    sub.add_parser("merge_variants", help="not part of this build (merges PEPPER and DeepVariant VCFs downstream of the hot path)")
    args = ap.parse_args(argv)
    if args.version:
        print("PEPPER VERSION: ", __version__)
        return 0
    if args.sub_command == "call_variant":
        from . import call_variant
        return call_variant.run(args)
    if args.sub_command == "make_images":
        from . import make_images
        return make_images.run(args)
    if args.sub_command == "run_inference":
        from . import run_inference
        return run_inference.run(args)
    if args.sub_command == "find_candidates":
        from . import find_candidates
        return find_candidates.run(args)
    if args.sub_command == "merge_variants":
        sys.stderr.write("ERROR: merge_variants is outside this build (SURVEY 2, row 24): use the reference's own script on the VCFs.\n")
        return 2
    sys.stderr.write("ERROR: NO SUBCOMMAND SELECTED. PLEASE SELECT ONE OF THE AVAIABLE SUB-COMMANDS.\n")
    ap.print_help()
    return 2
