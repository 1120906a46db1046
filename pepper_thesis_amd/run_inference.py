"""`run_inference` on the MI355X path: image HDF5 files in, prediction HDF5 files out.

Mirrors pepper_variant/modules/python/RunInference.py:125-138 -> distributed_gpu (:24-91) ->
predict_distributed_gpu.predict (models/predict_distributed_gpu.py:19-74): every `*.hdf5` in the image
directory is read (dataloader_predict.py:46-78), windows go through the P1 network in batches and the
results are written as `predictions/batch_<k>` groups (DataStorePredict.py:49-66) that the unmodified
`find_candidates` consumes. Flags keep the reference's names (RunInferenceArguments.py:7-124).

  python -m pepper_thesis_amd.run_inference -i IMAGE_DIR -m MODEL -o OUTPUT_DIR [-bs 512] [-per_gpu 16] [-d_ids 0]

MODEL is the reference's checkpoint (torch.save dict with 'model_state_dict', ModelHander.py:18-44; loaded
with weights_only=True) or an .npz of the same state dict. With several ranks (torchrun) files are dealt
round-robin (RunInference.py:101-106) and every rank writes pepper_prediction_<rank>.hdf.
"""
import argparse
import glob
import os
import sys
import time
from datetime import datetime

import numpy as np


def log(msg):
    sys.stderr.write("[" + datetime.now().strftime("%m-%d-%Y %H:%M:%S") + "] INFO: " + msg + "\n")
    sys.stderr.flush()


def load_state_dict(model_path: str) -> dict:
    if model_path.endswith(".npz"):
        with np.load(model_path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    import torch
    ckpt = torch.load(model_path, map_location="cpu", weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def predict_files(ctx, state_dict, input_files, output_file, batch_size=512, callers=16, timers=None, dtype=0):
    """predict() of predict_distributed_gpu.py:19-74 for a list of image files -> one prediction file.
    `timers` (optional dict) receives the stage times in seconds."""
    import time
    from .hdf5io import ImageStore, PredictionStore
    from .predict import Predictor
    T = dict(hdf5_read_s=0.0, predict_call_s=0.0, hdf5_write_s=0.0)
    t_start = time.perf_counter()
    predictor = Predictor(ctx, state_dict, "p1", dtype)
    T["load_weights_s"] = time.perf_counter() - t_start
    n_windows, batch_no = 0, 0
    with PredictionStore(output_file, "w") as out:
        for path in input_files:
            t0 = time.perf_counter()
            with ImageStore(path, "r") as store:
                parts = [store.read_summary(name) for name in store.summaries()]
            parts = [p for p in parts if len(p["positions"])]  # intervals with reads but no candidate leave empty groups
            T["hdf5_read_s"] += time.perf_counter() - t0
            if not parts:
                continue
            cat = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
            t0 = time.perf_counter()
            probs = predictor.predict(cat["images"], batch_size, callers)
            T["predict_call_s"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            for i in range(0, len(probs), batch_size):
                sl = slice(i, i + batch_size)
                out.write_prediction(batch_no, cat["contigs"][sl], cat["positions"][sl], cat["depths"][sl],
                                     cat["candidates"][sl], cat["candidate_frequency"][sl], probs[sl].astype(np.float64))
                batch_no += 1
            T["hdf5_write_s"] += time.perf_counter() - t0
            n_windows += len(probs)
    T["wall_s"] = time.perf_counter() - t_start
    T["windows"] = n_windows
    if timers is not None:
        timers.update(T)
    return n_windows


def run(args):
    from . import _ffi, cli
    from .dist import shard_regions
    from .runtime import Context
    if args.dry:
        sys.stderr.write("ERROR: --dry turns training labels into predictions (predict_distributed_cpu_fake.py:12-52); it needs labelled "
                         "training images and is not part of this build.\n")
        return 2
    if args.use_hp_info:
        sys.stderr.write("ERROR: run_inference -hp: the reference's network hard-codes 33-row images (simple_model.py:35); haplotag-aware "
                         "images have 21 rows and cannot be predicted by it.\n")
        return 2
    rank, world, device = cli.rank_world_device(args)
    files = sorted(glob.glob(os.path.join(args.image_dir, "*.hdf5")))
    mine = [files[i] for i in shard_regions(len(files), rank, world)]   # files to callers i % callers (RunInference.py:101-106)
    os.makedirs(args.output_dir, exist_ok=True)
    name = "pepper_prediction.hdf" if world == 1 else "pepper_prediction_%d.hdf" % rank
    t0 = time.time()
    log("INFERENCE STARTING ON DEVICE %d: %d FILES" % (device, len(mine)))
    ctx = Context(device)
    n = predict_files(ctx, load_state_dict(args.model_path), mine, os.path.join(args.output_dir, name),
                      args.batch_size, max(1, int(args.callers_per_gpu)) * 4,
                      dtype=_ffi.PV_DTYPE_BF16_INPUT_GEMM if args.bf16 else _ffi.PV_DTYPE_F32)
    ctx.close()
    log("FINISHED PREDICTION: %d WINDOWS IN %.2f SEC" % (n, time.time() - t0))
    return 0


def main(argv=None):
    from . import cli
    return run(cli.run_inference_parser().parse_args(argv))


if __name__ == "__main__":
    main()
