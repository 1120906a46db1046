"""Host-side mirror of the reference's inference loops above the C-ABI.

P1: `predict` of pepper_variant/modules/python/models/predict_distributed_gpu.py:19-74 — batches of
int8 [B,33,26] images -> float32 [B,3] softmax (the reference writes them as float64 `base_prediction`,
DataStorePredict.py:49-66). The reference's DataParallel fan-out / `callers_per_gpu` is replaced by
handing `callers * batch_size` windows to one launch chain.
P2: the 19-window sliding loop of pepper/modules/python/models/predict.py:47-97 is a single call.
"""
from typing import Iterable, Iterator, Tuple

import numpy as np

from .runtime import Context


class Predictor:
    def __init__(self, ctx: Context, state_dict: dict, plan: str = "p1", dtype: int = 0):
        """state_dict: the checkpoint's 'model_state_dict' with the 'module.' prefixes stripped
        (ModelHander.py:30-41) as numpy arrays (or anything np.asarray accepts, e.g. CPU torch tensors)."""
        self.ctx, self.plan = ctx, plan
        w = {k[7:] if k.startswith("module.") else k: np.asarray(v, dtype=np.float32) for k, v in state_dict.items()}
        if plan == "p1":
            ctx.load_p1(w, dtype)
        elif plan == "p2":
            ctx.load_p2(w, dtype)
        else:
            raise ValueError(plan)

    def predict(self, images: np.ndarray, batch_size: int = 512, callers: int = 16) -> np.ndarray:
        """images [N,33,26] int8 -> [N,3] float32; `callers` batches are fused per device call"""
        assert self.plan == "p1"
        step = max(1, batch_size * callers)
        out = np.zeros((images.shape[0], 3), np.float32)
        for i in range(0, images.shape[0], step):
            out[i:i + step] = self.ctx.forward_p1(images[i:i + step])
        return out

    def predict_batches(self, batches: Iterable[np.ndarray]) -> Iterator[Tuple[int, np.ndarray]]:
        """the reference's `for images in data_loader` shape: yields (batch_no, float64 [B,3]) ready for
        DataStore.write_prediction (np.float == float64, DataStorePredict.py:66)"""
        for n, images in enumerate(batches):
            yield n, self.ctx.forward_p1(np.asarray(images, dtype=np.int8)).astype(np.float64)

    def call_consensus(self, images: np.ndarray) -> np.ndarray:
        """images [B,1000,10] uint8 -> labels [B,1000] uint8 (predict.py:91-97)"""
        assert self.plan == "p2"
        return self.ctx.forward_p2(images)
