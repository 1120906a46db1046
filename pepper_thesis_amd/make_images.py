"""The harness around the image builder: batches of regions -> the reference's image HDF5 files.

Mirrors generate_image_and_save_to_file (pepper_variant/modules/python/ImageGenerationUI.py:191-274):
one group `summaries/<contig>_<start>_<end>` per interval (:249), "no group when the interval yields
nothing" (AlignmentSummarizer.py:212-213), interval i handled by process i % threads (:211). Reading
BAM/FASTA is outside this round's path, so the input here is the flat region batch the
C-ABI takes; `interval_arithmetic` restates the caller's coordinate rules.
"""
from typing import List, Sequence, Tuple

import numpy as np

from .batch import Params, RegionBatch

REGION_SAFE_BASES = 100  # ConsensCandidateFinder.REGION_SAFE_BASES, Options.py:2


def interval_arithmetic(start: int, end: int) -> Tuple[int, int, int, int]:
    """AlignmentSummarizer.py:181-218: reads and reference are fetched for [start-100, end+100],
    candidates are restricted to [start, end]. -> (region_start, region_end, cand_start, cand_end)"""
    return max(0, start - REGION_SAFE_BASES), end + REGION_SAFE_BASES, start, end


def split_intervals(contig: str, start: int, end: int, region_size: int = 100_000) -> List[Tuple[str, int, int]]:
    """ImageGenerationUI.py:307-315: consecutive intervals share their boundary position"""
    out = []
    pos = start
    while pos < end:
        pos_end = min(end, pos + region_size)
        out.append((contig, pos, pos_end))
        pos = pos_end
    return out


def downsample_indices(n_reads: int, downsample_rate: float = 1.0, max_reads: int = 5000, seed: int = 2719747673) -> np.ndarray:
    """reservoir sampling of AlignmentSummarizer.py:191-208 (NumPy legacy RandomState, fresh per region);
    returns the kept read indices in output order"""
    limit = int(min(max_reads, downsample_rate * n_reads))
    if n_reads <= limit:
        return np.arange(n_reads)
    rng = np.random.RandomState(seed)
    kept = list(range(limit))
    for i in range(limit, n_reads):
        j = rng.randint(0, i + 1)
        if j < limit:
            kept[j] = i
    return np.asarray(kept)


def write_image_file(ctx, path: str, batch: RegionBatch, intervals: Sequence[Tuple[str, int, int]], params: Params) -> int:
    """runs the builder over the batch and writes one summary group per interval that produced windows"""
    from .hdf5io import ImageStore
    assert len(intervals) == batch.n_regions
    out = ctx.summarize(batch, params)
    with ImageStore(path, "w") as store:
        for g, (contig, start, end) in enumerate(intervals):
            sel = np.flatnonzero(out.region == g)
            if sel.size == 0:
                continue
            store.write_summary("%s_%d_%d" % (contig, start, end), [contig] * sel.size, out.position[sel], out.depth[sel],
                                [[out.candidates[i]] for i in sel], out.cand_freq[sel].reshape(-1, 1), out.images[sel])
    return len(out)
