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


# ---- `make_images` from BAM + FASTA (native readers) ---------------------------------------------------

def parse_region(region: str):
    """'chr20', 'chr20:1000-2000' (ImageGenerationUI.py:138-151) -> (contig, start|None, end|None)"""
    if ":" not in region:
        return region, None, None
    contig, span = region.split(":")
    a, b = span.replace(",", "").split("-")
    return contig, int(a), int(b)


def expand_region_names(region: str) -> List[str]:
    """the `-r` grammar of ImageGenerationUtils.get_chromosome_list (ImageGenerationUI.py:131-169): a comma list of `name`
    or `name:start-end`; a name containing '-' is a RANGE (`chr1-22`, `1-22`): prefix = its leading non-digits, the numbers
    of its pieces sorted, every contig prefix + n for n in first..last, each with the same `:start-end` if one was given"""
    out = []
    for part in region.strip().split(","):
        part = part.strip()
        if not part:
            continue
        name, span = (part.split(":") + [None])[:2] if ":" in part else (part, None)
        pieces = name.split("-")
        if len(pieces) > 1:
            prefix = ""
            for ch in name:
                if ch.isdigit():
                    break
                prefix += ch
            nums = sorted(int("".join(c for c in piece if c.isdigit())) for piece in pieces)
            names = ["%s%d" % (prefix, k) for k in range(nums[0], nums[-1] + 1)]
        else:
            names = [name]
        out += [n if span is None else n + ":" + span for n in names]
    return out


def read_bed(path: str):
    """`--region_bed` (ImageGenerationUI.py:171-185): tab-separated contig, start, end per line -> {contig: [[lo, hi], ...]}.
    The reference consults the list in TRAIN mode only (AlignmentSummarizer.py:78-91: truth regions); an inference run parses
    the file (a malformed one still stops it) and then ignores it - and so does this one."""
    out = {}
    with open(path) as fh:
        for line in fh:
            if not line.strip():
                continue
            f = line.rstrip().split("\t")
            out.setdefault(f[0], []).append(sorted([int(f[1]), int(f[2])]))
    return out


def list_intervals(fasta, bam, region: str = None, region_size: int = 100_000, region_bed: str = None) -> List[Tuple[str, int, int]]:
    """the interval list of generate_images (ImageGenerationUI.py:286-316): a whole contig is [0, length-1], a user region is
    clamped to [max(0, start), min(end, length-1)], both cut into region_size pieces that share their boundary position"""
    if region_bed:
        read_bed(region_bed)   # validated, not applied: see read_bed
    todo = []
    if region:
        for part in expand_region_names(region):
            contig, a, b = parse_region(part)
            last = fasta.get_chromosome_sequence_length(contig) - 1
            if a is None:
                a, b = 0, last
            else:
                a, b = max(0, a), min(b, last)
            todo += split_intervals(contig, a, b, region_size)
    else:
        # contigs common to FASTA and BAM, natural-sorted (ImageGenerationUI.py:111-130; the reference also drops the names of
        # its 3417-entry EXCLUDED_HUMAN_CONTIGS table of decoys / alts, which this build does not carry: pass -r to restrict)
        import re
        in_bam = set(bam.get_chromosome_sequence_names())
        common = [n for n in fasta.get_chromosome_names() if n in in_bam]
        for n in sorted(common, key=lambda s_: [int(x) if x.isdigit() else x for x in re.split(r"(\d+)", s_)]):
            todo += split_intervals(n, 0, fasta.get_chromosome_sequence_length(n) - 1, region_size)
    return todo


def cpu_share() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota where one is set (a container
    with a 16-core quota on a 256-thread host reports 256 in its mask; 32 reader threads there ran at half the rate of 16)"""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                      # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:   # v1
                q, per = int(fq.read()), int(fp.read())
                if q > 0 and per > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def region_batches(bam_path: str, fasta_path: str, region: str = None, region_size: int = 100_000, min_mapq: int = 5,
                   include_supplementary: bool = False, downsample_rate: float = 1.0, intervals_per_call: int = 16,
                   rank: int = 0, world: int = 1, reader_threads: int = None, intervals_per_read: int = 1, T: dict = None,
                   region_bed: str = None, merge: bool = True, inflate_helpers: int = None):
    """The reader side of generate_images (ImageGenerationUI.py:277-345): returns an iterator of (RegionBatch, interval of every
    batch region) pairs, `intervals_per_call` intervals per batch; interval i belongs to rank i % world (:211).

    The reference gives every worker PROCESS its own BAM/FASTA handles and lets it fetch, summarise and write one interval
    at a time (:222-260). Here `reader_threads` threads (default: the CPU share) each own a handle pair and fill
    `intervals_per_read` intervals at a time straight into the flat pv_batch_in arrays in native code (bamio.fill_batch, GIL
    released), running ahead of the consumer; the consuming thread merges `intervals_per_call` of them per batch (array
    concatenation). The arrays of a yielded batch stay valid until the generator is resumed.
    merge=False yields the LIST of per-read batches instead of their concatenation (Context.upload_batches lays them end to
    end on the device: no host copy of the ~15 MB per interval). A short job (fewer than four full calls' worth of intervals)
    gets smaller calls, so that the device starts on the first intervals while the last are still being read.
    T (optional dict) accumulates the reader-side stage times."""
    import os
    import threading
    import time
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    from .bamio import BamHandler, FastaHandler, fill_batch
    from .batch import merge_batches
    T = T if T is not None else {}
    for k in ("read_inflate_cpu_s", "read_decode_cpu_s", "reader_stall_s", "merge_s"):
        T.setdefault(k, 0.0)
    for k in ("bytes_inflated", "bases", "reads"):
        T.setdefault(k, 0)
    bam, fasta = BamHandler(bam_path), FastaHandler(fasta_path)
    todo = list_intervals(fasta, bam, region, region_size, region_bed)
    mine = [iv for i, iv in enumerate(todo) if i % world == rank]
    intervals_per_call = max(1, min(int(intervals_per_call), max(4, (len(mine) + 3) // 4)))
    ipr = max(1, min(int(intervals_per_read), int(intervals_per_call)))
    groups = [mine[k:k + ipr] for k in range(0, len(mine), ipr)]
    reads_per_call = max(1, int(intervals_per_call) // ipr)
    budget = max(1, int(reader_threads or min(cpu_share(), 16)))     # host threads (beyond 16 the readers stop scaling)
    # Every thread of the budget reads its own interval. `inflate_helpers` > 0 instead gives every reader that many helper
    # threads which inflate BGZF blocks ahead of it (BamHandler.set_threads; htslib's hts_set_threads): an interval is then
    # read ~3x sooner and fewer are in flight. Measured on the MI355X box's 16-CPU quota (3.2 Mbp job, fused pipeline): 16 x 0
    # 107-118 ms, 4 x 3 137-260 ms, 8 x 1 160-170 ms - helpers sleep and wake once per few blocks, and threads that hop between
    # the host's CPUs strand the cgroup's quota slices, which starves the thread that feeds the GPU; so it stays opt-in.
    inflate_helpers = max(0, int(inflate_helpers or 0))
    n_thr = max(1, min(budget // (inflate_helpers + 1), max(len(groups), 1)))
    T["reader_threads"], T["inflate_helpers"], T["intervals"] = n_thr, inflate_helpers, len(mine)
    T.setdefault("read_helper_cpu_s", 0.0)
    tls = threading.local()

    def read_group(ivs):
        if not hasattr(tls, "h"):
            tls.h = (BamHandler(bam_path), FastaHandler(fasta_path))   # one handle pair per reader thread
            if inflate_helpers:
                tls.h[0].set_threads(inflate_helpers)
        return fill_batch(tls.h[0], tls.h[1], ivs, min_mapq, include_supplementary, downsample_rate, REGION_SAFE_BASES)

    # the pool is started and the first reads are submitted HERE, not at the first next(): the caller can do its own set-up
    # (load the model) while the first intervals are being read
    pool = ThreadPoolExecutor(n_thr)
    pending = deque()
    nxt = 0
    ahead = reads_per_call + n_thr + 2                               # reads in flight: one batch's worth + the pool
    while nxt < len(groups) and len(pending) < ahead:
        pending.append(pool.submit(read_group, groups[nxt]))
        nxt += 1

    def batches():
        nonlocal nxt
        try:
            while pending:
                fbs = []
                t0 = time.perf_counter()
                # the next read (waiting for it if need be) and whatever has completed behind it: when the readers are the
                # slower side every interval goes to the device the moment it is there, when the device is, the calls fill up
                # by themselves
                while pending and len(fbs) < reads_per_call and (not fbs or pending[0].done()):
                    fbs.append(pending.popleft().result())
                    if nxt < len(groups):
                        pending.append(pool.submit(read_group, groups[nxt]))
                        nxt += 1
                T["reader_stall_s"] += time.perf_counter() - t0
                names = []                                           # interval of every batch region, in batch order
                for fb in fbs:
                    T["read_inflate_cpu_s"] += fb.t_inflate
                    T["read_decode_cpu_s"] += fb.t_total - fb.t_inflate
                    T["read_helper_cpu_s"] += fb.t_helpers
                    T["bytes_inflated"] += fb.bytes_inflated
                    names += [fb.intervals[int(i)] for i in fb.interval_index]
                t0 = time.perf_counter()
                if merge:
                    batch = merge_batches([fb.batch for fb in fbs])
                else:
                    batch = [fb.batch for fb in fbs if fb.batch.n_regions]
                T["merge_s"] += time.perf_counter() - t0
                n_reg = batch.n_regions if merge else len(names)
                if n_reg:                                            # "no group when no reads" (AlignmentSummarizer.py:212-213)
                    T["bases"] += batch.n_bases if merge else sum(b.n_bases for b in batch)
                    T["reads"] += batch.n_reads if merge else sum(b.n_reads for b in batch)
                    yield batch, names
                del batch
                for fb in fbs:
                    fb.close()
        finally:
            for f in pending:
                f.cancel()
            pool.shutdown(wait=True)
    return batches()


def generate_images(ctx, bam_path: str, fasta_path: str, output_dir: str, params: Params, region: str = None,
                    region_size: int = 100_000, min_mapq: int = 5, include_supplementary: bool = False,
                    downsample_rate: float = 1.0, intervals_per_call: int = 16, rank: int = 0, world: int = 1,
                    reader_threads: int = None, timers: dict = None, intervals_per_read: int = 1,
                    use_hp_info: bool = False, region_bed: str = None) -> int:
    """generate_images (ImageGenerationUI.py:277-345) on the MI355X path: intervals of region_size, interval i handled
    by rank i % world (:211), `intervals_per_call` intervals per builder launch chain, one HDF5 file per rank; the readers
    are `region_batches` above. `timers` (optional dict) receives the stage times in seconds.
    use_hp_info (`-hp`, ImageGenerationUI.py:48-71,206-207): the haplotag-aware builder (AlignmentSummarizerHP.py:176-233:
    the same fetch, RegionalSummaryGeneratorHP with window 20 / 48 planes, the reads' HP tags); the file name gets "_hp"."""
    import os
    import time
    from .hdf5io import ImageStore
    t_start = time.perf_counter()
    os.makedirs(output_dir, exist_ok=True)
    T = dict(builder_call_s=0.0, hdf5_write_s=0.0)
    n_windows = 0
    if use_hp_info:
        from .batch import hp_params
        params = hp_params(params)
    fname = "pepper_variants_images_thread_%d%s.hdf5" % (rank, "_hp" if use_hp_info else "")
    if not use_hp_info and intervals_per_read == 1:
        # the builder stage of the fused call_variant pipeline with the image file as its only output: the readers' per-interval
        # arrays go straight to the device (no host concatenation), every interval as soon as it is read, the file is written
        # by its own thread
        from . import pipeline
        T2 = {}
        n_windows = pipeline.call_variant_fused(ctx, None, bam_path, fasta_path, None, params, region, region_size, min_mapq,
                                                include_supplementary, downsample_rate, 512, intervals_per_call, rank, world,
                                                reader_threads, os.path.join(output_dir, fname), T2, region_bed=region_bed)
        T2["builder_call_s"] = T2["upload_s"] + T2["device_call_s"] + T2["readback_s"]
        T2["wall_s"] = time.perf_counter() - t_start
        if timers is not None:
            timers.update(T2)
        return n_windows
    with ImageStore(os.path.join(output_dir, fname), "w") as store:
        for batch, names in region_batches(bam_path, fasta_path, region, region_size, min_mapq, include_supplementary,
                                           downsample_rate, intervals_per_call, rank, world, reader_threads, intervals_per_read, T,
                                           region_bed):
            t0 = time.perf_counter()
            out = ctx.summarize_hp(batch, params) if use_hp_info else ctx.summarize(batch, params)
            T["builder_call_s"] += time.perf_counter() - t0
            t0 = time.perf_counter()
            for g, (contig, start, end) in enumerate(names):
                sel = np.flatnonzero(out.region == g)
                # an interval with reads but no candidate still gets its (empty) group, as the reference's write_summary
                # of empty lists does (ImageGenerationUI.py:228-259)
                store.write_summary("%s_%d_%d" % (contig, start, end), [contig] * sel.size, out.position[sel], out.depth[sel],
                                    [[out.candidates[j]] for j in sel], out.cand_freq[sel].reshape(-1, 1), out.images[sel])
            T["hdf5_write_s"] += time.perf_counter() - t0
            n_windows += len(out)
    T["wall_s"] = time.perf_counter() - t_start
    T["windows"] = n_windows
    if timers is not None:
        timers.update(T)
    return n_windows


# CLI name -> Params field of the per-threshold overrides (CallVariantsArguments.py / MakeImagesArguments.py; None = preset)
_IMAGE_OVERRIDES = (("min_snp_baseq", "min_snp_baseq", float), ("min_indel_baseq", "min_indel_baseq", float),
                    ("snp_frequency", "snp_freq_threshold", float), ("insert_frequency", "insert_freq_threshold", float),
                    ("delete_frequency", "delete_freq_threshold", float), ("min_coverage_threshold", "min_coverage_threshold", float),
                    ("candidate_support_threshold", "candidate_support_threshold", float),
                    ("snp_candidate_frequency_threshold", "snp_candidate_freq_threshold", float),
                    ("indel_candidate_frequency_threshold", "indel_candidate_freq_threshold", float))


def add_image_arguments(ap):
    for cli, _, typ in _IMAGE_OVERRIDES:
        ap.add_argument("--" + cli, type=typ, default=None)
    ap.add_argument("--skip_indels", action="store_true", default=False)


def image_options_from_args(args, preset: str):
    """-> (Params, min_mapq): the preset's scalars with any explicit override applied (SetParameters.py `if X is None`)"""
    import dataclasses
    from .batch import PRESET_MIN_MAPQ, PRESETS
    over = {field: getattr(args, cli) for cli, field, _ in _IMAGE_OVERRIDES if getattr(args, cli, None) is not None}
    if getattr(args, "skip_indels", False):
        over["skip_indels"] = True
    params = dataclasses.replace(PRESETS[preset], **over)
    mq = getattr(args, "min_mapq", None)
    return params, (PRESET_MIN_MAPQ[preset] if mq is None else int(mq))


def run(args):
    import sys
    from . import cli
    from .runtime import Context
    preset = cli.preset_of(args)
    params, min_mapq = image_options_from_args(args, preset)
    rank, world, device = cli.rank_world_device(args)
    ctx = Context(device)
    n = generate_images(ctx, args.bam, args.fasta, args.output_dir, params, args.region, args.region_size,
                        min_mapq, args.include_supplementary, args.downsample_rate, rank=rank, world=world,
                        reader_threads=args.threads, use_hp_info=args.use_hp_info, region_bed=args.region_bed)
    ctx.close()
    sys.stderr.write("INFO: FINISHED IMAGE GENERATION: %d WINDOWS\n" % n)
    return 0


def main(argv=None):
    from . import cli
    return run(cli.make_images_parser().parse_args(argv))


if __name__ == "__main__":
    main()
