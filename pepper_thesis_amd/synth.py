"""Deterministic synthetic inputs of the shapes SURVEY.md section 8(d) fixes.

No real BAMs or checkpoints exist offline, so the benchmark and the large parity tests run on
region batches produced here: R-column regions at depth C with region-clipped reads that obey the
``BAM_handler.get_reads`` output contract (first and last retained op are
M-like, bases upper-case IUPAC, raw phred qualities), per-base noise and planted variant sites.
"""
from typing import List, Optional

import numpy as np

from .batch import Read, Region, RegionBatch, pack_regions

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
OP_M, OP_I, OP_D = 0, 1, 2


def _rle_ops(ops: np.ndarray) -> np.ndarray:
    """run-length encode a stream of per-base op codes into BAM-packed CIGAR words"""
    if ops.size == 0:
        return np.zeros(0, np.uint32)
    change = np.flatnonzero(ops[1:] != ops[:-1]) + 1
    starts = np.concatenate(([0], change))
    lens = np.diff(np.concatenate((starts, [ops.size])))
    return ((lens.astype(np.uint32) << 4) | ops[starts].astype(np.uint32)).astype(np.uint32)


def synth_read(rng: np.random.Generator, ref: np.ndarray, region_start: int, s: int, e: int, planted,
               mismatch=0.03, ins_rate=0.02, del_rate=0.03, qual_lo=5, qual_hi=35, n_rate=0.0) -> Optional[Read]:
    """One read covering region columns [s, e) (already clipped to the region).

    planted: list of (col, kind, payload, carried_by_this_read)
    """
    n = e - s
    if n < 2:
        return None
    show = ref[s:e].copy()                       # base the read shows at each column
    deleted = np.zeros(n, bool)
    ins_len = np.zeros(n, np.int64)
    ins_payload = {}

    # sequencing noise
    mm = rng.random(n) < mismatch
    if mm.any():
        shift = rng.integers(1, 4, size=int(mm.sum()))
        code = np.searchsorted(_ACGT, show[mm])
        code = np.where(_ACGT[np.clip(code, 0, 3)] == show[mm], code, 0)
        show[mm] = _ACGT[(code + shift) % 4]
    if n_rate > 0:
        nn = rng.random(n) < n_rate
        show[nn] = ord("N")
    dstart = np.flatnonzero(rng.random(n) < del_rate)
    for d in dstart:
        L = int(rng.integers(1, 4))
        deleted[d:d + L] = True
    ist = np.flatnonzero(rng.random(n) < ins_rate)
    ins_len[ist] = rng.integers(1, 4, size=ist.size)

    # planted variants carried by this read
    for col, kind, payload, carried in planted:
        i = col - s
        if not carried or i < 0 or i >= n:
            continue
        if kind == "snp":
            show[i] = payload
            deleted[i] = False
        elif kind == "ins":
            deleted[i] = False
            ins_len[i] = len(payload)
            ins_payload[i] = np.frombuffer(payload, dtype=np.uint8)
        elif kind == "del":
            deleted[i] = False
            deleted[i + 1:i + 1 + payload] = True
            ins_len[i:i + 1 + payload] = 0

    # the get_reads contract: first and last kept op are aligned bases, no insert after the last
    deleted[0] = False
    deleted[-1] = False
    ins_len[-1] = 0
    ins_len[deleted] = 0  # keep CIGARs simple: no insert hanging off a deleted column

    # per-column emitted symbols: [aligned base]? + inserted bases
    cnt = (~deleted).astype(np.int64) + ins_len
    total = int(cnt.sum())
    off = np.concatenate(([0], np.cumsum(cnt)))
    bases = rng.choice(_ACGT, size=total)
    keep = np.flatnonzero(~deleted)
    bases[off[keep]] = show[keep]
    for i, payload in ins_payload.items():
        if ins_len[i] != payload.size or deleted[i]:
            continue
        st = off[i] + 1
        bases[st:st + payload.size] = payload
    # op stream: per column M or D, then I * ins_len
    per_col = 1 + ins_len
    ops = np.full(int(per_col.sum()), OP_I, np.uint8)
    ooff = np.concatenate(([0], np.cumsum(per_col)))[:-1]
    ops[ooff] = np.where(deleted, OP_D, OP_M)
    cigar = _rle_ops(ops)
    quals = rng.integers(qual_lo, qual_hi + 1, size=total).astype(np.uint8)
    return Read(region_start + s, cigar, bases.astype(np.uint8).tobytes(), quals,
                bool(rng.random() < 0.5), int(rng.choice([60, 60, 60, 60, 30, 10, 0], p=[.3, .3, .2, .1, .05, .04, .01])))


def synth_region(seed: int, region_len: int = 100_200, depth: int = 60, read_len: int = 10_000,
                 site_every: int = 200, ref_start: int = 1_000_000, safe: int = 100,
                 mismatch=0.03, ins_rate=0.02, del_rate=0.03, n_rate=0.0, ref_n_rate=0.0,
                 contig: str = "chr20") -> Region:
    """SURVEY 8(d) image-builder workload: R columns, depth C, reads of read_len clipped to the region,
    3 % mismatch / 2 % insert / 3 % delete noise, quals U[5,35], planted sites every ~site_every bp."""
    rng = np.random.default_rng(seed)
    R = int(region_len)
    ref = rng.choice(_ACGT, size=R).astype(np.uint8)
    if ref_n_rate > 0:
        ref[rng.random(R) < ref_n_rate] = ord("N")
    # planted sites
    sites = []
    col = int(rng.integers(20, max(21, site_every)))
    while col < R - 40:
        kind = rng.choice(["snp", "snp", "ins", "del"])
        af = float(rng.choice([0.5, 1.0, 0.3, 0.15]))
        if kind == "snp":
            alt = _ACGT[(int(np.searchsorted(_ACGT, ref[col])) + int(rng.integers(1, 4))) % 4]
            payload = int(alt)
        elif kind == "ins":
            payload = rng.choice(_ACGT, size=int(rng.integers(1, 6))).astype(np.uint8).tobytes()
        else:
            payload = int(rng.integers(1, 8))
        sites.append((col, kind, payload, af))
        col += int(rng.integers(max(2, site_every // 2), site_every * 3 // 2 + 1))
    n_reads = max(1, int(round(depth * (R + read_len) / read_len)))
    starts = np.sort(rng.integers(-read_len + 1, R, size=n_reads))
    reads: List[Read] = []
    site_cols = np.asarray([s[0] for s in sites], dtype=np.int64)
    for st in starts:
        L = int(max(50, rng.normal(read_len, read_len * 0.1)))
        s, e = max(0, int(st)), min(R, int(st) + L)
        if e - s < 2:
            continue
        lo, hi = np.searchsorted(site_cols, s), np.searchsorted(site_cols, e)
        planted = [(sites[k][0], sites[k][1], sites[k][2], bool(rng.random() < sites[k][3])) for k in range(lo, hi)]
        rd = synth_read(rng, ref, ref_start, s, e, planted, mismatch, ins_rate, del_rate, n_rate=n_rate)
        if rd is not None:
            reads.append(rd)
    return Region(ref_start, ref_start + R - 1, ref.tobytes(), reads,
                  ref_start + safe, ref_start + R - 1 - safe, contig)


def synth_batch(seed: int, n_regions: int, **kw) -> RegionBatch:
    step = kw.get("region_len", 100_200) - 200
    base = kw.pop("ref_start", 1_000_000)
    return pack_regions([synth_region(seed + 7919 * g, ref_start=base + g * step, **kw) for g in range(n_regions)])


def synth_windows(seed: int, n: int) -> np.ndarray:
    """SURVEY 8(d) RNN-P1 workload: int8 [n,33,26] windows with the statistics of real summaries:
    col 0 ~ U{1..5}; cols 4,15 ~ -Binomial(30,.9); one dominant base plane per row per strand
    ~ -Binomial(30,.9), the others -Poisson(.5); row 16 carries a SNP overlay."""
    rng = np.random.default_rng(seed)
    img = np.zeros((n, 33, 26), np.int32)
    img[:, :, 0] = rng.integers(1, 6, size=(n, 33))
    for s0, rf in ((8, 4), (19, 15)):
        img[:, :, rf] = -rng.binomial(30, 0.9, size=(n, 33))
        img[:, :, s0:s0 + 7] = -rng.poisson(0.5, size=(n, 33, 7))
        dom = rng.integers(0, 4, size=(n, 33))
        val = -rng.binomial(30, 0.9, size=(n, 33))
        np.put_along_axis(img[:, :, s0:s0 + 7], dom[..., None], val[..., None], axis=2)
    alt = rng.integers(0, 4, size=n)
    idx = np.arange(n)
    img[idx, 16, 1] = alt + 1
    img[idx, 16, 5] = rng.integers(2, 20, size=n)
    img[idx, 16, 16] = rng.integers(2, 20, size=n)
    img[idx, 16, 8 + alt] *= -1
    img[idx, 16, 19 + alt] *= -1
    return np.clip(img, -128, 127).astype(np.int8)


def synth_p2_images(seed: int, n: int, seq_len: int = 1000, features: int = 10) -> np.ndarray:
    """SURVEY 8(d) RNN-P2 workload: uint8 [n, 1000, 10] ~ U[0,254]."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 255, size=(n, seq_len, features), dtype=np.uint8)


# ---- deterministic synthetic weights (no checkpoints exist offline) --------------------------------

def _splitmix_uniform(seed: int, n: int) -> np.ndarray:
    """n doubles in [0,1), counter-based splitmix64: identical on every NumPy version/platform."""
    with np.errstate(over="ignore"):
        z = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _uinit(seed, shape, bound):
    n = int(np.prod(shape))
    return ((2.0 * _splitmix_uniform(seed, n) - 1.0) * bound).astype(np.float32).reshape(shape)


def _rnn_weights(prefix, seed, gates, in_size, hidden):
    """PyTorch's default LSTM/GRU init law: U(-1/sqrt(H), 1/sqrt(H)) for every tensor."""
    k = 1.0 / np.sqrt(hidden)
    w = {}
    for d, suffix in enumerate(("", "_reverse")):
        s = seed + 1000 * d
        w["%s.weight_ih_l0%s" % (prefix, suffix)] = _uinit(s + 1, (gates * hidden, in_size), k)
        w["%s.weight_hh_l0%s" % (prefix, suffix)] = _uinit(s + 2, (gates * hidden, hidden), k)
        w["%s.bias_ih_l0%s" % (prefix, suffix)] = _uinit(s + 3, (gates * hidden,), k)
        w["%s.bias_hh_l0%s" % (prefix, suffix)] = _uinit(s + 4, (gates * hidden,), k)
    return w


def make_weights_p1(seed: int = 1234, head_gain: float = 1.0) -> dict:
    """state_dict (numpy fp32) of the pepper_variant TransducerGRU(26,1,256,28,3)
    (pepper_variant/modules/python/models/simple_model.py:23-46; shapes in SURVEY Appendix B).
    head_gain > 1 scales the Linear weights so that the random-init softmax is not almost uniform."""
    w = {}
    w.update(_rnn_weights("encoder", seed + 10, 4, 26, 256))
    w.update(_rnn_weights("decoder", seed + 5000, 4, 512, 256))
    sizes = [(512, 33 * 512), (512, 512), (512, 512), (512, 512), (512, 512)]
    for i, (o, k) in enumerate(sizes):
        b = 1.0 / np.sqrt(k)
        w["linear_%d.weight" % (i + 1)] = _uinit(seed + 9000 + 10 * i, (o, k), b * head_gain)
        w["linear_%d.bias" % (i + 1)] = _uinit(seed + 9001 + 10 * i, (o,), b)
    b = 1.0 / np.sqrt(512)
    w["output_layer_type.weight"] = _uinit(seed + 9900, (3, 512), b * head_gain)
    w["output_layer_type.bias"] = _uinit(seed + 9901, (3,), b)
    return w


def make_weights_p2(seed: int = 4321, dense_gain: float = 1.0) -> dict:
    """state_dict (numpy fp32) of the polisher TransducerGRU(1,10,1,128,5)
    (pepper/modules/python/models/simple_model.py:5-25)."""
    w = {}
    w.update(_rnn_weights("gru_encoder", seed + 10, 3, 10, 128))
    w.update(_rnn_weights("gru_decoder", seed + 5000, 3, 256, 128))
    b = 1.0 / np.sqrt(256)
    w["dense1.weight"] = _uinit(seed + 9900, (5, 256), b * dense_gain)
    w["dense1.bias"] = _uinit(seed + 9901, (5,), b)
    return w
