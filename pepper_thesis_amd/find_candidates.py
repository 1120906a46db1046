"""`find_candidates`: prediction HDF files -> candidate selection -> VCF records.

CPU-side consumer of the hot path, restated from the reference so that the pipeline BAM -> images ->
predictions -> VCF runs end to end on this code base:
  small_chunk_stitch        pepper_variant/modules/python/CandidateFinder.py:356-529
  find_candidates (dedupe)  CandidateFinder.py:532-581
  candidate_list_to_variant pepper_variant/modules/python/VcfWriter.py:48-138
  write_vcf_records         VcfWriter.py:140-218   (header fields :220-289)
The reference writes bgzipped + tabix-indexed VCFs through pysam (VcfWriter.py:21-46); pysam/htslib are not available
here, so the same five `.vcf.gz` files (FULL, PEPPER, VARIANT_CALLING, *_SNPs, *_INDEL) and their `.tbi` indexes are written
by the native BGZF / tabix writer of csrc/pv_io.cpp (pvio_write_vcf_gz), with the same records and FILTER/FORMAT fields.
Parity note: the reference module cannot be imported here (needs pysam/h5py/the pybind build) and ships no
tests or fixtures for it: this restatement is pinned by nothing ("parity unpinned"); tests check its
invariants and that GPU and oracle probabilities give identical records.
"""
import math
import os
from collections import defaultdict
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Tuple

import numpy as np


@dataclass
class CandidateOptions:
    """--ont_r9_guppy5_sup candidate-finding preset (SetParameters.py:39-66)"""
    allowed_multiallelics: int = 4
    snp_p_value: float = 0.1
    insert_p_value: float = 0.1
    delete_p_value: float = 0.1
    snp_q_cutoff: int = 20
    indel_q_cutoff: int = 15
    report_snp_above_freq: float = 0
    report_indel_above_freq: float = 0
    snp_p_value_in_lc: float = 0.1
    insert_p_value_in_lc: float = 0.15
    delete_p_value_in_lc: float = 0.1
    snp_q_cutoff_in_lc: int = 20
    indel_q_cutoff_in_lc: int = 10


# candidate-finding scalars per platform preset, in CandidateOptions field order
# (SetParameters.py:39-66 guppy5_sup, :93-121 guppy4_hac, :148-176 r10_q20, :202-230 hifi, :256-283 clr)
CANDIDATE_PRESETS = {
    "ont_r9_guppy5_sup": CandidateOptions(4, 0.1, 0.1, 0.1, 20, 15, 0, 0, 0.1, 0.15, 0.1, 20, 10),
    "ont_r9_guppy4_hac": CandidateOptions(4, 0.10, 0.25, 0.25, 20, 15, 0, 0, 0.05, 0.01, 0.01, 20, 10),
    "ont_r10_q20": CandidateOptions(4, 0.00001, 0.001, 0.001, 15, 30, 0, 0, 0.000001, 0.001, 0.001, 20, 35),
    "hifi": CandidateOptions(4, 0, 0, 0, 15, 20, 0, 0, 0, 0, 0, 15, 20),
    "clr": CandidateOptions(4, 0.1, 0.2, 0.2, 20, 20, 0, 0, 0.05, 0.05, 0.05, 20, 20),
}


def add_candidate_arguments(ap):
    """the per-threshold overrides of FindCandidatesArguments / CallVariantsArguments (None = take the preset's value,
    SetParameters.py `if options.X is None`)"""
    import dataclasses
    for f in dataclasses.fields(CandidateOptions):
        ap.add_argument("--" + f.name, type=int if f.type in (int, "int") else float, default=None)


def candidate_options_from_args(args, preset: str) -> CandidateOptions:
    import dataclasses
    base = CANDIDATE_PRESETS[preset]
    over = {f.name: getattr(args, f.name) for f in dataclasses.fields(CandidateOptions)
            if getattr(args, f.name, None) is not None}
    return dataclasses.replace(base, **over)


def repeat_annotation(sequence: str, kmer_size: int) -> List[int]:
    """CandidateFinder.py:277-297"""
    max_observed_repeats = [1] * len(sequence)
    for i in range(len(sequence) - (kmer_size - 1)):
        kmer_count = 0
        end_index = i + (kmer_size - 1)
        for j in range(i, len(sequence), kmer_size):
            if sequence[i:i + kmer_size] == sequence[j:j + kmer_size]:
                kmer_count += 1
            else:
                break
            end_index = j + kmer_size
        for k in range(i, min(len(sequence), end_index)):
            max_observed_repeats[k] = max(max_observed_repeats[k], kmer_count)
    return max_observed_repeats


def _valid(allele: str) -> bool:
    return all(b in "ACGT" for b in allele)


def select_candidates(records: Iterable[dict], get_ref: Callable[[str, int, int], str], opt: CandidateOptions):
    """small_chunk_stitch for an iterable of prediction records
    dict(contig, position, depth, candidates [str], candidate_frequency [int], prediction [3 floats]);
    get_ref(contig, start, stop) = FASTA_handler.get_reference_sequence. -> list of DeepVariant-side tuples
    (contig, start, end, ref_allele, alt_alleles, genotype, depth, supports, prediction_value, predictions,
     non_alt_predictions, in_repeat)"""
    selected = []
    for c in records:
        contig, pos = c["contig"], int(c["position"])
        reference_base = get_ref(contig, pos, pos + 1).upper()
        upstream = get_ref(contig, pos, pos + 10).upper()
        downstream = get_ref(contig, max(0, pos - 10), pos).upper()
        full = downstream + upstream
        hp = repeat_annotation(full, 1)
        pidx = len(downstream)
        up_i, down_i = min(len(hp), pidx + 4), max(0, pidx - 5)
        in_repeat = max(hp[down_i:up_i]) >= 5
        if reference_base not in ("A", "C", "G", "T"):
            continue
        pred = np.asarray(c["prediction"], dtype=np.float64)
        g = int(np.argmax(pred))
        genotype = [0, 0] if g == 0 else ([0, 1] if g == 1 else [1, 1])
        prediction_value = pred[g]
        alt_alleles, supports, non_alt_predictions = [], [], []
        reference_allele = reference_base
        depth = int(c["depth"])
        for alt_allele, freq in zip(c["candidates"], c["candidate_frequency"]):
            alt_type, allele = alt_allele[0], alt_allele[1:]
            if not _valid(allele):
                continue
            vaf = float(freq) / float(depth)
            nap = max(pred[1], pred[2])
            non_alt_predictions.append(nap)
            if alt_type == "1":
                if (not in_repeat and nap >= opt.snp_p_value) or (in_repeat and nap >= opt.snp_p_value_in_lc) or \
                        (0 < opt.report_snp_above_freq <= vaf):
                    alt_alleles.append(allele)
                    supports.append(int(freq))
            elif alt_type == "2":
                if (not in_repeat and nap >= opt.insert_p_value) or (in_repeat and nap >= opt.insert_p_value_in_lc) or \
                        (0 < opt.report_indel_above_freq <= vaf):
                    alt_alleles.append(allele)
                    supports.append(int(freq))
            elif alt_type == "3":
                if (not in_repeat and nap >= opt.delete_p_value) or (in_repeat and nap >= opt.delete_p_value_in_lc):
                    alt_alleles.append(reference_allele)  # deletion: REF = anchor + deleted, ALT = previous REF
                    reference_allele = allele
                    supports.append(int(freq))
                elif 0 < opt.report_indel_above_freq <= vaf:
                    alt_alleles.append(allele)
                    supports.append(int(freq))
        if alt_alleles:
            selected.append((contig, pos, pos + len(reference_allele), reference_allele, alt_alleles, genotype, depth,
                             supports, prediction_value, pred, non_alt_predictions, in_repeat))
    return selected


def _in_repeat_rows(W: np.ndarray, pidx: np.ndarray, full_len: np.ndarray) -> np.ndarray:
    """`max(repeat_annotation(full, 1)[max(0, pidx - 5):min(len(full), pidx + 4)]) >= 5` for many sites at once. W [n,20] holds
    `full` (the <= 10 bases before the site + the <= 10 from it on) left-aligned, the rest filled with values that equal nothing
    next to them. With k-mer size 1 the annotation of a base is the length of the homopolymer run it lies in, so the test is:
    does a run of >= 5 equal bases touch the slice?"""
    eq = W[:, 1:] == W[:, :-1]
    run5 = eq[:, 0:16] & eq[:, 1:17] & eq[:, 2:18] & eq[:, 3:19]   # bases a .. a + 4 are equal
    lo = np.maximum(0, pidx - 5)
    hi = np.minimum(full_len, pidx + 4)
    a = np.arange(16)
    return (run5 & (a[None, :] <= hi[:, None] - 1) & (a[None, :] + 4 >= lo[:, None])).any(axis=1)


def select_candidates_batch(batch: Dict[str, np.ndarray], get_ref: Callable[[str, int, int], str], opt: CandidateOptions):
    """select_candidates for one prediction batch as read from the file (arrays contigs S[n], positions, depths, candidates
    [n,k] of str / bytes, candidate_frequency [n,k], base_prediction [n,3]): the same tuples in the same order, computed for
    all single-candidate windows of a contig at once (one reference fetch per contig instead of three per window, the
    homopolymer test on a [n,20] matrix). Windows with several candidates, a zero depth, or on a contig with fewer than 20
    bases take the per-record path."""
    n = len(batch["positions"])
    if n == 0:
        return []
    cands = batch["candidates"]

    def rec(i):
        return dict(contig=batch["contigs"][i].decode() if isinstance(batch["contigs"][i], bytes) else str(batch["contigs"][i]),
                    position=int(batch["positions"][i]), depth=int(batch["depths"][i]),
                    candidates=[x.decode() if isinstance(x, bytes) else str(x) for x in cands[i]],
                    candidate_frequency=[int(x) for x in np.atleast_1d(batch["candidate_frequency"][i])], prediction=batch["base_prediction"][i])
    if cands.ndim != 2 or cands.shape[1] != 1:
        return select_candidates((rec(i) for i in range(n)), get_ref, opt)
    pos_all = np.asarray(batch["positions"]).astype(np.int64)
    depth_all = np.asarray(batch["depths"]).astype(np.int64)
    freq_all = np.asarray(batch["candidate_frequency"]).reshape(n).astype(np.int64)
    pred_all = np.asarray(batch["base_prediction"], dtype=np.float64)
    contig_col = np.asarray(batch["contigs"])
    out = [None] * n            # per window: None (nothing selected) or its tuple
    slow = []
    acgt = np.zeros(256, bool)
    acgt[[65, 67, 71, 84]] = True
    for cname in np.unique(contig_col):
        idx = np.flatnonzero(contig_col == cname)
        contig = cname.decode() if isinstance(cname, bytes) else str(cname)
        pos = pos_all[idx]
        p_lo, p_hi = int(pos.min()), int(pos.max())
        span0 = max(0, p_lo - 10)
        ref = get_ref(contig, span0, p_hi + 10).upper().encode()
        R = np.frombuffer(ref, np.uint8)
        clen = span0 + len(R)                      # bases known to exist (the fetch clamps at the contig end)
        if len(R) < 20 or (depth_all[idx] == 0).any() or (pos < 0).any():
            slow.extend(idx.tolist())
            continue
        # full = ref[max(0, pos - 10):pos] + ref[pos:pos + 10], left-aligned in 20 columns
        start = np.maximum(0, pos - 10)
        pidx = pos - start
        stop = np.minimum(clen, pos + 10)
        full_len = np.maximum(0, stop - start)
        cols = np.arange(20)
        gi = start[:, None] + cols[None, :] - span0
        inside = cols[None, :] < full_len[:, None]
        W = np.where(inside, R[np.clip(gi, 0, len(R) - 1)], (128 + cols)[None, :].astype(np.uint8))
        in_rep = _in_repeat_rows(W, pidx, full_len)
        has_base = pos < clen
        ref_base = np.where(has_base, R[np.clip(pos - span0, 0, len(R) - 1)], 0).astype(np.uint8)
        ref_ok = acgt[ref_base]
        pred = pred_all[idx]
        g = np.argmax(pred, axis=1)
        nap = np.maximum(pred[:, 1], pred[:, 2])
        freq = freq_all[idx]
        depth = depth_all[idx]
        vaf = freq.astype(np.float64) / depth.astype(np.float64)
        thr = {"1": np.where(in_rep, opt.snp_p_value_in_lc, opt.snp_p_value), "2": np.where(in_rep, opt.insert_p_value_in_lc, opt.insert_p_value),
               "3": np.where(in_rep, opt.delete_p_value_in_lc, opt.delete_p_value)}
        by_p = {t: nap >= thr[t] for t in thr}
        by_f_snp = (vaf >= opt.report_snp_above_freq) if opt.report_snp_above_freq > 0 else np.zeros(len(idx), bool)
        by_f_indel = (vaf >= opt.report_indel_above_freq) if opt.report_indel_above_freq > 0 else np.zeros(len(idx), bool)
        gts = ([0, 0], [0, 1], [1, 1])
        cl = cands[idx, 0]
        for j in np.flatnonzero(ref_ok):
            c = cl[j]
            if isinstance(c, bytes):
                c = c.decode()
            t, allele = c[:1], c[1:]
            if allele.strip("ACGT"):           # an allele with anything but A/C/G/T is skipped (and adds no non-alt prediction)
                continue
            rb = chr(ref_base[j])
            if t == "1":
                if not (by_p["1"][j] or by_f_snp[j]):
                    continue
                ref_allele, alt = rb, allele
            elif t == "2":
                if not (by_p["2"][j] or by_f_indel[j]):
                    continue
                ref_allele, alt = rb, allele
            elif t == "3":
                if by_p["3"][j]:
                    ref_allele, alt = allele, rb   # deletion: REF = anchor + deleted, ALT = previous REF
                elif by_f_indel[j]:
                    ref_allele, alt = rb, allele
                else:
                    continue
            else:
                continue
            p = int(pos[j])
            gj = int(g[j])
            out[idx[j]] = (contig, p, p + len(ref_allele), ref_allele, [alt], list(gts[gj]), int(depth[j]), [int(freq[j])], pred[j][gj],
                           pred[j], [nap[j]], bool(in_rep[j]))
    if slow:
        for i in slow:
            r = select_candidates([rec(i)], get_ref, opt)
            out[i] = r[0] if r else None
    return [t for t in out if t is not None]


def dedupe_by_position(selected) -> Dict[Tuple[str, int], list]:
    """find_candidates tail (:548-574): sort by (contig, pos), keep the first record per (ref, alt)"""
    out, seen = defaultdict(list), defaultdict(list)
    for cand in sorted(selected, key=lambda x: (x[0], x[1])):
        key, ra = (cand[0], cand[1]), (cand[3], cand[4][0])
        if ra in seen[key]:
            continue
        seen[key].append(ra)
        out[key].append(cand)
    return out


def candidate_list_to_variant(candidates, opt: CandidateOptions):
    """VcfWriter.py:48-138"""
    if len(candidates) == 1 and opt.allowed_multiallelics >= 1:   # the usual site: what the general path below gives for one record
        contig, rs, re_, ref, alts, gt, depth, sup, gp, preds, naps, rep = candidates[0]
        g = int(np.argmax(preds))
        gt_qual = preds[g] if g != 0 else max(preds[1], preds[2])
        return contig, rs, rs + len(ref), ref, [alts[0]], ([0, 0], [0, 1], [1, 1])[g], depth, [sup[0]], gt_qual, list(naps), bool(rep)
    candidates = sorted(candidates, key=lambda x: (x[5], x[8]), reverse=True)[:opt.allowed_multiallelics]
    max_ref = max((c[3] for c in candidates), key=len)
    norm = []
    for c in candidates:
        contig, rs, re_, ref, alts, gt, depth, sup, gp, preds, naps, rep = c
        if len(ref) < len(max_ref):
            suffix = max_ref[-(len(max_ref) - len(ref)):]
            ref, alts = ref + suffix, [a + suffix for a in alts]
        norm.append((contig, rs, re_, ref, alts, gt, depth, sup, gp, preds, naps, rep))
    gt_qual, hp1, hp2 = -1.0, [], []
    site = None
    site_alts, site_sup, site_naps, site_rep, site_depth = [], [], [], False, 0
    for i, c in enumerate(norm):
        contig, rs, re_, ref, alts, gt, depth, sup, gp, preds, naps, rep = c
        site_rep = rep or site_rep
        g = int(np.argmax(preds))
        if g != 0:
            gt_qual = preds[g] if gt_qual < 0 else min(gt_qual, preds[g])
        elif gt_qual < 0:
            gt_qual = max(preds[1], preds[2])
        if site is None:
            site = (contig, rs, rs + len(ref), ref)
            site_depth = depth
        site_depth = min(site_depth, depth)
        site_alts.append(alts[0])
        site_sup.append(sup[0])
        site_naps.extend(naps)
        if g == 1:
            hp1.append(i + 1)
        elif g == 2:
            hp1.append(i + 1)
            hp2.append(i + 1)
    if 0 < len(hp1) + len(hp2) <= 2:
        gt = hp1 + hp2
        if len(gt) == 1:
            gt = [0, gt[0]]
    else:
        gt = [0, 0]
    return site[0], site[1], site[2], site[3], site_alts, gt, site_depth, site_sup, gt_qual, site_naps, site_rep


def _fmt(v) -> str:
    if isinstance(v, (float, np.floating)):
        return ("%.6g" % float(v))
    return str(v)


VCF_HEADER = """##fileformat=VCFv4.2
##FILTER=<ID=PASS,Description="All filters passed">
##FILTER=<ID=refCall,Description="Call is homozygous">
##FILTER=<ID=lowGQ,Description="Low genotype quality">
##FILTER=<ID=lowQUAL,Description="Low variant call quality">
##FILTER=<ID=conflictPos,Description="Overlapping record">
##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">
##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Depth">
##FORMAT=<ID=AD,Number=A,Type=Integer,Description="Allele depth">
##FORMAT=<ID=VAF,Number=A,Type=Float,Description="Variant allele fractions.">
##FORMAT=<ID=AP,Number=A,Type=Float,Description="Maximum variant allele probability for each allele.">
##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype Quality">
##FORMAT=<ID=REP,Number=1,Type=String,Description="If set to 1 then variant site is considered to be ina LowCompexity repeat region">
"""


def variant_records(variants: Dict[Tuple[str, int], list], opt: CandidateOptions):
    """write_vcf_records (:140-218) as data: yields (line, selected_for_variant_calling, is_snp)"""
    last_position = -1
    log10 = math.log10
    for key in sorted(variants):
        contig, rs, re_, ref, alleles, genotype, depth, sup, gp, naps, rep = candidate_list_to_variant(variants[key], opt)
        if len(alleles) <= 0 or rs == last_position:
            continue
        last_position = rs
        max_alt_len = max(len(ref), max(len(x) for x in alleles))
        qual = max(1, int(-10 * log10(max(0.000000001, 1.0 - gp))))
        is_snp = max_alt_len == 1
        if is_snp:
            failed = qual <= (opt.snp_q_cutoff_in_lc if rep else opt.snp_q_cutoff)
        else:
            failed = qual <= (opt.indel_q_cutoff_in_lc if rep else opt.indel_q_cutoff)
        selected = genotype == [0, 0] or failed
        d1 = max(1, depth)
        # (floats as "%.6g", integers as they are: _fmt)
        line = "%s\t%d\t.\t%s\t%s\t%d\t%s\t.\tGT:AP:GQ:DP:AD:VAF:REP\t%d/%d:%s:%d:%d:%s:%s:%s" % (
            contig, rs + 1, ref, ",".join(alleles), qual, "refCall" if genotype == [0, 0] else "PASS", genotype[0], genotype[1],
            ",".join(["%.6g" % float(x) for x in naps]), qual, depth, ",".join([str(x) for x in sup]),
            ",".join(["%.6g" % round(ad / d1, 3) for ad in sup]), "1" if rep else "0")
        yield line, selected, is_snp


def read_prediction_records(prediction_dir: str):
    """every batch of every *.hdf file (FindCandidates.py:145-166)"""
    from .hdf5io import PredictionStore
    for fn in sorted(os.listdir(prediction_dir)):
        if not fn.endswith("hdf"):
            continue
        with PredictionStore(os.path.join(prediction_dir, fn), "r") as st:
            for _, b in st.batches():
                for i in range(len(b["positions"])):
                    yield dict(contig=b["contigs"][i].decode(), position=int(b["positions"][i]), depth=int(b["depths"][i]),
                               candidates=[str(x) for x in b["candidates"][i]],
                               candidate_frequency=[int(x) for x in b["candidate_frequency"][i]], prediction=b["base_prediction"][i])


def read_prediction_batches(prediction_dir: str):
    """every batch of every *.hdf file as arrays (FindCandidates.py:145-166)"""
    from .hdf5io import PredictionStore
    for fn in sorted(os.listdir(prediction_dir)):
        if not fn.endswith("hdf"):
            continue
        with PredictionStore(os.path.join(prediction_dir, fn), "r") as st:
            for _, b in st.batches():
                yield b


class CandidateCollector:
    """select_candidates_batch over prediction batches as they are produced (the fused call_variant hands every call's windows
    over on its writer thread): `selected` is what process_candidates would have selected from the prediction file"""

    def __init__(self, fasta_path: str, opt: CandidateOptions):
        from .bamio import FastaHandler
        self.fasta, self.opt, self.selected = FastaHandler(fasta_path), opt, []

    def __call__(self, batch: Dict[str, np.ndarray]):
        self.selected += select_candidates_batch(batch, self.fasta.get_reference_sequence, self.opt)


def process_candidates(prediction_dir: str, fasta_path: str, sample_name: str, output_dir: str,
                       opt: CandidateOptions = CandidateOptions(), selected: list = None) -> Dict[str, int]:
    """candidate_finder (FindCandidates.py:131-190): prediction files -> five VCFs; returns the record counts.
    selected: the candidates already chosen from these predictions (CandidateCollector), then the files are not read again"""
    from .bamio import FastaHandler
    fasta = FastaHandler(fasta_path)
    if selected is None:
        selected = []
        for b in read_prediction_batches(prediction_dir):
            selected += select_candidates_batch(b, fasta.get_reference_sequence, opt)
    variants = dedupe_by_position(selected)
    os.makedirs(output_dir, exist_ok=True)
    contigs = []
    for c in sorted(variants):
        if c[0] not in contigs:
            contigs.append(c[0])
    header = VCF_HEADER + "".join("##contig=<ID=%s,length=%d>\n" % (n, fasta.get_chromosome_sequence_length(n))
                                  for n in fasta.get_chromosome_names()) + \
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s\n" % sample_name
    names = {"full": "PEPPER_VARIANT_FULL.vcf.gz", "pepper": "PEPPER_VARIANT_OUTPUT_PEPPER.vcf.gz",
             "vc": "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING.vcf.gz", "snp": "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_SNPs.vcf.gz",
             "indel": "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_INDEL.vcf.gz"}
    texts = {k: [header] for k in names}
    counts = dict(total=0, pepper=0, variant_calling=0, snp=0, indel=0)
    for line, selected_vc, is_snp in variant_records(variants, opt):
        texts["full"].append(line + "\n")
        counts["total"] += 1
        if selected_vc:
            texts["snp" if is_snp else "indel"].append(line + "\n")
            counts["snp" if is_snp else "indel"] += 1
            texts["vc"].append(line + "\n")
            counts["variant_calling"] += 1
        else:
            texts["pepper"].append(line + "\n")
            counts["pepper"] += 1
    from concurrent.futures import ThreadPoolExecutor
    from .bamio import write_vcf_gz
    # bgzip + tabix index, as VariantFile(..., 'w') + pysam.tabix_index (VcfWriter.py:21-46); the five files side by side (native
    # code, GIL released)
    with ThreadPoolExecutor(len(names)) as pool:
        for f in [pool.submit(write_vcf_gz, os.path.join(output_dir, fname), "".join(texts[k])) for k, fname in names.items()]:
            f.result()
    return counts


def run(args):
    import sys
    from . import cli
    preset = cli.preset_of(args)
    c = process_candidates(args.input_dir, args.fasta, args.sample_name, args.output_dir,
                           candidate_options_from_args(args, preset))
    sys.stderr.write("INFO: FINISHED PROCESSING, TOTAL CANDIDATES FOUND: %d (PEPPER %d, RE-GENOTYPING %d: SNP %d INDEL %d)\n" %
                     (c["total"], c["pepper"], c["variant_calling"], c["snp"], c["indel"]))
    return 0


def main(argv=None):
    from . import cli
    return run(cli.find_candidates_parser().parse_args(argv))


if __name__ == "__main__":
    main()
