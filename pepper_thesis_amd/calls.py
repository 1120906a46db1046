"""The hard decisions the consumer takes on the softmax output, restated so that "identical
candidate-variant calls" can be checked without the reference's find_candidates (which needs pysam / h5py):

  * genotype = argmax(prediction_base)                                  CandidateFinder.py:424-431
  * non_alt_prediction = max(p[1], p[2]) >= {snp,insert,delete}_p_value CandidateFinder.py:456-515
  * qual = max(1, int(-10 * log10(max(1e-9, 1 - p[genotype]))))         VcfWriter.py:156-157
    failed = qual <= {snp,indel}_q_cutoff                               VcfWriter.py:160-171

SURVEY D5: a 1e-4 agreement on probabilities does not by itself guarantee identical decisions at a
threshold edge, so `compare` reports exact mismatches and how many of them sit within `tol` of an edge.
Parity note: this restatement is pinned by nothing in the reference (no tests, module not importable here)."""
import math
from dataclasses import dataclass

import numpy as np


@dataclass
class CallOptions:
    """ONT R9 guppy5 sup preset, SetParameters.py:39-66"""
    snp_p_value: float = 0.1
    insert_p_value: float = 0.1
    delete_p_value: float = 0.1
    snp_q_cutoff: int = 20
    indel_q_cutoff: int = 15


def decision_signature(probs: np.ndarray, candidates, opt: CallOptions = CallOptions()) -> np.ndarray:
    """-> int array [N,4]: genotype, passes p-value, qual, failed-by-qual"""
    probs = np.asarray(probs, dtype=np.float64)
    n = probs.shape[0]
    out = np.zeros((n, 4), np.int64)
    for i in range(n):
        p = probs[i]
        g = int(np.argmax(p))
        t = candidates[i][0]
        thr = {"1": opt.snp_p_value, "2": opt.insert_p_value, "3": opt.delete_p_value}[t]
        qual = max(1, int(-10 * math.log10(max(0.000000001, 1.0 - p[g]))))
        cut = opt.snp_q_cutoff if t == "1" else opt.indel_q_cutoff
        out[i] = (g, int(max(p[1], p[2]) >= thr), qual, int(qual <= cut))
    return out


def near_edge(probs: np.ndarray, candidates, tol: float, opt: CallOptions = CallOptions()) -> np.ndarray:
    """bool [N]: some decision of this window is within `tol` (in probability) of flipping"""
    probs = np.asarray(probs, dtype=np.float64)
    n = probs.shape[0]
    edge = np.zeros(n, bool)
    for i in range(n):
        p = np.sort(probs[i])
        t = candidates[i][0]
        thr = {"1": opt.snp_p_value, "2": opt.insert_p_value, "3": opt.delete_p_value}[t]
        e = (p[2] - p[1]) < 2 * tol or abs(max(probs[i][1], probs[i][2]) - thr) < tol
        # phred edges: 1 - p crosses 10^(-q/10) for an integer q
        x = max(1e-9, 1.0 - p[2])
        q = -10 * math.log10(x)
        for qq in (math.floor(q), math.ceil(q)):
            if abs(x - 10 ** (-qq / 10)) < tol:
                e = True
        edge[i] = e
    return edge


def compare(probs_a, probs_b, candidates, tol=1e-4, opt: CallOptions = CallOptions()):
    """-> dict(n, mismatches, mismatches_near_edge, mismatches_off_edge)"""
    sa, sb = decision_signature(probs_a, candidates, opt), decision_signature(probs_b, candidates, opt)
    mism = (sa != sb).any(axis=1)
    edge = near_edge(probs_b, candidates, tol, opt)
    return dict(n=len(sa), mismatches=int(mism.sum()), mismatches_near_edge=int((mism & edge).sum()),
                mismatches_off_edge=int((mism & ~edge).sum()))
