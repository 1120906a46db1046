"""Drop-in mirror of the pybind11 class PEPPER_VARIANT.RegionalSummaryGenerator
(reference binding: pepper_variant/modules/cpp/pybind_api.h:55-62; call site
pepper_variant/modules/python/AlignmentSummarizer.py:220-238) on top of the C-ABI.

Same constructor, same two methods, same argument order and meaning; results come back as objects
with the attributes ImageGenerationUI.py:239-247 reads (contig, position, depth, candidates,
candidate_frequency, image_matrix, base_label, type_label). `image_matrix` is a numpy int8 [33,26]
array (the reference returns nested lists of int that DataStore.py:68 immediately casts to int8).
"""
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

from .batch import Params, Read, Region, pack_regions, pack_cigar
from .runtime import Context


@dataclass
class CandidateImageSummary:
    """region_summary.h:88-111"""
    contig: str
    position: int
    depth: int
    candidates: List[str]
    candidate_frequency: List[int]
    image_matrix: np.ndarray
    base_label: int = 0
    type_label: int = 0


def read_from_type_read(r) -> Read:
    """accepts anything with the type_read fields the builder uses (read.h:60-71): pos, sequence,
    base_qualities, cigar_tuples [(operation, length) or objects with .operation/.length],
    mapping_quality, flags.is_reverse, hp_tag (haplotag-aware builder only)"""
    if isinstance(r, Read):
        return r
    cig = [(c.operation, c.length) if hasattr(c, "operation") else (c[0], c[1]) for c in r.cigar_tuples]
    seq = r.sequence.encode() if isinstance(r.sequence, str) else bytes(r.sequence)
    return Read(int(r.pos), pack_cigar(cig), seq, np.asarray(r.base_qualities, dtype=np.uint8),
                bool(r.flags.is_reverse), int(r.mapping_quality), int(getattr(r, "hp_tag", 0)))


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class RegionalSummaryGenerator:
    def __init__(self, contig: str, region_start: int, region_end: int, reference_sequence, ctx: Context = None):
        self.contig = contig
        self.ref_start = int(region_start)
        self.ref_end = int(region_end)
        self.reference_sequence = reference_sequence.encode() if isinstance(reference_sequence, str) else bytes(reference_sequence)
        self.ctx = ctx

    def generate_max_insert_summary(self, reads: Sequence) -> None:
        """region_summary.cpp:69-96 with GENERATE_INDELS == false (region_summary.h:50): positions[i] =
        ref_start + i, index[i] = 0 — the identity map, nothing to materialise."""
        return None

    def generate_summary(self, reads: Sequence, min_snp_baseq, min_indel_baseq, snp_freq_threshold,
                         insert_freq_threshold, delete_freq_threshold, min_coverage_threshold,
                         snp_candidate_freq_threshold, indel_candidate_freq_threshold, candidate_support_threshold,
                         skip_indels, candidate_region_start, candidate_region_end, candidate_window_size,
                         feature_size, train_mode) -> List[CandidateImageSummary]:
        if train_mode:
            raise NotImplementedError("train_mode labels are outside the inference hot path (SURVEY 2, #21)")
        params = Params(min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold,
                        delete_freq_threshold, min_coverage_threshold, snp_candidate_freq_threshold,
                        indel_candidate_freq_threshold, candidate_support_threshold, bool(skip_indels),
                        int(candidate_window_size), int(feature_size))
        region = Region(self.ref_start, self.ref_end, self.reference_sequence, [read_from_type_read(r) for r in reads],
                        int(candidate_region_start), int(candidate_region_end), self.contig)
        out = (self.ctx or default_context()).summarize(pack_regions([region]), params)
        return [CandidateImageSummary(self.contig, int(out.position[i]), int(out.depth[i]), [out.candidates[i]],
                                      [int(out.cand_freq[i])], out.images[i]) for i in range(len(out))]


class RegionalSummaryGeneratorHP(RegionalSummaryGenerator):
    """Mirror of PEPPER_VARIANT.RegionalSummaryGeneratorHP (pybind_api.h:64-71; call site AlignmentSummarizerHP.py:215-233):
    the haplotag-aware builder, 48 planes x 21 rows, reads carry type_read::hp_tag. Same constructor and methods;
    candidate_window_size must be 20 and feature_size 48 (ImageSizeOptionsHP, Options.py:17-22)."""

    def generate_summary(self, reads: Sequence, min_snp_baseq, min_indel_baseq, snp_freq_threshold,
                         insert_freq_threshold, delete_freq_threshold, min_coverage_threshold,
                         snp_candidate_freq_threshold, indel_candidate_freq_threshold, candidate_support_threshold,
                         skip_indels, candidate_region_start, candidate_region_end, candidate_window_size,
                         feature_size, train_mode) -> List[CandidateImageSummary]:
        if train_mode:
            raise NotImplementedError("train_mode labels are outside the inference hot path (SURVEY 2, #21)")
        params = Params(min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold,
                        delete_freq_threshold, min_coverage_threshold, snp_candidate_freq_threshold,
                        indel_candidate_freq_threshold, candidate_support_threshold, bool(skip_indels),
                        int(candidate_window_size), int(feature_size))
        region = Region(self.ref_start, self.ref_end, self.reference_sequence, [read_from_type_read(r) for r in reads],
                        int(candidate_region_start), int(candidate_region_end), self.contig)
        out = (self.ctx or default_context()).summarize_hp(pack_regions([region]), params)
        return [CandidateImageSummary(self.contig, int(out.position[i]), int(out.depth[i]), [out.candidates[i]],
                                      [int(out.cand_freq[i])], out.images[i]) for i in range(len(out))]
