/*
 * ref_driver_hp.cpp — TEST INFRASTRUCTURE. Builds the REFERENCE's own haplotag-aware image builder
 * (region_summary_hp.cpp), from the sources where they lie under /root/reference (nothing is copied into this
 * repository), into oracle/_ref/libref_region_summary_hp.so behind the flat C interface of
 * pv_summarize_regions_hp, so that reference, oracle and product can be compared on identical inputs.
 *
 * Only compiled where /root/reference exists (this container). Like region_summary.cpp the translation unit needs
 * cigar.h, read.h and the three AlleleType constants of candidate_finder.h:23-27 (data, declared here because that
 * header pulls in htslib). It is a separate library because the two builders define clashing free functions.
 */
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "/root/reference/pepper_variant/modules/cpp/cigar.h"
#include "/root/reference/pepper_variant/modules/cpp/read.h"
namespace AlleleType {
static constexpr int SNP_ALLELE = 1;
static constexpr int INSERT_ALLELE = 2;
static constexpr int DELETE_ALLELE = 3;
}  // namespace AlleleType
#include "/root/reference/pepper_variant/modules/cpp/region_summary_hp.cpp"

#include "../include/pepper_hip.h"

extern "C" int ref_summarize_regions_hp(const pv_batch_in* in, const int32_t* read_hp, const pv_params* p, pv_batch_out* out) {
    out->n_out = 0;
    out->str_bytes = 0;
    if (out->capacity > 0) out->cand_off[0] = 0;
    for (int g = 0; g < in->n_regions; g++) {
        const int64_t ref_len = in->ref_off[g + 1] - in->ref_off[g];
        std::string ref((const char*)in->ref + in->ref_off[g], (size_t)ref_len);
        std::vector<type_read> reads;
        for (int64_t r = in->read_off[g]; r < in->read_off[g + 1]; r++) {
            type_read rd;
            rd.pos = in->read_pos[r];
            rd.pos_end = rd.pos;
            rd.flags.is_reverse = (in->read_flags[r] & 1) != 0;
            rd.mapping_quality = in->read_mapq[r];
            rd.read_id = (int)(r - in->read_off[g]);
            rd.hp_tag = read_hp ? read_hp[r] : 0;
            const int64_t b0 = in->base_off[r], b1 = in->base_off[r + 1];
            rd.sequence.assign((const char*)in->bases + b0, (size_t)(b1 - b0));
            rd.base_qualities.assign(in->quals + b0, in->quals + b1);
            for (int64_t c = in->cigar_off[r]; c < in->cigar_off[r + 1]; c++)
                rd.cigar_tuples.push_back(CigarOp((int)(in->cigar[c] & 0xF), (int)(in->cigar[c] >> 4)));
            reads.push_back(rd);
        }
        RegionalSummaryGeneratorHP gen("contig", in->ref_start[g], in->ref_end[g], ref);
        gen.generate_max_insert_summary(reads);
        std::vector<CandidateImageSummaryHP> res = gen.generate_summary(
            reads, p->min_snp_baseq, p->min_indel_baseq, p->snp_freq_threshold, p->insert_freq_threshold,
            p->delete_freq_threshold, p->min_coverage_threshold, p->snp_candidate_freq_threshold,
            p->indel_candidate_freq_threshold, p->candidate_support_threshold, p->skip_indels != 0,
            in->cand_start[g], in->cand_end[g], p->candidate_window_size, p->feature_size, false);
        for (size_t i = 0; i < res.size(); i++) {
            const CandidateImageSummaryHP& s = res[i];
            const std::string& key = s.candidates.empty() ? std::string() : s.candidates[0];
            const int64_t k = out->n_out, so = out->str_bytes;
            out->n_out += 1;
            out->str_bytes += (int64_t)key.size();
            if (k >= out->capacity || so + (int64_t)key.size() > out->str_capacity) continue;
            out->region[k] = g;
            out->position[k] = s.position;
            out->depth[k] = (uint8_t)s.depth;
            out->cand_freq[k] = s.candidate_frequency.empty() ? 0 : (uint8_t)s.candidate_frequency[0];
            for (int r = 0; r < PV_HP_WINDOW_ROWS; r++)
                for (int j = 0; j < PV_HP_FEATURES; j++) {
                    const int v = s.image_matrix[r][j];
                    out->images[k * PV_HP_WINDOW_BYTES + r * PV_HP_FEATURES + j] = (int8_t)(uint8_t)(v & 0xFF);
                    if (out->images_i32) out->images_i32[k * PV_HP_WINDOW_BYTES + r * PV_HP_FEATURES + j] = v;
                }
            out->cand_off[k] = so;
            memcpy(out->cand_str + so, key.data(), key.size());
            out->cand_off[k + 1] = so + (int64_t)key.size();
        }
    }
    if (out->n_out > out->capacity || out->str_bytes > out->str_capacity) return PV_ERR_CAPACITY;
    return PV_OK;
}
