/* TEST INFRASTRUCTURE ONLY - never linked into or called from the product path.
 *
 * CPU restatement of the polisher's (P2) summary-image builder:
 *   SummaryGenerator::iterate_over_read   pepper/modules/src/pileup_summary/summary_generator.cpp:47-121
 *   SummaryGenerator::generate_image      ... :274-304
 *   SummaryGenerator::generate_summary    ... :371-392
 *   AlignmentSummarizer.chunk_images      pepper/modules/python/AlignmentSummarizer.py:19-56
 *
 * PARITY UNPINNED: summary_generator.cpp cannot be compiled in this container (summary_generator.h:11 includes
 * dataio/bam_handler.h, which includes htslib's sam.h/hts.h/cram.h/hts_endian.h; htslib is fetched from a URL by the
 * reference's cmake and is absent), and the reference holds no test or fixture for it. This file follows the source
 * statement by statement; the std::map containers become dense arrays indexed by position - ref_start.
 *
 * One deliberate reading: `uint8_t pixel_value = <double>` (:281, :293) is undefined for values > 255, which happen
 * when deletions cover a column that no aligned base covers (count / max(1, 0) * 254). The x86-64 build converts with
 * cvttsd2si and keeps the low byte; that is what is restated here and in the HIP kernel.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pepper_hip.h"

static int p2_upper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

/* get_feature_index, :16-33 */
static int p2_feature(int base, int is_reverse) {
    base = p2_upper(base);
    if (is_reverse) {
        if (base == 'A') return 0;
        if (base == 'C') return 1;
        if (base == 'G') return 2;
        if (base == 'T') return 3;
        return 8;
    }
    if (base == 'A') return 4;
    if (base == 'C') return 5;
    if (base == 'G') return 6;
    if (base == 'T') return 7;
    return 9;
}

static uint8_t p2_pixel(double count, double coverage) { /* :281 */
    const double v = (count / (coverage > 1.0 ? coverage : 1.0)) * 254.0;
    return (uint8_t)(uint32_t)(int32_t)v;
}

typedef struct {
    int64_t ref_start, ref_end, R;
    double* base_summaries; /* [R][10] */
    double* coverage;       /* [R] */
    int64_t* longest;       /* [R] longest_insert_count */
    int64_t* ins_off;       /* [R+1] */
    double* insert_summaries; /* [ins rows][10] */
} p2_state;

/* iterate_over_read; pass 0 fills everything except insert_summaries, pass 1 fills insert_summaries (the reference
 * does both at once in maps; the sums do not depend on the order) */
static int p2_iterate(p2_state* s, int pass, int64_t pos, int is_reverse, const uint8_t* seq, int64_t seq_len,
                      const uint32_t* cigar, int64_t n_cigar) {
    int64_t read_index = 0;
    int64_t ref_position = pos;
    for (int64_t ci = 0; ci < n_cigar; ci++) {
        const int op = (int)(cigar[ci] & 0xF);
        const int64_t length = (int64_t)(cigar[ci] >> 4);
        if (ref_position > s->ref_end) break; /* :55 (region_end == ref_end in generate_summary's only caller) */
        switch (op) {
            case PV_CIGAR_EQUAL:
            case PV_CIGAR_DIFF:
            case PV_CIGAR_MATCH: {
                int64_t cigar_index = 0;
                if (ref_position < s->ref_start) { /* :61-65 */
                    cigar_index = s->ref_start - ref_position < length ? s->ref_start - ref_position : length;
                    read_index += cigar_index;
                    ref_position += cigar_index;
                }
                for (int64_t i = cigar_index; i < length; i++) {
                    if (ref_position >= s->ref_start && ref_position <= s->ref_end) {
                        if (read_index >= seq_len) return PV_ERR_INVALID;
                        if (pass == 0) {
                            const int f = p2_feature(seq[read_index], is_reverse);
                            s->base_summaries[(ref_position - s->ref_start) * 10 + f] += 1.0;
                            s->coverage[ref_position - s->ref_start] += 1.0;
                        }
                    }
                    read_index += 1;
                    ref_position += 1;
                }
                break;
            }
            case PV_CIGAR_IN:
                if (ref_position - 1 >= s->ref_start && ref_position - 1 <= s->ref_end) { /* :85-86 */
                    const int64_t a = ref_position - 1 - s->ref_start;
                    if (read_index + length > seq_len) return PV_ERR_INVALID; /* alt[i] past the substr */
                    if (pass == 0) {
                        if (length > s->longest[a]) s->longest[a] = length;
                    } else {
                        for (int64_t i = 0; i < length; i++)
                            s->insert_summaries[(s->ins_off[a] + i) * 10 + p2_feature(seq[read_index + i], is_reverse)] += 1.0;
                    }
                }
                read_index += length;
                break;
            case PV_CIGAR_REF_SKIP:
            case PV_CIGAR_PAD:
            case PV_CIGAR_DEL:
                for (int64_t i = 0; i < length; i++) {
                    if (ref_position + i >= s->ref_start && ref_position + i <= s->ref_end) {
                        if (pass == 0) {
                            s->base_summaries[(ref_position + i - s->ref_start) * 10 + p2_feature('*', is_reverse)] += 1.0;
                            /* :110 - keyed by ref_position, not ref_position + i */
                            if (ref_position >= s->ref_start && ref_position <= s->ref_end)
                                s->coverage[ref_position - s->ref_start] += 1.0;
                        }
                    }
                }
                ref_position += length;
                break;
            case PV_CIGAR_SOFT_CLIP:
                read_index += length;
                break;
            default: /* HARD_CLIP and anything else: nothing (:118-120) */
                break;
        }
    }
    return PV_OK;
}

/* Same contract as pv_polish_summarize_regions (host buffers). */
int oracle_polish_summarize_regions(const pv_batch_in* in, int seq_length, int seq_overlap, pv_polish_out* out) {
    int64_t rows_total = 0, chunks_total = 0;
    int rc = PV_OK;
    if (seq_length < 1 || seq_overlap < 0 || seq_overlap >= seq_length) return PV_ERR_INVALID;
    for (int g = 0; g < in->n_regions; g++) {
        p2_state s;
        s.ref_start = in->ref_start[g];
        s.ref_end = in->ref_end[g];
        s.R = s.ref_end - s.ref_start + 1;
        if (s.R < 1) return PV_ERR_INVALID;
        s.base_summaries = (double*)calloc((size_t)s.R * 10, sizeof(double));
        s.coverage = (double*)calloc((size_t)s.R, sizeof(double));
        s.longest = (int64_t*)calloc((size_t)s.R, sizeof(int64_t));
        s.ins_off = (int64_t*)calloc((size_t)s.R + 1, sizeof(int64_t));
        s.insert_summaries = NULL;
        for (int pass = 0; pass < 2 && rc == PV_OK; pass++) {
            if (pass == 1) {
                for (int64_t i = 0; i < s.R; i++) s.ins_off[i + 1] = s.ins_off[i] + s.longest[i];
                s.insert_summaries = (double*)calloc((size_t)(s.ins_off[s.R] + 1) * 10, sizeof(double));
            }
            for (int64_t r = in->read_off[g]; r < in->read_off[g + 1] && rc == PV_OK; r++) {
                if (in->read_mapq[r] == 0) continue; /* :378 */
                rc = p2_iterate(&s, pass, in->read_pos[r], in->read_flags[r] & 1, in->bases + in->base_off[r],
                                in->base_off[r + 1] - in->base_off[r], in->cigar + in->cigar_off[r],
                                in->cigar_off[r + 1] - in->cigar_off[r]);
            }
        }
        if (rc == PV_OK) {
            /* genomic_pos (:383-390) and generate_image (:274-304), written straight into the flat arrays */
            const int64_t n = s.R + s.ins_off[s.R];
            const int64_t row0 = rows_total;
            if (out->region_row_off) out->region_row_off[g] = row0;
            uint8_t* img = (uint8_t*)malloc((size_t)n * 10);
            int64_t* gp = (int64_t*)malloc((size_t)n * sizeof(int64_t));
            int32_t* gi = (int32_t*)malloc((size_t)n * sizeof(int32_t));
            int64_t row = 0;
            for (int64_t i = 0; i < s.R; i++) {
                for (int j = 0; j <= 9; j++) img[row * 10 + j] = p2_pixel(s.base_summaries[i * 10 + j], s.coverage[i]);
                gp[row] = s.ref_start + i;
                gi[row] = 0;
                row++;
                for (int64_t ii = 0; ii < s.longest[i]; ii++) {
                    for (int j = 0; j <= 9; j++)
                        img[row * 10 + j] = p2_pixel(s.insert_summaries[(s.ins_off[i] + ii) * 10 + j], s.coverage[i]);
                    gp[row] = s.ref_start + i;
                    gi[row] = (int32_t)(ii + 1);
                    row++;
                }
            }
            if (out->flat_images) {
                for (int64_t k = 0; k < n; k++) {
                    if (row0 + k >= out->row_capacity) break;
                    memcpy(out->flat_images + (row0 + k) * 10, img + k * 10, 10);
                    out->flat_position[row0 + k] = gp[k];
                    out->flat_index[row0 + k] = gi[k];
                }
            }
            rows_total += n;
            /* chunk_images (AlignmentSummarizer.py:19-56) */
            int64_t chunk_start = 0, chunk_id = 0;
            int64_t chunk_end = n < seq_length ? n : seq_length;
            for (;;) {
                if (out->images && chunks_total < out->chunk_capacity) {
                    const int64_t k = chunks_total;
                    uint8_t* dst = out->images + k * seq_length * 10;
                    for (int64_t j = 0; j < seq_length; j++) {
                        if (chunk_start + j < chunk_end) {
                            memcpy(dst + j * 10, img + (chunk_start + j) * 10, 10);
                            out->position[k * seq_length + j] = gp[chunk_start + j];
                            out->index[k * seq_length + j] = gi[chunk_start + j];
                        } else { /* padding: zero rows, positions (-1, -1) (:36-40) */
                            memset(dst + j * 10, 0, 10);
                            out->position[k * seq_length + j] = -1;
                            out->index[k * seq_length + j] = -1;
                        }
                    }
                    out->region[k] = g;
                    out->chunk_id[k] = (int32_t)chunk_id;
                }
                chunks_total++;
                chunk_id++;
                if (chunk_end == n) break;
                chunk_start = chunk_end - seq_overlap;
                chunk_end = n < chunk_start + seq_length ? n : chunk_start + seq_length;
            }
            free(img); free(gp); free(gi);
        }
        free(s.base_summaries); free(s.coverage); free(s.longest); free(s.ins_off); free(s.insert_summaries);
        if (rc != PV_OK) return rc;
    }
    if (out->region_row_off) out->region_row_off[in->n_regions] = rows_total;
    out->n_chunks = chunks_total;
    out->n_rows = rows_total;
    if (chunks_total > out->chunk_capacity || (out->flat_images && rows_total > out->row_capacity)) return PV_ERR_CAPACITY;
    return PV_OK;
}
