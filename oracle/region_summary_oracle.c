/*
 * region_summary_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's pileup summary-image builder
 *   RegionalSummaryGenerator::generate_summary      /root/reference/pepper_variant/modules/cpp/region_summary.cpp:568-916
 *   RegionalSummaryGenerator::populate_summary_matrix                                   region_summary.cpp:337-566
 * followed by the int8 cast of DataStore.write_summary (pepper_variant/modules/python/DataStore.py:68).
 *
 * It works on the flat SoA batch layout of include/pepper_hip.h and is used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / CPU baseline. The product
 * path (pepper_thesis_amd/csrc) never calls into this file.
 *
 * Parity pinning: this restatement is checked bit-for-bit against the reference's own
 * region_summary.cpp compiled by oracle/Makefile into oracle/_ref/ (tests/test_oracle_vs_reference.py,
 * runs wherever /root/reference exists) and against the committed fixtures under tests/golden/
 * that were produced by that reference build (tests/golden/make_summary_golden.py).
 *
 * Structure (deliberately unlike the reference, which keeps std::map<string,int> per position):
 *   pass 1  walk every read's CIGAR once; bump dense per-position int counters and append one
 *           fixed-size "allele event" per SNP / insert / delete observation;
 *   pass 2  sort the events by (position, key) where key order == std::string order of
 *           "<type digit><allele bytes>" and run-length them into per-position allele tables;
 *   pass 3  per-position threshold scan + clamp (region_summary.cpp:634-654);
 *   pass 4  per site, per allele: filters, 33x26 window copy, overlays (region_summary.cpp:669-912).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pepper_hip.h"

typedef struct {
    int32_t pos;         /* region index of the anchor position */
    uint8_t type;        /* 1 SNP, 2 INS, 3 DEL (AlleleType, candidate_finder.h:23-27) */
    uint8_t rev;         /* read strand */
    int32_t len;         /* allele bytes (without the type digit) */
    const uint8_t* bytes;
} ora_event;

typedef struct {
    ora_event* v;
    int64_t n, cap;
} ora_events;

static int push_event(ora_events* e, int32_t pos, uint8_t type, uint8_t rev, int32_t len, const uint8_t* bytes) {
    if (e->n == e->cap) {
        int64_t nc = e->cap ? e->cap * 2 : 4096;
        ora_event* nv = (ora_event*)realloc(e->v, (size_t)nc * sizeof(ora_event));
        if (!nv) return -1;
        e->v = nv;
        e->cap = nc;
    }
    ora_event* x = &e->v[e->n++];
    x->pos = pos;
    x->type = type;
    x->rev = rev;
    x->len = len;
    x->bytes = bytes;
    return 0;
}

/* std::string operator< on "<digit><bytes>": type first, then unsigned bytes, then length */
static int key_cmp(const ora_event* a, const ora_event* b) {
    if (a->type != b->type) return a->type < b->type ? -1 : 1;
    int32_t m = a->len < b->len ? a->len : b->len;
    int c = m > 0 ? memcmp(a->bytes, b->bytes, (size_t)m) : 0;
    if (c) return c < 0 ? -1 : 1;
    if (a->len != b->len) return a->len < b->len ? -1 : 1;
    return 0;
}

static int event_cmp(const void* pa, const void* pb) {
    const ora_event* a = (const ora_event*)pa;
    const ora_event* b = (const ora_event*)pb;
    if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
    return key_cmp(a, b);
}

static int upper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

/* get_reference_feature_value, region_summary.cpp:165-172 */
static int ref_feature_value(int base) {
    base = upper(base);
    if (base == 'A') return 1;
    if (base == 'C') return 2;
    if (base == 'G') return 3;
    if (base == 'T') return 4;
    return 5;
}

/* get_feature_index, region_summary.cpp:201-230 */
static int feature_index(int ref_base, int base, int is_reverse) {
    base = upper(base);
    ref_base = upper(ref_base);
    if (!(ref_base == 'A' || ref_base == 'C' || ref_base == 'G' || ref_base == 'T')) return -1;
    int start = is_reverse ? 18 : 7;
    if (base == 'A') return start + 1;
    if (base == 'C') return start + 2;
    if (base == 'G') return start + 3;
    if (base == 'T') return start + 4;
    if (base == 'I') return start + 5;
    if (base == 'D') return start + 6;
    return start + 7;
}

static int imin(int a, int b) { return a < b ? a : b; }

typedef struct {
    int64_t R;        /* region columns */
    int32_t* image;   /* [(R+1)][26] */
    int32_t* cov;     /* [R] */
    int32_t* snp;     /* [R] */
    int32_t* ins;     /* [R] */
    int32_t* del;     /* [R] */
} ora_counters;

/* pass 1 for one read — populate_summary_matrix, region_summary.cpp:337-566 */
static int walk_read(const pv_batch_in* in, int64_t read, int64_t ref_start, int64_t ref_end,
                     const uint8_t* ref, int64_t ref_len, const pv_params* p, ora_counters* c, ora_events* ev) {
    const uint8_t* seq = in->bases + in->base_off[read];
    const uint8_t* qual = in->quals + in->base_off[read];
    const int64_t seq_len = in->base_off[read + 1] - in->base_off[read];
    const uint32_t* cig = in->cigar + in->cigar_off[read];
    const int64_t n_cig = in->cigar_off[read + 1] - in->cigar_off[read];
    const int rev = in->read_flags[read] & 1;
    int64_t read_index = 0;
    int64_t ref_position = in->read_pos[read];

    for (int64_t ci = 0; ci < n_cig; ci++) {
        const int op = (int)(cig[ci] & 0xF);
        const int64_t len = (int64_t)(cig[ci] >> 4);
        if (ref_position > ref_end) break; /* :355 */
        switch (op) {
            case PV_CIGAR_EQUAL:
            case PV_CIGAR_DIFF:
            case PV_CIGAR_MATCH: {
                int64_t i0 = 0;
                if (ref_position < ref_start) { /* :361-365 */
                    i0 = ref_start - ref_position;
                    if (i0 > len) i0 = len;
                    read_index += i0;
                    ref_position += i0;
                }
                for (int64_t i = i0; i < len; i++) {
                    if (ref_position >= ref_start && ref_position <= ref_end) {
                        if (read_index >= seq_len) return PV_ERR_INVALID;
                        const int64_t ri = ref_position - ref_start;
                        const int base = seq[read_index];
                        const int ref_base = ref[ri];
                        const double bq = (double)qual[read_index];
                        const int fi = feature_index(ref_base, base, rev);
                        const int qok = bq >= p->min_snp_baseq;
                        if (qok) { /* :378-392 */
                            c->cov[ri] += 1;
                            int anchor = 0;
                            if (i == len - 1 && ci != n_cig - 1) {
                                const int nop = (int)(cig[ci + 1] & 0xF);
                                if (nop == PV_CIGAR_IN || nop == PV_CIGAR_DEL) anchor = 1;
                            }
                            if (!anchor) c->image[ri * PV_FEATURES + (rev ? 15 : 4)] -= 1;
                        }
                        if (qok && fi >= 0) c->image[ri * PV_FEATURES + fi] -= 1; /* :396,423 */
                        if (ref_base != base && qok) {                            /* :394-421 */
                            c->snp[ri] += 1;
                            if (push_event(ev, (int32_t)ri, 1, (uint8_t)rev, 1, seq + read_index)) return PV_ERR_INVALID;
                        }
                    }
                    read_index += 1;
                    ref_position += 1;
                }
                break;
            }
            case PV_CIGAR_IN: { /* :431-490 */
                const int64_t anchor = ref_position - 1;
                if (anchor >= ref_start && anchor <= ref_end && read_index - 1 >= 0) {
                    const int64_t ri = anchor - ref_start;
                    const int ref_base = ref[ri];
                    const int fi = feature_index(ref_base, 'I', rev);
                    const int64_t start = read_index - 1;
                    const int64_t L = len + 1;
                    if (start + L > seq_len) return PV_ERR_INVALID;
                    double bq = 0;
                    for (int64_t i = start; i < start + L; i++) bq += (double)qual[i];
                    const int qok = bq >= p->min_indel_baseq * (double)L;
                    if (qok && (double)qual[start] < p->min_snp_baseq) c->cov[ri] += 1; /* :453-454 */
                    if (1 + L <= PV_MAX_ALLELE_KEY && qok) {
                        if (fi >= 0) c->image[ri * PV_FEATURES + fi] -= 1;
                        c->ins[ri] += 1;
                        if (push_event(ev, (int32_t)ri, 2, (uint8_t)rev, (int32_t)L, seq + start)) return PV_ERR_INVALID;
                    }
                }
                read_index += len;
                break;
            }
            case PV_CIGAR_DEL: { /* :491-555 */
                const int64_t anchor = ref_position - 1;
                if (anchor >= ref_start && anchor <= ref_end) {
                    const int64_t ri = anchor - ref_start;
                    const int ref_base = ref[ri];
                    const int fi = feature_index(ref_base, 'D', rev);
                    if (fi >= 0) c->image[ri * PV_FEATURES + fi] -= 1; /* unconditional, :496-497 */
                    int64_t L = len + 1; /* reference_sequence.substr(anchor, len+1) truncates */
                    if (ri + L > ref_len) L = ref_len - ri;
                    if (1 + L <= PV_MAX_ALLELE_KEY) {
                        c->del[ri] += 1;
                        if (push_event(ev, (int32_t)ri, 3, (uint8_t)rev, (int32_t)L, ref + ri)) return PV_ERR_INVALID;
                    }
                }
                for (int64_t i = 0; i < len; i++) { /* :542-552 */
                    const int64_t pos = ref_position + i;
                    if (pos >= ref_start && pos <= ref_end) {
                        const int64_t ri = pos - ref_start;
                        const int fi = feature_index(ref[ri], '*', rev);
                        if (fi >= 0) c->image[ri * PV_FEATURES + fi] -= 1;
                    }
                }
                ref_position += len;
                break;
            }
            case PV_CIGAR_REF_SKIP:
            case PV_CIGAR_PAD:
                ref_position += len; /* falls through, :556-561 */
                read_index += len;
                break;
            case PV_CIGAR_SOFT_CLIP:
                read_index += len;
                break;
            default: /* HARD_CLIP, BACK, unknown: no state change */
                break;
        }
    }
    return PV_OK;
}

/* one region; appends to out starting at out->n_out / out->str_bytes. Counts even past capacity. */
static int summarize_one(const pv_batch_in* in, int g, const pv_params* p, pv_batch_out* out) {
    const int64_t ref_start = in->ref_start[g], ref_end = in->ref_end[g];
    const int64_t R = ref_end - ref_start + 1;
    const uint8_t* ref = in->ref + in->ref_off[g];
    const int64_t ref_len = in->ref_off[g + 1] - in->ref_off[g];
    if (R <= 0 || ref_len < R) return PV_ERR_INVALID;
    const int W = p->candidate_window_size, F = p->feature_size;
    if (W != 32 || F != PV_FEATURES) return PV_ERR_INVALID;

    ora_counters c;
    c.R = R;
    c.image = (int32_t*)calloc((size_t)(R + 1) * PV_FEATURES, sizeof(int32_t));
    c.cov = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    c.snp = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    c.ins = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    c.del = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    uint8_t* pass = (uint8_t*)calloc((size_t)R, 1); /* bit0 site, bit1 snp, bit2 ins, bit3 del */
    int64_t* ev_begin = (int64_t*)calloc((size_t)R + 1, sizeof(int64_t));
    ora_events ev = {0, 0, 0};
    int rc = PV_OK;
    if (!c.image || !c.cov || !c.snp || !c.ins || !c.del || !pass || !ev_begin) { rc = PV_ERR_INVALID; goto done; }

    /* encode_reference_bases, :174-191 (GENERATE_INDELS == false: base_index == region index) */
    for (int64_t i = 0; i < R; i++) c.image[i * PV_FEATURES + 0] = ref_feature_value(ref[i]);

    /* pass 1 */
    for (int64_t r = in->read_off[g]; r < in->read_off[g + 1]; r++) {
        if (in->read_mapq[r] == 0) continue; /* :619 */
        if (in->base_off[r + 1] - in->base_off[r] <= 0) { rc = PV_ERR_INVALID; goto done; } /* :352 would be UB */
        rc = walk_read(in, r, ref_start, ref_end, ref, ref_len, p, &c, &ev);
        if (rc) goto done;
    }
    /* pass 2 */
    if (ev.n) qsort(ev.v, (size_t)ev.n, sizeof(ora_event), event_cmp);
    {
        int64_t e = 0;
        for (int64_t i = 0; i <= R; i++) {
            while (e < ev.n && ev.v[e].pos < i) e++;
            ev_begin[i] = e;
        }
    }

    /* pass 3 — :634-654 */
    for (int64_t i = 0; i < R; i++) {
        const double cv = (double)c.cov[i] > 1.0 ? (double)c.cov[i] : 1.0;
        const double fs = c.snp[i] / cv, fi = c.ins[i] / cv, fd = c.del[i] / cv;
        if (fs >= p->snp_freq_threshold || fi >= p->insert_freq_threshold || fd >= p->delete_freq_threshold) {
            const int64_t pos = ref_start + i;
            if (pos >= in->cand_start[g] && pos <= in->cand_end[g] && (double)c.cov[i] >= p->min_coverage_threshold) {
                pass[i] = 1;
                if (fs >= p->snp_freq_threshold) pass[i] |= 2;
                if (fi >= p->insert_freq_threshold) pass[i] |= 4;
                if (fd >= p->delete_freq_threshold) pass[i] |= 8;
            }
        }
        for (int j = 11; j < 25; j++) { /* BASE_INDEX_START=11, BASE_INDEX_SIZE=14 */
            int32_t* v = &c.image[i * PV_FEATURES + j];
            if (*v > PV_MAX_COLOR) *v = PV_MAX_COLOR;
            if (*v < -PV_MAX_COLOR) *v = -PV_MAX_COLOR;
        }
    }

    /* pass 4 — :669-912 */
    for (int64_t i = 0; i < R; i++) {
        if (!(pass[i] & 1)) continue;
        const int depth = imin(c.cov[i], PV_MAX_COLOR);
        const int ref_base = ref[i];
        int64_t e = ev_begin[i];
        const int64_t e_end = ev_begin[i + 1];
        while (e < e_end) {
            /* one allele = one run of equal keys */
            int64_t e2 = e;
            int total = 0, fwd = 0, rvs = 0;
            while (e2 < e_end && key_cmp(&ev.v[e], &ev.v[e2]) == 0) {
                total++;
                if (ev.v[e2].rev) rvs++; else fwd++;
                e2++;
            }
            const ora_event* a = &ev.v[e];
            e = e2;
            const double freq = (double)total / ((double)depth > 1.0 ? (double)depth : 1.0);
            if ((double)total < p->candidate_support_threshold) continue;                  /* :693 */
            if (a->type != 1 && freq < p->indel_candidate_freq_threshold) continue;         /* :697 */
            if (a->type == 1 && freq < p->snp_candidate_freq_threshold) continue;           /* :700 */
            if (a->type != 1 && p->skip_indels) continue;                                   /* :704 */
            if ((a->type == 1 && !(pass[i] & 2)) || (a->type == 2 && !(pass[i] & 4)) ||
                (a->type == 3 && !(pass[i] & 8))) continue;                                 /* :708-712 */

            const int64_t k = out->n_out;
            const int64_t so = out->str_bytes;
            out->n_out += 1;
            out->str_bytes += 1 + a->len;
            if (k >= out->capacity || so + 1 + a->len > out->str_capacity) continue; /* count only */

            int32_t win[PV_WINDOW_ROWS][PV_FEATURES];
            const int64_t left = i - W / 2;
            for (int r = 0; r <= W; r++) { /* :828-841; row R exists and is zero */
                const int64_t src = left + r;
                for (int j = 0; j < PV_FEATURES; j++)
                    win[r][j] = (src < 0 || src > R) ? 0 : c.image[src * PV_FEATURES + j];
            }
            const int mid = W / 2;
            const int cf = imin(total, PV_MAX_COLOR);
            if (a->type == 1) { /* :848-862 */
                const int ff = feature_index(ref_base, a->bytes[0], 0);
                const int fr = feature_index(ref_base, a->bytes[0], 1);
                win[mid][1] = ref_feature_value(a->bytes[0]);
                win[mid][5] = imin(fwd, PV_MAX_COLOR);
                win[mid][16] = imin(rvs, PV_MAX_COLOR);
                if (ff >= 0) win[mid][ff] = -win[mid][ff]; /* ff < 0 is UB in the reference: fenced off */
                if (fr >= 0) win[mid][fr] = -win[mid][fr];
            } else if (a->type == 2) { /* :863-877 */
                const int ff = feature_index(ref_base, 'I', 0);
                const int fr = feature_index(ref_base, 'I', 1);
                win[mid][2] = imin(a->len, PV_MAX_COLOR);
                win[mid][6] = imin(fwd, PV_MAX_COLOR);
                win[mid][17] = imin(rvs, PV_MAX_COLOR);
                if (ff >= 0) win[mid][ff] = -win[mid][ff];
                if (fr >= 0) win[mid][fr] = -win[mid][fr];
            } else { /* :878-905 */
                const int del_len = a->len;
                const int end_index = imin(mid + del_len - 1, W - 1);
                int ff = feature_index(ref_base, 'D', 0);
                int fr = feature_index(ref_base, 'D', 1);
                win[mid][3] = imin(del_len, PV_MAX_COLOR);
                win[mid][7] = imin(fwd, PV_MAX_COLOR);
                win[mid][18] = imin(rvs, PV_MAX_COLOR);
                if (ff >= 0) win[mid][ff] = -win[mid][ff];
                if (fr >= 0) win[mid][fr] = -win[mid][fr];
                ff = feature_index(ref_base, '*', 0);
                fr = feature_index(ref_base, '*', 1);
                for (int idx = mid + 1; idx <= end_index; idx++) {
                    win[idx][3] = imin(del_len, PV_MAX_COLOR);
                    win[idx][7] = imin(fwd, PV_MAX_COLOR);
                    win[idx][18] = imin(rvs, PV_MAX_COLOR);
                    if (ff >= 0) win[idx][ff] = -win[idx][ff];
                    if (fr >= 0) win[idx][fr] = -win[idx][fr];
                }
            }
            out->region[k] = g;
            out->position[k] = ref_start + i;
            out->depth[k] = (uint8_t)depth;
            out->cand_freq[k] = (uint8_t)cf;
            for (int r = 0; r < PV_WINDOW_ROWS; r++)
                for (int j = 0; j < PV_FEATURES; j++) {
                    out->images[k * PV_WINDOW_BYTES + r * PV_FEATURES + j] = (int8_t)(uint8_t)(win[r][j] & 0xFF);
                    if (out->images_i32) out->images_i32[k * PV_WINDOW_BYTES + r * PV_FEATURES + j] = win[r][j];
                }
            out->cand_off[k] = so;
            out->cand_str[so] = (char)('0' + a->type);
            memcpy(out->cand_str + so + 1, a->bytes, (size_t)a->len);
            out->cand_off[k + 1] = so + 1 + a->len;
        }
    }

done:
    free(c.image); free(c.cov); free(c.snp); free(c.ins); free(c.del);
    free(pass); free(ev_begin); free(ev.v);
    return rc;
}

int oracle_summarize_regions(const pv_batch_in* in, const pv_params* params, pv_batch_out* out) {
    if (!in || !params || !out) return PV_ERR_INVALID;
    out->n_out = 0;
    out->str_bytes = 0;
    if (out->capacity > 0) out->cand_off[0] = 0;
    for (int g = 0; g < in->n_regions; g++) {
        int rc = summarize_one(in, g, params, out);
        if (rc) return rc;
    }
    if (out->n_out > out->capacity || out->str_bytes > out->str_capacity) return PV_ERR_CAPACITY;
    return PV_OK;
}

/* Debug tap for kernel bring-up: the dense counters of ONE region after pass 1 (no clamp).
 * image [(R+1)*26], cov/snp/ins/del [R]. */
int oracle_region_counters(const pv_batch_in* in, int g, const pv_params* p, int32_t* image, int32_t* cov,
                           int32_t* snp, int32_t* ins, int32_t* del) {
    const int64_t ref_start = in->ref_start[g], ref_end = in->ref_end[g];
    const int64_t R = ref_end - ref_start + 1;
    const uint8_t* ref = in->ref + in->ref_off[g];
    const int64_t ref_len = in->ref_off[g + 1] - in->ref_off[g];
    ora_counters c = {R, image, cov, snp, ins, del};
    ora_events ev = {0, 0, 0};
    memset(image, 0, (size_t)(R + 1) * PV_FEATURES * sizeof(int32_t));
    memset(cov, 0, (size_t)R * sizeof(int32_t));
    memset(snp, 0, (size_t)R * sizeof(int32_t));
    memset(ins, 0, (size_t)R * sizeof(int32_t));
    memset(del, 0, (size_t)R * sizeof(int32_t));
    for (int64_t i = 0; i < R; i++) image[i * PV_FEATURES] = ref_feature_value(ref[i]);
    int rc = PV_OK;
    for (int64_t r = in->read_off[g]; r < in->read_off[g + 1] && !rc; r++) {
        if (in->read_mapq[r] == 0) continue;
        rc = walk_read(in, r, ref_start, ref_end, ref, ref_len, p, &c, &ev);
    }
    free(ev.v);
    return rc;
}
