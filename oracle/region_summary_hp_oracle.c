/*
 * region_summary_hp_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded CPU restatement of the reference's HAPLOTAG-AWARE pileup summary-image builder
 *   RegionalSummaryGeneratorHP::generate_summary        /root/reference/pepper_variant/modules/cpp/region_summary_hp.cpp:665-1012
 *   RegionalSummaryGeneratorHP::populate_summary_matrix                                      region_summary_hp.cpp:350-663
 *   RegionalSummaryGeneratorHP::get_feature_index                                            region_summary_hp.cpp:191-243
 * (48 planes, 21-row windows: ImageSizeOptionsHP, pepper_variant/modules/python/Options.py:17-22) followed by the int8
 * cast of DataStore.write_summary. Same flat batch layout as region_summary_oracle.c plus one hp_tag per read.
 *
 * Used ONLY by tests/ as the checker. Parity pinning: checked bit-for-bit against the reference's own
 * region_summary_hp.cpp compiled in place by oracle/Makefile (oracle/_ref/libref_region_summary_hp.so; live comparison
 * wherever /root/reference exists) and against tests/golden/summary_hp_golden.npz produced by that build.
 *
 * How this builder differs from the 26-plane one (every line below cites where):
 *   - a read contributes to haplotype set 1 and/or 2: REF-count planes and the per-strand allele maps use
 *     "hp_tag == 0 || hp_tag == k"; the symbol planes use "hp_tag == 0 -> both, hp_tag == 1 -> set 1, ANYTHING ELSE -> set 2";
 *   - matching bases DEcrement their symbol plane, mismatching bases touch no symbol plane, I / D / * INcrement;
 *   - no anchor rule; inserts sum the qualities of the inserted bases only and REMOVE the anchor's coverage when the
 *     insert fails the quality bar; every plane is clamped; windows carry no deletion tail and no sign flips.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pepper_hip.h"

typedef struct {
    int32_t pos;
    uint8_t type;  /* 1 SNP, 2 INS, 3 DEL */
    uint8_t rev;
    uint8_t sets;  /* bit0: counts for haplotype 1 (hp_tag 0 or 1), bit1: haplotype 2 (hp_tag 0 or 2) */
    int32_t len;
    const uint8_t* bytes;
} hp_event;

typedef struct {
    hp_event* v;
    int64_t n, cap;
} hp_events;

static int hp_push(hp_events* e, int32_t pos, int type, int rev, int sets, int32_t len, const uint8_t* bytes) {
    if (e->n == e->cap) {
        int64_t nc = e->cap ? e->cap * 2 : 4096;
        hp_event* nv = (hp_event*)realloc(e->v, (size_t)nc * sizeof(hp_event));
        if (!nv) return -1;
        e->v = nv;
        e->cap = nc;
    }
    hp_event* x = &e->v[e->n++];
    x->pos = pos; x->type = (uint8_t)type; x->rev = (uint8_t)rev; x->sets = (uint8_t)sets; x->len = len; x->bytes = bytes;
    return 0;
}

static int hp_key_cmp(const hp_event* a, const hp_event* b) { /* std::string order of "<digit><bytes>" */
    if (a->type != b->type) return a->type < b->type ? -1 : 1;
    int32_t m = a->len < b->len ? a->len : b->len;
    int c = m > 0 ? memcmp(a->bytes, b->bytes, (size_t)m) : 0;
    if (c) return c < 0 ? -1 : 1;
    if (a->len != b->len) return a->len < b->len ? -1 : 1;
    return 0;
}
static int hp_event_cmp(const void* pa, const void* pb) {
    const hp_event* a = (const hp_event*)pa;
    const hp_event* b = (const hp_event*)pb;
    if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
    return hp_key_cmp(a, b);
}

static int hp_upper(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
static int hp_ref_value(int base) { /* get_reference_feature_value, :155-162 */
    base = hp_upper(base);
    return base == 'A' ? 1 : base == 'C' ? 2 : base == 'G' ? 3 : base == 'T' ? 4 : 5;
}
/* get_feature_index, :191-243: set 1 forward starts at 7, set 1 reverse 18, set 2 forward 29, set 2 reverse 40 */
static int hp_feature_index(int ref_base, int base, int is_reverse, int hp_tag) {
    base = hp_upper(base);
    ref_base = hp_upper(ref_base);
    if (!(ref_base == 'A' || ref_base == 'C' || ref_base == 'G' || ref_base == 'T')) return -1;
    const int start = (hp_tag == 1 ? 7 : 29) + (is_reverse ? 11 : 0);
    if (base == 'A') return start + 1;
    if (base == 'C') return start + 2;
    if (base == 'G') return start + 3;
    if (base == 'T') return start + 4;
    if (base == 'I') return start + 5;
    if (base == 'D') return start + 6;
    return start + 7;
}
/* "if hp_tag == 0: both sets, else the read's own tag" (:454-462, :498-506, :561-569, :637-645) */
static void hp_bump_symbol(int32_t* row, int ref_base, int sym, int rev, int hp_tag, int delta) {
    if (hp_tag == 0) {
        int fi = hp_feature_index(ref_base, sym, rev, 1);
        if (fi >= 0) row[fi] += delta;
        fi = hp_feature_index(ref_base, sym, rev, 2);
        if (fi >= 0) row[fi] += delta;
    } else {
        const int fi = hp_feature_index(ref_base, sym, rev, hp_tag);
        if (fi >= 0) row[fi] += delta;
    }
}
static int hp_imin(int a, int b) { return a < b ? a : b; }

typedef struct {
    int64_t R;
    int32_t* image; /* [(R+1)][48] */
    int32_t *cov, *snp, *ins, *del;
} hp_counters;

static int hp_walk_read(const pv_batch_in* in, int64_t read, int hp_tag, int64_t ref_start, int64_t ref_end,
                        const uint8_t* ref, int64_t ref_len, const pv_params* p, hp_counters* c, hp_events* ev) {
    const uint8_t* seq = in->bases + in->base_off[read];
    const uint8_t* qual = in->quals + in->base_off[read];
    const int64_t seq_len = in->base_off[read + 1] - in->base_off[read];
    const uint32_t* cig = in->cigar + in->cigar_off[read];
    const int64_t n_cig = in->cigar_off[read + 1] - in->cigar_off[read];
    const int rev = in->read_flags[read] & 1;
    const int F = PV_HP_FEATURES;
    const int sets = ((hp_tag == 0 || hp_tag == 1) ? 1 : 0) | ((hp_tag == 0 || hp_tag == 2) ? 2 : 0);
    int64_t read_index = 0;
    int64_t ref_position = in->read_pos[read];

    for (int64_t ci = 0; ci < n_cig; ci++) {
        const int op = (int)(cig[ci] & 0xF);
        const int64_t len = (int64_t)(cig[ci] >> 4);
        if (ref_position > ref_end) break; /* :370 */
        switch (op) {
            case PV_CIGAR_EQUAL:
            case PV_CIGAR_DIFF:
            case PV_CIGAR_MATCH: {
                int64_t i0 = 0;
                if (ref_position < ref_start) { /* :376-380 */
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
                        const int qok = (double)qual[read_index] >= p->min_snp_baseq;
                        int32_t* row = c->image + ri * F;
                        if (qok) { /* :393-403 */
                            c->cov[ri] += 1;
                            if (sets & 1) row[rev ? 15 : 4] -= 1;
                            if (sets & 2) row[rev ? 37 : 26] -= 1;
                        }
                        if (ref_base != base && qok) { /* :406-452: raw bytes */
                            c->snp[ri] += 1;
                            if (hp_push(ev, (int32_t)ri, 1, rev, sets, 1, seq + read_index)) return PV_ERR_INVALID;
                        } else if (qok) { /* :453-463 */
                            hp_bump_symbol(row, ref_base, base, rev, hp_tag, -1);
                        }
                    }
                    read_index += 1;
                    ref_position += 1;
                }
                break;
            }
            case PV_CIGAR_IN: { /* :469-553 */
                const int64_t anchor = ref_position - 1;
                if (anchor >= ref_start && anchor <= ref_end && read_index - 1 >= 0) {
                    const int64_t ri = anchor - ref_start;
                    if (read_index + len > seq_len) return PV_ERR_INVALID; /* the quality loop of :482-484 would read past the read */
                    double bq = 0;
                    for (int64_t i = 0; i < len; i++) bq += (double)qual[read_index + i];
                    const int qok = bq >= p->min_indel_baseq * (double)len;
                    if (!qok && (double)qual[read_index - 1] >= p->min_snp_baseq) c->cov[ri] -= 1; /* :487-488 */
                    const int64_t L = len + 1; /* anchor base + inserted bases, :477 */
                    if (1 + L <= PV_MAX_ALLELE_KEY && qok) { /* :496 */
                        hp_bump_symbol(c->image + ri * F, ref[ri], 'I', rev, hp_tag, +1);
                        c->ins[ri] += 1;
                        if (hp_push(ev, (int32_t)ri, 2, rev, sets, (int32_t)L, seq + read_index - 1)) return PV_ERR_INVALID;
                    }
                }
                read_index += len;
                break;
            }
            case PV_CIGAR_DEL: { /* :556-649 */
                const int64_t anchor = ref_position - 1;
                if (anchor >= ref_start && anchor <= ref_end) {
                    const int64_t ri = anchor - ref_start;
                    hp_bump_symbol(c->image + ri * F, ref[ri], 'D', rev, hp_tag, +1); /* unconditional, :561-569 */
                    int64_t L = len + 1; /* reference_sequence.substr(anchor, len + 1) truncates */
                    if (ri + L > ref_len) L = ref_len - ri;
                    if (1 + L <= PV_MAX_ALLELE_KEY) {
                        c->del[ri] += 1;
                        if (hp_push(ev, (int32_t)ri, 3, rev, sets, (int32_t)L, ref + ri)) return PV_ERR_INVALID;
                    }
                }
                for (int64_t i = 0; i < len; i++) { /* :631-647 */
                    const int64_t pos = ref_position + i;
                    if (pos >= ref_start && pos <= ref_end) {
                        const int64_t ri = pos - ref_start;
                        hp_bump_symbol(c->image + ri * F, ref[ri], '*', rev, hp_tag, +1);
                    }
                }
                ref_position += len;
                break;
            }
            case PV_CIGAR_REF_SKIP:
            case PV_CIGAR_PAD:
                ref_position += len; /* falls through into SOFT_CLIP, :652-657 */
                read_index += len;
                break;
            case PV_CIGAR_SOFT_CLIP:
                read_index += len;
                break;
            default:
                break;
        }
    }
    return PV_OK;
}

static int hp_summarize_one(const pv_batch_in* in, const int32_t* read_hp, int g, const pv_params* p, pv_batch_out* out) {
    const int64_t ref_start = in->ref_start[g], ref_end = in->ref_end[g];
    const int64_t R = ref_end - ref_start + 1;
    const uint8_t* ref = in->ref + in->ref_off[g];
    const int64_t ref_len = in->ref_off[g + 1] - in->ref_off[g];
    if (R <= 0 || ref_len < R) return PV_ERR_INVALID;
    const int W = p->candidate_window_size, F = PV_HP_FEATURES;
    if (W != PV_HP_WINDOW_ROWS - 1 || p->feature_size != F) return PV_ERR_INVALID;

    hp_counters c;
    c.R = R;
    c.image = (int32_t*)calloc((size_t)(R + 1) * F, sizeof(int32_t));
    c.cov = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    c.snp = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    c.ins = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    c.del = (int32_t*)calloc((size_t)R, sizeof(int32_t));
    uint8_t* pass = (uint8_t*)calloc((size_t)R, 1);
    int64_t* ev_begin = (int64_t*)calloc((size_t)R + 1, sizeof(int64_t));
    hp_events ev = {0, 0, 0};
    int rc = PV_OK;
    if (!c.image || !c.cov || !c.snp || !c.ins || !c.del || !pass || !ev_begin) { rc = PV_ERR_INVALID; goto done; }

    for (int64_t i = 0; i < R; i++) c.image[i * F + 0] = hp_ref_value(ref[i]); /* encode_reference_bases, :164-181 */

    for (int64_t r = in->read_off[g]; r < in->read_off[g + 1]; r++) {
        if (in->read_mapq[r] == 0) continue; /* :720 */
        if (in->base_off[r + 1] - in->base_off[r] <= 0) { rc = PV_ERR_INVALID; goto done; }
        rc = hp_walk_read(in, r, read_hp ? read_hp[r] : 0, ref_start, ref_end, ref, ref_len, p, &c, &ev);
        if (rc) goto done;
    }
    if (ev.n) qsort(ev.v, (size_t)ev.n, sizeof(hp_event), hp_event_cmp);
    {
        int64_t e = 0;
        for (int64_t i = 0; i <= R; i++) {
            while (e < ev.n && ev.v[e].pos < i) e++;
            ev_begin[i] = e;
        }
    }

    for (int64_t i = 0; i < R; i++) { /* :748-769 */
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
        for (int j = 0; j < F; j++) { /* every plane, :762-767 */
            int32_t* v = &c.image[i * F + j];
            if (*v > PV_MAX_COLOR) *v = PV_MAX_COLOR;
            if (*v < -PV_MAX_COLOR) *v = -PV_MAX_COLOR;
        }
    }

    for (int64_t i = 0; i < R; i++) { /* :783-1009 */
        if (!(pass[i] & 1)) continue;
        const int depth = hp_imin(c.cov[i], PV_MAX_COLOR);
        int64_t e = ev_begin[i];
        const int64_t e_end = ev_begin[i + 1];
        while (e < e_end) {
            int64_t e2 = e;
            int total = 0, cnt[4] = {0, 0, 0, 0}; /* fwd set 1, fwd set 2, rev set 1, rev set 2 */
            while (e2 < e_end && hp_key_cmp(&ev.v[e], &ev.v[e2]) == 0) {
                total++;
                if (ev.v[e2].sets & 1) cnt[ev.v[e2].rev ? 2 : 0]++;
                if (ev.v[e2].sets & 2) cnt[ev.v[e2].rev ? 3 : 1]++;
                e2++;
            }
            const hp_event* a = &ev.v[e];
            e = e2;
            const double freq = (double)total / ((double)depth > 1.0 ? (double)depth : 1.0);
            if ((double)total < p->candidate_support_threshold) continue;           /* :808 */
            if (a->type != 1 && freq < p->indel_candidate_freq_threshold) continue;  /* :812 */
            if (a->type == 1 && freq < p->snp_candidate_freq_threshold) continue;    /* :815 */
            if (a->type != 1 && p->skip_indels) continue;                            /* :819 */
            if ((a->type == 1 && !(pass[i] & 2)) || (a->type == 2 && !(pass[i] & 4)) ||
                (a->type == 3 && !(pass[i] & 8))) continue;                          /* :823-827 */

            const int64_t k = out->n_out;
            const int64_t so = out->str_bytes;
            out->n_out += 1;
            out->str_bytes += 1 + a->len;
            if (k >= out->capacity || so + 1 + a->len > out->str_capacity) continue;

            int32_t win[PV_HP_WINDOW_ROWS][PV_HP_FEATURES];
            const int64_t left = i - W / 2;
            for (int r = 0; r <= W; r++) { /* :943-957; row R exists and is zero */
                const int64_t src = left + r;
                for (int j = 0; j < F; j++) win[r][j] = (src < 0 || src > R) ? 0 : c.image[src * F + j];
            }
            const int mid = W / 2;
            const int base_plane = a->type; /* 1, 2, 3 */
            win[mid][base_plane] = a->type == 1 ? hp_ref_value(a->bytes[0]) : hp_imin(a->len, PV_MAX_COLOR); /* :970, :983, :996 */
            win[mid][4 + a->type] = hp_imin(cnt[0], PV_MAX_COLOR);        /* 5 / 6 / 7   forward, set 1 */
            win[mid][26 + a->type] = hp_imin(cnt[1], PV_MAX_COLOR);       /* 27 / 28 / 29 forward, set 2 */
            win[mid][15 + a->type] = hp_imin(cnt[2], PV_MAX_COLOR);       /* 16 / 17 / 18 reverse, set 1 */
            win[mid][37 + a->type] = hp_imin(cnt[3], PV_MAX_COLOR);       /* 38 / 39 / 40 reverse, set 2 */

            out->region[k] = g;
            out->position[k] = ref_start + i;
            out->depth[k] = (uint8_t)depth;
            out->cand_freq[k] = (uint8_t)hp_imin(total, PV_MAX_COLOR);
            for (int r = 0; r < PV_HP_WINDOW_ROWS; r++)
                for (int j = 0; j < F; j++) {
                    out->images[k * PV_HP_WINDOW_BYTES + r * F + j] = (int8_t)(uint8_t)(win[r][j] & 0xFF);
                    if (out->images_i32) out->images_i32[k * PV_HP_WINDOW_BYTES + r * F + j] = win[r][j];
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

int oracle_summarize_regions_hp(const pv_batch_in* in, const int32_t* read_hp, const pv_params* params, pv_batch_out* out) {
    if (!in || !params || !out) return PV_ERR_INVALID;
    out->n_out = 0;
    out->str_bytes = 0;
    if (out->capacity > 0) out->cand_off[0] = 0;
    for (int g = 0; g < in->n_regions; g++) {
        int rc = hp_summarize_one(in, read_hp, g, params, out);
        if (rc) return rc;
    }
    if (out->n_out > out->capacity || out->str_bytes > out->str_capacity) return PV_ERR_CAPACITY;
    return PV_OK;
}
