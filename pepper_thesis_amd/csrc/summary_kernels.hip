// summary_kernels.hip — the pileup summary-image builder on gfx950 (MI355X).
//
// Replaces RegionalSummaryGenerator::generate_summary + populate_summary_matrix
// (reference: pepper_variant/modules/cpp/region_summary.cpp:337-566, 568-916) for a whole BATCH of
// regions per call. It is an HBM-bound integer pipeline; nothing here is GEMM-shaped.
//
// Data layout in HBM (column = one reference position; global column id = ref_off[g] + i):
//   cnt[n_cols][CNT_STRIDE] int16, COLUMN-MAJOR since round 3 (the 21 counters of a column are 48 contiguous bytes: only the
//   columns something reads - sites and the windows around them, ~20 % - are written at all, by one thread each in three 16-byte
//   stores, and a window's 33 columns are one 1.6 KB run; they were [NCNT][n_cols] planes, every column of every plane written)
//   (cnt_t: a count is bounded by the reads of its region, which k_init holds to <= 32767 - the
//   reference's caller keeps at most MAX_READS_IN_REGION = 5000, pepper_variant/modules/python/Options.py:98); counter index inside a column:
//     0 coverage  1 snp_count  2 insert_count  3 delete_count  4 rare-event count
//     5 + 8*strand + {0 REF, 1 A, 2 C, 3 G, 4 T, 5 I, 6 D, 7 *}   (the 16 accumulated planes of the
//     reference's 26; planes 0-3,5-7,16-18 are constants or overlays and are never stored)
//   the clamp of planes 11..24 (region_summary.cpp:648-653) is applied when windows are gathered, so
//   the raw counters stay available as exact SNP allele counts.
//
// Pipeline (all on one stream, no host round trip in the middle):
//   k_cigar_scan     wave per read: prefix sums over CIGAR ops -> per-op (column, read index)
//   k_tile_fill      lane per (read, 512-column tile): op range of the read that can touch the tile
//   k_pileup_tiles   workgroup per tile: counters in LDS, aligned bases dealt to lanes in padded groups of 4
//   (site flags: frequency thresholds per column, per-tile site counts - in the flush of k_pileup_tiles)
//   k_scan_*         single-block exclusive scans (tiny arrays)
//   k_site_rank      site columns -> site list (rank = tile offset + rank inside the 1024-column block), per-site event bucket sizes
//   k_collect        wave per site, lane per overlapping read: the read's ops at that column by binary search; allele events
//   k_site_alleles   wave per site: dedupe + order alleles like std::set<std::string>, filters
//   k_write_windows  wave per site: gather 33x26, clamp, overlays, int8 cast, metadata, keys
//
// Allele keys never leave their source: an allele is (type, length, pointer into bases/ref), compared
// bytewise exactly as std::string operator< would compare "<type digit><bytes>".
#include "pv_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int NCNT = 21;   // counters of a column
typedef uint32_t cnt_u32x4 __attribute__((ext_vector_type(4)));
constexpr int CNT_STRIDE = 24, CNT_STRIDE_HP = 40;   // counters of a column, padded to whole 16-byte granules (48 / 80 bytes)
static_assert(NCNT <= CNT_STRIDE, "the 21 counters of the 26-plane form fit its stride");
typedef int16_t cnt_t;               // element of the global counter planes (they were int32: half the flush and gather bytes)
constexpr int MAX_REGION_READS = 32767;
constexpr int C_COV = 0, C_SNP = 1, C_INS = 2, C_DEL = 3, C_RARE = 4, C_PLANE = 5;
// haplotag-aware builder (region_summary_hp.cpp): the same four site counters, then 4 groups (set 1 fwd, set 1 rev,
// set 2 fwd, set 2 rev) x {REF count, A, C, G, T, I, D, *} holding the FINAL signed plane values (window plane
// 4 + 11*group for the REF count, 8 + 11*group + k for the symbols)
constexpr int NCNT_HP = 36, HC_PLANE = 4;
static_assert(NCNT_HP <= CNT_STRIDE_HP, "the 36 counters of the haplotag form fit its stride");
constexpr int32_t OP_INACTIVE = 0x7fffffff;
constexpr int UMAX = 1024;  // distinct alleles per site held in LDS
constexpr int UM_SMALL = 96; // table of the k_site_alleles instantiation for sites with few events
constexpr int TILE_COLS = 512;  // columns per pileup tile (one workgroup accumulates a tile in LDS)

enum { D_SPARE = 8 };
enum { D_NSITES = 0, D_NEVENTS = 1, D_NOUT = 2, D_STRBYTES = 3, D_STATUS = 4, D_NPAIRS = 5, D_NINS = 6, D_NROWS = 7, D_NCHUNKS = 8, D_NBIG = 9, D_NDIAG = 10 };
enum { D_DEPTH = D_NDIAG + 7 };      // set (with status PV_ERR_LIMIT) when a region holds more reads than the 16-bit planes can count

struct Event {  // 16 B
    int64_t src;  // index into bases (kind 1) or ref (kind 2)
    int32_t len;
    uint8_t type;   // 1 SNP 2 INS 3 DEL
    uint8_t rev;
    uint8_t kind;   // 1 bases, 2 ref
    uint8_t flags;  // bit0: is an allele observation; bit1: plane correction (lower-case acgt counted in an ACGT plane);
                    // bits 2-3 (haplotag form): the haplotype sets whose per-strand allele counts the observation joins
};

struct AlleleRec {  // 32 B
    int64_t src;
    int32_t len;
    int32_t total;
    int32_t fwd;
    int32_t rev;
    uint8_t type;
    uint8_t kind;  // 0 immediate byte, 1 bases, 2 ref
    uint8_t imm;
    uint8_t pad;
    int32_t pad2;
};

constexpr int SUB_COLS = 64;                      // k_collect's sub-tile index: op offsets at every 64th column of the tile
constexpr int SUB_N = TILE_COLS / SUB_COLS;       // 8
struct PairRec {  // 64 B: everything a tile workgroup needs to walk one (read, tile) pair
    int32_t read, op_lo, op_hi, col_base;
    int32_t R, c_last, ref_len, rev;   // rev: bit0 strand; haplotag builder: bits 1-2 count sets, bits 3-4 symbol sets
    int64_t base0, seq_end;
    // byte k of subw, k = 0 .. SUB_N: (first op that starts at or behind column 64 k of the tile) - op_lo, saturated at 255; byte
    // SUB_N + 1: 1 when the index is there. A site at column c of the tile searches ops [op_lo + sub[c / 64], op_lo + sub[c / 64 + 1]]
    // only (~11 ops in one or two cache lines instead of ~90 in seven scattered probes).
    uint32_t subw[4];
    __device__ __forceinline__ int sub(int k) const {
        const uint32_t w = k < 4 ? subw[0] : (k < 8 ? subw[1] : subw[2]);
        return (int)((w >> (8 * (k & 3))) & 0xFFu);
    }
};
static_assert(sizeof(PairRec) == 64, "PairRec layout");

// What the per-site kernels need to know of a site before they can request anything else, as ONE 48-byte record written by
// k_site_rank: they are chains of dependent round trips at full occupancy (~3 us each under that load), and column -> region
// -> region geometry / tile range was two of those levels in each of them.
struct SiteHdr {
    int32_t col, col_base, R, g;      // global column, first column and length of its region, region
    int32_t p0, np, cov, flags;       // pair list of its tile, coverage, flags: bits 0-7 reference byte, bit 8: base observations wanted
                                      // (rare ones, or - haplotag form - SNP ones, were counted), bits 16-23: the site flag
    int64_t ref_start;                // of its region
    int32_t nev, pad;
};
static_assert(sizeof(SiteHdr) == 48, "SiteHdr layout");

struct SumArgs {
    pv_batch_in in;
    pv_params p;
    int64_t n_reads, n_bases, n_cigar, n_cols;
    int32_t* op_ref;
    int32_t* op_rd;
    int32_t* op_read;
    uint8_t* op_flag;
    int32_t* read_region;
    int32_t* read_t0;
    int32_t* read_t1;
    int32_t* tile_cnt;   // [n_tiles] (read, tile) pairs per tile
    int32_t* tile_off;   // [n_tiles] exclusive scan
    int32_t* tile_fill;
    PairRec* pairs;      // [n_pairs]
    int64_t n_tiles;
    int64_t max_pairs;
    int32_t qmin_snp;    // smallest integer quality q with (double)q >= min_snp_baseq (exact: q is an integer)
    cnt_t* cnt;
    uint8_t* flags;
    int32_t* blk_cnt;
    int32_t* site_col;
    int32_t* site_region;
    int32_t* site_nev;
    int32_t* site_evoff;
    int32_t* site_fill;
    int32_t* site_nemit;
    int64_t* site_strbytes;
    int32_t* site_outoff;
    int64_t* site_stroff;
    Event* ev;
    AlleleRec* rec;
    int64_t* diag;
    int64_t max_sites;
    int64_t max_events;
    pv_batch_out out;
    int64_t* d_counts;
    // ---- haplotag-aware builder only ----
    int32_t hp;               // 1: RegionalSummaryGeneratorHP semantics (48 planes, 21 rows)
    const int32_t* read_hp;   // [n_reads] type_read::hp_tag, or null (all 0)
    // ---- P2 (polisher) summary only ----
    int32_t polish;       // 1: CIGAR semantics of SummaryGenerator::iterate_over_read (N and P consume the reference only)
    int32_t seq_len, seq_step;  // chunk length, chunk length - overlap
    int32_t* pcnt;        // [PC_N][n_cols] plane-major: 10 features, coverage, longest insert
    int32_t* tile_g0;     // [n_tiles] region of each tile's first column
    SiteHdr* site_hdr;    // [max_sites]
    int32_t* big_sites;   // [max_sites] ranks of the sites whose events need the large allele table (diag[D_NBIG] of them)
    int32_t* ins_blk;     // [n_blk] insert rows per 1024-column block
    int32_t* ins_blkoff;  // [n_blk] exclusive scan
    int32_t* ins_off;     // [n_cols + 1] insert rows before every column
    int32_t* ins_cnt;     // [max_ins_rows][10]
    int64_t max_ins_rows;
    int64_t* reg_rows;    // [n_regions + 1] first flat row of every region
    int64_t* reg_chunks;  // [n_regions + 1] first chunk of every region
    uint8_t* flat_img;    // [flat_cap][10]
    int64_t* flat_pos;
    int32_t* flat_idx;
    int64_t flat_cap;
    pv_polish_out pout;
};

__device__ __forceinline__ int up(int c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
__device__ __forceinline__ bool is_acgt(int c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
// offset of get_feature_index's result from its strand start (region_summary.cpp:208-215): A1 C2 G3 T4 I5 D6 other 7
__device__ __forceinline__ int sym_of(int c) {
    c = up(c);
    return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : c == 'I' ? 5 : c == 'D' ? 6 : 7;
}
__device__ __forceinline__ int refcode(int c) {  // get_reference_feature_value, :165-172
    c = up(c);
    return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 5;
}
__device__ __forceinline__ void set_status(int64_t* diag, int code) {
    atomicCAS((unsigned long long*)&diag[D_STATUS], 0ull, (unsigned long long)(long long)code);
}
__device__ __forceinline__ int upper_bound_i64(const int64_t* a, int n, int64_t v) {  // first idx with a[idx] > v
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// The same answer (number of entries <= v) for the short ascending offset tables of a batch (regions + 1 entries) in ONE round
// trip: every entry is requested at once and compared, instead of five dependent probes. wave_: v uniform over the wave, an
// entry per lane; thread_: independent loads in a short loop. Longer tables take the search.
__device__ __forceinline__ int wave_count_le(const int64_t* a, int n, int64_t v, int lane) {
    if (n > 64) return upper_bound_i64(a, n, v);
    const bool le = lane < n && a[lane] <= v;
    return __popcll(__ballot(le));
}
__device__ __forceinline__ int thread_count_le(const int64_t* a, int n, int64_t v) {
    if (n > 32) return upper_bound_i64(a, n, v);
    int c = 0;
    for (int i = 0; i < n; i++) c += a[i] <= v ? 1 : 0;
    return c;
}

__device__ __forceinline__ int64_t last_lane(int64_t v) {   // lane 63's value in every lane (scalar reads, no LDS permute)
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, 63), hi = __builtin_amdgcn_readlane((unsigned)((uint64_t)v >> 32), 63);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// inclusive wave prefix sums on the DPP path: four row shifts inside the 16-lane rows, then the row totals (row_bcast:15 into
// rows 1 and 3, row_bcast:31 into rows 2 and 3) - six DPP adds instead of six ds_bpermute round trips (twelve for the
// 64-bit form, whose halves move separately and are added as one number)
template <int CTRL, int ROWS>
__device__ __forceinline__ int64_t dpp_move64(int64_t x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)x, CTRL, ROWS, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)((uint64_t)x >> 32), CTRL, ROWS, 0xf, false);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t wave_incl_scan(int64_t v, int) {
    int64_t x = v;
    x += dpp_move64<0x111, 0xf>(x);   // row_shr:1
    x += dpp_move64<0x112, 0xf>(x);   // row_shr:2
    x += dpp_move64<0x114, 0xf>(x);   // row_shr:4
    x += dpp_move64<0x118, 0xf>(x);   // row_shr:8
    x += dpp_move64<0x142, 0xa>(x);   // row_bcast:15
    x += dpp_move64<0x143, 0xc>(x);   // row_bcast:31
    return x;
}
__device__ __forceinline__ int wave_incl_scan32(int v, int) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31
    return x;
}

// ---- K1 -------------------------------------------------------------------------------------------
// One wave per read. CIGAR semantics of populate_summary_matrix (:353-565): M/=/X consume both,
// I and S consume the read, D consumes the reference, N and P consume BOTH (the REF_SKIP/PAD cases
// fall through into SOFT_CLIP, :556-561), H/B/unknown consume nothing. The walk stops at the first
// op that starts beyond ref_end (:355); those ops are marked inactive.
__global__ __launch_bounds__(256) void k_cigar_scan(SumArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.n_reads) return;
    const int g = wave_count_le(a.in.read_off, a.in.n_regions + 1, r, lane) - 1;
    if (lane == 0) a.read_region[r] = g;
    const int64_t c0 = a.in.cigar_off[r], c1 = a.in.cigar_off[r + 1];
    const bool skip = a.in.read_mapq[r] == 0;  // :619
    const int64_t R = a.in.ref_end[g] - a.in.ref_start[g] + 1;
    if (!skip && !a.polish && a.in.base_off[r + 1] - a.in.base_off[r] <= 0 && lane == 0) set_status(a.diag, PV_ERR_INVALID);
    int64_t ref_rel = a.in.read_pos[r] - a.in.ref_start[g];
    int64_t rd = 0;
    // Two trips of 64 ops per loop pass, each with its own CIGAR-word register that is reloaded (for two trips on) right
    // after its trip has used it: the words are requested about one and a half trips ahead and no copy between registers
    // makes the wave wait for a load it has just issued (a rotating pair did: s_waitcnt vmcnt(0) at every loop end).
    auto trip = [&](const uint32_t w, const int64_t cb) {
        const int64_t c = cb + lane;
        const int op = w & 0xF;
        const int64_t len = c < c1 ? (int64_t)(w >> 4) : 0;
        const bool cr = (op == 0 || op == 7 || op == 8 || op == 2 || op == 3 || op == 6);
        // P2: REF_SKIP and PAD share the DEL case (summary_generator.cpp:100-114) and consume the reference only
        const bool cq = (op == 0 || op == 7 || op == 8 || op == 1 || op == 4 || (!a.polish && (op == 3 || op == 6)));
        const int64_t dr = cr ? len : 0, dq = cq ? len : 0;
        // 64 lengths below 2^25 sum to less than 2^31: the 32-bit DPP scan is exact for every real CIGAR; anything longer takes
        // the 64-bit shuffle scan
        int64_t ir, iq;
        if (__ballot(len >= (1ll << 25)) == 0) {
            ir = wave_incl_scan32((int)dr, lane);
            iq = wave_incl_scan32((int)dq, lane);
        } else {
            ir = wave_incl_scan(dr, lane);
            iq = wave_incl_scan(dq, lane);
        }
        const int64_t my_ref = ref_rel + ir - dr, my_rd = rd + iq - dq;
        if (c < c1) {
            const bool active = !skip && my_ref < R;
            if (active && (my_ref < -(1ll << 30) || my_rd > (1ll << 30))) set_status(a.diag, PV_ERR_LIMIT);
            a.op_ref[c] = active ? (int32_t)my_ref : OP_INACTIVE;
            a.op_rd[c] = (int32_t)my_rd;
            if (a.polish) a.op_read[c] = (int32_t)r;   // only k_polish_insert walks op -> read
            if (a.polish) a.op_flag[c] = 0;   // (the image builders no longer keep a per-op flag: k_collect repeats the test)
        }
        ref_rel += last_lane(ir);
        rd += last_lane(iq);
    };
    auto fetch = [&](int64_t cb) -> uint32_t { return cb + lane < c1 ? a.in.cigar[cb + lane] : 0u; };
    uint32_t w_a = fetch(c0), w_b = fetch(c0 + 64);
    for (int64_t cb = c0; cb < c1; cb += 128) {
        trip(w_a, cb);
        w_a = fetch(cb + 128);
        if (cb + 64 >= c1) break;
        trip(w_b, cb + 64);
        w_b = fetch(cb + 192);
    }
    // Column span that this read can touch: every effect of populate_summary_matrix lies between the
    // column before its first position (an insert anchored at pos-1 after a leading soft clip) and its
    // last reference-consumed column, clipped to the region. One (read, tile) pair per overlapped tile.
    int64_t lo = a.in.read_pos[r] - a.in.ref_start[g] - 1, hi = ref_rel - 1;
    if (lo < 0) lo = 0;
    if (hi > R - 1) hi = R - 1;
    const int64_t cb0 = a.in.ref_off[g];
    int32_t t0 = 0, t1 = -1;
    if (!skip && hi >= lo) { t0 = (int32_t)((cb0 + lo) / TILE_COLS); t1 = (int32_t)((cb0 + hi) / TILE_COLS); }
    if (lane == 0) { a.read_t0[r] = t0; a.read_t1[r] = t1; }
    for (int32_t t = t0 + lane; t <= t1; t += 64) atomicAdd(&a.tile_cnt[t], 1);
}

// One wave per read: claim a slot in the pair list of every tile the read overlaps and record the op range [op_lo, op_hi)
// of the read that can touch the tile (an op starting one column past the tile may still anchor an indel on the tile's last
// column). The ranges come from ONE coalesced pass over the read's per-op start columns: an op whose start column and its
// predecessor's lie on different sides of a tile boundary is that boundary's lower / upper bound (start columns ascend), so
// each lane looks at its op and its left neighbour's and writes the boundaries between them to a per-wave LDS table. That was
// two binary searches per (read, tile) lane - ~20 dependent scattered probes - and 31 us per 16 regions; reads over more
// than TF_CAP tiles (regions beyond 130 kb) still search.
constexpr int TF_CAP = 256;
__global__ __launch_bounds__(256) void k_tile_fill(SumArgs a) {
    // per wave: lower bound of every 64-column boundary from the first column of tile t0 to that of tile t1 + 1, upper bound of
    // the tile boundaries
    // (16-bit op offsets from the read's first op: 18 KB per workgroup, so that eight of them still fit a CU; a read with more
    // than 65535 ops - or over more than TF_CAP tiles - takes the searches and leaves no sub-tile index)
    __shared__ uint16_t s_lo[4][TF_CAP * SUB_N + 2];
    __shared__ int32_t s_hi[4][TF_CAP + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= a.n_reads || a.diag[D_STATUS] != 0) return;
    const int32_t t0 = a.read_t0[r], t1 = a.read_t1[r];
    if (t1 < t0) return;
    const int g = a.read_region[r];
    const int64_t cb0 = a.in.ref_off[g];
    const int32_t c0 = (int32_t)a.in.cigar_off[r], c1 = (int32_t)a.in.cigar_off[r + 1];
    const int nb = t1 - t0 + 2;   // boundaries: first columns of tiles t0 .. t1 + 1
    const bool table = nb <= TF_CAP + 1 && c1 - c0 <= 65535;
    if (table) {
        for (int i = lane; i < nb; i += 64) s_hi[wv][i] = c1;                         // no op behind the boundary
        for (int i = lane; i < (nb - 1) * SUB_N + 1; i += 64) s_lo[wv][i] = (uint16_t)(c1 - c0);   // no op at or behind the boundary
        auto fetch = [&](int32_t cb) -> int32_t { return cb + lane < c1 ? a.op_ref[cb + lane] : OP_INACTIVE; };
        int32_t carry = -0x7fffffff - 1;   // "start column" of the op before the first
        auto trip = [&](const int32_t x, const int32_t cb) {
            const int32_t c = cb + lane;
            int32_t p = __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false);   // wave_shr:1: the left neighbour's start column
            if (lane == 0) p = carry;
            if (c < c1 && p != x) {
                // boundaries b_t = t * TILE_COLS - cb0 (region-relative), t0 <= t <= t1 + 1
                const int64_t pp = (int64_t)p + cb0, xx = (int64_t)x + cb0;
                // lower bound of the 64-column boundaries u (column 64 u, counted from the first column of tile t0): p < b_u <= x
                const int64_t org = (int64_t)t0 * TILE_COLS;
                int64_t lo_a = ((pp - org) >> 6) + 1, lo_b = (xx - org) >> 6;
                int64_t hi_a = (pp + 511) >> 9, hi_b = ((xx + 511) >> 9) - 1;    // upper bound of the tile boundaries: p <= b_t < x
                static_assert(TILE_COLS == 512 && SUB_COLS == 64, ">> 9, >> 6");
                const int64_t u_last = (int64_t)(nb - 1) * SUB_N;
                if (lo_a < 0) lo_a = 0;
                if (hi_a < t0) hi_a = t0;
                if (lo_b > u_last) lo_b = u_last;
                if (hi_b > (int64_t)t1 + 1) hi_b = (int64_t)t1 + 1;
                for (int64_t u = lo_a; u <= lo_b; u++) s_lo[wv][u] = (uint16_t)(c - c0);
                for (int64_t t = hi_a; t <= hi_b; t++) s_hi[wv][t - t0] = c;
            }
            carry = __builtin_amdgcn_readlane(x, 63);
        };
        int32_t x_a = fetch(c0), x_b = fetch(c0 + 64);   // (two trips per pass, registers reloaded after use: see k_cigar_scan)
        for (int32_t cb = c0; cb < c1; cb += 128) {
            trip(x_a, cb);
            x_a = fetch(cb + 128);
            if (cb + 64 >= c1) break;
            trip(x_b, cb + 64);
            x_b = fetch(cb + 192);
        }
    }
    for (int32_t t = t0 + lane; t <= t1; t += 64) {
        int32_t op_lo, op_hi;
        PairRec pr;
        pr.subw[0] = pr.subw[1] = pr.subw[2] = pr.subw[3] = 0u;
        if (table) {
            const uint16_t* lo_t = &s_lo[wv][(t - t0) * SUB_N];
            const int32_t lower = c0 + lo_t[0];                     // first op with op_ref >= first column of tile t
            op_lo = lower > c0 ? lower - 1 : c0;
            op_hi = s_hi[wv][t + 1 - t0];                           // first op with op_ref > first column of tile t + 1
#pragma unroll
            for (int k = 0; k <= SUB_N; k++) {
                const int32_t d = c0 + (int32_t)lo_t[k] - op_lo;
                pr.subw[k >> 2] |= (uint32_t)(d > 255 ? 255 : d) << (8 * (k & 3));
            }
            pr.subw[(SUB_N + 1) >> 2] |= 1u << (8 * ((SUB_N + 1) & 3));
        } else {
            const int64_t tlo = (int64_t)t * TILE_COLS - cb0, thi = tlo + TILE_COLS - 1;  // region-relative columns
            int32_t lo = c0, hi = c1;  // first op with op_ref >= tlo
            while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if ((int64_t)a.op_ref[mid] < tlo) lo = mid + 1; else hi = mid; }
            op_lo = lo > c0 ? lo - 1 : c0;
            lo = op_lo; hi = c1;  // first op with op_ref > thi + 1
            while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if ((int64_t)a.op_ref[mid] <= thi + 1) lo = mid + 1; else hi = mid; }
            op_hi = lo;
        }
        const int32_t slot = a.tile_off[t] + atomicAdd(&a.tile_fill[t], 1);
        pr.read = (int32_t)r; pr.op_lo = op_lo; pr.op_hi = op_hi; pr.col_base = (int32_t)cb0;
        pr.R = (int32_t)(a.in.ref_end[g] - a.in.ref_start[g] + 1);
        pr.c_last = c1 - 1;
        pr.ref_len = (int32_t)(a.in.ref_off[g + 1] - cb0);
        pr.rev = a.in.read_flags[r] & 1;
        if (a.hp) {
            // region_summary_hp.cpp: REF-count planes and the allele maps take "hp_tag == 0 || hp_tag == k" (:395-402, :415-422);
            // the symbol planes take both sets for tag 0, set 1 for tag 1 and set 2 for ANY other tag (:454-462, get_feature_index :197)
            const int32_t tag = a.read_hp ? a.read_hp[r] : 0;
            const int cs = ((tag == 0 || tag == 1) ? 1 : 0) | ((tag == 0 || tag == 2) ? 2 : 0);
            const int ss = tag == 0 ? 3 : (tag == 1 ? 1 : 2);
            pr.rev |= (cs << 1) | (ss << 3);
        }
        pr.base0 = a.in.base_off[r];
        pr.seq_end = a.in.base_off[r + 1];
        a.pairs[slot] = pr;
    }
}

// ---- K2 -------------------------------------------------------------------------------------------
// One workgroup per TILE of TILE_COLS columns. All 21 counters of the tile live in LDS for the whole
// kernel (ds_add instead of global atomics) and are written out once with coalesced stores, so the
// counter planes need no memset and see no global atomics.
// The tile's work is FLATTENED across the whole workgroup so that no latency chain is per read:
//   pair batch  : up to PT_PB (read, op-range) pair records -> LDS, block prefix sum of their op counts
//   op batch    : one THREAD per op over all pairs of the batch (512 ops at a time): CIGAR word, start
//                 column and read index are fetched with independent loads; indel bookkeeping per thread;
//                 block-wide prefix sum of the in-tile aligned-base counts
//   expansion   : the aligned bases of the 512 ops are dealt to the threads 4 consecutive bases at a
//                 time (one LDS binary search per 4 bases, all 12 byte loads issued before first use),
//                 so lanes stay busy whatever the CIGAR run lengths are and bytes are read coalesced.
constexpr int PT_THREADS = 512;
constexpr int PT_PB = 128;  // pairs per batch
constexpr int PT_GPL = 1;   // groups of 4 consecutive bases per thread per trip (groups strided by the block size)
// counters of a tile live in LDS with the column index swizzled so that the 64 lanes of a wave, which hold columns
// c, c+4, c+8, ... for the same group element, hit 64 consecutive banks
// (= ((lc & 3) << 7) | (lc >> 2) for 0 <= lc < 512, as one multiply-add and one bit-field extract: lc * 513 = lc | lc << 9)
#define SW(lc) ((int)((((unsigned)(lc) * 513u) >> 2) & 511u))
static_assert(TILE_COLS == 512, "SW() assumes 512-column tiles");

// (s_wsum: two sets of wave totals used in turn - `turn` counts the calls - so that one barrier per call is enough: a set is
// written again two calls later, and every thread has passed the barrier of the call in between only after all of them have
// read this one)
__device__ __forceinline__ int block_incl_scan512(int v, int* s_wsum, int tid, int& turn) {
    const int lane = tid & 63, wv = tid >> 6;
    int* ws = s_wsum + (turn & 1) * (PT_THREADS / 64);
    turn++;
    const int inc = wave_incl_scan32(v, lane);
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int k = 0; k < PT_THREADS / 64; k++) off += (k < wv) ? ws[k] : 0;
    return inc + off;
}

// LDS counters of the tile kernel (all non-negative; converted to the global plane-major layout at
// flush time). The common case - a quality-passing A/C/G/T base over an A/C/G/T reference - costs ONE
// ds_add: coverage and the REFF/REFR planes are derived as sums (every counted base lands in exactly
// one symbol plane), anchors / odd symbols / non-ACGT reference columns use the side counters.
enum {
    L_P = 0,      // [2 strands][4]: base A,C,G,T counted over a valid reference
    L_X = 8,      // [2]: counted bases that are NOT in L_P (odd symbol, or reference not ACGT)
    L_O = 10,     // [2][3]: planes I, D, * (ops and odd symbols)
    L_ANC = 16,   // [2]: counted bases that anchor an indel (no REFF/REFR decrement, :381-391)
    L_COVI = 18,  // coverage bumps of the insert-anchor rule (:452-454)
    L_SNP = 19, L_INS = 20, L_DEL = 21, L_RARE = 22, L_N = 23
};

// LDS counters of the haplotag-aware form (region_summary_hp.cpp:393-463): a counted base costs TWO ds_adds whatever the
// read's tag is - one into its count-set class (coverage and the REF-count planes of both haplotypes are sums of classes),
// one into either the SNP counter or its symbol-set class (a match can only land in the plane of the reference's own symbol).
enum {
    HL_REFC = 0,   // [4 count-set classes: none, set 1, set 2, both][2 strands]: quality-passing aligned bases
    HL_M = 8,      // [3 symbol-set classes: set 1, set 2, both][2 strands]: bases equal to a valid reference base
    HL_O = 14,     // [2 sets][2 strands][3]: planes I, D, *
    HL_COVD = 26,  // coverage taken back by inserts that fail the quality bar (:487-488)
    HL_SNP = 27, HL_INS = 28, HL_DEL = 29, HL_N = 30
};

// Four bytes at a time: bit 7 of a result byte is set where the byte is NOT one of A/C/G/T (swar_not_acgt_upper) or not one of
// A/C/G/T/a/c/g/t (swar_not_acgt): two bits of the byte index a four-entry v_perm_b32 table of the expected letters, and what
// differs from it is non-zero
__device__ __forceinline__ uint32_t swar_nonzero(uint32_t z) { return (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u; }
__device__ __forceinline__ uint32_t swar_not_acgt_upper(uint32_t w) {
    return swar_nonzero(__builtin_amdgcn_perm(0u, 0x47544341u, (w >> 1) & 0x03030303u) ^ w);   // index (b >> 1) & 3: A 0, C 1, T 2, G 3
}
__device__ __forceinline__ uint32_t swar_not_acgt(uint32_t w) { return swar_not_acgt_upper(w & 0xDFDFDFDFu); }

// Site flag of one column from its four counters: frequency thresholds of :634-646 (bit 0 site, bits 1-3 which of the
// SNP / insert / delete thresholds passed). Runs in the flush of k_pileup_tiles, where the counters still sit in LDS
// (it was a kernel of its own, k_site_scan, re-reading four planes: 16 us per 1.6 M columns, mostly round trips).
// A tile may run across region boundaries: the region of its first column is looked up when the workgroup starts
// (SiteRegion, off the tile's critical path), a column beyond it walks on from there.
struct SiteRegion { int g; int64_t off, next, R, start, cand_lo, cand_hi; };
__device__ __forceinline__ SiteRegion site_region_load(const SumArgs& a, int g) {
    SiteRegion r;
    r.g = g;
    const bool ok = g >= 0 && g < a.in.n_regions;
    r.off = ok ? a.in.ref_off[g] : 0;
    r.next = ok ? a.in.ref_off[g + 1] : 0;
    r.start = ok ? a.in.ref_start[g] : 0;
    r.R = ok ? a.in.ref_end[g] - r.start + 1 : 0;
    r.cand_lo = ok ? a.in.cand_start[g] : 1;
    r.cand_hi = ok ? a.in.cand_end[g] : 0;
    return r;
}
__device__ __forceinline__ uint8_t site_flag(const SumArgs& a, const SiteRegion& r0, int64_t col, int cov, int n_snp, int n_ins,
                                             int n_del) {
    SiteRegion r = r0;
    if (col >= r0.next) {  // (columns are >= the tile's first: only forwards)
        int g = r0.g;
        while (g + 1 <= a.in.n_regions && a.in.ref_off[g + 1] <= col) g++;
        r = site_region_load(a, g);
    }
    const int64_t i = col - r.off;
    if (i >= r.R) return 0;
    const double cv = (double)cov > 1.0 ? (double)cov : 1.0;
    const double fs = (double)n_snp / cv;
    const double fi = (double)n_ins / cv;
    const double fd = (double)n_del / cv;
    const bool ps = fs >= a.p.snp_freq_threshold, pi = fi >= a.p.insert_freq_threshold, pd = fd >= a.p.delete_freq_threshold;
    const int64_t pos = r.start + i;
    if ((ps || pi || pd) && pos >= r.cand_lo && pos <= r.cand_hi && (double)cov >= a.p.min_coverage_threshold)
        return (uint8_t)(1 | (ps ? 2 : 0) | (pi ? 4 : 0) | (pd ? 8 : 0));
    return 0;
}

// Does the insert of `len` bases whose anchor base is bases[ins_start] count (region_summary.cpp:431-490 /
// region_summary_hp.cpp:469-553)? The quality sum runs over the anchor base and the inserted bases (26-plane form) or over the
// inserted bases only (haplotag form). k_pileup_tiles counts it into the planes of its anchor column, k_collect repeats the
// test at site columns instead of reading a per-op flag (scattered one-byte stores: ~30 MB of HBM writes per 16 regions).
__device__ __forceinline__ bool insert_counts(const SumArgs& a, int64_t ins_start, int32_t len, bool hp) {
    const int64_t L = (int64_t)len + 1;
    int64_t qs_all = 0;
    for (int64_t i = 0; i < L; i++) qs_all += a.in.quals[ins_start + i];
    const int q0 = a.in.quals[ins_start];
    if (hp) return 2 + (int64_t)len <= PV_MAX_ALLELE_KEY && (double)(qs_all - q0) >= a.p.min_indel_baseq * (double)len;
    return 1 + L <= PV_MAX_ALLELE_KEY && (double)qs_all >= a.p.min_indel_baseq * (double)L;
}

template <bool HP>
__global__ __launch_bounds__(PT_THREADS, HP ? 2 : 4) void k_pileup_tiles(SumArgs a) {   // 26-plane form: two workgroups per CU (<= 128 VGPRs)
    __shared__ int32_t s_cnt[HP ? (int)HL_N : (int)L_N][TILE_COLS];
    __shared__ __attribute__((aligned(4))) uint8_t s_ref[TILE_COLS + 16];  // the tile's reference bytes (+16: groups of 4 / 8 columns are read as two / three aligned words from any column of the tile)
    __shared__ uint8_t s_lut[256];           // byte class: bits0-2 plane symbol 1..7, 8 = upper ACGT, 16 = lower acgt, 32 = valid reference
    constexpr int SB_N = 2048;               // 16-slot blocks with an owner entry (32 k slots per op batch; beyond: a search)
    __shared__ uint16_t s_blk[SB_N];         // op that owns the first slot of every 16-slot block: a padded op is ~12 slots, so the
                                             // walk from there is one step or none (64-slot blocks: two or three dependent reads)
    // per-op staging (one op batch)
    __shared__ int32_t s_pref[PT_THREADS];   // inclusive prefix of the in-tile aligned bases, every op padded to whole groups of 4
    // what the expansion needs of an op, as one 32-byte record (two ds_read_b128 per group instead of nine scalar reads):
    struct OpSt {
        int32_t i0s;     // i = j + i0s: offset in the op of slot j
        int32_t iend;    // one past the op's last in-tile base offset
        int32_t lcoff;   // tile-local column of op offset 0
        int32_t meta;    // len - 1
        int32_t base_lo, base_hi;  // global base index of op offset 0
        int32_t bleft;   // bases from there to the end of the read's sequence (saturated)
        int32_t fl;      // bit0 rev, bit1 anchor_next, haplotag form: bits 2-5
    };
    __shared__ __attribute__((aligned(16))) OpSt s_op[PT_THREADS];
    // per-pair staging (one pair batch)
    __shared__ int32_t p_off[PT_PB + 1];     // exclusive prefix of op counts
    constexpr int PB_BLK = 768;              // 32-op blocks of a pair batch with an owner entry (24 k ops; beyond: binary search)
    __shared__ uint8_t p_blk[PB_BLK];
    __shared__ int32_t p_oplo[PT_PB], p_colbase[PT_PB], p_R[PT_PB], p_clast[PT_PB], p_reflen[PT_PB], p_rev[PT_PB];
    __shared__ int64_t p_base0[PT_PB], p_seqend[PT_PB];
    __shared__ int32_t s_wsum[2 * (PT_THREADS / 64)];
    int scan_turn = 0;
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
#ifdef PV_PSTAMPS
    unsigned long long ps_t0, ps_t1, ps_acc[6] = {0, 0, 0, 0, 0, 0};
#define PSTAMP(i) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ps_t1) :: "memory"); ps_acc[i] += ps_t1 - ps_t0; ps_t0 = ps_t1; }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ps_t0) :: "memory");
#else
#define PSTAMP(i)
#endif
    const int64_t tlo = tile * TILE_COLS, thi = tlo + TILE_COLS - 1;  // global columns of this tile
    __shared__ SiteRegion s_sreg;  // region of the tile's first column, for the flush (looked up by k_init)
    if (tid == 0) s_sreg = site_region_load(a, a.tile_g0[tile]);
    for (int i = tid; i < (HP ? (int)HL_N : (int)L_N) * TILE_COLS; i += PT_THREADS) (&s_cnt[0][0])[i] = 0;
    for (int i = tid; i < TILE_COLS + 16; i += PT_THREADS) s_ref[i] = (i < TILE_COLS && tlo + i < a.n_cols) ? a.in.ref[tlo + i] : (uint8_t)'N';
    if (tid < 256) s_lut[tid] = (uint8_t)(sym_of(tid) | (is_acgt(tid) ? 8 : 0) | ((tid != up(tid) && is_acgt(up(tid))) ? 16 : 0) | (is_acgt(up(tid)) ? 32 : 0));
    const int32_t p0 = a.tile_off[tile];
    const int32_t np = a.tile_cnt[tile];
    // quality bar as a per-byte compare: q >= qmin  <=>  high bits decide, or are equal and the low seven bits decide
    [[maybe_unused]] const uint32_t q_low = (uint32_t)(a.qmin_snp & 0x7F) * 0x01010101u;
    [[maybe_unused]] const bool q_hi = a.qmin_snp >= 128, q_all = a.qmin_snp <= 0, q_none = a.qmin_snp > 255;
    __syncthreads();
    for (int32_t pb = 0; pb < np; pb += PT_PB) {
        const int npb = (np - pb) < PT_PB ? (np - pb) : PT_PB;
        // ---- pair batch -> LDS -----------------------------------------------------------------------
        int nops = 0;
        if (tid < npb) {
            const PairRec pr = a.pairs[p0 + pb + tid];
            nops = pr.op_hi - pr.op_lo;
            p_oplo[tid] = pr.op_lo; p_colbase[tid] = pr.col_base; p_R[tid] = pr.R; p_clast[tid] = pr.c_last;
            p_reflen[tid] = pr.ref_len; p_rev[tid] = pr.rev; p_base0[tid] = pr.base0; p_seqend[tid] = pr.seq_end;
        }
        const int incl_ops = block_incl_scan512(nops, s_wsum, tid, scan_turn);
        if (tid < npb) {
            p_off[tid + 1] = incl_ops;
            // pair that owns the first op of every 32-op block that starts inside this pair's range (op -> pair lookups start there)
            for (int bb = (incl_ops - nops + 31) >> 5; (bb << 5) < incl_ops && bb < PB_BLK; bb++) p_blk[bb] = (uint8_t)tid;
        }
        if (tid == 0) p_off[0] = 0;
        __syncthreads();
        const int total_ops = p_off[npb];
        PSTAMP(0)  // pair batch
        // the four words of an op (start column, CIGAR word, read offset, next CIGAR word) are requested one op batch AHEAD:
        // the batch's first phase is otherwise a chain of dependent round trips (pair lookup -> op words -> indel qualities)
        struct OpWords { int pslot; int32_t c, c_last, rr, rdv; uint32_t w, wn; };
        auto op_fetch = [&](int kk) {
            OpWords o;
            o.pslot = 0; o.c = 0; o.c_last = 0; o.rr = OP_INACTIVE; o.rdv = 0; o.w = 15u; o.wn = 15u;
            if (kk < total_ops) {
                int lo;  // last pair slot with p_off[slot] <= kk: the owner of the op's 32-op block, then a short walk
                if ((kk >> 5) < PB_BLK) {
                    lo = p_blk[kk >> 5];
                    while (p_off[lo + 1] <= kk) lo++;
                } else {
                    int hi = npb;
                    lo = 0;
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p_off[mid] <= kk) lo = mid; else hi = mid; }
                }
                o.pslot = lo;
                o.c = p_oplo[lo] + (kk - p_off[lo]);
                o.c_last = p_clast[lo];
                o.rr = a.op_ref[o.c];
                o.w = a.in.cigar[o.c];
                o.rdv = a.op_rd[o.c];
                o.wn = a.in.cigar[o.c < o.c_last ? o.c + 1 : o.c];
            }
            return o;
        };
        OpWords ow_next = op_fetch(tid);
        for (int ob = 0; ob < total_ops; ob += PT_THREADS) {
            // ---- op batch: one thread per op -----------------------------------------------------------
            const int k = ob + tid;
            int32_t ref_rel = 0, rd = 0, len = 0, op = 15, col_base = 0;
            bool active = false, anchor_next = false, rev = false;
            int pslot = 0, hpbits = 0;  // hpbits: bits 0-1 count sets, bits 2-3 symbol sets (haplotag form only)
            int32_t c = 0;
            int64_t clo = 0, chi = -1;
            const OpWords ow = ow_next;
            if (k < total_ops) {
                pslot = ow.pslot;
                c = ow.c;
                const int32_t c_last = ow.c_last;
                const int32_t rr = ow.rr;
                const uint32_t w = ow.w;
                const int32_t rdv = ow.rdv;
                const uint32_t wn = ow.wn;
                col_base = p_colbase[pslot];
                rev = (p_rev[pslot] & 1) != 0;
                hpbits = p_rev[pslot] >> 1;
                active = rr != OP_INACTIVE;
                if (active) {
                    ref_rel = rr; rd = rdv; op = w & 0xF; len = (int32_t)(w >> 4);
                    const int nop = wn & 0xF;
                    anchor_next = (c < c_last) && (nop == PV_CIGAR_IN || nop == PV_CIGAR_DEL);  // :381-391
                }
                clo = tlo - col_base; chi = thi - col_base;  // tile columns relative to the region, clipped to it
                if (clo < 0) clo = 0;
                if (chi > p_R[pslot] - 1) chi = p_R[pslot] - 1;
            }
            [[maybe_unused]] const int so = L_O + (rev ? 3 : 0);
            // (0) an insert anchored in this tile: the qualities of its anchor base and its inserted bases (bytes [start, start + L),
            // start = the base before the insert, L = len + 1) are REQUESTED here and - haplotag form - used behind the scans and the
            // staging below: the one HBM round trip of this phase that nothing used to cover
            bool ins_here = false, ins_long = false;   // (start and column are recomputed when used: the kernel sits at its register cap)
            uint32_t ins_qlo = 0, ins_qhi = 0;
            if (active && op == PV_CIGAR_IN) {  // region_summary.cpp:431-490 / region_summary_hp.cpp:469-553
                const int64_t anchor = (int64_t)ref_rel - 1;
                if (anchor >= clo && anchor <= chi && rd >= 1) {
                    const int64_t ins_start = p_base0[pslot] + rd - 1;
                    const int64_t L = (int64_t)len + 1;
                    if (ins_start + L > p_seqend[pslot]) {
                        set_status(a.diag, PV_ERR_INVALID);
                    } else {
                        ins_here = true;
                        if (L <= 8 && ins_start + 8 <= a.n_bases) {  // the usual short insert: ONE round trip instead of L dependent ones
                            ins_qlo = *reinterpret_cast<const uint32_t*>(a.in.quals + ins_start);
                            ins_qhi = *reinterpret_cast<const uint32_t*>(a.in.quals + ins_start + 4);
                        } else {
                            ins_long = true;
                        }
                    }
                }
            }
            auto ins_use = [&]() {
                if (!ins_here) return;
                const int64_t L = (int64_t)len + 1;
                int64_t qs_all = 0;   // anchor base + inserted bases
                int q0;               // the anchor base
                if (!ins_long) {
                    uint32_t lo = ins_qlo, hi = ins_qhi;
                    if (L <= 4) { hi = 0; if (L < 4) lo &= (1u << (8 * (int)L)) - 1u; }
                    else if (L < 8) hi &= (1u << (8 * ((int)L - 4))) - 1u;
                    q0 = (int)(lo & 0xFF);
                    qs_all = (int64_t)__builtin_amdgcn_sad_u8(lo, 0u, __builtin_amdgcn_sad_u8(hi, 0u, 0u));
                } else {
                    const int64_t ins_start = p_base0[pslot] + rd - 1;
                    for (int64_t i = 0; i < L; i++) qs_all += a.in.quals[ins_start + i];
                    q0 = a.in.quals[ins_start];
                }
                const int lc = (int)((int64_t)col_base + ref_rel - 1 - tlo);
                if constexpr (HP) {
                    const int st = rev ? 1 : 0, ss = hpbits >> 2;
                    const bool qok = (double)(qs_all - q0) >= a.p.min_indel_baseq * (double)len;   // inserted bases only, :482-484
                    if (!qok && (double)q0 >= a.p.min_snp_baseq) atomicAdd(&s_cnt[HL_COVD][SW(lc)], 1);
                    if (2 + (int64_t)len <= PV_MAX_ALLELE_KEY && qok) {
                        if (is_acgt(up(s_ref[lc]))) {
                            if (ss & 1) atomicAdd(&s_cnt[HL_O + (0 + st) * 3 + 0][SW(lc)], 1);
                            if (ss & 2) atomicAdd(&s_cnt[HL_O + (2 + st) * 3 + 0][SW(lc)], 1);
                        }
                        atomicAdd(&s_cnt[HL_INS][SW(lc)], 1);
                    }
                } else {
                    const bool qok = (double)qs_all >= a.p.min_indel_baseq * (double)L;
                    if (qok && (double)q0 < a.p.min_snp_baseq) atomicAdd(&s_cnt[L_COVI][SW(lc)], 1);  // :453
                    if (1 + L <= PV_MAX_ALLELE_KEY && qok) {
                        if (is_acgt(up(s_ref[lc]))) atomicAdd(&s_cnt[so + 0][SW(lc)], 1);
                        atomicAdd(&s_cnt[L_INS][SW(lc)], 1);
                    }
                }
            };
            // The 26-plane form sits at its 128-register cap (two more live values spill): it uses the words at once, as before;
            // the haplotag form (one workgroup per CU, 256 registers) uses them behind the scans.
            // the next batch's op words are requested BEHIND the insert's qualities (so that waiting for those leaves these in
            // flight) and ahead of the delete bookkeeping, which is LDS work: its time and the pair lookup of the prefetch cover
            // most of the qualities' round trip before the 26-plane form uses them
            if (ob + PT_THREADS < total_ops) ow_next = op_fetch(k + PT_THREADS);
            // (1) delete ops; an op belongs to the tile that owns its anchor column
            if constexpr (HP) {
                const int st = rev ? 1 : 0, ss = hpbits >> 2;
                if (active && op == PV_CIGAR_DEL) {  // :556-649
                    const int64_t anchor = (int64_t)ref_rel - 1;
                    if (anchor >= clo && anchor <= chi) {
                        const int lc = (int)(col_base + anchor - tlo);
                        if (is_acgt(up(s_ref[lc]))) {  // unconditional, :561-569
                            if (ss & 1) atomicAdd(&s_cnt[HL_O + (0 + st) * 3 + 1][SW(lc)], 1);
                            if (ss & 2) atomicAdd(&s_cnt[HL_O + (2 + st) * 3 + 1][SW(lc)], 1);
                        }
                        int64_t L = (int64_t)len + 1;
                        if (anchor + L > p_reflen[pslot]) L = p_reflen[pslot] - anchor;
                        if (1 + L <= PV_MAX_ALLELE_KEY) {
                            atomicAdd(&s_cnt[HL_DEL][SW(lc)], 1);
                            }
                    }
                    int64_t i0 = clo - ref_rel; if (i0 < 0) i0 = 0;
                    int64_t i1 = chi + 1 - ref_rel; if (i1 > len) i1 = len;
                    for (int64_t i = i0; i < i1; i += 8) {  // :631-647, eight deleted columns per pass (see the 26-plane form)
                        const int lcb = (int)((int64_t)col_base + ref_rel + i - tlo);
                        const int n = (int)(i1 - i < 8 ? i1 - i : 8);
                        const uint32_t* wp = reinterpret_cast<const uint32_t*>(s_ref) + (lcb >> 2);
                        const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                        const uint32_t bad0 = swar_not_acgt(__builtin_amdgcn_alignbyte(w1, w0, (unsigned)lcb & 3u));
                        const uint32_t bad1 = swar_not_acgt(__builtin_amdgcn_alignbyte(w2, w1, (unsigned)lcb & 3u));
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int inc = e < n ? (int)((((e < 4 ? bad0 : bad1) >> (8 * (e & 3) + 7)) & 1u) ^ 1u) : 0;
                            const int sw = SW(lcb + e);
                            atomicAdd(&s_cnt[HL_O + (0 + st) * 3 + 2][sw], (ss & 1) ? inc : 0);
                            atomicAdd(&s_cnt[HL_O + (2 + st) * 3 + 2][sw], (ss & 2) ? inc : 0);
                        }
                    }
                }
            } else {
            if (active && op == PV_CIGAR_DEL) {  // :491-555
                const int64_t anchor = (int64_t)ref_rel - 1;
                if (anchor >= clo && anchor <= chi) {
                    const int lc = (int)(col_base + anchor - tlo);
                    if (is_acgt(up(s_ref[lc]))) atomicAdd(&s_cnt[so + 1][SW(lc)], 1);  // unconditional, :496
                    int64_t L = (int64_t)len + 1;
                    if (anchor + L > p_reflen[pslot]) L = p_reflen[pslot] - anchor;  // substr truncation, :500
                    if (1 + L <= PV_MAX_ALLELE_KEY) {
                        atomicAdd(&s_cnt[L_DEL][SW(lc)], 1);
                    }
                }
                int64_t i0 = clo - ref_rel; if (i0 < 0) i0 = 0;
                int64_t i1 = chi + 1 - ref_rel; if (i1 > len) i1 = len;
                // :542-552, the '*' plane of the deleted columns inside the tile where the reference is A/C/G/T. Eight columns per
                // pass: their reference bytes arrive as three aligned words and are tested together, and the adds (ZERO where a
                // column does not count; SW() keeps any column inside the plane) follow each other with no read between them -
                // a column at a time was a chain of dependent LDS round trips, the longest part of this phase
                for (int64_t i = i0; i < i1; i += 8) {
                    const int lcb = (int)((int64_t)col_base + ref_rel + i - tlo);
                    const int n = (int)(i1 - i < 8 ? i1 - i : 8);
                    const uint32_t* wp = reinterpret_cast<const uint32_t*>(s_ref) + (lcb >> 2);
                    const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2];
                    const uint32_t bad0 = swar_not_acgt(__builtin_amdgcn_alignbyte(w1, w0, (unsigned)lcb & 3u));
                    const uint32_t bad1 = swar_not_acgt(__builtin_amdgcn_alignbyte(w2, w1, (unsigned)lcb & 3u));
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int inc = e < n ? (int)((((e < 4 ? bad0 : bad1) >> (8 * (e & 3) + 7)) & 1u) ^ 1u) : 0;
                        atomicAdd(&s_cnt[so + 2][SW(lcb + e)], inc);
                    }
                }
            }
            }
            if constexpr (!HP) ins_use();
            // (2) aligned bases of the batch's M/=/X ops, clipped to tile and region
            PSTAMP(1)  // op lookup + indel ops
            const bool is_m = active && (op == PV_CIGAR_MATCH || op == PV_CIGAR_EQUAL || op == PV_CIGAR_DIFF);
            int32_t i0 = 0, eff = 0;
            if (is_m) {
                int64_t lo = clo - ref_rel; if (lo < 0) lo = 0;
                int64_t hi = chi + 1 - ref_rel; if (hi > len) hi = len;
                if (hi > lo) { i0 = (int32_t)lo; eff = (int32_t)(hi - lo); }
            }
            // Slots: every op's in-tile bases are padded to whole groups of 4 slots, so that a GROUP never straddles two ops:
            // one owner lookup, one dword load of bases and one of qualities serve 4 consecutive bases / columns.
            const int32_t effp = (eff + 3) & ~3;
            const int32_t incl = block_incl_scan512(effp, s_wsum, tid, scan_turn);
            s_pref[tid] = incl;
            {
                const int64_t base = (k < total_ops ? p_base0[pslot] : 0) + rd;
                int64_t bleft = (k < total_ops ? p_seqend[pslot] : 0) - base;
                bleft = bleft > 0x7fffffff ? 0x7fffffff : (bleft < -0x7fffffff ? -0x7fffffff : bleft);
                OpSt o;
                o.i0s = i0 - (incl - effp);
                o.iend = i0 + eff;
                o.lcoff = (int32_t)((int64_t)col_base + ref_rel - tlo);
                o.meta = len - 1;
                o.base_lo = (int32_t)(uint32_t)base;
                o.base_hi = (int32_t)(base >> 32);
                o.bleft = (int32_t)bleft;
                o.fl = (rev ? 1 : 0) | (anchor_next ? 2 : 0) | (HP ? hpbits << 2 : 0);
                s_op[tid] = o;
            }
            for (int32_t bb = (incl - effp + 15) >> 4; (bb << 4) < incl && bb < SB_N; bb++) s_blk[bb] = (uint16_t)tid;  // blocks starting inside this op
            __syncthreads();
            const int32_t total = s_pref[PT_THREADS - 1];
            if constexpr (HP) ins_use();   // (0, continued) the insert's qualities have arrived by now
            PSTAMP(2)  // scan + staging + barrier
            // ---- expansion: PT_GPL groups of 4 consecutive bases per thread per trip; the owner lookups and the loads of
            // trip t+1 are issued before trip t is counted, so the HBM round trip of the bases hides behind the ds_adds ----
            struct Grp { int lc, nv, fl, last; uint32_t bw, qw, rw; };
            auto look = [&](int32_t jb, Grp (&g)[PT_GPL]) {
#pragma unroll
                for (int u = 0; u < PT_GPL; u++) {
                    // consecutive lanes take consecutive groups: the dword loads of a wave cover 256 consecutive bytes of a run
                    const int32_t j = jb + (u * PT_THREADS + tid) * 4;
                    const bool ok = j < total;
                    int owc = 0;                       // owner of the block's first slot, then a short probe
                    if (ok) {
                        if ((j >> 4) < SB_N) {
                            owc = s_blk[j >> 4];
                        } else {                       // first op whose inclusive prefix exceeds j
                            int hi = PT_THREADS - 1;
                            while (owc < hi) { const int mid = (owc + hi) >> 1; if (s_pref[mid] <= j) owc = mid + 1; else hi = mid; }
                        }
                        while (s_pref[owc] <= j) owc++;
                    }
                    const OpSt o = s_op[owc];
                    const int32_t i = j + o.i0s;
                    int nv = o.iend - i;               // valid bases of the group (the rest is padding)
                    nv = ok ? (nv > 4 ? 4 : nv) : 0;
                    const int64_t bi = (int64_t)(((uint64_t)(uint32_t)o.base_hi << 32) | (uint32_t)o.base_lo) + i;
                    const int64_t left = (int64_t)o.bleft - i;
                    if (nv > 0 && nv > left) { set_status(a.diag, PV_ERR_INVALID); nv = left > 0 ? (int)left : 0; }
                    const int lc = o.lcoff + i;
                    g[u].lc = lc;
                    g[u].nv = nv;
                    const int f = o.fl;
                    g[u].fl = HP ? f : (f & 1);
                    g[u].last = (f & 2) ? o.meta - i : -1;  // group position of the op's last base, if that base anchors an indel
                    uint32_t b4 = 0, q4 = 0;
                    if (nv > 0) {
                        if (bi + 4 <= a.n_bases) {  // unaligned dword loads
                            b4 = *reinterpret_cast<const uint32_t*>(a.in.bases + bi);
                            q4 = *reinterpret_cast<const uint32_t*>(a.in.quals + bi);
                        } else {
                            for (int e = 0; e < nv; e++) {
                                b4 |= (uint32_t)a.in.bases[bi + e] << (8 * e);
                                q4 |= (uint32_t)a.in.quals[bi + e] << (8 * e);
                            }
                        }
                    }
                    g[u].bw = b4; g[u].qw = q4;
                    const int lcr = nv > 0 ? lc : 0;   // four reference bytes from lcr on: two aligned words, shifted together
                    const uint32_t* rwp = reinterpret_cast<const uint32_t*>(s_ref) + (lcr >> 2);
                    g[u].rw = __builtin_amdgcn_alignbyte(rwp[1], rwp[0], (unsigned)lcr & 3u);
                }
            };
            // general classification of one counted base (any byte over any reference byte), :379-423
            auto count_general = [&](const Grp& G, int e, int st) {
                const int base = (G.bw >> (8 * e)) & 0xFF, refb = (G.rw >> (8 * e)) & 0xFF;
                const int lc = G.lc + e;
                const int cb = s_lut[base];
                const bool refvalid = (s_lut[refb] & 32) != 0;
                const int sy = cb & 7;                                           // 1..7
                if (refvalid && sy <= 4) {
                    atomicAdd(&s_cnt[L_P + 4 * st + (sy - 1)][SW(lc)], 1);       // :379 + :381-391 + :396,423 in one
                } else {
                    atomicAdd(&s_cnt[L_X + st][SW(lc)], 1);
                    if (refvalid) atomicAdd(&s_cnt[L_O + 3 * st + (sy - 5)][SW(lc)], 1);
                }
                if (e == G.last) atomicAdd(&s_cnt[L_ANC + st][SW(lc)], 1);
                const bool mism = refb != base;                                  // raw bytes, :394
                if (mism) atomicAdd(&s_cnt[L_SNP][SW(lc)], 1);
                const bool rare = mism && !(refvalid && (cb & 8));
                const bool corr = refvalid && (cb & 16);
                if (rare || corr) atomicAdd(&s_cnt[L_RARE][SW(lc)], 1);
            };
            auto count = [&](const Grp (&g)[PT_GPL]) {
#pragma unroll
                for (int u = 0; u < PT_GPL; u++) {
                    // quality bar and the group's valid bases, a byte per base (bit 7 = counts)
                    constexpr uint32_t H = 0x80808080u;
                    const Grp& G = g[u];
                    const uint32_t tq = (G.qw | H) - q_low;                              // bit 7: low seven bits of q >= those of qmin
                    uint32_t ge = q_hi ? (G.qw & tq) : (G.qw | tq);
                    ge = q_all ? H : (q_none ? 0u : ge);
                    const uint32_t vm = G.nv >= 4 ? H : ((H >> 8) >> (24 - 8 * (G.nv < 0 ? 0 : G.nv)));   // the group's valid bases
                    const uint32_t ok = ge & vm;
                    if constexpr (HP) {  // region_summary_hp.cpp:393-463: a counted base adds to its count-set class and to either the
                        // SNP counter (raw bytes differ, :406) or - over a valid reference - its symbol-set class; branch-free like the
                        // 26-plane form (a base that does not count adds zero)
                        const uint32_t differs = swar_nonzero(G.bw ^ G.rw);
                        const uint32_t second = ok & (differs | ~swar_not_acgt(G.rw));
                        const int st = G.fl & 1, cs = (G.fl >> 2) & 3, ss = (G.fl >> 4) & 3;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int sw = SW(G.lc + e);
                            atomicAdd(&s_cnt[HL_REFC + 2 * cs + st][sw], (int)((ok >> (8 * e + 7)) & 1u));
                            const bool mm = ((differs >> (8 * e + 7)) & 1u) != 0;
                            atomicAdd(&s_cnt[mm ? (int)HL_SNP : HL_M + 2 * (ss - 1) + st][sw], (int)((second >> (8 * e + 7)) & 1u));
                        }
                    } else {
                        // The four bases of a group are classified together, a byte per base in 32-bit operations (bit 7 of a byte =
                        // the answer for that base), so that the usual base - A/C/G/T in upper case over an A/C/G/T reference of
                        // either case, quality passing - costs a bit test, an address and its one ds_add; bases that are anything
                        // else take count_general, one by one.
                        const uint32_t selb = (G.bw >> 1) & 0x03030303u;                     // A 0, C 1, T 2, G 3
                        const uint32_t b_bad = swar_not_acgt_upper(G.bw);
                        const uint32_t r_bad = swar_not_acgt(G.rw);
                        const uint32_t fast = ok & ~(b_bad | r_bad);
                        const uint32_t slow = ok & (b_bad | r_bad);
                        const uint32_t mism = swar_nonzero(G.bw ^ G.rw) & fast;                   // raw bytes, :394 (never rare: both are A/C/G/T)
                        const uint32_t pidx = selb ^ ((selb >> 1) & 0x01010101u);            // -> A 0, C 1, G 2, T 3
                        const int st = G.fl;
                        // no branches: a base that does not count adds ZERO (SW() keeps any column inside the plane, the plane
                        // index is two bits of the byte), which costs the LDS nothing it was not already doing - some lane of
                        // the wave nearly always counts - and saves the exec-mask bookkeeping per base
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int sw = SW(G.lc + e);
                            atomicAdd(&s_cnt[L_P + 4 * st + (int)((pidx >> (8 * e)) & 3u)][sw], (int)((fast >> (8 * e + 7)) & 1u));
                            atomicAdd(&s_cnt[L_SNP][sw], (int)((mism >> (8 * e + 7)) & 1u));
                        }
                        {
                            const unsigned la = (unsigned)G.last < 4u ? (unsigned)G.last : 0u;
                            const int inc = (unsigned)G.last < 4u ? (int)((fast >> (8 * la + 7)) & 1u) : 0;
                            atomicAdd(&s_cnt[L_ANC + st][SW(G.lc + (int)la)], inc);
                        }
                        if (slow) {
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                if (slow & (0x80u << (8 * e))) count_general(G, e, st);
                        }
                    }
                }
            };
            constexpr int32_t TRIP = PT_THREADS * PT_GPL * 4;
            Grp ga[PT_GPL], gb[PT_GPL];
            if (total > 0) look(0, ga);
            for (int32_t jb = 0; jb < total; jb += 2 * TRIP) {
                if (jb + TRIP < total) look(jb + TRIP, gb);
                count(ga);
                if (jb + TRIP < total) {
                    if (jb + 2 * TRIP < total) look(jb + 2 * TRIP, ga);
                    count(gb);
                }
            }
            PSTAMP(3)  // expansion
            __syncthreads();  // staging arrays are rewritten by the next op batch
            PSTAMP(4)
        }
        __syncthreads();  // pair arrays are rewritten by the next pair batch
    }
    __syncthreads();
    // flush: derive the global plane-major counters (negative counts, as the reference keeps them)
    PSTAMP(4)
    const int64_t NC = a.n_cols;
    int64_t ncol = NC - tlo;
    if (ncol > TILE_COLS) ncol = TILE_COLS;
    static_assert(TILE_COLS <= PT_THREADS, "one column per thread: the site count below is a ballot");
    int site = 0;
    const SiteRegion sreg = s_sreg;
    // Pass 1: the four site counters of this thread's column, its flag, and - a ballot per wave - which columns of the tile are
    // sites. Pass 2 writes the counter planes ONLY where something will read them: the planes are read at site columns
    // (k_site_rank, k_site_alleles) and in the windows around them (k_write_windows: W columns to either side), i.e. ~20 % of the
    // columns at one site per ~190 columns; a column within W of the tile's edge is written anyway, because the site that needs it
    // may lie in the next tile. (Before: every column of every plane, 85 MB per 16 regions, the largest write of the chain.)
    constexpr int W = HP ? (PV_HP_WINDOW_ROWS - 1) / 2 : (PV_WINDOW_ROWS - 1) / 2;
    __shared__ unsigned long long s_sitebits[PT_THREADS / 64];
    int cov = 0, n_snp = 0, n_ins = 0, n_del = 0;
    const int lc = tid;                      // one column per thread (TILE_COLS == PT_THREADS)
    const bool have = lc < ncol;
    const int64_t g = tlo + lc;
    if (have) {
        if constexpr (HP) {
            cov = -s_cnt[HL_COVD][SW(lc)];
#pragma unroll
            for (int k = 0; k < 8; k++) cov += s_cnt[HL_REFC + k][SW(lc)];
            n_snp = s_cnt[HL_SNP][SW(lc)]; n_ins = s_cnt[HL_INS][SW(lc)]; n_del = s_cnt[HL_DEL][SW(lc)];
        } else {
            cov = s_cnt[L_COVI][SW(lc)];
#pragma unroll
            for (int st = 0; st < 2; st++) {
#pragma unroll
                for (int k = 0; k < 4; k++) cov += s_cnt[L_P + 4 * st + k][SW(lc)];
                cov += s_cnt[L_X + st][SW(lc)];
            }
            n_snp = s_cnt[L_SNP][SW(lc)]; n_ins = s_cnt[L_INS][SW(lc)]; n_del = s_cnt[L_DEL][SW(lc)];
        }
        const uint8_t f = site_flag(a, sreg, g, cov, n_snp, n_ins, n_del);
        a.flags[g] = f;
        site = f & 1;
    }
    const unsigned long long site_m = __ballot(site);
    if ((tid & 63) == 0) s_sitebits[tid >> 6] = site_m;
    __syncthreads();
    bool need = have && (lc < W || lc >= (int)ncol - W);
    if (have && !need) {
        const int c0 = lc - W, c1 = lc + W;   // inside [0, ncol) here
#pragma unroll
        for (int wdx = 0; wdx < PT_THREADS / 64; wdx++) {
            const int lo = wdx * 64, hi = lo + 63;
            if (c1 < lo || c0 > hi) continue;
            const int b0_ = c0 > lo ? c0 - lo : 0, b1_ = c1 < hi ? c1 - lo : 63;
            const unsigned long long mask = (b1_ - b0_ == 63) ? ~0ull : (((1ull << (b1_ - b0_ + 1)) - 1ull) << b0_);
            need = need || (s_sitebits[wdx] & mask) != 0;
        }
    }
    if (need) {
        if constexpr (HP) {
            cnt_t v[CNT_STRIDE_HP];
#pragma unroll
            for (int k = 0; k < CNT_STRIDE_HP; k++) v[k] = 0;
            const int rsym = s_lut[s_ref[lc]];  // bits0-2: plane symbol of the reference byte, bit 5: valid reference
#pragma unroll
            for (int set = 0; set < 2; set++) {
#pragma unroll
                for (int st = 0; st < 2; st++) {
                    const int grp = 2 * set + st;
                    cnt_t* dst = v + HC_PLANE + 8 * grp;
                    dst[0] = (cnt_t)-(s_cnt[HL_REFC + 2 * (1 + set) + st][SW(lc)] + s_cnt[HL_REFC + 2 * 3 + st][SW(lc)]);
                    const int m = s_cnt[HL_M + 2 * set + st][SW(lc)] + s_cnt[HL_M + 2 * 2 + st][SW(lc)];
#pragma unroll
                    for (int k = 0; k < 4; k++) dst[1 + k] = (cnt_t)(((rsym & 32) && (rsym & 7) == k + 1) ? -m : 0);
#pragma unroll
                    for (int k = 0; k < 3; k++) dst[5 + k] = (cnt_t)s_cnt[HL_O + grp * 3 + k][SW(lc)];
                }
            }
            v[C_COV] = (cnt_t)cov; v[C_SNP] = (cnt_t)n_snp; v[C_INS] = (cnt_t)n_ins; v[C_DEL] = (cnt_t)n_del;
            cnt_u32x4* dst4 = reinterpret_cast<cnt_u32x4*>(a.cnt + g * CNT_STRIDE_HP);
#pragma unroll
            for (int k = 0; k < CNT_STRIDE_HP / 8; k++) {
                cnt_u32x4 w4;
#pragma unroll
                for (int j = 0; j < 4; j++) w4[j] = (uint32_t)(uint16_t)v[8 * k + 2 * j] | ((uint32_t)(uint16_t)v[8 * k + 2 * j + 1] << 16);
                dst4[k] = w4;
            }
        } else {
            cnt_t v[CNT_STRIDE];
#pragma unroll
            for (int k = 0; k < CNT_STRIDE; k++) v[k] = 0;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                int sp = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int x = s_cnt[L_P + 4 * st + k][SW(lc)];
                    sp += x;
                    v[C_PLANE + 8 * st + 1 + k] = (cnt_t)-x;
                }
                const int counted = sp + s_cnt[L_X + st][SW(lc)];
                v[C_PLANE + 8 * st] = (cnt_t)-(counted - s_cnt[L_ANC + st][SW(lc)]);
#pragma unroll
                for (int k = 0; k < 3; k++) v[C_PLANE + 8 * st + 5 + k] = (cnt_t)-s_cnt[L_O + 3 * st + k][SW(lc)];
            }
            v[C_COV] = (cnt_t)cov; v[C_SNP] = (cnt_t)n_snp; v[C_INS] = (cnt_t)n_ins; v[C_DEL] = (cnt_t)n_del;
            v[C_RARE] = (cnt_t)s_cnt[L_RARE][SW(lc)];
            cnt_u32x4* dst4 = reinterpret_cast<cnt_u32x4*>(a.cnt + g * CNT_STRIDE);
#pragma unroll
            for (int k = 0; k < CNT_STRIDE / 8; k++) {
                cnt_u32x4 w4;
#pragma unroll
                for (int j = 0; j < 4; j++) w4[j] = (uint32_t)(uint16_t)v[8 * k + 2 * j] | ((uint32_t)(uint16_t)v[8 * k + 2 * j + 1] << 16);
                dst4[k] = w4;
            }
        }
    }
    if ((tid & 63) == 0 && site_m) atomicAdd(&a.blk_cnt[tile], __popcll(site_m));
#ifdef PV_PSTAMPS
    PSTAMP(5)  // flush
    if (tid == 0 && a.site_strbytes) {  // debug: reuse a workspace array that is written later in the pipeline
        for (int i = 0; i < 6; i++) atomicAdd((unsigned long long*)&a.diag[D_NDIAG + i], ps_acc[i]);
    }
#endif
#undef PSTAMP
}

// ---- K3 -------------------------------------------------------------------------------------------
// Exclusive scans of the pipeline's small arrays (tiles, 1024-column blocks, sites). A thread owns SCAN_V consecutive
// values per pass, a 1024-thread workgroup 8192. These kernels are chains of dependent memory round trips, not work,
// so the chains are kept short:
//  * arrays whose length the host knows (tiles, blocks) run as one workgroup looping over passes, every thread
//    summing the 16 wave totals itself (no carry cell, two barriers per pass);
//  * the per-site arrays (length = diag[D_NSITES], 17.9 k in the benchmark's 16-region batches: three passes) run one
//    workgroup per 8192-entry chunk. A chunk's carry is the sum of everything before it, which the workgroup adds up
//    itself from the input (independent coalesced loads: at most n^2 / 16 k loads in all, nothing at these sizes)
//    instead of waiting for its neighbours; the chunk's own values are loaded together WITH the length (the arrays
//    are max_sites long, the grid covers max_sites) and masked once it has arrived. The workgroup holding the last
//    entry publishes the totals. k_scan_outputs runs its two scans side by side.
// What used to be separate one-thread kernels behind a scan (limit checks, publishing the result counters) runs in the
// scan's first thread.
constexpr int SCAN_V = 8;
constexpr int64_t SCAN_PASS = 1024 * SCAN_V;
constexpr int SCAN_SPEC_CHUNKS = 3;  // chunks below this add up their carry before the length has arrived
// Loads go through a sized raw buffer: entries past `elems` read as zero without a branch per load, so all of a
// thread's loads are in flight together. (Byte offsets are 32-bit: arrays here stay below 2^31 bytes, n_cols < 2^31.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t scan_rsrc(const void* p, int64_t elems, int esz) {
    const int64_t bytes = elems > 0 ? elems * esz : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffff ? 0x7fffffff : bytes),
                                             0x00020000);
}
__device__ __forceinline__ unsigned scan_voff(int64_t i, int esz) {
    const int64_t b = i * esz;
    return b > 0x7fffffff ? 0x80000000u : (unsigned)b;
}
__device__ __forceinline__ int32_t scan_ld(__amdgpu_buffer_rsrc_t r, int64_t i, int32_t) {
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(r, scan_voff(i, 4), 0, 0);
}
__device__ __forceinline__ int64_t scan_ld(__amdgpu_buffer_rsrc_t r, int64_t i, int64_t) {
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r, scan_voff(i, 8), 0, 0);
    const unsigned lo = w[0], hi = w[1];
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
typedef unsigned scan_u32x4 __attribute__((ext_vector_type(4)));
// sum of the 16 bytes at entry i (4 int32 or 2 int64 values)
__device__ __forceinline__ int64_t scan_ld16_sum(__amdgpu_buffer_rsrc_t r, int64_t i, int32_t) {
    const scan_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, scan_voff(i, 4), 0, 0);
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    return (int64_t)(int32_t)w0 + (int32_t)w1 + (int64_t)(int32_t)w2 + (int32_t)w3;
}
__device__ __forceinline__ int64_t scan_ld16_sum(__amdgpu_buffer_rsrc_t r, int64_t i, int64_t) {
    const scan_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, scan_voff(i, 8), 0, 0);
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    return (int64_t)(((uint64_t)w1 << 32) | w0) + (int64_t)(((uint64_t)w3 << 32) | w2);
}
// A thread's SCAN_V consecutive values. The vector memory pipe of the one CU a scan workgroup sits on is what these
// kernels wait for (about 18 cycles per load instruction, 16 waves), so the values come as 16-byte loads; only a thread
// whose run crosses `lim` loads entry by entry.
__device__ __forceinline__ void scan_load(const int32_t* in, int64_t i0, int64_t lim, int32_t (&v)[SCAN_V]) {
    const __amdgpu_buffer_rsrc_t r = scan_rsrc(in, lim, 4);
    if (i0 + SCAN_V <= lim || i0 >= lim) {
#pragma unroll
        for (int q = 0; q < SCAN_V / 4; q++) {
            const scan_u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(r, scan_voff(i0 + 4 * q, 4), 0, 0);
            const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            v[4 * q] = (int32_t)w0; v[4 * q + 1] = (int32_t)w1; v[4 * q + 2] = (int32_t)w2; v[4 * q + 3] = (int32_t)w3;
        }
    } else {
#pragma unroll
        for (int e = 0; e < SCAN_V; e++) v[e] = scan_ld(r, i0 + e, int32_t());
    }
}
// (8-byte values: one load each, whole or not at all, so no branch - a branch here makes the compiler wait for the
// loads at its end, ahead of the loads that should follow them out)
__device__ __forceinline__ void scan_load(const int64_t* in, int64_t i0, int64_t lim, int64_t (&v)[SCAN_V]) {
    const __amdgpu_buffer_rsrc_t r = scan_rsrc(in, lim, 8);
#pragma unroll
    for (int e = 0; e < SCAN_V; e++) v[e] = scan_ld(r, i0 + e, int64_t());
}
template <typename T>
__device__ __forceinline__ void scan_mask(int64_t i0, int64_t n, T (&v)[SCAN_V]) {
#pragma unroll
    for (int e = 0; e < SCAN_V; e++) v[e] = i0 + e < n ? v[e] : (T)0;
}
// this thread's share of sum(in[0 .. c0)), c0 a multiple of SCAN_PASS and at most `lim`: 16 bytes per load, lanes side
// by side
template <typename T>
__device__ __forceinline__ int64_t scan_carry_part(const T* in, int64_t c0, int64_t lim) {
    constexpr int PER = 16 / (int)sizeof(T);  // values per load
    const __amdgpu_buffer_rsrc_t r = scan_rsrc(in, lim, (int)sizeof(T));
    int64_t pre = 0;
    for (int64_t j0 = (int64_t)threadIdx.x * PER; j0 < c0; j0 += SCAN_PASS) {
        int64_t q[SCAN_V / PER];
#pragma unroll
        for (int e = 0; e < SCAN_V / PER; e++) q[e] = scan_ld16_sum(r, j0 + 1024 * PER * e, T());
#pragma unroll
        for (int e = 0; e < SCAN_V / PER; e++) pre += q[e];
    }
    return pre;
}
// the same for the first SCAN_SPEC_CHUNKS chunks: a fixed number of loads (those at or past c0 read as zero: c0 is a
// multiple of every load's span), all in flight at once
template <typename T>
__device__ __forceinline__ int64_t scan_carry_part_spec(const T* in, int64_t c0) {
    constexpr int PER = 16 / (int)sizeof(T);
    constexpr int NL = SCAN_V * (SCAN_SPEC_CHUNKS - 1) / PER;
    const __amdgpu_buffer_rsrc_t r = scan_rsrc(in, c0, (int)sizeof(T));
    int64_t q[NL];
#pragma unroll
    for (int k = 0; k < NL; k++) q[k] = scan_ld16_sum(r, ((int64_t)threadIdx.x + 1024 * k) * PER, T());
    int64_t pre = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) pre += q[k];
    return pre;
}
// the scans' length: a vector load (kept in program order with the value loads around it: issued before the carry
// loads, waited for after them), then made uniform
__device__ __forceinline__ int64_t scan_len_issue(const int64_t* p) { return scan_ld(scan_rsrc(p, 1, 8), 0, int64_t()); }
__device__ __forceinline__ int64_t scan_len_uniform(int64_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
// one pass over v[] (zero beyond n): out[i0..] = carry + exclusive prefix; returns carry + the pass total.
// carry = `carry` (same in every thread) + the sum of `part` over the workgroup. s_w: 32 cells, reusable after the
// closing barrier, which only a caller with another pass to run asks for (it also waits for the stores).
template <typename T>
__device__ __forceinline__ int64_t scan_pass(const T (&v)[SCAN_V], T* out, int64_t i0, int64_t n, int64_t carry,
                                             int64_t part, int64_t* s_w, bool again = false) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t sum = 0;
#pragma unroll
    for (int e = 0; e < SCAN_V; e++) sum += (int64_t)v[e];
    const int64_t inc = wave_incl_scan(sum, lane);
    const int64_t pinc = wave_incl_scan(part, lane);
    if (lane == 63) { s_w[wv] = inc; s_w[16 + wv] = pinc; }
    __syncthreads();
    int64_t woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int64_t w = s_w[k];
        woff += k < wv ? w : 0;
        tot += w;
        carry += s_w[16 + k];
    }
    int64_t run = carry + woff + inc - sum;
    T o[SCAN_V];
#pragma unroll
    for (int e = 0; e < SCAN_V; e++) { o[e] = (T)run; run += (int64_t)v[e]; }
    if (i0 + SCAN_V <= n) {  // 16-byte stores (i0 is a multiple of SCAN_V, the arrays are 256-byte aligned)
        typedef T ovec __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int PER = 16 / (int)sizeof(T);
#pragma unroll
        for (int q = 0; q < SCAN_V / PER; q++) {
            ovec w;
#pragma unroll
            for (int e = 0; e < PER; e++) w[e] = o[PER * q + e];
            *reinterpret_cast<ovec*>(out + i0 + PER * q) = w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < SCAN_V; e++) if (i0 + e < n) out[i0 + e] = o[e];
    }
    if (again) __syncthreads();
    return carry + tot;
}
// one workgroup, n known to the host
template <typename T>
__device__ __forceinline__ int64_t block_excl_scan(const T* in, T* out, int64_t n, int64_t* s_w) {
    int64_t carry = 0;
    for (int64_t b = 0; b == 0 || b < n; b += SCAN_PASS) {
        const int64_t i0 = b + (int64_t)threadIdx.x * SCAN_V;
        T v[SCAN_V];
        scan_load(in, i0, n, v);
        carry = scan_pass(v, out, i0, n, carry, 0, s_w, b + SCAN_PASS < n);
    }
    return carry;
}
// grid of the per-site scans
static inline unsigned scan_chunks(int64_t cap) { return (unsigned)std::max<int64_t>(1, (cap + SCAN_PASS - 1) / SCAN_PASS); }

// tile pair counts -> offsets, total -> diag[D_NPAIRS]; over the pair workspace: nothing is filled or walked
__global__ __launch_bounds__(1024) void k_scan_tiles(SumArgs a) {
    __shared__ int64_t s_w[32];
    const int64_t total = block_excl_scan<int32_t>(a.tile_cnt, a.tile_off, a.n_tiles, s_w);
    if (total > a.max_pairs) {
        if (threadIdx.x == 0) set_status(a.diag, PV_ERR_LIMIT);
        for (int64_t t = threadIdx.x; t < a.n_tiles; t += 1024) a.tile_cnt[t] = 0;
    }
    if (threadIdx.x == 0) a.diag[D_NPAIRS] = total;
}


// events per site -> offsets, total -> diag[D_NEVENTS]; site / event workspace limits. Grid: scan_chunks(max_sites).
__global__ __launch_bounds__(1024) void k_scan_events(SumArgs a) {
    __shared__ int64_t s_w[32];
    const int64_t c0 = (int64_t)blockIdx.x * SCAN_PASS, i0 = c0 + (int64_t)threadIdx.x * SCAN_V;
    const bool spec = blockIdx.x < SCAN_SPEC_CHUNKS;
    int32_t v[SCAN_V];
    scan_load(a.site_nev, i0, a.max_sites, v);
    const int64_t n_raw = scan_len_issue(a.diag + D_NSITES);
    int64_t part = spec ? scan_carry_part_spec(a.site_nev, c0) : 0;
    const int64_t n_sites = scan_len_uniform(n_raw);
    const int64_t n = n_sites > a.max_sites ? a.max_sites : n_sites;
    if (c0 >= n && blockIdx.x > 0) return;
    if (!spec) part = scan_carry_part(a.site_nev, c0, a.max_sites);
    scan_mask(i0, n, v);
    const int64_t total = scan_pass(v, a.site_evoff, i0, n, 0, part, s_w);
    if (threadIdx.x == 0 && n <= c0 + SCAN_PASS) {
        a.diag[D_NEVENTS] = total;
        if (n_sites > a.max_sites || total > a.max_events) set_status(a.diag, PV_ERR_LIMIT);
    }
}

// windows and key bytes per site -> offsets, totals -> diag[D_NOUT], diag[D_STRBYTES]; result counters of the call.
// Grid: scan_chunks(max_sites).
__global__ __launch_bounds__(1024) void k_scan_outputs(SumArgs a) {
    __shared__ int64_t s_w[2][32];
    const int64_t c0 = (int64_t)blockIdx.x * SCAN_PASS, i0 = c0 + (int64_t)threadIdx.x * SCAN_V;
    const bool spec = blockIdx.x < SCAN_SPEC_CHUNKS;
    int32_t ve[SCAN_V];
    int64_t vs[SCAN_V];
    scan_load(a.site_nemit, i0, a.max_sites, ve);
    scan_load(a.site_strbytes, i0, a.max_sites, vs);
    const int64_t n_raw = scan_len_issue(a.diag + D_NSITES);
    int64_t pe = 0, ps = 0;
    if (spec) {
        pe = scan_carry_part_spec(a.site_nemit, c0);
        ps = scan_carry_part_spec(a.site_strbytes, c0);
    }
    const int64_t n_sites = scan_len_uniform(n_raw);
    const int64_t n = n_sites > a.max_sites ? a.max_sites : n_sites;
    if (c0 >= n && blockIdx.x > 0) return;
    if (!spec) {
        pe = scan_carry_part(a.site_nemit, c0, a.max_sites);
        ps = scan_carry_part(a.site_strbytes, c0, a.max_sites);
    }
    const int64_t status = a.diag[D_STATUS];  // (k_write_windows, the only kernel behind this one, sets no status)
    scan_mask(i0, n, ve);
    scan_mask(i0, n, vs);
    const int64_t n_out = scan_pass(ve, a.site_outoff, i0, n, 0, pe, s_w[0]);
    const int64_t n_str = scan_pass(vs, a.site_stroff, i0, n, 0, ps, s_w[1]);
    if (threadIdx.x == 0 && n <= c0 + SCAN_PASS) {
        a.diag[D_NOUT] = n_out;
        a.diag[D_STRBYTES] = n_str;
        a.d_counts[0] = n_out;
        a.d_counts[1] = n_str;
        a.d_counts[2] = status;
        a.d_counts[3] = n_sites;
    }
}

// single-block exclusive scan of n int32 values; total -> *total_out (int64) (polisher pipeline)
__global__ __launch_bounds__(1024) void k_scan_i32(const int32_t* in, int32_t* out, int64_t n_fixed,
                                                   const int64_t* n_ptr, int64_t n_cap, int64_t* total_out) {
    __shared__ int64_t s_w[32];
    int64_t n = n_ptr ? *n_ptr : n_fixed;
    if (n > n_cap) n = n_cap;
    const int64_t total = block_excl_scan<int32_t>(in, out, n, s_w);
    if (threadIdx.x == 0 && total_out) *total_out = total;
}

__global__ __launch_bounds__(1024) void k_site_rank(SumArgs a) {
    __shared__ int32_t s_w[16], s_p[16];
    const int64_t col = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int f = col < a.n_cols ? a.flags[col] : 0;
    // sites before this block = the per-tile counts (k_pileup_tiles) of the tiles before it, added up here (at most 3 k values,
    // three coalesced loads per thread) instead of by a scan kernel of its own in front of this one
    const int64_t tiles_before = (int64_t)blockIdx.x * (1024 / TILE_COLS);
    int part = 0;
    for (int64_t i = threadIdx.x; i < tiles_before; i += 1024) part += a.blk_cnt[i];
    const int pinc = wave_incl_scan32(part, lane);
    const int site = f & 1;
    const unsigned long long m = __ballot(site);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wv] = __popcll(m);
    if (lane == 63) s_p[wv] = pinc;
    __syncthreads();
    int woff = 0, own = 0, base = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        woff += k < wv ? s_w[k] : 0;
        own += s_w[k];
        base += s_p[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.diag[D_NSITES] = (int64_t)base + own;   // the last block knows the total
    const int32_t rank = base + woff + before;
    if (site && rank < a.max_sites) {
        int g = a.tile_g0[col / TILE_COLS];   // region of the tile's first column (k_init), then forwards: two dependent loads, not five
        while (g + 1 <= a.in.n_regions && a.in.ref_off[g + 1] <= col) g++;
        a.site_col[rank] = (int32_t)col;
        a.site_region[rank] = g;
        // events a site will receive: every insert / delete observation, and either the rare SNP observations (the
        // common ones are read off the symbol planes) or, in the haplotag form, every SNP observation
        const cnt_t* cc = a.cnt + (int64_t)col * (a.hp ? CNT_STRIDE_HP : CNT_STRIDE);   // this column's counters
        const int n_base = cc[a.hp ? C_SNP : C_RARE];
        const int nev = cc[C_INS] + cc[C_DEL] + n_base;
        a.site_nev[rank] = nev;
        a.site_fill[rank] = 0;
        SiteHdr h;
        h.col = (int32_t)col; h.col_base = (int32_t)a.in.ref_off[g]; h.g = g;
        h.ref_start = a.in.ref_start[g];
        h.R = (int32_t)(a.in.ref_end[g] - h.ref_start + 1);
        const int64_t t = col / TILE_COLS;
        h.p0 = a.tile_off[t]; h.np = a.tile_cnt[t];
        h.cov = cc[C_COV];
        h.flags = (int32_t)a.in.ref[col] | (n_base != 0 ? 256 : 0) | (f << 16);
        h.nev = nev; h.pad = 0;
        a.site_hdr[rank] = h;
        // the few sites whose events exceed the small allele table are listed for the large-table launch of k_site_alleles
        if (nev + 4 > UM_SMALL) a.big_sites[atomicAdd((unsigned long long*)&a.diag[D_NBIG], 1ull)] = rank;
    }
}

// ---- K5 -------------------------------------------------------------------------------------------
// Site-indexed kernels walk the site list XCD by XCD: workgroup b runs on XCD b & 7 (round-robin dispatch), so giving
// each XCD one contiguous eighth of the sites, in order, keeps the sites of a tile - which read the same pair records,
// the same op ranges (the upper probes of their binary searches are the same words) and neighbouring counter columns -
// in ONE 4 MB L2 at about the same time instead of fetching them into up to eight. Grids are multiples of 8.
__device__ __forceinline__ int64_t xcd_chunk(int64_t n_sites) { return (n_sites + 7) >> 3; }
__device__ __forceinline__ int64_t xcd_site(int64_t j, int64_t n_sites) { return (int64_t)(blockIdx.x & 7) * xcd_chunk(n_sites) + j; }

// (nev, evoff: the site's bucket size and offset, read once per site by the caller)
__device__ __forceinline__ void push_event(const SumArgs& a, int32_t s, int32_t nev, int64_t evoff, int64_t src, int32_t len,
                                           int type, bool rev, int kind, int flags) {
    const int32_t slot = atomicAdd(&a.site_fill[s], 1);
    if (slot >= nev) { set_status(a.diag, PV_ERR_INVALID); return; }  // cannot happen: exact bucket sizes
    Event e;
    e.src = src; e.len = len; e.type = (uint8_t)type; e.rev = rev ? 1 : 0; e.kind = (uint8_t)kind; e.flags = (uint8_t)flags;
    a.ev[evoff + slot] = e;
}

// One WAVE per SITE, one lane per (read, tile) pair of the site's tile: the allele observations a read contributes at that
// column. Sites are ~1 column in 200, so walking the CIGAR stream a second time (a workgroup per read, five loads per op,
// 20 M ops per launch) spent nearly all its loads on ops that touch no site; here a lane finds the read's ops at the site
// column with one binary search over the pair's op range (start columns are sorted), 8 k sites x ~70 reads x 7 steps.
//   ops that START right behind the column and are inserts / deletes anchor on it (counted by k_pileup_tiles: insert_counts());
//   the aligned op that contains the column gives the read's base there (rare observations only, or - haplotag form - every
//   mismatch).
constexpr int KC_WAVES = 2;  // waves per site: the benchmark's tiles hold ~70 pairs, which one wave would walk as two trips in a row
__global__ __launch_bounds__(64 * KC_WAVES) void k_collect(SumArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (a.diag[D_STATUS] != 0) return;
    int64_t n_sites = a.diag[D_NSITES];
    if (n_sites > a.max_sites) n_sites = a.max_sites;
    for (int64_t sj = blockIdx.x >> 3; sj < xcd_chunk(n_sites); sj += gridDim.x >> 3) {
        const int64_t s = xcd_site(sj, n_sites);
        if (s >= n_sites) break;
        const SiteHdr h = a.site_hdr[s];
        const int64_t evoff = a.site_evoff[s];
        const int64_t col = h.col;
        const int64_t col_base = h.col_base;
        const int32_t col_rel = (int32_t)(col - col_base);
        const bool need_base = (h.flags & 256) != 0;
        const int refb = h.flags & 0xFF;
        const int sub_k = (int)(col & (TILE_COLS - 1)) / SUB_COLS;
        const int32_t p0 = h.p0, np = h.np;
        for (int32_t pb = 64 * wv; pb < np; pb += 64 * KC_WAVES) {
            if (pb + lane >= np) continue;
            const PairRec pr = a.pairs[p0 + pb + lane];
            if (pr.col_base != (int32_t)col_base) continue;   // a tile can hold the end of one region and the start of the next
            const bool rev = (pr.rev & 1) != 0;
            int obs = 1;  // flags of an allele observation
            if (a.hp) obs |= ((pr.rev >> 1) & 3) << 2;         // count sets of the read (k_tile_fill)
            // first op of the pair's range that starts behind the column; the pair's sub-tile index narrows the range to the ops
            // between the 64-column boundaries around the site (every op before the first starts before the boundary at or before
            // the column, no op from the second on starts at or before the column)
            int32_t lo = pr.op_lo, hi = pr.op_hi;
            if (pr.sub(SUB_N + 1)) {
                lo += pr.sub(sub_k);
                const int s1 = pr.sub(sub_k + 1);
                if (s1 < 255 && pr.op_lo + s1 < hi) hi = pr.op_lo + s1;
            }
            while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (a.op_ref[mid] <= col_rel) lo = mid + 1; else hi = mid; }
            const int32_t f = lo;
            for (int32_t o = f; o < pr.op_hi; o++) {            // inserts / deletes anchored on the column
                if (a.op_ref[o] != col_rel + 1) break;
                const uint32_t w = a.in.cigar[o];
                const int op = w & 0xF;
                const int32_t len = (int32_t)(w >> 4);
                if (op == PV_CIGAR_IN) {   // the conditions under which k_pileup_tiles counted it (INS plane, insert_count)
                    const int32_t rdv = a.op_rd[o];
                    const int64_t ins_start = pr.base0 + rdv - 1;
                    if (rdv >= 1 && ins_start + (int64_t)len + 1 <= pr.seq_end && insert_counts(a, ins_start, len, a.hp != 0))
                        push_event(a, (int32_t)s, h.nev, evoff, ins_start, len + 1, 2, rev, 1, obs);
                } else if (op == PV_CIGAR_DEL) {
                    int64_t L = (int64_t)len + 1;
                    if ((int64_t)col_rel + L > pr.ref_len) L = pr.ref_len - col_rel;
                    if (1 + L <= PV_MAX_ALLELE_KEY) push_event(a, (int32_t)s, h.nev, evoff, col, (int32_t)L, 3, rev, 2, obs);
                }
            }
            if (!need_base) continue;
            for (int32_t o = f - 1; o >= pr.op_lo; o--) {       // the op that holds the column, skipping ops that consume no reference
                const uint32_t w = a.in.cigar[o];
                const int op = w & 0xF;
                const bool aligned = op == PV_CIGAR_MATCH || op == PV_CIGAR_EQUAL || op == PV_CIGAR_DIFF;
                if (!aligned) {
                    if (op == PV_CIGAR_DEL || op == PV_CIGAR_REF_SKIP || op == PV_CIGAR_PAD) break;   // the column lies in a gap of this read
                    continue;
                }
                const int32_t rr = a.op_ref[o];
                const int64_t i = (int64_t)col_rel - rr;
                if (rr == OP_INACTIVE || i < 0 || i >= (int64_t)(w >> 4)) break;
                const int64_t bi = pr.base0 + a.op_rd[o] + i;
                if (bi >= pr.seq_end) break;  // already reported by k_pileup
                const int base = a.in.bases[bi];
                if (!((double)a.in.quals[bi] >= a.p.min_snp_baseq)) break;
                if (a.hp) {  // every mismatch (raw bytes, region_summary_hp.cpp:406) is an allele observation
                    if (refb != base) push_event(a, (int32_t)s, h.nev, evoff, bi, 1, 1, rev, 1, obs);
                    break;
                }
                const bool refvalid = is_acgt(up(refb));
                const bool rare = (refb != base) && !(refvalid && is_acgt(base));
                const bool corr = refvalid && base != up(base) && is_acgt(up(base));
                if (rare || corr) push_event(a, (int32_t)s, h.nev, evoff, bi, 1, 1, rev, 1, (rare ? 1 : 0) | (corr ? 2 : 0));
                break;
            }
        }
    }
}

// ---- K6 -------------------------------------------------------------------------------------------
struct Key {
    int64_t src;
    int32_t len;
    uint8_t type, kind, imm;
};
__device__ __forceinline__ int key_byte(const SumArgs& a, const Key& k, int i) {
    return k.kind == 0 ? k.imm : (k.kind == 1 ? a.in.bases[k.src + i] : a.in.ref[k.src + i]);
}
// the first (up to) 8 bytes of a key, big-endian and zero-padded, so that integer order is byte order: fetched ONCE per
// allele; nearly every comparison (SNP keys are one byte, most indels a few) is then decided in registers / LDS instead of
// with dependent byte loads from the bases / reference
__device__ __forceinline__ uint64_t key_prefix(const SumArgs& a, const Key& k) {
    uint64_t p = 0;
    const int n = k.len < 8 ? k.len : 8;
    for (int i = 0; i < n; i++) p |= (uint64_t)(uint8_t)key_byte(a, k, i) << (56 - 8 * i);
    return p;
}
// std::string compare of "<type digit><bytes>" given the prefixes: equal prefixes mean the first min(len, 8) bytes agree
// (where a zero byte meets padding the shorter key is a prefix of the longer, which the length rule orders the same way)
__device__ __forceinline__ int key_cmp(const SumArgs& a, const Key& x, uint64_t px, const Key& y, uint64_t py) {
    if (x.type != y.type) return x.type < y.type ? -1 : 1;
    if (px != py) return px < py ? -1 : 1;
    const int m = x.len < y.len ? x.len : y.len;
    for (int i = 8; i < m; i++) {
        const int bx = key_byte(a, x, i), by = key_byte(a, y, i);
        if (bx != by) return bx < by ? -1 : 1;
    }
    if (x.len != y.len) return x.len < y.len ? -1 : 1;
    return 0;
}

// HP: the haplotag form keeps four per-strand counts per allele (forward / reverse x haplotype set 1 / 2,
// region_summary_hp.cpp:415-447) next to the total, and no allele count comes from the planes.
// UM = alleles the LDS table of a wave holds. The table is what limits the waves per CU (1024 entries are 34 KB: four waves
// per CU, one per SIMD, and a site is a chain of dependent loads), while a site can never hold more distinct alleles than it
// has events + 4: sites with few events (all but the deepest) run in the instantiation with a UM_SMALL-entry table, BIG = the
// others.
template <bool HP, int UM, bool BIG>
__global__ __launch_bounds__(64) void k_site_alleles(SumArgs a) {
    __shared__ int64_t u_src[UM];
    __shared__ uint64_t u_pre[UM];  // key_prefix of the allele
    __shared__ int32_t u_len[UM];
    __shared__ int32_t u_fwd[UM];   // HP: total observations
    __shared__ int32_t u_rev[UM];   // HP: unused (0), so that u_fwd + u_rev is the total in both forms
    __shared__ int32_t u_hc[HP ? 4 : 1][HP ? UM : 1];  // HP: forward set 1, forward set 2, reverse set 1, reverse set 2
    __shared__ uint8_t u_type[UM];
    __shared__ uint8_t u_kind[UM];
    __shared__ uint8_t u_imm[UM];
    __shared__ uint8_t u_ok[UM];
    __shared__ int16_t u_order[UM];
    __shared__ int32_t s_nU;
    const int lane = threadIdx.x;
    if (a.diag[D_STATUS] != 0) return;
    int64_t n_sites = a.diag[D_NSITES];
    if (n_sites > a.max_sites) n_sites = a.max_sites;
    const int64_t n_big = BIG ? a.diag[D_NBIG] : 0;
    for (int64_t sj = BIG ? blockIdx.x : blockIdx.x >> 3; sj < (BIG ? n_big : xcd_chunk(n_sites)); sj += BIG ? gridDim.x : gridDim.x >> 3) {
        const int64_t s = BIG ? a.big_sites[sj] : xcd_site(sj, n_sites);
        if (s >= n_sites) { if (BIG) continue; else break; }
        const SiteHdr h = a.site_hdr[s];
        const int64_t eoff = a.site_evoff[s];          // (requested together with the header)
        if ((h.nev + 4 > UM_SMALL) != BIG) continue;   // the other instantiation's site
        const int64_t col = h.col;
        const int f = (h.flags >> 16) & 0xFF;
        const int cov = h.cov;
        const int depth = cov < PV_MAX_COLOR ? cov : PV_MAX_COLOR;  // :682
        const int refraw = h.flags & 0xFF;
        const bool refvalid = is_acgt(up(refraw));
        __syncthreads();
        // slots 0..3: SNP alleles whose counts are the (negated, un-clamped) A/C/G/T planes
        if (!HP && lane < 4) {
            const int b = "ACGT"[lane];
            u_src[lane] = 0; u_len[lane] = 1; u_type[lane] = 1; u_kind[lane] = 0; u_imm[lane] = (uint8_t)b;
            u_pre[lane] = (uint64_t)(uint8_t)b << 56;
            const bool ok = refvalid && b != refraw;
            u_ok[lane] = ok;
            u_fwd[lane] = ok ? -a.cnt[(int64_t)col * CNT_STRIDE + C_PLANE + 1 + lane] : 0;
            u_rev[lane] = ok ? -a.cnt[(int64_t)col * CNT_STRIDE + C_PLANE + 8 + 1 + lane] : 0;
        }
        if (lane == 0) s_nU = HP ? 0 : 4;
        __syncthreads();
        const int nev = h.nev;
        for (int eb = 0; eb < nev; eb += 64) {
            const bool have = eb + lane < nev;
            Event e;
            e.src = 0; e.len = 0; e.type = 0; e.rev = 0; e.kind = 1; e.flags = 0;
            if (have) e = a.ev[eoff + eb + lane];
            if (!HP && have && (e.flags & 2)) {  // lower-case acgt was counted in plane toupper(): take it back out
                const int ub = up(a.in.bases[e.src]);
                const int sl = ub == 'A' ? 0 : ub == 'C' ? 1 : ub == 'G' ? 2 : 3;
                if (u_ok[sl]) atomicAdd(e.rev ? &u_rev[sl] : &u_fwd[sl], -1);
            }
            bool pending = have && (e.flags & 1);
            Key ke; ke.src = e.src; ke.len = e.len; ke.type = e.type; ke.kind = e.kind; ke.imm = 0;
            const uint64_t pe = pending ? key_prefix(a, ke) : 0;
            int checked = HP ? 0 : 4;  // slots 0..3 can never equal an event key (see k_pileup: those are not events)
            [[maybe_unused]] const int hs = (e.flags >> 2) & 3, hst = e.rev ? 2 : 0;
            while (true) {
                const int nU = s_nU;
                if (pending) {
                    for (int k = checked; k < nU; k++) {
                        Key ku; ku.src = u_src[k]; ku.len = u_len[k]; ku.type = u_type[k]; ku.kind = u_kind[k]; ku.imm = u_imm[k];
                        if (ku.type == ke.type && ku.len == ke.len && u_pre[k] == pe && key_cmp(a, ku, pe, ke, pe) == 0) {
                            if constexpr (HP) {
                                atomicAdd(&u_fwd[k], 1);
                                if (hs & 1) atomicAdd(&u_hc[hst + 0][k], 1);
                                if (hs & 2) atomicAdd(&u_hc[hst + 1][k], 1);
                            } else
                            atomicAdd(e.rev ? &u_rev[k] : &u_fwd[k], 1);
                            pending = false;
                            break;
                        }
                    }
                }
                checked = nU;
                const unsigned long long m = __ballot(pending);
                if (m == 0) break;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) {
                    if (nU < UM) {
                        u_src[nU] = ke.src; u_len[nU] = ke.len; u_type[nU] = ke.type; u_kind[nU] = ke.kind; u_imm[nU] = 0;
                        u_pre[nU] = pe;
                        if constexpr (HP) {
                            u_ok[nU] = 1; u_fwd[nU] = 1; u_rev[nU] = 0;
                            u_hc[0][nU] = (!e.rev && (hs & 1)) ? 1 : 0; u_hc[1][nU] = (!e.rev && (hs & 2)) ? 1 : 0;
                            u_hc[2][nU] = (e.rev && (hs & 1)) ? 1 : 0;  u_hc[3][nU] = (e.rev && (hs & 2)) ? 1 : 0;
                        } else {
                        u_ok[nU] = 1; u_fwd[nU] = e.rev ? 0 : 1; u_rev[nU] = e.rev ? 1 : 0;
                        }
                        s_nU = nU + 1;
                    } else {
                        set_status(a.diag, PV_ERR_LIMIT);
                    }
                    pending = false;
                }
                __syncthreads();
            }
            __syncthreads();
        }
        __syncthreads();
        const int nU = s_nU;
        // order like std::set<std::string> (:670): rank among the observed alleles
        int nV = 0;
        for (int kb = 0; kb < nU; kb += 64) {
            const int k = kb + lane;
            const bool live = k < nU && u_ok[k] && (u_fwd[k] + u_rev[k]) > 0;
            if (live) {
                Key kk; kk.src = u_src[k]; kk.len = u_len[k]; kk.type = u_type[k]; kk.kind = u_kind[k]; kk.imm = u_imm[k];
                const uint64_t pk = u_pre[k];
                int rank = 0;
                for (int j = 0; j < nU; j++) {
                    if (j == k || !u_ok[j] || (u_fwd[j] + u_rev[j]) <= 0) continue;
                    Key kj; kj.src = u_src[j]; kj.len = u_len[j]; kj.type = u_type[j]; kj.kind = u_kind[j]; kj.imm = u_imm[j];
                    if (key_cmp(a, kj, u_pre[j], kk, pk) < 0) rank++;
                }
                u_order[rank] = (int16_t)k;
            }
            nV += __popcll(__ballot(live));
        }
        __syncthreads();
        // filters (:682-712) in set order; survivors become allele records
        const int64_t recbase = eoff + 4 * s;
        int nemit = 0;
        int64_t sbytes = 0;
        for (int rb = 0; rb < nV; rb += 64) {
            const int r = rb + lane;
            bool keep = false;
            int k = 0;
            if (r < nV) {
                k = u_order[r];
                const int total = u_fwd[k] + u_rev[k];
                const int t = u_type[k];
                const double dd = (double)depth > 1.0 ? (double)depth : 1.0;
                const double freq = (double)total / dd;
                keep = true;
                if ((double)total < a.p.candidate_support_threshold) keep = false;
                if (t != 1 && freq < a.p.indel_candidate_freq_threshold) keep = false;
                if (t == 1 && freq < a.p.snp_candidate_freq_threshold) keep = false;
                if (t != 1 && a.p.skip_indels) keep = false;
                if ((t == 1 && !(f & 2)) || (t == 2 && !(f & 4)) || (t == 3 && !(f & 8))) keep = false;
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int e = nemit + __popcll(m & ((1ull << lane) - 1ull));
                AlleleRec rc;
                rc.src = u_src[k]; rc.len = u_len[k]; rc.total = u_fwd[k] + u_rev[k]; rc.fwd = u_fwd[k]; rc.rev = u_rev[k];
                if constexpr (HP) {  // the four overlay values, already clamped (region_summary_hp.cpp:971-974)
                    auto c8 = [](int v) { return (uint32_t)(v < PV_MAX_COLOR ? v : PV_MAX_COLOR); };
                    rc.fwd = (int32_t)(c8(u_hc[0][k]) | (c8(u_hc[1][k]) << 8) | (c8(u_hc[2][k]) << 16) | (c8(u_hc[3][k]) << 24));
                    rc.rev = 0;
                }
                rc.type = u_type[k]; rc.kind = u_kind[k]; rc.imm = u_imm[k]; rc.pad = 0; rc.pad2 = 0;
                a.rec[recbase + e] = rc;
            }
            nemit += __popcll(m);
            int64_t b = keep ? 1 + u_len[k] : 0;
            for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d, 64);
            sbytes += b;
        }
        if (lane == 0) {
            a.site_nemit[s] = nemit;
            a.site_strbytes[s] = sbytes;
        }
    }
}

// ---- K8 -------------------------------------------------------------------------------------------
// a window is a chain of gathers per lane: WW_THREADS lanes share one site's windows, so that a lane walks 4 elements of a
// window instead of 14 and four times as many chains are in flight per CU
constexpr int WW_THREADS = 256;
__global__ __launch_bounds__(WW_THREADS) void k_write_windows(SumArgs a) {
    __shared__ int32_t s_win[PV_WINDOW_BYTES];
    const int lane = threadIdx.x;
    if (a.diag[D_STATUS] != 0) return;
    int64_t n_sites = a.diag[D_NSITES];
    if (n_sites > a.max_sites) n_sites = a.max_sites;
    for (int64_t sj = blockIdx.x >> 3; sj < xcd_chunk(n_sites); sj += gridDim.x >> 3) {
        const int64_t s = xcd_site(sj, n_sites);
        if (s >= n_sites) break;
        // (everything a site's windows start from is requested at once, also for the two sites in three that emit nothing:
        // one round trip instead of two for those that do)
        const int nemit = a.site_nemit[s];
        const SiteHdr h = a.site_hdr[s];
        const int64_t evoff_s = a.site_evoff[s], stroff_s = a.site_stroff[s], outoff_s = a.site_outoff[s];
        if (nemit == 0) continue;
        const int64_t col = h.col;
        const int g = h.g;
        const int64_t col_base = h.col_base;
        const int64_t R = h.R;
        const int64_t ci = col - col_base;
        const int cov = h.cov;
        const int depth = cov < PV_MAX_COLOR ? cov : PV_MAX_COLOR;
        const int refraw = h.flags & 0xFF;
        const bool refvalid = is_acgt(up(refraw));
        const int64_t recbase = evoff_s + 4 * s;
        int64_t so = stroff_s;
        for (int e = 0; e < nemit; e++) {
            const AlleleRec rc = a.rec[recbase + e];
            const int64_t k = outoff_s + e;
            const int64_t send = so + 1 + rc.len;
            if (k < a.out.capacity && send <= a.out.str_capacity) {
                const int cfwd = rc.fwd < PV_MAX_COLOR ? rc.fwd : PV_MAX_COLOR;
                const int crev = rc.rev < PV_MAX_COLOR ? rc.rev : PV_MAX_COLOR;
                const int clen = rc.len < PV_MAX_COLOR ? rc.len : PV_MAX_COLOR;
                int alt = 0, ff = -1, fr = -1;
                if (rc.type == 1) {
                    alt = rc.kind == 0 ? rc.imm : a.in.bases[rc.src];
                    if (refvalid) { ff = 7 + sym_of(alt); fr = 18 + sym_of(alt); }
                } else if (rc.type == 2) {
                    if (refvalid) { ff = 12; fr = 23; }
                } else {
                    if (refvalid) { ff = 13; fr = 24; }
                }
                int end_index = 16 + rc.len - 1;  // :885
                if (end_index > 31) end_index = 31;
                // gather with the ROW running fastest across lanes: the counters are plane-major, so the lanes of a wave then read
                // two or three planes x 33 consecutive columns (a few lines) instead of one value from each of 26 planes 6 MB
                // apart; the finished window goes through LDS and leaves in its own (row, feature) order, coalesced
                constexpr int WW_TRIPS = (PV_WINDOW_BYTES + WW_THREADS - 1) / WW_THREADS;
                int raw[WW_TRIPS];
#pragma unroll
                for (int u = 0; u < WW_TRIPS; u++) {   // every load of the window is requested before the first is used
                    const int t = lane + u * WW_THREADS;
                    const int pl = t / PV_WINDOW_ROWS, row = t - pl * PV_WINDOW_ROWS;
                    const int64_t i = ci - 16 + row;
                    const bool in = t < PV_WINDOW_BYTES && i >= 0 && i < R;  // row R of the reference's matrix exists and is all zero (:835)
                    const int64_t c2 = col_base + (in ? i : 0);
                    // feature -> counter plane: REF count 4 / 15, symbol planes 8..14 / 19..25; the rest is written per candidate
                    const int plane = pl == 4 ? C_PLANE : (pl >= 8 && pl <= 14) ? C_PLANE + 1 + (pl - 8)
                                    : pl == 15 ? C_PLANE + 8 : pl >= 19 ? C_PLANE + 8 + 1 + (pl - 19) : -1;
                    int v = 0;
                    if (in && plane >= 0) v = a.cnt[c2 * CNT_STRIDE + plane];
                    if (in && pl == 0) v = a.in.ref[c2];
                    raw[u] = v;
                }
#pragma unroll
                for (int u = 0; u < WW_TRIPS; u++) {
                    const int t = lane + u * WW_THREADS;
                    if (t >= PV_WINDOW_BYTES) continue;
                    const int pl = t / PV_WINDOW_ROWS, row = t - pl * PV_WINDOW_ROWS;
                    const int64_t i = ci - 16 + row;
                    int v = raw[u];
                    if (i >= 0 && i < R) {
                        if (pl == 0) v = refcode(v);
                        if (pl >= 11 && pl <= 24) v = v > PV_MAX_COLOR ? PV_MAX_COLOR : (v < -PV_MAX_COLOR ? -PV_MAX_COLOR : v);  // :648-653
                    }
                    if (row == 16) {  // :848-894
                        if (rc.type == 1) {
                            if (pl == 1) v = refcode(alt);
                            if (pl == 5) v = cfwd;
                            if (pl == 16) v = crev;
                        } else if (rc.type == 2) {
                            if (pl == 2) v = clen;
                            if (pl == 6) v = cfwd;
                            if (pl == 17) v = crev;
                        } else {
                            if (pl == 3) v = clen;
                            if (pl == 7) v = cfwd;
                            if (pl == 18) v = crev;
                        }
                        if (pl == ff || pl == fr) v = -v;
                    } else if (rc.type == 3 && row > 16 && row <= end_index) {  // :895-904
                        if (pl == 3) v = clen;
                        if (pl == 7) v = cfwd;
                        if (pl == 18) v = crev;
                        if (refvalid && (pl == 14 || pl == 25)) v = -v;
                    }
                    s_win[row * PV_FEATURES + pl] = v;
                }
                __syncthreads();
                for (int el = lane; el < PV_WINDOW_BYTES; el += WW_THREADS) {
                    const int v = s_win[el];
                    a.out.images[k * PV_WINDOW_BYTES + el] = (int8_t)(uint8_t)(v & 0xFF);  // DataStore.py:68 wrap
                    if (a.out.images_i32) a.out.images_i32[k * PV_WINDOW_BYTES + el] = v;
                }
                __syncthreads();
                if (lane == 0) {
                    a.out.region[k] = g;
                    a.out.position[k] = h.ref_start + ci;
                    a.out.depth[k] = (uint8_t)depth;
                    a.out.cand_freq[k] = (uint8_t)(rc.total < PV_MAX_COLOR ? rc.total : PV_MAX_COLOR);
                    a.out.cand_off[k] = so;
                    a.out.cand_off[k + 1] = send;
                    a.out.cand_str[so] = (char)('0' + rc.type);
                }
                for (int i = lane; i < rc.len; i += WW_THREADS) {
                    const int b = rc.kind == 0 ? rc.imm : (rc.kind == 1 ? a.in.bases[rc.src + i] : a.in.ref[rc.src + i]);
                    a.out.cand_str[so + 1 + i] = (char)b;
                }
            }
            so = send;
        }
    }
}

// Haplotag form of K8 (region_summary_hp.cpp:943-1003): 21 rows x 48 planes around the site, every plane clamped, the
// five overlay values of the candidate on the middle row; no deletion tail, no sign flips.
__global__ __launch_bounds__(WW_THREADS) void k_write_windows_hp(SumArgs a) {
    __shared__ int32_t s_win[PV_HP_WINDOW_BYTES];
    const int lane = threadIdx.x;
    if (a.diag[D_STATUS] != 0) return;
    int64_t n_sites = a.diag[D_NSITES];
    if (n_sites > a.max_sites) n_sites = a.max_sites;
    constexpr int MID = (PV_HP_WINDOW_ROWS - 1) / 2;
    for (int64_t sj = blockIdx.x >> 3; sj < xcd_chunk(n_sites); sj += gridDim.x >> 3) {
        const int64_t s = xcd_site(sj, n_sites);
        if (s >= n_sites) break;
        // (everything a site's windows start from is requested at once, also for the two sites in three that emit nothing:
        // one round trip instead of two for those that do)
        const int nemit = a.site_nemit[s];
        const SiteHdr h = a.site_hdr[s];
        const int64_t evoff_s = a.site_evoff[s], stroff_s = a.site_stroff[s], outoff_s = a.site_outoff[s];
        if (nemit == 0) continue;
        const int64_t col = h.col;
        const int g = h.g;
        const int64_t col_base = h.col_base;
        const int64_t R = h.R;
        const int64_t ci = col - col_base;
        const int cov = h.cov;
        const int depth = cov < PV_MAX_COLOR ? cov : PV_MAX_COLOR;
        const int64_t recbase = evoff_s + 4 * s;
        int64_t so = stroff_s;
        for (int e = 0; e < nemit; e++) {
            const AlleleRec rc = a.rec[recbase + e];
            const int64_t k = outoff_s + e;
            const int64_t send = so + 1 + rc.len;
            if (k < a.out.capacity && send <= a.out.str_capacity) {
                const int t = rc.type;  // 1 SNP, 2 INS, 3 DEL
                const int v1 = t == 1 ? refcode(a.in.bases[rc.src]) : (rc.len < PV_MAX_COLOR ? rc.len : PV_MAX_COLOR);
                const uint32_t hc = (uint32_t)rc.fwd;  // forward set 1, forward set 2, reverse set 1, reverse set 2
                // row fastest across lanes, every load requested before the first is used, window out through LDS (see
                // k_write_windows)
                constexpr int WW_TRIPS = (PV_HP_WINDOW_BYTES + WW_THREADS - 1) / WW_THREADS;
                int raw[WW_TRIPS];
#pragma unroll
                for (int u = 0; u < WW_TRIPS; u++) {
                    const int tt = lane + u * WW_THREADS;
                    const int pl = tt / PV_HP_WINDOW_ROWS, row = tt - pl * PV_HP_WINDOW_ROWS;
                    const int64_t i = ci - MID + row;
                    const bool in = tt < PV_HP_WINDOW_BYTES && i >= 0 && i < R;  // row R of the reference's matrix exists and is all zero
                    const int64_t c2 = col_base + (in ? i : 0);
                    const int grp = (pl - 4) / 11, w = (pl - 4) - 11 * grp;  // 0 REF count, 1-3 overlays, 4-10 symbols
                    const int plane = pl < 4 ? -1 : (w == 0 ? HC_PLANE + 8 * grp : (w >= 4 ? HC_PLANE + 8 * grp + (w - 3) : -1));
                    int v = 0;
                    if (in && plane >= 0) v = a.cnt[c2 * CNT_STRIDE_HP + plane];
                    if (in && pl == 0) v = a.in.ref[c2];
                    raw[u] = v;
                }
#pragma unroll
                for (int u = 0; u < WW_TRIPS; u++) {
                    const int tt = lane + u * WW_THREADS;
                    if (tt >= PV_HP_WINDOW_BYTES) continue;
                    const int pl = tt / PV_HP_WINDOW_ROWS, row = tt - pl * PV_HP_WINDOW_ROWS;
                    const int64_t i = ci - MID + row;
                    int v = raw[u];
                    if (i >= 0 && i < R) {
                        if (pl == 0) v = refcode(v);
                        v = v > PV_MAX_COLOR ? PV_MAX_COLOR : (v < -PV_MAX_COLOR ? -PV_MAX_COLOR : v);  // :762-767
                    }
                    if (row == MID) {  // :970-974, :983-987, :996-1000
                        if (pl == t) v = v1;
                        if (pl == 4 + t) v = (int)(hc & 0xFF);
                        if (pl == 26 + t) v = (int)((hc >> 8) & 0xFF);
                        if (pl == 15 + t) v = (int)((hc >> 16) & 0xFF);
                        if (pl == 37 + t) v = (int)((hc >> 24) & 0xFF);
                    }
                    s_win[row * PV_HP_FEATURES + pl] = v;
                }
                __syncthreads();
                for (int el = lane; el < PV_HP_WINDOW_BYTES; el += WW_THREADS) {
                    const int v = s_win[el];
                    a.out.images[k * PV_HP_WINDOW_BYTES + el] = (int8_t)(uint8_t)(v & 0xFF);
                    if (a.out.images_i32) a.out.images_i32[k * PV_HP_WINDOW_BYTES + el] = v;
                }
                __syncthreads();
                if (lane == 0) {
                    a.out.region[k] = g;
                    a.out.position[k] = h.ref_start + ci;
                    a.out.depth[k] = (uint8_t)depth;
                    a.out.cand_freq[k] = (uint8_t)(rc.total < PV_MAX_COLOR ? rc.total : PV_MAX_COLOR);
                    a.out.cand_off[k] = so;
                    a.out.cand_off[k + 1] = send;
                    a.out.cand_str[so] = (char)('0' + rc.type);
                }
                for (int i = lane; i < rc.len; i += WW_THREADS)
                    a.out.cand_str[so + 1 + i] = (char)(rc.kind == 1 ? a.in.bases[rc.src + i] : a.in.ref[rc.src + i]);
            }
            so = send;
        }
    }
}

// ==== P2 (polisher) summary images ====================================================================
// SummaryGenerator::iterate_over_read / generate_image (pepper/modules/src/pileup_summary/summary_generator.cpp:47-121,
// 274-304). Same tile-owner scheme as k_pileup_tiles (pairs -> ops -> bases, counters of a 512-column tile in LDS), with
// the polisher's much simpler per-base rule: one ds_add into one of ten (symbol, strand) planes, no qualities.
constexpr int PC_COV = 10, PC_LONG = 11, PC_N = 12;  // global planes: 0-9 features, coverage, longest insert
enum { Q_F = 0, Q_STAR = 10 /* [rev, fwd] deleted columns */, Q_DCOV = 12, Q_LONG = 13, Q_N = 14 };

// get_feature_index (summary_generator.cpp:16-33): toupper, then reverse A0 C1 G2 T3 else 8, forward A4 C5 G6 T7 else 9
__device__ __forceinline__ int polish_sym(int c) { c = up(c); return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
__device__ __forceinline__ int polish_feature(int sym, bool rev) { return sym < 4 ? (rev ? sym : 4 + sym) : (rev ? 8 : 9); }

__global__ __launch_bounds__(PT_THREADS) void k_polish_tiles(SumArgs a) {
    __shared__ int32_t s_cnt[Q_N][TILE_COLS];
    __shared__ uint8_t s_lut[256];           // polish_sym of every byte
    __shared__ uint16_t s_blk[PT_THREADS * (TILE_COLS + 4) / 64 + 2];
    __shared__ int32_t s_pref[PT_THREADS];   // inclusive prefix of in-tile aligned bases, every op padded to whole groups of 4
    __shared__ int32_t s_iend[PT_THREADS];
    __shared__ int32_t s_col0[PT_THREADS];
    __shared__ int64_t s_base[PT_THREADS];
    __shared__ int32_t s_i0[PT_THREADS];
    __shared__ uint8_t s_opfl[PT_THREADS];   // bit0 rev
    __shared__ uint8_t s_opair[PT_THREADS];
    __shared__ int32_t p_off[PT_PB + 1];
    __shared__ int32_t p_oplo[PT_PB], p_colbase[PT_PB], p_R[PT_PB], p_rev[PT_PB];
    __shared__ int64_t p_base0[PT_PB], p_seqend[PT_PB];
    __shared__ int32_t s_wsum[2 * (PT_THREADS / 64)];
    int scan_turn = 0;
    const int tid = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int64_t tlo = tile * TILE_COLS, thi = tlo + TILE_COLS - 1;
    for (int i = tid; i < Q_N * TILE_COLS; i += PT_THREADS) (&s_cnt[0][0])[i] = 0;
    if (tid < 256) s_lut[tid] = (uint8_t)polish_sym(tid);
    const int32_t p0 = a.tile_off[tile];
    const int32_t np = a.tile_cnt[tile];
    __syncthreads();
    for (int32_t pb = 0; pb < np; pb += PT_PB) {
        const int npb = (np - pb) < PT_PB ? (np - pb) : PT_PB;
        int nops = 0;
        if (tid < npb) {
            const PairRec pr = a.pairs[p0 + pb + tid];
            nops = pr.op_hi - pr.op_lo;
            p_oplo[tid] = pr.op_lo; p_colbase[tid] = pr.col_base; p_R[tid] = pr.R;
            p_rev[tid] = pr.rev; p_base0[tid] = pr.base0; p_seqend[tid] = pr.seq_end;
        }
        const int incl_ops = block_incl_scan512(nops, s_wsum, tid, scan_turn);
        if (tid < npb) p_off[tid + 1] = incl_ops;
        if (tid == 0) p_off[0] = 0;
        __syncthreads();
        const int total_ops = p_off[npb];
        for (int ob = 0; ob < total_ops; ob += PT_THREADS) {
            const int k = ob + tid;
            int32_t ref_rel = 0, rd = 0, len = 0, op = 15, col_base = 0, R = 0;
            bool active = false, rev = false;
            int pslot = 0;
            int32_t c = 0;
            int64_t clo = 0, chi = -1;
            if (k < total_ops) {
                int lo = 0, hi = npb;
                while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p_off[mid] <= k) lo = mid; else hi = mid; }
                pslot = lo;
                c = p_oplo[pslot] + (k - p_off[pslot]);
                const int32_t rr = a.op_ref[c];
                const uint32_t w = a.in.cigar[c];
                const int32_t rdv = a.op_rd[c];
                col_base = p_colbase[pslot];
                rev = p_rev[pslot] != 0;
                R = p_R[pslot];
                active = rr != OP_INACTIVE;
                if (active) { ref_rel = rr; rd = rdv; op = w & 0xF; len = (int32_t)(w >> 4); }
                clo = tlo - col_base; chi = thi - col_base;
                if (clo < 0) clo = 0;
                if (chi > R - 1) chi = R - 1;
            }
            if (active && op == PV_CIGAR_IN) {  // :82-98; the tile that owns the anchor column takes the op
                const int64_t anchor = (int64_t)ref_rel - 1;
                if (anchor >= clo && anchor <= chi) {
                    const int lc = (int)(col_base + anchor - tlo);
                    if (p_base0[pslot] + rd + (int64_t)len > p_seqend[pslot]) {
                        set_status(a.diag, PV_ERR_INVALID);  // alt[i] past the end of the read
                    } else {
                        atomicMax(&s_cnt[Q_LONG][SW(lc)], len);
                        a.op_flag[c] = 1;  // counted per insert row by k_polish_insert once the row layout is known
                    }
                }
            } else if (active && (op == PV_CIGAR_DEL || op == PV_CIGAR_REF_SKIP || op == PV_CIGAR_PAD)) {  // :100-114
                int64_t i0 = clo - ref_rel; if (i0 < 0) i0 = 0;
                int64_t i1 = chi + 1 - ref_rel; if (i1 > len) i1 = len;
                for (int64_t i = i0; i < i1; i++)
                    atomicAdd(&s_cnt[Q_STAR + (rev ? 0 : 1)][SW((int)((int64_t)col_base + ref_rel + i - tlo))], 1);
                // "coverage[ref_position] += 1.0" sits INSIDE the loop over the deleted columns but is keyed by the
                // START of the deletion (:110): that column gains one per in-region deleted column, the others nothing.
                if ((int64_t)ref_rel >= clo && (int64_t)ref_rel <= chi) {
                    int64_t n = (int64_t)R - ref_rel; if (n > len) n = len;
                    if (n > 0) atomicAdd(&s_cnt[Q_DCOV][SW((int)(col_base + ref_rel - tlo))], (int)n);
                }
            }
            const bool is_m = active && (op == PV_CIGAR_MATCH || op == PV_CIGAR_EQUAL || op == PV_CIGAR_DIFF);
            int32_t i0 = 0, eff = 0;
            if (is_m) {
                int64_t lo = clo - ref_rel; if (lo < 0) lo = 0;
                int64_t hi = chi + 1 - ref_rel; if (hi > len) hi = len;
                if (hi > lo) { i0 = (int32_t)lo; eff = (int32_t)(hi - lo); }
            }
            const int32_t effp = (eff + 3) & ~3;  // groups of 4 slots never straddle two ops (see k_pileup_tiles)
            const int32_t incl = block_incl_scan512(effp, s_wsum, tid, scan_turn);
            s_pref[tid] = incl;
            s_col0[tid] = col_base + ref_rel;
            s_base[tid] = (k < total_ops ? p_base0[pslot] : 0) + rd;
            s_i0[tid] = i0 - (incl - effp);
            s_iend[tid] = i0 + eff;
            s_opfl[tid] = (uint8_t)(rev ? 1 : 0);
            s_opair[tid] = (uint8_t)pslot;
            for (int32_t bb = (incl - effp + 63) >> 6; (bb << 6) < incl; bb++) s_blk[bb] = (uint16_t)tid;
            __syncthreads();
            const int32_t total = s_pref[PT_THREADS - 1];
            for (int32_t jb = 0; jb < total; jb += PT_THREADS * PT_GPL * 4) {
                int lcv[PT_GPL], nvv[PT_GPL], rv[PT_GPL];
                uint32_t bw[PT_GPL];
#pragma unroll
                for (int u = 0; u < PT_GPL; u++) {
                    const int32_t j = jb + (u * PT_THREADS + tid) * 4;
                    const bool ok = j < total;
                    int owc = ok ? s_blk[j >> 6] : 0;
                    while (ok && s_pref[owc] <= j) owc++;
                    const int32_t i = j + s_i0[owc];
                    int nv = s_iend[owc] - i;
                    nv = ok ? (nv > 4 ? 4 : nv) : 0;
                    const int64_t bi = s_base[owc] + i;
                    const int64_t left = p_seqend[s_opair[owc]] - bi;
                    if (nv > 0 && nv > left) { set_status(a.diag, PV_ERR_INVALID); nv = left > 0 ? (int)left : 0; }
                    lcv[u] = (int)((int64_t)s_col0[owc] + i - tlo);
                    nvv[u] = nv;
                    rv[u] = s_opfl[owc] & 1;
                    uint32_t b4 = 0;
                    if (nv > 0) {
                        if (bi + 4 <= a.n_bases) b4 = *reinterpret_cast<const uint32_t*>(a.in.bases + bi);
                        else for (int e = 0; e < nv; e++) b4 |= (uint32_t)a.in.bases[bi + e] << (8 * e);
                    }
                    bw[u] = b4;
                }
#pragma unroll
                for (int u = 0; u < PT_GPL; u++) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (e >= nvv[u]) continue;
                        atomicAdd(&s_cnt[Q_F + polish_feature(s_lut[(bw[u] >> (8 * e)) & 0xFF], rv[u] != 0)][SW(lcv[u] + e)], 1);  // :70-74
                    }
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }
    __syncthreads();
    const int64_t NC = a.n_cols;
    int64_t ncol = NC - tlo;
    if (ncol > TILE_COLS) ncol = TILE_COLS;
    for (int lc = tid; lc < ncol; lc += PT_THREADS) {
        const int64_t g = tlo + lc;
        int cov = s_cnt[Q_DCOV][SW(lc)];
#pragma unroll
        for (int f = 0; f < 10; f++) {
            const int v = s_cnt[Q_F + f][SW(lc)];
            cov += v;  // every aligned base bumps coverage once (:72-73)
            a.pcnt[(int64_t)f * NC + g] = v + (f == 8 ? s_cnt[Q_STAR][SW(lc)] : f == 9 ? s_cnt[Q_STAR + 1][SW(lc)] : 0);
        }
        a.pcnt[(int64_t)PC_COV * NC + g] = cov;
        a.pcnt[(int64_t)PC_LONG * NC + g] = s_cnt[Q_LONG][SW(lc)];
    }
}

// insert rows per 1024-column block (columns of the reference buffer beyond R never get an insert: the tile kernel clips)
__global__ __launch_bounds__(1024) void k_polish_blk(SumArgs a) {
    __shared__ int32_t s_w[16];
    const int64_t col = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int32_t v = col < a.n_cols ? a.pcnt[(int64_t)PC_LONG * a.n_cols + col] : 0;
    const int32_t inc = wave_incl_scan32(v, lane);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    if (threadIdx.x == 0) { int32_t t = 0; for (int k = 0; k < 16; k++) t += s_w[k]; a.ins_blk[blockIdx.x] = t; }
}

__global__ __launch_bounds__(1024) void k_polish_insoff(SumArgs a) {
    __shared__ int32_t s_w[16];
    const int64_t col = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int32_t v = col < a.n_cols ? a.pcnt[(int64_t)PC_LONG * a.n_cols + col] : 0;
    const int32_t inc = wave_incl_scan32(v, lane);
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int k = 0; k < wv; k++) woff += s_w[k];
    const int32_t excl = a.ins_blkoff[blockIdx.x] + woff + inc - v;
    if (col < a.n_cols) a.ins_off[col] = excl;
    if (col == a.n_cols - 1) a.ins_off[a.n_cols] = excl + v;
}

// row / chunk layout of the regions (sequential over the few regions of a batch), limits, counters
__global__ void k_polish_regions(SumArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t rows = 0, chunks = 0;
    for (int g = 0; g < a.in.n_regions; g++) {
        a.reg_rows[g] = rows;
        a.reg_chunks[g] = chunks;
        const int64_t R = a.in.ref_end[g] - a.in.ref_start[g] + 1;
        const int64_t c0 = a.in.ref_off[g];
        const int64_t n = R + (a.ins_off[c0 + R] - a.ins_off[c0]);
        rows += n;
        // AlignmentSummarizer.chunk_images (AlignmentSummarizer.py:19-56): starts 0, L-O, 2(L-O), ... until a chunk ends at n
        chunks += n <= a.seq_len ? 1 : 1 + (n - a.seq_len + a.seq_step - 1) / a.seq_step;
    }
    a.reg_rows[a.in.n_regions] = rows;
    a.reg_chunks[a.in.n_regions] = chunks;
    a.diag[D_NROWS] = rows;
    a.diag[D_NCHUNKS] = chunks;
    if (a.diag[D_NINS] > a.max_ins_rows) set_status(a.diag, PV_ERR_LIMIT);
    a.d_counts[0] = chunks;
    a.d_counts[1] = rows;
    a.d_counts[2] = a.diag[D_STATUS];
    a.d_counts[3] = a.diag[D_NINS];
}

// thread per CIGAR op: the bases of every in-region insert, counted on its insert rows (:88-93)
__global__ __launch_bounds__(256) void k_polish_insert(SumArgs a) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n_cigar || a.diag[D_STATUS] != 0 || a.op_flag[c] != 1) return;
    const uint32_t w = a.in.cigar[c];
    const int32_t len = (int32_t)(w >> 4);
    const int32_t r = a.op_read[c];
    const int g = a.read_region[r];
    const int64_t col = a.in.ref_off[g] + a.op_ref[c] - 1;
    const int64_t row = a.ins_off[col];
    const int64_t b0 = a.in.base_off[r] + a.op_rd[c];
    const bool rev = (a.in.read_flags[r] & 1) != 0;
    for (int32_t i = 0; i < len; i++)
        atomicAdd(&a.ins_cnt[(row + i) * 10 + polish_feature(polish_sym(a.in.bases[b0 + i]), rev)], 1);
}

__device__ __forceinline__ uint8_t polish_pixel(int32_t cnt, int32_t cov) {  // generate_image, :281 / :293-294
    const double v = ((double)cnt / ((double)cov > 1.0 ? (double)cov : 1.0)) * 254.0;
    return (uint8_t)(uint32_t)(int32_t)v;
}

// thread per column: its base row and its insert rows
__global__ __launch_bounds__(256) void k_polish_image(SumArgs a) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= a.n_cols || a.diag[D_STATUS] != 0) return;
    const int g = upper_bound_i64(a.in.ref_off, a.in.n_regions + 1, col) - 1;
    if (g < 0 || g >= a.in.n_regions) return;
    const int64_t c0 = a.in.ref_off[g];
    const int64_t i = col - c0;
    if (i >= a.in.ref_end[g] - a.in.ref_start[g] + 1) return;
    const int64_t NC = a.n_cols;
    const int64_t ins0 = a.ins_off[col];
    int64_t row = a.reg_rows[g] + i + (ins0 - a.ins_off[c0]);
    const int32_t cov = a.pcnt[(int64_t)PC_COV * NC + col];
    const int32_t nl = a.pcnt[(int64_t)PC_LONG * NC + col];
    const int64_t pos = a.in.ref_start[g] + i;
    if (row < a.flat_cap) {
#pragma unroll
        for (int f = 0; f < 10; f++) a.flat_img[row * 10 + f] = polish_pixel(a.pcnt[(int64_t)f * NC + col], cov);
        a.flat_pos[row] = pos;
        a.flat_idx[row] = 0;
    }
    for (int32_t ii = 0; ii < nl; ii++) {
        row++;
        if (row >= a.flat_cap) break;
#pragma unroll
        for (int f = 0; f < 10; f++) a.flat_img[row * 10 + f] = polish_pixel(a.ins_cnt[(ins0 + ii) * 10 + f], cov);
        a.flat_pos[row] = pos;
        a.flat_idx[row] = ii + 1;
    }
}

// thread per (chunk, row): gather from the flat rows, or pad
__global__ __launch_bounds__(256) void k_polish_chunks(SumArgs a) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a.diag[D_STATUS] != 0) return;
    int64_t nck = a.diag[D_NCHUNKS];
    if (nck > a.pout.chunk_capacity) nck = a.pout.chunk_capacity;
    const int64_t k = t / a.seq_len;
    if (k >= nck) return;
    const int j = (int)(t - k * a.seq_len);
    const int g = upper_bound_i64(a.reg_chunks, a.in.n_regions + 1, k) - 1;
    const int64_t kk = k - a.reg_chunks[g];
    const int64_t rows = a.reg_rows[g + 1] - a.reg_rows[g];
    const int64_t r = kk * a.seq_step + j;
    if (j == 0) { a.pout.region[k] = g; a.pout.chunk_id[k] = (int32_t)kk; }
    uint8_t* dst = a.pout.images + t * 10;
    if (r < rows && a.reg_rows[g] + r < a.flat_cap) {
        const int64_t src = a.reg_rows[g] + r;
#pragma unroll
        for (int f = 0; f < 10; f++) dst[f] = a.flat_img[src * 10 + f];
        a.pout.position[t] = a.flat_pos[src];
        a.pout.index[t] = a.flat_idx[src];
    } else {
#pragma unroll
        for (int f = 0; f < 10; f++) dst[f] = 0;
        a.pout.position[t] = -1;
        a.pout.index[t] = -1;
    }
}

// start of a call: the diagnostics block and the per-tile pair counters / fill cursors (one launch instead of a kernel
// and two memsets)
__global__ __launch_bounds__(256) void k_init(SumArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < D_NDIAG + 8) a.diag[i] = 0;
    if (i < a.n_tiles) {
        a.tile_cnt[i] = 0;
        a.tile_fill[i] = 0;
        if (a.blk_cnt) {  // (builder pipelines; the polisher's has no site lists)
            a.blk_cnt[i] = 0;
            a.tile_g0[i] = thread_count_le(a.in.ref_off, a.in.n_regions + 1, i * TILE_COLS) - 1;
        }
    }
}

// a read reaches a column at most once, so a region's read count bounds every counter of its columns: the 16-bit planes are
// exact while it stays within MAX_REGION_READS. (Runs behind k_init, which zeroes the status words.)
__global__ __launch_bounds__(256) void k_check_depth(SumArgs a) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < a.in.n_regions && a.in.read_off[g + 1] - a.in.read_off[g] > MAX_REGION_READS) {
        a.diag[D_DEPTH] = 1;
        set_status(a.diag, PV_ERR_LIMIT);
    }
}

}  // namespace

static inline unsigned int grid_for(int64_t n, int per) { return (unsigned int)((n + per - 1) / per); }

// Workspace + launch sequence. Everything asynchronous on `st`.
static int summarize_launch(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params, int64_t n_reads,
                            int64_t n_bases, int64_t n_cigar, int64_t n_cols, int64_t max_sites, int64_t max_events,
                            int64_t max_pairs, const pv_batch_out* out, int64_t* d_counts, hipStream_t st,
                            bool hp = false, const int32_t* read_hp = nullptr) {
    if (hp)
        PV_CHECK(params->candidate_window_size == PV_HP_WINDOW_ROWS - 1 && params->feature_size == PV_HP_FEATURES,
                 PV_ERR_INVALID, "haplotag builder: candidate_window_size must be 20 and feature_size 48 (got %d, %d)",
                 params->candidate_window_size, params->feature_size);
    else
    PV_CHECK(params->candidate_window_size == 32 && params->feature_size == PV_FEATURES, PV_ERR_INVALID,
             "candidate_window_size must be 32 and feature_size 26 (got %d, %d)", params->candidate_window_size,
             params->feature_size);
    PV_CHECK(n_cols < (1ll << 31) - 2048 && n_cigar < (1ll << 31) && n_reads < (1ll << 31), PV_ERR_LIMIT,
             "batch too large for 32-bit column/op indices (cols %lld, ops %lld)", (long long)n_cols, (long long)n_cigar);
    SumArgs a;
    memset(&a, 0, sizeof(a));
    a.in = *in;
    a.p = *params;
    a.n_reads = n_reads; a.n_bases = n_bases; a.n_cigar = n_cigar; a.n_cols = n_cols;
    a.max_sites = max_sites; a.max_events = max_events;
    a.out = *out;
    a.d_counts = d_counts;
    a.hp = hp ? 1 : 0;
    a.read_hp = read_hp;
    const int64_t n_blk = (n_cols + 1023) / 1024;
    int rc;
    const int64_t nc1 = n_cigar > 0 ? n_cigar : 1, nr1 = n_reads > 0 ? n_reads : 1;
    if ((rc = pv_get(ctx, "sum.op_ref", nc1, &a.op_ref))) return rc;
    if ((rc = pv_get(ctx, "sum.op_rd", nc1, &a.op_rd))) return rc;
    if ((rc = pv_get(ctx, "sum.op_flag", nc1, &a.op_flag))) return rc;
    if ((rc = pv_get(ctx, "sum.read_region", nr1, &a.read_region))) return rc;
    a.n_tiles = (n_cols + TILE_COLS - 1) / TILE_COLS;
    {
        const double t = params->min_snp_baseq;
        a.qmin_snp = t <= 0.0 ? 0 : (t > 255.0 ? 256 : (int32_t)ceil(t));
    }
    // every read overlaps at most span/TILE_COLS + 2 tiles; the exact pair count is only known on the
    // device, so bound it: sum over reads of (read span)/TILE + 2 <= (bases + deleted cols)/TILE + 2 reads
    a.max_pairs = max_pairs;
    if ((rc = pv_get(ctx, "sum.read_t0", nr1, &a.read_t0))) return rc;
    if ((rc = pv_get(ctx, "sum.read_t1", nr1, &a.read_t1))) return rc;
    if ((rc = pv_get(ctx, "sum.tile_cnt", a.n_tiles, &a.tile_cnt))) return rc;
    if ((rc = pv_get(ctx, "sum.tile_off", a.n_tiles, &a.tile_off))) return rc;
    if ((rc = pv_get(ctx, "sum.tile_fill", a.n_tiles, &a.tile_fill))) return rc;
    if ((rc = pv_get(ctx, "sum.pairs", max_pairs, &a.pairs))) return rc;
    if ((rc = pv_get(ctx, "sum.cnt", (size_t)(hp ? CNT_STRIDE_HP : CNT_STRIDE) * n_cols, &a.cnt))) return rc;
    if ((rc = pv_get(ctx, "sum.flags", n_cols, &a.flags))) return rc;
    if ((rc = pv_get(ctx, "sum.blk_cnt", (size_t)a.n_tiles + 2, &a.blk_cnt))) return rc;   // per tile
    if ((rc = pv_get(ctx, "sum.tile_g0", (size_t)a.n_tiles + 2, &a.tile_g0))) return rc;
    if ((rc = pv_get(ctx, "sum.site_col", max_sites, &a.site_col))) return rc;
    if ((rc = pv_get(ctx, "sum.site_hdr", max_sites, &a.site_hdr))) return rc;
    if ((rc = pv_get(ctx, "sum.big_sites", max_sites, &a.big_sites))) return rc;
    if ((rc = pv_get(ctx, "sum.site_region", max_sites, &a.site_region))) return rc;
    if ((rc = pv_get(ctx, "sum.site_nev", max_sites, &a.site_nev))) return rc;
    if ((rc = pv_get(ctx, "sum.site_evoff", max_sites, &a.site_evoff))) return rc;
    if ((rc = pv_get(ctx, "sum.site_fill", max_sites, &a.site_fill))) return rc;
    if ((rc = pv_get(ctx, "sum.site_nemit", max_sites, &a.site_nemit))) return rc;
    if ((rc = pv_get(ctx, "sum.site_strbytes", max_sites, &a.site_strbytes))) return rc;
    if ((rc = pv_get(ctx, "sum.site_outoff", max_sites, &a.site_outoff))) return rc;
    if ((rc = pv_get(ctx, "sum.site_stroff", max_sites, &a.site_stroff))) return rc;
    if ((rc = pv_get(ctx, "sum.ev", max_events, &a.ev))) return rc;
    if ((rc = pv_get(ctx, "sum.rec", max_events + 4 * max_sites, &a.rec))) return rc;
    if ((rc = pv_get(ctx, "sum.diag", (size_t)D_NDIAG + D_SPARE, &a.diag))) return rc;

    pv_prof_scope ps_all(ctx, "summary_pipeline", st);
    k_init<<<grid_for(std::max<int64_t>(a.n_tiles, D_NDIAG + 8), 256), 256, 0, st>>>(a);
    if (in->n_regions > 0) k_check_depth<<<grid_for(in->n_regions, 256), 256, 0, st>>>(a);
    if (n_reads > 0) { pv_prof_scope ps(ctx, "k_cigar_scan", st); k_cigar_scan<<<grid_for(n_reads, 4), 256, 0, st>>>(a); }
    k_scan_tiles<<<1, 1024, 0, st>>>(a);
    if (n_reads > 0) { pv_prof_scope ps(ctx, "k_tile_fill", st); k_tile_fill<<<grid_for(n_reads, 4), 256, 0, st>>>(a); }
    {
        pv_prof_scope ps(ctx, "k_pileup", st);
        if (hp) k_pileup_tiles<true><<<(unsigned)a.n_tiles, PT_THREADS, 0, st>>>(a);
        else k_pileup_tiles<false><<<(unsigned)a.n_tiles, PT_THREADS, 0, st>>>(a);
    }
    k_site_rank<<<(unsigned)n_blk, 1024, 0, st>>>(a);
    k_scan_events<<<scan_chunks(a.max_sites), 1024, 0, st>>>(a);
    // per-site kernels are chains of dependent loads per wave: as many workgroups as can be resident (one site each for the
    // benchmark's ~8 k sites per launch)
    // (swept in round 2, 2048 .. 32768 workgroups: k_collect and k_write_windows are flat from 4096 / 8192 up, k_site_alleles
    // gains 5 us at 16384)
    const unsigned site_grid = (unsigned)(max_sites < 16384 ? (max_sites > 0 ? (max_sites + 7) / 8 * 8 : 8) : 16384);   // multiples of 8: xcd_site()
    const unsigned collect_grid = site_grid < 4096 ? site_grid : 4096;
    const unsigned ww_grid = site_grid < 8192 ? site_grid : 8192;
    if (n_cigar > 0 && n_reads > 0) { pv_prof_scope ps(ctx, "k_collect", st); k_collect<<<collect_grid, 64 * KC_WAVES, 0, st>>>(a); }
    {
        pv_prof_scope ps(ctx, "k_site_alleles", st);
        const unsigned big_grid = site_grid < 1024 ? site_grid : 1024;
        if (hp) {
            k_site_alleles<true, UM_SMALL, false><<<site_grid, 64, 0, st>>>(a);
            k_site_alleles<true, UMAX, true><<<big_grid, 64, 0, st>>>(a);
        } else {
            k_site_alleles<false, UM_SMALL, false><<<site_grid, 64, 0, st>>>(a);
            k_site_alleles<false, UMAX, true><<<big_grid, 64, 0, st>>>(a);
        }
    }
    k_scan_outputs<<<scan_chunks(a.max_sites), 1024, 0, st>>>(a);
    {
        pv_prof_scope ps(ctx, "k_write_windows", st);
        if (hp) k_write_windows_hp<<<ww_grid, WW_THREADS, 0, st>>>(a);
        else k_write_windows<<<ww_grid, WW_THREADS, 0, st>>>(a);
    }
    PV_HIP(hipGetLastError());
    return PV_OK;
}

static void default_limits(int64_t n_cols, int64_t n_cigar, int64_t n_bases, int64_t n_reads, int64_t capacity,
                           int64_t* max_sites, int64_t* max_events, int64_t* max_pairs) {
    // a read touches span/TILE_COLS + 2 tiles at most and its in-region span is bounded by its aligned
    // bases plus deleted columns; deletions are rare, so allow 2x and let the device report overflow
    *max_pairs = 2 * (n_bases / TILE_COLS) + 3 * n_reads + 64;
    int64_t s = n_cols / 8 + 1024;
    if (s < 2 * capacity) s = 2 * capacity;
    if (s > n_cols) s = n_cols;
    if (s < 1) s = 1;
    *max_sites = s;
    *max_events = n_cigar + n_bases / 64 + 4096;
}

extern "C" int pv_summarize_regions_dev(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params, int64_t n_reads,
                                        int64_t n_bases, int64_t n_cigar, int64_t n_ref_bytes, int64_t max_region_len,
                                        pv_batch_out* out, int64_t* d_counts, void* stream) {
    PV_CHECK(ctx && in && params && out && d_counts, PV_ERR_INVALID, "null argument");
    PV_CHECK(in->n_regions >= 0 && n_ref_bytes >= 0, PV_ERR_INVALID, "negative sizes");
    (void)max_region_len;
    PV_HIP(hipSetDevice(ctx->device));
    int64_t ms, me;
    int64_t mp;
    default_limits(n_ref_bytes, n_cigar, n_bases, n_reads, out->capacity, &ms, &me, &mp);
    return summarize_launch(ctx, in, params, n_reads, n_bases, n_cigar, n_ref_bytes > 0 ? n_ref_bytes : 1, ms, me, mp, out,
                            d_counts, pv_pick_stream(ctx, stream));
}

extern "C" int pv_summarize_regions_hp_dev(pv_ctx* ctx, const pv_batch_in* in, const int32_t* read_hp, const pv_params* params,
                                           int64_t n_reads, int64_t n_bases, int64_t n_cigar, int64_t n_ref_bytes,
                                           pv_batch_out* out, int64_t* d_counts, void* stream) {
    PV_CHECK(ctx && in && params && out && d_counts, PV_ERR_INVALID, "null argument");
    PV_CHECK(in->n_regions >= 0 && n_ref_bytes >= 0, PV_ERR_INVALID, "negative sizes");
    PV_HIP(hipSetDevice(ctx->device));
    int64_t ms, me, mp;
    default_limits(n_ref_bytes, n_cigar, n_bases, n_reads, out->capacity, &ms, &me, &mp);
    me += n_bases / 16;  // every SNP observation at a site is an event in this form
    return summarize_launch(ctx, in, params, n_reads, n_bases, n_cigar, n_ref_bytes > 0 ? n_ref_bytes : 1, ms, me, mp, out,
                            d_counts, pv_pick_stream(ctx, stream), true, read_hp);
}

template <typename T>
static int upload(pv_ctx* ctx, const char* name, const T* h, size_t n, const T** d, hipStream_t st) {
    T* p = nullptr;
    int rc = pv_get(ctx, name, n ? n : 1, &p);
    if (rc) return rc;
    if (n) PV_HIP(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, st));
    *d = p;
    return PV_OK;
}

// The arrays of a HOST batch -> the context's workspace (asynchronous copies on `stream`); `dev` receives the same struct with
// DEVICE pointers, valid until the next upload on this context. The offset arrays are validated on the host first (cheap:
// O(regions + reads)). totals4 = {n_reads, n_bases, n_cigar, n_ref_bytes}: what the *_dev entry points take next to the struct.
extern "C" int pv_upload_batch(pv_ctx* ctx, const pv_batch_in* in, pv_batch_in* dev, int64_t* totals4, void* stream) {
    PV_CHECK(ctx && in && dev && totals4, PV_ERR_INVALID, "null argument");
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = pv_pick_stream(ctx, stream);
    const int G = in->n_regions;
    PV_CHECK(G >= 0, PV_ERR_INVALID, "negative region count");
    *dev = *in;
    totals4[0] = totals4[1] = totals4[2] = totals4[3] = 0;
    if (G == 0) return PV_OK;
    const int64_t n_reads = in->read_off[G], n_cols = in->ref_off[G];
    PV_CHECK(in->read_off[0] == 0 && in->ref_off[0] == 0 && n_reads >= 0, PV_ERR_INVALID, "offset arrays must start at 0");
    for (int g = 0; g < G; g++) {
        const int64_t R = in->ref_end[g] - in->ref_start[g] + 1;
        PV_CHECK(R >= 1 && in->ref_off[g + 1] - in->ref_off[g] >= R, PV_ERR_INVALID,
                 "region %d: reference shorter than ref_end-ref_start+1", g);
        PV_CHECK(in->read_off[g + 1] >= in->read_off[g], PV_ERR_INVALID, "read_off not monotone");
    }
    const int64_t n_bases = n_reads ? in->base_off[n_reads] : 0, n_cigar = n_reads ? in->cigar_off[n_reads] : 0;
    for (int64_t r = 0; r < n_reads; r++)
        PV_CHECK(in->base_off[r + 1] >= in->base_off[r] && in->cigar_off[r + 1] >= in->cigar_off[r], PV_ERR_INVALID,
                 "read %lld: offsets not monotone", (long long)r);
    pv_batch_in& d = *dev;
    int rc;
    if ((rc = upload(ctx, "in.ref_start", in->ref_start, G, &d.ref_start, st))) return rc;
    if ((rc = upload(ctx, "in.ref_end", in->ref_end, G, &d.ref_end, st))) return rc;
    if ((rc = upload(ctx, "in.cand_start", in->cand_start, G, &d.cand_start, st))) return rc;
    if ((rc = upload(ctx, "in.cand_end", in->cand_end, G, &d.cand_end, st))) return rc;
    if ((rc = upload(ctx, "in.ref_off", in->ref_off, G + 1, &d.ref_off, st))) return rc;
    if ((rc = upload(ctx, "in.ref", in->ref, n_cols, &d.ref, st))) return rc;
    if ((rc = upload(ctx, "in.read_off", in->read_off, G + 1, &d.read_off, st))) return rc;
    if ((rc = upload(ctx, "in.read_pos", in->read_pos, n_reads, &d.read_pos, st))) return rc;
    if ((rc = upload(ctx, "in.read_flags", in->read_flags, n_reads, &d.read_flags, st))) return rc;
    if ((rc = upload(ctx, "in.read_mapq", in->read_mapq, n_reads, &d.read_mapq, st))) return rc;
    if ((rc = upload(ctx, "in.base_off", in->base_off, n_reads + 1, &d.base_off, st))) return rc;
    if ((rc = upload(ctx, "in.bases", in->bases, n_bases, &d.bases, st))) return rc;
    if ((rc = upload(ctx, "in.quals", in->quals, n_bases, &d.quals, st))) return rc;
    if ((rc = upload(ctx, "in.cigar_off", in->cigar_off, n_reads + 1, &d.cigar_off, st))) return rc;
    if ((rc = upload(ctx, "in.cigar", in->cigar, n_cigar, &d.cigar, st))) return rc;
    totals4[0] = n_reads; totals4[1] = n_bases; totals4[2] = n_cigar; totals4[3] = n_cols;
    return PV_OK;
}

// The same for a batch that arrives in PARTS (e.g. one part per interval from the reader threads): the parts are laid end to
// end on the device - the large arrays (reference, bases, qualities, CIGAR) are copied part by part straight to their offsets,
// only the small per-region / per-read arrays are rebased on the host - so the caller never concatenates ~15 MB per interval.
extern "C" int pv_upload_batches(pv_ctx* ctx, int n_parts, const pv_batch_in* const* parts, pv_batch_in* dev, int64_t* totals4, void* stream) {
    PV_CHECK(ctx && parts && dev && totals4 && n_parts >= 0, PV_ERR_INVALID, "null argument");
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = pv_pick_stream(ctx, stream);
    int64_t G = 0, n_reads = 0, n_bases = 0, n_cigar = 0, n_cols = 0;
    for (int k = 0; k < n_parts; k++) {
        const pv_batch_in* in = parts[k];
        PV_CHECK(in && in->n_regions >= 0, PV_ERR_INVALID, "part %d: bad region count", k);
        const int g = in->n_regions;
        if (g == 0) continue;
        PV_CHECK(in->read_off[0] == 0 && in->ref_off[0] == 0 && in->read_off[g] >= 0, PV_ERR_INVALID, "part %d: offset arrays must start at 0", k);
        for (int i = 0; i < g; i++) {
            const int64_t R = in->ref_end[i] - in->ref_start[i] + 1;
            PV_CHECK(R >= 1 && in->ref_off[i + 1] - in->ref_off[i] >= R, PV_ERR_INVALID, "part %d region %d: reference shorter than ref_end-ref_start+1", k, i);
            PV_CHECK(in->read_off[i + 1] >= in->read_off[i], PV_ERR_INVALID, "read_off not monotone");
        }
        const int64_t nr = in->read_off[g];
        for (int64_t r = 0; r < nr; r++)
            PV_CHECK(in->base_off[r + 1] >= in->base_off[r] && in->cigar_off[r + 1] >= in->cigar_off[r], PV_ERR_INVALID, "part %d read %lld: offsets not monotone", k, (long long)r);
        G += g; n_reads += nr; n_cols += in->ref_off[g];
        n_bases += nr ? in->base_off[nr] : 0; n_cigar += nr ? in->cigar_off[nr] : 0;
    }
    PV_CHECK(G < (1ll << 31), PV_ERR_LIMIT, "too many regions");
    memset(dev, 0, sizeof(*dev));
    dev->n_regions = (int32_t)G;
    totals4[0] = n_reads; totals4[1] = n_bases; totals4[2] = n_cigar; totals4[3] = n_cols;
    if (G == 0) return PV_OK;
    // small arrays: rebased on the host into one staging vector per array (kept alive in the context until the next upload)
    std::vector<int64_t>& hs = ctx->upload_i64;
    std::vector<uint8_t>& hb = ctx->upload_u8;
    const size_t n64 = (size_t)(4 * G + 2 * (G + 1) + n_reads + 2 * (n_reads + 1));
    hs.resize(n64);
    hb.resize((size_t)(2 * n_reads));
    int64_t* h_ref_start = hs.data(); int64_t* h_ref_end = h_ref_start + G; int64_t* h_cand_start = h_ref_end + G; int64_t* h_cand_end = h_cand_start + G;
    int64_t* h_ref_off = h_cand_end + G; int64_t* h_read_off = h_ref_off + (G + 1); int64_t* h_read_pos = h_read_off + (G + 1);
    int64_t* h_base_off = h_read_pos + n_reads; int64_t* h_cigar_off = h_base_off + (n_reads + 1);
    uint8_t* h_flags = hb.data(); uint8_t* h_mapq = h_flags + n_reads;
    uint8_t *d_ref = nullptr, *d_bases = nullptr, *d_quals = nullptr;
    uint32_t* d_cigar = nullptr;
    int rc;
    if ((rc = pv_get(ctx, "in.ref", (size_t)std::max<int64_t>(n_cols, 1), &d_ref)) || (rc = pv_get(ctx, "in.bases", (size_t)std::max<int64_t>(n_bases, 1), &d_bases)) ||
        (rc = pv_get(ctx, "in.quals", (size_t)std::max<int64_t>(n_bases, 1), &d_quals)) || (rc = pv_get(ctx, "in.cigar", (size_t)std::max<int64_t>(n_cigar, 1), &d_cigar)))
        return rc;
    int64_t g0 = 0, r0 = 0, b0 = 0, c0 = 0, col0 = 0;
    h_ref_off[0] = 0; h_read_off[0] = 0; h_base_off[0] = 0; h_cigar_off[0] = 0;
    for (int k = 0; k < n_parts; k++) {
        const pv_batch_in* in = parts[k];
        const int g = in->n_regions;
        if (g == 0) continue;
        const int64_t nr = in->read_off[g], nb = nr ? in->base_off[nr] : 0, nc = nr ? in->cigar_off[nr] : 0, ncol = in->ref_off[g];
        for (int i = 0; i < g; i++) {
            h_ref_start[g0 + i] = in->ref_start[i]; h_ref_end[g0 + i] = in->ref_end[i];
            h_cand_start[g0 + i] = in->cand_start[i]; h_cand_end[g0 + i] = in->cand_end[i];
            h_ref_off[g0 + i + 1] = col0 + in->ref_off[i + 1];
            h_read_off[g0 + i + 1] = r0 + in->read_off[i + 1];
        }
        for (int64_t r = 0; r < nr; r++) {
            h_read_pos[r0 + r] = in->read_pos[r];
            h_flags[r0 + r] = in->read_flags[r]; h_mapq[r0 + r] = in->read_mapq[r];
            h_base_off[r0 + r + 1] = b0 + in->base_off[r + 1];
            h_cigar_off[r0 + r + 1] = c0 + in->cigar_off[r + 1];
        }
        if (ncol) PV_HIP(hipMemcpyAsync(d_ref + col0, in->ref, (size_t)ncol, hipMemcpyHostToDevice, st));
        if (nb) {
            PV_HIP(hipMemcpyAsync(d_bases + b0, in->bases, (size_t)nb, hipMemcpyHostToDevice, st));
            PV_HIP(hipMemcpyAsync(d_quals + b0, in->quals, (size_t)nb, hipMemcpyHostToDevice, st));
        }
        if (nc) PV_HIP(hipMemcpyAsync(d_cigar + c0, in->cigar, (size_t)nc * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        g0 += g; r0 += nr; b0 += nb; c0 += nc; col0 += ncol;
    }
    dev->ref = d_ref; dev->bases = d_bases; dev->quals = d_quals; dev->cigar = d_cigar;
    if ((rc = upload(ctx, "in.ref_start", h_ref_start, G, &dev->ref_start, st))) return rc;
    if ((rc = upload(ctx, "in.ref_end", h_ref_end, G, &dev->ref_end, st))) return rc;
    if ((rc = upload(ctx, "in.cand_start", h_cand_start, G, &dev->cand_start, st))) return rc;
    if ((rc = upload(ctx, "in.cand_end", h_cand_end, G, &dev->cand_end, st))) return rc;
    if ((rc = upload(ctx, "in.ref_off", h_ref_off, G + 1, &dev->ref_off, st))) return rc;
    if ((rc = upload(ctx, "in.read_off", h_read_off, G + 1, &dev->read_off, st))) return rc;
    if ((rc = upload(ctx, "in.read_pos", h_read_pos, n_reads, &dev->read_pos, st))) return rc;
    if ((rc = upload(ctx, "in.base_off", h_base_off, n_reads + 1, &dev->base_off, st))) return rc;
    if ((rc = upload(ctx, "in.cigar_off", h_cigar_off, n_reads + 1, &dev->cigar_off, st))) return rc;
    {
        const uint8_t* t = nullptr;
        if ((rc = upload(ctx, "in.read_flags", (const uint8_t*)h_flags, n_reads, &t, st))) return rc;
        dev->read_flags = t;
        if ((rc = upload(ctx, "in.read_mapq", (const uint8_t*)h_mapq, n_reads, &t, st))) return rc;
        dev->read_mapq = t;
    }
    return PV_OK;
}

static int summarize_host(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params, pv_batch_out* out, bool hp,
                          const int32_t* read_hp);
extern "C" int pv_summarize_regions(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params, pv_batch_out* out) {
    return summarize_host(ctx, in, params, out, false, nullptr);
}
extern "C" int pv_summarize_regions_hp(pv_ctx* ctx, const pv_batch_in* in, const int32_t* read_hp, const pv_params* params,
                                       pv_batch_out* out) {
    return summarize_host(ctx, in, params, out, true, read_hp);
}

// host buffers in and out; `hp` selects the haplotag-aware builder (window bytes, one more input array)
static int summarize_host(pv_ctx* ctx, const pv_batch_in* in, const pv_params* params, pv_batch_out* out, bool hp,
                          const int32_t* read_hp) {
    PV_CHECK(ctx && in && params && out, PV_ERR_INVALID, "null argument");
    const size_t WB = hp ? PV_HP_WINDOW_BYTES : PV_WINDOW_BYTES;
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int G = in->n_regions;
    out->n_out = 0;
    out->str_bytes = 0;
    if (out->capacity > 0 && out->cand_off) out->cand_off[0] = 0;
    if (G <= 0) return PV_OK;
    for (int g = 0; g < G; g++)
        PV_CHECK(in->read_off[g + 1] - in->read_off[g] <= MAX_REGION_READS, PV_ERR_LIMIT,
                 "region %d holds %lld reads: the counter planes are 16-bit (at most %d reads per region; the reference's caller "
                 "down-samples to 5000)", g, (long long)(in->read_off[g + 1] - in->read_off[g]), MAX_REGION_READS);
    pv_batch_in d;
    int64_t totals[4];
    int rc = pv_upload_batch(ctx, in, &d, totals, st);
    if (rc) return rc;
    const int64_t n_reads = totals[0], n_bases = totals[1], n_cigar = totals[2], n_cols = totals[3];
    const int32_t* d_hp = nullptr;
    if (hp && read_hp)
        if ((rc = upload(ctx, "in.read_hp", read_hp, n_reads, &d_hp, st))) return rc;

    const int64_t cap = out->capacity > 0 ? out->capacity : 0, scap = out->str_capacity > 0 ? out->str_capacity : 0;
    pv_batch_out dout = *out;
    if ((rc = pv_get(ctx, "out.region", cap + 1, &dout.region))) return rc;
    if ((rc = pv_get(ctx, "out.position", cap + 1, &dout.position))) return rc;
    if ((rc = pv_get(ctx, "out.depth", cap + 1, &dout.depth))) return rc;
    if ((rc = pv_get(ctx, "out.cand_freq", cap + 1, &dout.cand_freq))) return rc;
    if ((rc = pv_get(ctx, "out.images", (size_t)(cap + 1) * WB, &dout.images))) return rc;
    dout.images_i32 = nullptr;
    if (out->images_i32)
        if ((rc = pv_get(ctx, "out.images_i32", (size_t)(cap + 1) * WB, &dout.images_i32))) return rc;
    if ((rc = pv_get(ctx, "out.cand_str", scap + 1, &dout.cand_str))) return rc;
    if ((rc = pv_get(ctx, "out.cand_off", cap + 2, &dout.cand_off))) return rc;
    int64_t* d_counts = nullptr;
    if ((rc = pv_get(ctx, "out.counts", (size_t)4, &d_counts))) return rc;

    int64_t ms, me, mp;
    default_limits(n_cols, n_cigar, n_bases, n_reads, cap, &ms, &me, &mp);
    if (hp) me += n_bases / 16;
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = summarize_launch(ctx, &d, params, n_reads, n_bases, n_cigar, n_cols, ms, me, mp, &dout, d_counts, st, hp, d_hp);
        if (rc) return rc;
        PV_HIP(hipMemcpyAsync(ctx->h_counts, d_counts, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipStreamSynchronize(st));
        if (ctx->h_counts[2] == PV_ERR_LIMIT && attempt < 2) {  // workspace heuristics too small: take exact bounds
            ms = n_cols;
            me = n_cigar + n_bases;
            mp = n_reads * ((n_cols + TILE_COLS - 1) / TILE_COLS + 2);
            continue;
        }
        break;
    }
    const int64_t status = ctx->h_counts[2];
    PV_CHECK(status != PV_ERR_INVALID, PV_ERR_INVALID, "malformed read: CIGAR walks past the end of its bases");
    PV_CHECK(status != PV_ERR_LIMIT, PV_ERR_LIMIT, "more than %d distinct alleles at one site, or index range exceeded", UMAX);
    PV_CHECK(status == 0, (int)status, "device status %lld", (long long)status);
    out->n_out = ctx->h_counts[0];
    out->str_bytes = ctx->h_counts[1];
    if (out->n_out > cap || out->str_bytes > scap) {
        pv_set_error("output capacity too small: need %lld windows, %lld key bytes", (long long)out->n_out,
                     (long long)out->str_bytes);
        return PV_ERR_CAPACITY;
    }
    const int64_t n = out->n_out;
    if (n > 0) {
        PV_HIP(hipMemcpyAsync(out->region, dout.region, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->position, dout.position, n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->depth, dout.depth, n, hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->cand_freq, dout.cand_freq, n, hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->images, dout.images, n * WB, hipMemcpyDeviceToHost, st));
        if (out->images_i32)
            PV_HIP(hipMemcpyAsync(out->images_i32, dout.images_i32, n * WB * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->cand_str, dout.cand_str, out->str_bytes, hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->cand_off, dout.cand_off, (n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipStreamSynchronize(st));
    }
    return PV_OK;
}

// ==== P2 (polisher) summary images: launch sequence and C-ABI ===========================================
static int polish_launch(pv_ctx* ctx, const pv_batch_in* in, int64_t n_reads, int64_t n_bases, int64_t n_cigar,
                         int64_t n_cols, int64_t max_pairs, int64_t max_ins_rows, int seq_length, int seq_overlap,
                         const pv_polish_out* out, int64_t* d_counts, hipStream_t st) {
    PV_CHECK(seq_length >= 1 && seq_overlap >= 0 && seq_overlap < seq_length, PV_ERR_INVALID,
             "need 0 <= seq_overlap < seq_length (got %d, %d)", seq_overlap, seq_length);
    PV_CHECK(n_cols < (1ll << 31) - 2048 && n_cigar < (1ll << 31) && n_reads < (1ll << 31), PV_ERR_LIMIT,
             "batch too large for 32-bit column/op indices (cols %lld, ops %lld)", (long long)n_cols, (long long)n_cigar);
    SumArgs a;
    memset(&a, 0, sizeof(a));
    a.in = *in;
    a.polish = 1;
    a.seq_len = seq_length;
    a.seq_step = seq_length - seq_overlap;
    a.n_reads = n_reads; a.n_bases = n_bases; a.n_cigar = n_cigar; a.n_cols = n_cols;
    a.max_pairs = max_pairs;
    a.max_ins_rows = max_ins_rows;
    a.pout = *out;
    a.d_counts = d_counts;
    a.n_tiles = (n_cols + TILE_COLS - 1) / TILE_COLS;
    const int64_t n_blk = (n_cols + 1023) / 1024;
    const int64_t nc1 = n_cigar > 0 ? n_cigar : 1, nr1 = n_reads > 0 ? n_reads : 1, G = in->n_regions;
    int rc;
    if ((rc = pv_get(ctx, "sum.op_ref", nc1, &a.op_ref))) return rc;
    if ((rc = pv_get(ctx, "sum.op_rd", nc1, &a.op_rd))) return rc;
    if ((rc = pv_get(ctx, "sum.op_read", nc1, &a.op_read))) return rc;
    if ((rc = pv_get(ctx, "sum.op_flag", nc1, &a.op_flag))) return rc;
    if ((rc = pv_get(ctx, "sum.read_region", nr1, &a.read_region))) return rc;
    if ((rc = pv_get(ctx, "sum.read_t0", nr1, &a.read_t0))) return rc;
    if ((rc = pv_get(ctx, "sum.read_t1", nr1, &a.read_t1))) return rc;
    if ((rc = pv_get(ctx, "sum.tile_cnt", a.n_tiles, &a.tile_cnt))) return rc;
    if ((rc = pv_get(ctx, "sum.tile_off", a.n_tiles, &a.tile_off))) return rc;
    if ((rc = pv_get(ctx, "sum.tile_fill", a.n_tiles, &a.tile_fill))) return rc;
    if ((rc = pv_get(ctx, "sum.pairs", max_pairs, &a.pairs))) return rc;
    if ((rc = pv_get(ctx, "pol.pcnt", (size_t)PC_N * n_cols, &a.pcnt))) return rc;
    if ((rc = pv_get(ctx, "pol.ins_blk", n_blk, &a.ins_blk))) return rc;
    if ((rc = pv_get(ctx, "pol.ins_blkoff", n_blk, &a.ins_blkoff))) return rc;
    if ((rc = pv_get(ctx, "pol.ins_off", n_cols + 1, &a.ins_off))) return rc;
    if ((rc = pv_get(ctx, "pol.ins_cnt", (size_t)(max_ins_rows > 0 ? max_ins_rows : 1) * 10, &a.ins_cnt))) return rc;
    if ((rc = pv_get(ctx, "pol.reg_rows", (size_t)G + 1, &a.reg_rows))) return rc;
    if ((rc = pv_get(ctx, "pol.reg_chunks", (size_t)G + 1, &a.reg_chunks))) return rc;
    if ((rc = pv_get(ctx, "sum.diag", (size_t)D_NDIAG + D_SPARE, &a.diag))) return rc;
    if (out->flat_images) {
        PV_CHECK(out->flat_position && out->flat_index, PV_ERR_INVALID, "flat_position / flat_index missing");
        a.flat_img = out->flat_images; a.flat_pos = out->flat_position; a.flat_idx = out->flat_index;
        a.flat_cap = out->row_capacity;
    } else {
        a.flat_cap = n_cols + max_ins_rows;
        if ((rc = pv_get(ctx, "pol.flat_img", (size_t)a.flat_cap * 10, &a.flat_img))) return rc;
        if ((rc = pv_get(ctx, "pol.flat_pos", (size_t)a.flat_cap, &a.flat_pos))) return rc;
        if ((rc = pv_get(ctx, "pol.flat_idx", (size_t)a.flat_cap, &a.flat_idx))) return rc;
    }

    pv_prof_scope ps_all(ctx, "polish_pipeline", st);
    k_init<<<grid_for(std::max<int64_t>(a.n_tiles, D_NDIAG + 8), 256), 256, 0, st>>>(a);
    { int rcz = pv_zero_async(a.ins_cnt, (size_t)(max_ins_rows > 0 ? max_ins_rows : 1) * 10 * sizeof(int32_t), st); if (rcz) return rcz; }
    if (n_reads > 0) { pv_prof_scope ps(ctx, "k_cigar_scan", st); k_cigar_scan<<<grid_for(n_reads, 4), 256, 0, st>>>(a); }
    k_scan_tiles<<<1, 1024, 0, st>>>(a);
    if (n_reads > 0) { pv_prof_scope ps(ctx, "k_tile_fill", st); k_tile_fill<<<grid_for(n_reads, 4), 256, 0, st>>>(a); }
    { pv_prof_scope ps(ctx, "k_polish_tiles", st); k_polish_tiles<<<(unsigned)a.n_tiles, PT_THREADS, 0, st>>>(a); }
    k_polish_blk<<<(unsigned)n_blk, 1024, 0, st>>>(a);
    k_scan_i32<<<1, 1024, 0, st>>>(a.ins_blk, a.ins_blkoff, n_blk, nullptr, n_blk, &a.diag[D_NINS]);
    k_polish_insoff<<<(unsigned)n_blk, 1024, 0, st>>>(a);
    k_polish_regions<<<1, 1, 0, st>>>(a);
    if (n_cigar > 0) { pv_prof_scope ps(ctx, "k_polish_insert", st); k_polish_insert<<<grid_for(n_cigar, 256), 256, 0, st>>>(a); }
    { pv_prof_scope ps(ctx, "k_polish_image", st); k_polish_image<<<grid_for(n_cols, 256), 256, 0, st>>>(a); }
    if (out->chunk_capacity > 0 && out->images) {
        PV_CHECK(out->position && out->index && out->region && out->chunk_id, PV_ERR_INVALID, "chunk output arrays missing");
        pv_prof_scope ps(ctx, "k_polish_chunks", st);
        k_polish_chunks<<<grid_for(out->chunk_capacity * seq_length, 256), 256, 0, st>>>(a);
    }
    if (out->region_row_off)
        PV_HIP(hipMemcpyAsync(out->region_row_off, a.reg_rows, (size_t)(G + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    PV_HIP(hipGetLastError());
    return PV_OK;
}

static void polish_limits(int64_t n_cols, int64_t n_bases, int64_t n_reads, int64_t* max_pairs, int64_t* max_ins_rows) {
    *max_pairs = 2 * (n_bases / TILE_COLS) + 3 * n_reads + 64;
    *max_ins_rows = 2 * n_cols + 4096;  // 60x ONT (2 % inserts of 1-3 bases) adds ~1.5 insert rows per column; the device reports overflow
}

extern "C" int pv_polish_summarize_regions_dev(pv_ctx* ctx, const pv_batch_in* in, int64_t n_reads, int64_t n_bases,
                                               int64_t n_cigar, int64_t n_ref_bytes, int seq_length, int seq_overlap,
                                               pv_polish_out* out, int64_t* d_counts, void* stream) {
    PV_CHECK(ctx && in && out && d_counts, PV_ERR_INVALID, "null argument");
    PV_CHECK(in->n_regions >= 0 && n_ref_bytes >= 0, PV_ERR_INVALID, "negative sizes");
    PV_HIP(hipSetDevice(ctx->device));
    int64_t mp, mi;
    polish_limits(n_ref_bytes, n_bases, n_reads, &mp, &mi);
    if (out->flat_images && out->row_capacity > n_ref_bytes && out->row_capacity - n_ref_bytes > mi) mi = out->row_capacity - n_ref_bytes;
    return polish_launch(ctx, in, n_reads, n_bases, n_cigar, n_ref_bytes > 0 ? n_ref_bytes : 1, mp, mi, seq_length, seq_overlap,
                         out, d_counts, pv_pick_stream(ctx, stream));
}

extern "C" int pv_polish_summarize_regions(pv_ctx* ctx, const pv_batch_in* in, int seq_length, int seq_overlap,
                                           pv_polish_out* out) {
    PV_CHECK(ctx && in && out, PV_ERR_INVALID, "null argument");
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int G = in->n_regions;
    out->n_chunks = 0;
    out->n_rows = 0;
    if (G <= 0) return PV_OK;
    const int64_t n_reads = in->read_off[G], n_cols = in->ref_off[G];
    PV_CHECK(in->read_off[0] == 0 && in->ref_off[0] == 0 && n_reads >= 0, PV_ERR_INVALID, "offset arrays must start at 0");
    for (int g = 0; g < G; g++) {
        const int64_t R = in->ref_end[g] - in->ref_start[g] + 1;
        PV_CHECK(R >= 1 && in->ref_off[g + 1] - in->ref_off[g] >= R, PV_ERR_INVALID,
                 "region %d: reference shorter than ref_end-ref_start+1", g);
        PV_CHECK(in->read_off[g + 1] >= in->read_off[g], PV_ERR_INVALID, "read_off not monotone");
    }
    const int64_t n_bases = n_reads ? in->base_off[n_reads] : 0, n_cigar = n_reads ? in->cigar_off[n_reads] : 0;
    for (int64_t r = 0; r < n_reads; r++)
        PV_CHECK(in->base_off[r + 1] >= in->base_off[r] && in->cigar_off[r + 1] >= in->cigar_off[r], PV_ERR_INVALID,
                 "read %lld: offsets not monotone", (long long)r);

    pv_batch_in d = *in;
    int rc;
    if ((rc = upload(ctx, "in.ref_start", in->ref_start, G, &d.ref_start, st))) return rc;
    if ((rc = upload(ctx, "in.ref_end", in->ref_end, G, &d.ref_end, st))) return rc;
    if ((rc = upload(ctx, "in.ref_off", in->ref_off, G + 1, &d.ref_off, st))) return rc;
    d.cand_start = d.ref_start; d.cand_end = d.ref_end; d.ref = nullptr; d.quals = nullptr;  // not read by the polisher kernels
    if ((rc = upload(ctx, "in.read_off", in->read_off, G + 1, &d.read_off, st))) return rc;
    if ((rc = upload(ctx, "in.read_pos", in->read_pos, n_reads, &d.read_pos, st))) return rc;
    if ((rc = upload(ctx, "in.read_flags", in->read_flags, n_reads, &d.read_flags, st))) return rc;
    if ((rc = upload(ctx, "in.read_mapq", in->read_mapq, n_reads, &d.read_mapq, st))) return rc;
    if ((rc = upload(ctx, "in.base_off", in->base_off, n_reads + 1, &d.base_off, st))) return rc;
    if ((rc = upload(ctx, "in.bases", in->bases, n_bases, &d.bases, st))) return rc;
    if ((rc = upload(ctx, "in.cigar_off", in->cigar_off, n_reads + 1, &d.cigar_off, st))) return rc;
    if ((rc = upload(ctx, "in.cigar", in->cigar, n_cigar, &d.cigar, st))) return rc;

    const int64_t ccap = out->chunk_capacity > 0 ? out->chunk_capacity : 0, rcap = out->flat_images ? out->row_capacity : 0;
    pv_polish_out dout = *out;
    dout.chunk_capacity = ccap;
    dout.row_capacity = rcap;
    const size_t L = (size_t)seq_length;
    if ((rc = pv_get(ctx, "pout.images", (ccap + 1) * L * 10, &dout.images))) return rc;
    if ((rc = pv_get(ctx, "pout.position", (ccap + 1) * L, &dout.position))) return rc;
    if ((rc = pv_get(ctx, "pout.index", (ccap + 1) * L, &dout.index))) return rc;
    if ((rc = pv_get(ctx, "pout.region", (size_t)ccap + 1, &dout.region))) return rc;
    if ((rc = pv_get(ctx, "pout.chunk_id", (size_t)ccap + 1, &dout.chunk_id))) return rc;
    dout.flat_images = nullptr; dout.flat_position = nullptr; dout.flat_index = nullptr; dout.region_row_off = nullptr;
    if (out->flat_images) {
        PV_CHECK(out->flat_position && out->flat_index, PV_ERR_INVALID, "flat_position / flat_index missing");
        if ((rc = pv_get(ctx, "pout.flat_images", (size_t)(rcap + 1) * 10, &dout.flat_images))) return rc;
        if ((rc = pv_get(ctx, "pout.flat_position", (size_t)rcap + 1, &dout.flat_position))) return rc;
        if ((rc = pv_get(ctx, "pout.flat_index", (size_t)rcap + 1, &dout.flat_index))) return rc;
    }
    if (out->region_row_off)
        if ((rc = pv_get(ctx, "pout.region_row_off", (size_t)G + 1, &dout.region_row_off))) return rc;
    int64_t* d_counts = nullptr;
    if ((rc = pv_get(ctx, "out.counts", (size_t)4, &d_counts))) return rc;

    int64_t mp, mi;
    polish_limits(n_cols, n_bases, n_reads, &mp, &mi);
    if (rcap > n_cols && rcap - n_cols > mi) mi = rcap - n_cols;
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = polish_launch(ctx, &d, n_reads, n_bases, n_cigar, n_cols, mp, mi, seq_length, seq_overlap, &dout, d_counts, st);
        if (rc) return rc;
        PV_HIP(hipMemcpyAsync(ctx->h_counts, d_counts, 4 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipStreamSynchronize(st));
        if (ctx->h_counts[2] == PV_ERR_LIMIT && attempt < 2) {  // workspace heuristics too small: take what the device measured
            mp = n_reads * ((n_cols + TILE_COLS - 1) / TILE_COLS + 2);
            if (ctx->h_counts[3] > mi) mi = ctx->h_counts[3];
            continue;
        }
        break;
    }
    const int64_t status = ctx->h_counts[2];
    PV_CHECK(status != PV_ERR_INVALID, PV_ERR_INVALID, "malformed read: CIGAR walks past the end of its bases");
    PV_CHECK(status == 0, (int)status, "device status %lld", (long long)status);
    out->n_chunks = ctx->h_counts[0];
    out->n_rows = ctx->h_counts[1];
    if (out->n_chunks > ccap || (out->flat_images && out->n_rows > rcap)) {
        pv_set_error("output capacity too small: need %lld chunks, %lld rows", (long long)out->n_chunks, (long long)out->n_rows);
        return PV_ERR_CAPACITY;
    }
    const size_t n = (size_t)out->n_chunks;
    if (n > 0 && out->images) {
        PV_HIP(hipMemcpyAsync(out->images, dout.images, n * L * 10, hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->position, dout.position, n * L * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->index, dout.index, n * L * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->region, dout.region, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->chunk_id, dout.chunk_id, n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    if (out->flat_images && out->n_rows > 0) {
        const size_t nr = (size_t)out->n_rows;
        PV_HIP(hipMemcpyAsync(out->flat_images, dout.flat_images, nr * 10, hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->flat_position, dout.flat_position, nr * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PV_HIP(hipMemcpyAsync(out->flat_index, dout.flat_index, nr * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    if (out->region_row_off)
        PV_HIP(hipMemcpyAsync(out->region_row_off, dout.region_row_off, (size_t)(G + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PV_HIP(hipStreamSynchronize(st));
    return PV_OK;
}

#ifdef PV_PSTAMPS
// diagnostic builds only: phase cycle sums of the last k_pileup_tiles launch (6 values)
extern "C" int pv_debug_read_pstamps(pv_ctx* ctx, unsigned long long* out) {
    int64_t* d = nullptr;
    if (pv_get(ctx, "sum.diag", (size_t)D_NDIAG + D_SPARE, &d)) return PV_ERR_HIP;
    PV_HIP(hipDeviceSynchronize());
        PV_HIP(hipMemcpy(out, d + D_NDIAG, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PV_OK;
}
#endif
