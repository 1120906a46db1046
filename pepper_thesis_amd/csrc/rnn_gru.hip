// rnn_gru.hip — P2: the polisher's bidirectional-GRU encoder/decoder on gfx950 (MI355X), fp32.
//
// Reference semantics (pepper/modules/python/models/simple_model.py:27-42 and the sliding loop of
// pepper/modules/python/models/predict.py:47-97): a chunk is 1000 pileup columns x 10 features (uint8).
// 19 windows of 100 columns (stride 50) are pushed through
//     enc = bi-GRU(10 -> 128)  with h0 = hidden carried from the previous window
//     dec = bi-GRU(256 -> 128) with h0 = final encoder state of the same direction
//     logits = Linear(256 -> 5)(dec_out);  acc[i : i+100] += softmax(logits);  hidden = final decoder state
// and labels = argmax(acc). 3800 strictly sequential GRU steps per chunk.
//
// One PERSISTENT workgroup per 32-chunk batch tile runs the whole 19-window loop in a single launch:
// 8 waves = 4 (forward direction) + 4 (reverse), wave w of a direction owns hidden units [32w,32w+32)
// for the r, z, n gates, so the cell update is in-register; h lives in registers (update) and in a
// double-buffered LDS tile (A operand of the next step's MFMA); the gate products run on the f32 MFMA
// (v_mfma_f32_32x32x2_f32) with host-packed weight fragments streamed from L2 (1.6 MB for the whole
// model); x_{t+1} is prefetched into registers behind step t's MFMAs. Encoder/decoder outputs of the
// current window go through a per-tile scratch in HBM/L2 (the decoder needs both directions of the
// encoder, the classifier both directions of the decoder); dense + softmax + accumulate and the final
// argmax are fused into the same launch.
#include "pv_common.hpp"
#include "mfma_tiles.hpp"
#include "rnn_bf16.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

using namespace pvdev;

constexpr int SEQ = 1000, WIN = 100, JUMP = 50, NWIN = 19, FEAT = 10, NCLS = 5;
constexpr int HG = 128;          // GRU hidden
constexpr int KPE = 32;          // encoder input (10 features) padded to four k-blocks of 8 (mma3_ring works in groups of 4)
constexpr int KPD = 2 * HG;      // decoder input
constexpr int LDH = HG + 4;      // LDS row strides (floats): % 64 == 4 keeps ds_read_b128 conflict-free
constexpr int LDXD = KPD + 4;

__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * rcpf_(e + 1.0f);
}

// ---- batch-tile geometry ---------------------------------------------------------------------------------------------
// TR = chunks (batch rows) per workgroup. 32 rows use v_mfma_f32_32x32x2_f32, 16 rows v_mfma_f32_16x16x4_f32: the same
// FLOP per cycle, half the cycles per GRU step and twice the workgroups, which is what small batches need (a launch of
// B chunks is a chain of 3800 dependent steps, so its duration is set by the per-step MFMA time of one workgroup, not by
// B, until the workgroups fill the chip); the price is that every weight fragment feeds half as many rows, so the
// 32-row form stays the choice once 32-row tiles fill all CUs.
// Per wave: hidden units [32w, 32w+32) of the gates r, z, n. One "gate accumulator" holds those 32 units for TR rows.
// (Gate<TR>, AFrag<TR>, gate_mma: mfma_tiles.hpp)

// One K loop over [x_t | h_{t-1}] for the three gates of a wave: r and z accumulate both products, n keeps its x- and
// h-parts apart (anx / anh, PyTorch's GRU: n = tanh(W_in x + b_in + r * (W_hn h + b_hn))). The packed weight stream is
// contiguous over both operands, one f32x4 per lane per gate per k-block of 8. B fragments come from L2 (~1 us away)
// while a k-block is only 12 MFMAs (0.33 us), so they run in a ring of four register sets requested THREE k-blocks
// ahead. A fragments (LDS) are fetched one block ahead. The ring wraps around: the last group of a step already requests
// k-blocks 0..2 of the next step (same weights every step), which arrive during the cell update and the barriers, so a
// step does not start with a cold L2 round trip. `bq` lives in the caller (slots 0..2 must hold k-blocks 0..2 on
// entry). nkbx and nkbh are multiples of 4.
template <int TR>
__device__ __forceinline__ void mma3_ring(Gate<TR>& ar, Gate<TR>& az, Gate<TR>& anx, Gate<TR>& anh,
                                          const float* __restrict__ X, int ldx, int nkbx, const float* __restrict__ Hh,
                                          int ldh, int nkbh, __amdgpu_buffer_rsrc_t wr, f32x4 (&bq)[4][3], int lane) {
    typedef typename AFrag<TR>::type afrag;
    constexpr int NJ = TR == 32 ? 4 : 2;
    const float* apx = afrag_ptr<TR>(X, ldx, lane);
    const float* aph = afrag_ptr<TR>(Hh, ldh, lane) - 8 * nkbx;
    const int nkb = nkbx + nkbh;
    const unsigned lane16 = (unsigned)lane * 16u;
    afrag aq[2];
#define G_B(slot, kbv) { _Pragma("unroll") for (int nt = 0; nt < 3; nt++) bq[slot][nt] = buf_load4(wr, lane16, (unsigned)(((kbv) * 3 + nt) * 1024)); }
#define G_A(slot, kbv) { aq[slot] = *reinterpret_cast<const afrag*>(((kbv) < nkbx ? apx : aph) + 8 * (kbv)); }
#define G_M(bs, as, CN)                                         \
    {                                                           \
        _Pragma("unroll") for (int j = 0; j < NJ; j++) {        \
            gate_mma<TR>(ar, aq[as], bq[bs][0], j);             \
            gate_mma<TR>(az, aq[as], bq[bs][1], j);             \
            gate_mma<TR>(CN, aq[as], bq[bs][2], j);             \
        }                                                       \
    }
#define G_FENCE __builtin_amdgcn_sched_barrier(0); /* keep hipcc from sinking the prefetches next to their uses */
#define G_GROUP(CN)                          \
    {                                        \
        G_B(3, kb + 3)                       \
        G_A(1, kb + 1)                       \
        G_FENCE                              \
        G_M(0, 0, CN)                        \
        G_FENCE                              \
        const int kw = kb + 4 < nkb ? kb + 4 : 0; /* wrap: next step's first blocks */ \
        G_B(0, kw)                           \
        G_A(0, kb + 2)                       \
        G_FENCE                              \
        G_M(1, 1, CN)                        \
        G_FENCE                              \
        G_B(1, kw + 1)                       \
        G_A(1, kb + 3)                       \
        G_FENCE                              \
        G_M(2, 0, CN)                        \
        G_FENCE                              \
        G_B(2, kw + 2)                       \
        if (kb + 4 < nkb) G_A(0, kb + 4)     \
        G_FENCE                              \
        G_M(3, 1, CN)                        \
        G_FENCE                              \
    }
    G_A(0, 0)
    int kb = 0;
#pragma nounroll
    for (; kb < nkbx; kb += 4) G_GROUP(anx)
#pragma nounroll
    for (; kb < nkb; kb += 4) G_GROUP(anh)
#undef G_B
#undef G_A
#undef G_M
#undef G_GROUP
#undef G_FENCE
}

struct GruArgs {
    const uint8_t* images;  // [B,1000,10]
    const float* enc_wp;    // packed [2 dirs][4 waves][(KPE+HG)/8 kb][3 gates][64 lanes][4] in the tile form of the launch
    const float* dec_wp;    // packed [2 dirs][4 waves][(KPD+HG)/8 kb][3][64][4]
    const float* enc_bias;  // [2 dirs][4][128]: b_ir+b_hr, b_iz+b_hz, b_in, b_hn
    const float* dec_bias;
    const float* dense_w;   // [5,256]
    const float* dense_b;   // [5]
    float* enc_out;         // scratch [n_tiles][100][TR][256]
    float* dec_out;         // scratch [n_tiles][100][TR][256]
    float* acc;             // [B,seq,5] (zero-initialised by the caller)
    uint8_t* labels;        // [B,seq] or NULL
    int64_t B;
    int seq;                // columns per chunk (1000; 100 for a single TransducerGRU.forward)
    int nwin;               // windows of 100 columns, stride 50 (19; 1)
    const float* hidden_in; // [B,2,128] or NULL (zeros)
    float* hidden_out;      // [B,2,128] or NULL
    float* logits;          // [B,100,5] raw dense1 output of the LAST window, or NULL
    int* pair_flags;        // split form only: [n_tiles][2] hand-off counters of the two direction workgroups of a tile (zeroed per launch)
    // unit-split form only (k_gru_us):
    u32x4* hx;              // [n_tiles][2 dirs][US_HX_QUADS] data-tagged h pairs
    int* quad_flags;        // [n_tiles][4][US_FLAG_STRIDE] layer-boundary counters of the four workgroups of a tile (zeroed per launch)
    unsigned tag_base;      // tags of this launch run from tag_base + 1 (set by the kernel from *epoch)
    unsigned* epoch;        // device word: launches of the unit-split form so far (bumped by k_gru_bump behind every launch, so a
                            // replayed hipGraph advances the tags like eager launches do)
    int* err;               // bumped when a bounded poll of the exchange or of a layer hand-off gave up
    int spin_limit;         // polls of the per-step exchange give up after this many tries; layer hand-offs after 256 x as many
    int drop_part;          // diagnostic (pv_opts::debug_drop_part): workgroup `sub` of every tile (unit-split form) or direction
                            // `drop_part & 1` (direction-split form) leaves at once; -1 = none
};

// ---- the two directions of a tile on two CUs (small batches) -----------------------------------------------------------
// A launch is a chain of 3800 dependent steps whose duration is the per-step time of ONE workgroup. With both directions
// in one workgroup every SIMD carries a forward and a reverse wave, i.e. twice the MFMA cycles per step. In the SPLIT
// form a workgroup is (tile, direction), 4 waves, one per SIMD; the directions meet only where the model joins them: the
// decoder reads both halves of the encoder output of its window, and dense1 + softmax read both halves of the decoder
// output - two hand-offs per 100-column window (38 per launch), each a release/acquire pair at agent scope through a
// monotonic counter per workgroup (MI355X_MICROARCH.md, inter-workgroup visibility): every wave drains its stores, the
// workgroup's barrier, one lane publishes (release fence, drained, then the relaxed counter store) and polls the partner's
// counter with relaxed loads, one acquire fence, barrier. Used only while both workgroups of every tile are resident at
// once (grid <= CUs, one workgroup per CU), so a poll can never wait for a workgroup that has not been dispatched.
__device__ __forceinline__ void pair_handoff(int* mine, const int* theirs, int value, int tid, int* err, int limit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ROCm 7.2 can drop the fence's own wait: keep this one, before the flag
        __hip_atomic_store(mine, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > limit) { atomicAdd(err, 1); break; }   // bounded: a lost partner ends in POISONED results (k_gru_finish), never in a hung device
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// one GRU layer over one 100-column window for this wave's direction.
// Addressing: every global access is a raw buffer access (wave-uniform resource + 32-bit lane offset computed once +
// scalar offset) and every LDS access a uniform base plus a lane offset; with per-lane 64-bit addresses for the x-loads,
// h-stores and weight loads of a step the kernel spilled 150 VGPRs around the MFMA loop.
template <int TR, int KP, bool ENC>
__device__ __forceinline__ void gru_window(const GruArgs& a, int win_start, int dir, int wq, int lane, int tid_dir,
                                           int64_t b0, float* xbuf, float* hbuf, int& cur, float (&hst)[TR / 2],
                                           const float* wp, const float* bias, const float* x_src, float* out_dst) {
    constexpr int LDX = KP + 4;
    constexpr int NKB_X = KP / 8, NKB_H = HG / 8;
    constexpr int NE = TR / 2;  // accumulator elements per lane per gate
    const int unit = 32 * wq + lane_unit<TR>(lane);
    float b_r[TR == 32 ? 1 : 2], b_z[TR == 32 ? 1 : 2], b_in[TR == 32 ? 1 : 2], b_hn[TR == 32 ? 1 : 2];
#pragma unroll
    for (int t = 0; t < (TR == 32 ? 1 : 2); t++) {
        b_r[t] = bias[0 * HG + unit + 16 * t]; b_z[t] = bias[1 * HG + unit + 16 * t];
        b_in[t] = bias[2 * HG + unit + 16 * t]; b_hn[t] = bias[3 * HG + unit + 16 * t];
    }
    // register staging of x_t for this direction (256 threads per direction)
    constexpr int XR = ENC ? 1 : TR / 4;   // decoder: float4 per thread, rows tid_dir/64 + 4u
    constexpr int XE = TR / 8;             // encoder: floats per thread, rows tid_dir/32 + 8u (KPE = 32 padded features)
    f32x4 xr[XR];
    unsigned xe[ENC ? XE : 1];             // raw bytes: converted when they are written to LDS, so the load is not waited for
                                           // at the top of the step (a conversion right behind the load makes hipcc drain the
                                           // whole vmcnt queue there, weight-fragment ring included)
    unsigned xoff[ENC ? XE : 1];           // encoder: byte offset of (row's chunk, feature k) from the tile's first chunk
    const unsigned xd_g = (unsigned)((tid_dir >> 6) * KPD + (tid_dir & 63) * 4);   // decoder: float offset in x_src[t]
    const unsigned xd_l = (unsigned)((tid_dir >> 6) * LDX + (tid_dir & 63) * 4);   // ... and in xbuf
    const unsigned xe_l = (unsigned)((tid_dir >> 5) * LDX + (tid_dir & 31));
    const bool xe_valid = (tid_dir & 31) < FEAT;
    if constexpr (ENC) {
#pragma unroll
        for (int u = 0; u < XE; u++) {
            int64_t r = (tid_dir >> 5) + 8 * u;
            if (b0 + r >= a.B) r = a.B - 1 - b0;
            xoff[u] = (unsigned)(r * a.seq * FEAT) + (unsigned)((tid_dir & 31) < FEAT ? (tid_dir & 31) : FEAT - 1);   // padding lanes re-read feature 9
        }
    }
    const uint8_t* img0 = a.images + ((size_t)b0 * a.seq + win_start) * FEAT;  // uniform
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wp), xsr = make_rsrc(ENC ? (const void*)wp : (const void*)x_src), osr = make_rsrc(out_dst);
    auto x_load = [&](int t) {
        if constexpr (ENC) {
            const uint8_t* src = img0 + t * FEAT;
#pragma unroll
            for (int u = 0; u < XE; u++) xe[u] = src[xoff[u]];
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) xr[u] = buf_load4(xsr, xd_g * 4u, (unsigned)((t * TR + u * 4) * KPD * 4));
        }
    };
    // 16-row form: x_t tiles alternate between two LDS buffers (slot s & 1 holds step s): x_{s+1} is written right behind
    // step s's MFMAs, so ONE barrier per step publishes both the new h and the next x. (Two 32-row x buffers of both
    // directions do not fit the 160 KB: that form keeps one buffer and a second barrier.)
    constexpr bool XDB = TR == 16;
    auto x_store = [&](int slot) {
        float* xb = xbuf + (XDB ? slot : 0) * TR * LDXD;
        if constexpr (ENC) {
#pragma unroll
            for (int u = 0; u < XE; u++) (xb + u * 8 * LDX)[xe_l] = xe_valid ? (float)xe[u] : 0.0f;
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) *reinterpret_cast<f32x4*>(xb + u * 4 * LDX + xd_l) = xr[u];
        }
    };
    f32x4 bq[4][3];  // weight-fragment ring of mma3_ring; slots 0..2 start with k-blocks 0..2
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++) bq[q][nt] = buf_load4(wr, (unsigned)lane * 16u, (unsigned)((q * 3 + nt) * 1024));
    const unsigned h_l = (unsigned)(lane_row<TR>(lane) * LDH + unit);   // lane part of the h tile offset
    const unsigned o_l = (unsigned)(lane_row<TR>(lane) * KPD + unit);   // lane part of the output offset
    x_load(dir ? WIN - 1 : 0);
    x_store(0);
    __syncthreads();
    for (int s = 0; s < WIN; s++) {
        const int t = dir ? (WIN - 1 - s) : s;
        const int nxt = cur ^ 1;
        if (s + 1 < WIN) x_load(dir ? (WIN - 2 - s) : (s + 1));
        Gate<TR> ar, az, anx, anh;  // r and z accumulate both products; n keeps its x and h parts apart
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int t2 = TR == 32 ? 0 : e >> 2;  // which 16-unit tile the element belongs to
            gate_set<TR>(ar, e, b_r[t2]); gate_set<TR>(az, e, b_z[t2]); gate_set<TR>(anx, e, b_in[t2]); gate_set<TR>(anh, e, b_hn[t2]);
        }
        static_assert(NKB_X % 4 == 0 && NKB_H % 4 == 0, "mma3_ring works in groups of four k-blocks");
        mma3_ring<TR>(ar, az, anx, anh, xbuf + (XDB ? (s & 1) : 0) * TR * LDXD, LDX, NKB_X, hbuf + cur * TR * LDH, LDH, NKB_H, wr, bq, lane);
        if (XDB && s + 1 < WIN) x_store((s + 1) & 1);   // that buffer was last read in step s-1: every wave is past that step's barrier
        float* hn = hbuf + nxt * TR * LDH;
        const unsigned ob = (unsigned)((t * TR * KPD + dir * HG) * 4);  // uniform byte offset of (t, direction) in the scratch
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const float rg = sigmoidf_(gate_get<TR>(ar, e));
            const float zg = sigmoidf_(gate_get<TR>(az, e));
            const float ng = tanhf_(gate_get<TR>(anx, e) + rg * gate_get<TR>(anh, e));
            const float h = (1.0f - zg) * ng + zg * hst[e];
            hst[e] = h;
            (hn + elem_row<TR>(e) * LDH + elem_unit<TR>(e))[h_l] = h;
            buf_store1(h, osr, o_l * 4u, ob + (unsigned)((elem_row<TR>(e) * KPD + elem_unit<TR>(e)) * 4));
        }
        cur = nxt;
        lds_barrier();
        if (!XDB && s + 1 < WIN) {
            x_store(0);
            lds_barrier();
        }
    }
    __syncthreads();   // the window's outputs (global scratch) are read by the next phase of this workgroup
}

// ---- overlapped window form (16-row tiles, one wave per SIMD: the split form) --------------------------------------------
// With one wave per SIMD nothing hides the cell update, the h exchange and the barrier of a step. But two thirds of a
// decoder step's MFMAs (W_ix x_t) do not depend on h at all: the x-part of step s+1 is issued right behind the h-part of
// step s, and the cell update of step s runs in its shadow, one element per k-block group (VALU and LDS / global stores
// issue between the MFMAs of the matrix pipe). The packed weight stream of this form is [h-part | x-part] per step and the
// fragment ring wraps from the x-part into the next step's h-part.
// K blocks [KB0, KB0 + NKB) of a stream of NTOT blocks (the requests wrap around), A from one LDS tile; hook(i) runs behind
// the MFMAs of block i. On entry bq slots 0..2 hold blocks KB0..KB0+2, on exit the three blocks behind the range.
// D = ring depth (register sets; requests run D-1 blocks ahead): one wave per SIMD has no partner wave to cover the L2
// round trip of a fragment, so the decoder uses 8 sets (7 blocks = 2700 MFMA cycles ahead); NTOT % D == 0 keeps a block's
// set (stream position % D) the same across the wrap.
template <int KB0, int NKB, int NTOT, int D, typename Hook>
__device__ __forceinline__ void ring16(Gate<16>& g0, Gate<16>& g1, Gate<16>& g2, const float* __restrict__ A, int lda,
                                       __amdgpu_buffer_rsrc_t wr, f32x4 (&bq)[D][3], int lane, Hook&& hook) {
    static_assert(NTOT % D == 0 && (D & (D - 1)) == 0, "ring depth must divide the stream length");
    const float* ap = afrag_ptr<16>(A, lda, lane);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x2 aq[2];
    aq[0] = *reinterpret_cast<const f32x2*>(ap);
#pragma unroll
    for (int i = 0; i < NKB; i++) {
        {   // request block i + D - 1 of this call's range (wrapping into the stream's next blocks)
            const int kbv = (KB0 + i + D - 1) % NTOT;
            const int slot = (KB0 + i + D - 1) % D;
#pragma unroll
            for (int nt = 0; nt < 3; nt++) bq[slot][nt] = buf_load4(wr, lane16, (unsigned)((kbv * 3 + nt) * 1024));
        }
        if (i + 1 < NKB) aq[(i + 1) & 1] = *reinterpret_cast<const f32x2*>(ap + 8 * (i + 1));
        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the prefetches next to their uses
        {   // gate by gate (4 MFMAs each); hook(i, third) places a third of a cell-update slice behind each group, so its
            // vector instructions issue while the group's MFMAs occupy the matrix pipe
            const int slot = (KB0 + i) % D;
#pragma unroll
            for (int j = 0; j < 2; j++) gate_mma<16>(g0, aq[i & 1], bq[slot][0], j);
            hook(i, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; j++) gate_mma<16>(g1, aq[i & 1], bq[slot][1], j);
            hook(i, 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; j++) gate_mma<16>(g2, aq[i & 1], bq[slot][2], j);
            hook(i, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int KP, bool ENC>
__device__ __forceinline__ void gru_window_ovl(const GruArgs& a, int win_start, int dir, int wq, int lane, int tid_dir,
                                               int64_t b0, float* xbuf, float* hbuf, int& cur, float (&hst)[8],
                                               const float* wp, const float* bias, const float* x_src, float* out_dst) {
    constexpr int TR = 16;
    constexpr int LDX = KP + 4;
    constexpr int NKB_X = KP / 8, NKB_H = HG / 8, NTOT = NKB_X + NKB_H;
    constexpr int NE = 8;
    const int unit = 32 * wq + lane_unit<TR>(lane);
    float b_r[2], b_z[2], b_in[2], b_hn[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        b_r[t] = bias[0 * HG + unit + 16 * t]; b_z[t] = bias[1 * HG + unit + 16 * t];
        b_in[t] = bias[2 * HG + unit + 16 * t]; b_hn[t] = bias[3 * HG + unit + 16 * t];
    }
    constexpr int XR = ENC ? 1 : TR / 4;   // decoder: float4 per thread, rows tid_dir/64 + 4u
    constexpr int XE = TR / 8;             // encoder: floats per thread, rows tid_dir/32 + 8u
    f32x4 xr[XR];
    unsigned xe[ENC ? XE : 1];             // raw bytes: converted when they are written to LDS, so the load is not waited for
                                           // at the top of the step (a conversion right behind the load makes hipcc drain the
                                           // whole vmcnt queue there, weight-fragment ring included)
    unsigned xoff[ENC ? XE : 1];
    const unsigned xd_g = (unsigned)((tid_dir >> 6) * KPD + (tid_dir & 63) * 4);
    const unsigned xd_l = (unsigned)((tid_dir >> 6) * LDX + (tid_dir & 63) * 4);
    const unsigned xe_l = (unsigned)((tid_dir >> 5) * LDX + (tid_dir & 31));
    const bool xe_valid = (tid_dir & 31) < FEAT;
    if constexpr (ENC) {
#pragma unroll
        for (int u = 0; u < XE; u++) {
            int64_t r = (tid_dir >> 5) + 8 * u;
            if (b0 + r >= a.B) r = a.B - 1 - b0;
            xoff[u] = (unsigned)(r * a.seq * FEAT) + (unsigned)((tid_dir & 31) < FEAT ? (tid_dir & 31) : FEAT - 1);   // padding lanes re-read feature 9
        }
    }
    const uint8_t* img0 = a.images + ((size_t)b0 * a.seq + win_start) * FEAT;  // uniform
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wp), xsr = make_rsrc(ENC ? (const void*)wp : (const void*)x_src), osr = make_rsrc(out_dst);
    auto x_load = [&](int s) {   // the input of step s (time index by direction)
        const int t = dir ? (WIN - 1 - s) : s;
        if constexpr (ENC) {
            const uint8_t* src = img0 + t * FEAT;
#pragma unroll
            for (int u = 0; u < XE; u++) xe[u] = src[xoff[u]];
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) xr[u] = buf_load4(xsr, xd_g * 4u, (unsigned)((t * TR + u * 4) * KPD * 4));
        }
    };
    auto x_store = [&](int slot) {
        float* xb = xbuf + slot * TR * LDXD;
        if constexpr (ENC) {
#pragma unroll
            for (int u = 0; u < XE; u++) (xb + u * 8 * LDX)[xe_l] = xe_valid ? (float)xe[u] : 0.0f;
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) *reinterpret_cast<f32x4*>(xb + u * 4 * LDX + xd_l) = xr[u];
        }
    };
    constexpr int D = NTOT % 8 == 0 ? 8 : 4;   // decoder: 48 blocks per step, 8 register sets; encoder: 20 blocks, 4 sets
    f32x4 bq[D][3];  // the ring starts with the first x-part blocks (the prologue runs the x-part of step 0)
#pragma unroll
    for (int q = 0; q < D - 1; q++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++)
            bq[(NKB_H + q) % D][nt] = buf_load4(wr, (unsigned)lane * 16u, (unsigned)((((NKB_H + q) % NTOT) * 3 + nt) * 1024));
    const unsigned h_l = (unsigned)(lane_row<TR>(lane) * LDH + unit);
    const unsigned o_l = (unsigned)(lane_row<TR>(lane) * KPD + unit);
    auto nohook = [](int, int) {};
    // prologue: x_0 and x_1 into the two slots, x-part of step 0
    x_load(0);
    x_store(0);
    x_load(1);
    x_store(1);
    __syncthreads();
    Gate<TR> xr_, xz_, xn_;   // x-part (+ biases) of the CURRENT step
#pragma unroll
    for (int e = 0; e < NE; e++) { gate_set<TR>(xr_, e, b_r[e >> 2]); gate_set<TR>(xz_, e, b_z[e >> 2]); gate_set<TR>(xn_, e, b_in[e >> 2]); }
    ring16<NKB_H, NKB_X, NTOT, D>(xr_, xz_, xn_, xbuf, LDX, wr, bq, lane, nohook);
#ifdef PV_GRU_STAMPS
    unsigned long long st0, st1, acc_t[4] = {0, 0, 0, 0};
#define GSTAMP(i) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st1) :: "memory"); acc_t[i] += st1 - st0; st0 = st1; }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st0) :: "memory");
#else
#define GSTAMP(i)
#endif
    for (int s = 0; s < WIN; s++) {
        const int t = dir ? (WIN - 1 - s) : s;
        const int nxt = cur ^ 1;
        if (s + 2 < WIN) x_load(s + 2);
        GSTAMP(0)
        Gate<TR> hr, hz, hn_;   // h-part of this step
#pragma unroll
        for (int e = 0; e < NE; e++) { gate_set<TR>(hr, e, 0.0f); gate_set<TR>(hz, e, 0.0f); gate_set<TR>(hn_, e, b_hn[e >> 2]); }
        ring16<0, NKB_H, NTOT, D>(hr, hz, hn_, hbuf + cur * TR * LDH, LDH, wr, bq, lane, nohook);
        GSTAMP(1)
        float* hnx = hbuf + nxt * TR * LDH;
        const unsigned ob = (unsigned)((t * TR * KPD + dir * HG) * 4);
        // the cell update of one element in three parts (state in c_r, c_z, c_n), so that each part fits behind one gate
        // group of MFMAs
        float c_r = 0.0f, c_z = 0.0f, c_n = 0.0f;
        auto cell3 = [&](int e, int third) {
            if (third == 0) {
                c_r = sigmoidf_(gate_get<TR>(xr_, e) + gate_get<TR>(hr, e));
                c_z = __expf(-(gate_get<TR>(xz_, e) + gate_get<TR>(hz, e)));
            } else if (third == 1) {
                c_z = rcpf_(1.0f + c_z);
                c_n = tanhf_(gate_get<TR>(xn_, e) + c_r * gate_get<TR>(hn_, e));
            } else {
                const float h = (1.0f - c_z) * c_n + c_z * hst[e];
                hst[e] = h;
                (hnx + elem_row<TR>(e) * LDH + elem_unit<TR>(e))[h_l] = h;
                buf_store1(h, osr, o_l * 4u, ob + (unsigned)((elem_row<TR>(e) * KPD + elem_unit<TR>(e)) * 4));
            }
        };
        auto cell = [&](int e) { cell3(e, 0); cell3(e, 1); cell3(e, 2); };
        if (s + 1 < WIN) {
            Gate<TR> nr, nz, nn;   // x-part of the NEXT step, issued now; the cell update of this step runs behind its MFMAs
#pragma unroll
            for (int e = 0; e < NE; e++) { gate_set<TR>(nr, e, b_r[e >> 2]); gate_set<TR>(nz, e, b_z[e >> 2]); gate_set<TR>(nn, e, b_in[e >> 2]); }
            constexpr int EPB = NKB_X >= NE ? 1 : NE / NKB_X;   // elements per hooked block
            constexpr int STRIDE = NKB_X >= NE ? NKB_X / NE : 1; // hooked block every STRIDE blocks
            ring16<NKB_H, NKB_X, NTOT, D>(nr, nz, nn, xbuf + ((s + 1) & 1) * TR * LDXD, LDX, wr, bq, lane, [&](int i, int third) {
                if (i % STRIDE == 0) {
                    if constexpr (EPB == 1) {
                        cell3(i / STRIDE, third);
                    } else {   // encoder: four blocks for eight elements, whole elements behind the gate groups
                        if (third < EPB) cell((i / STRIDE) * EPB + third);
                    }
                }
            });
            xr_ = nr; xz_ = nz; xn_ = nn;
        } else {
#pragma unroll
            for (int e = 0; e < NE; e++) cell(e);
        }
        GSTAMP(2)
        if (s + 2 < WIN) x_store(s & 1);   // x_{s+2} into the slot x_s left
        cur = nxt;
        lds_barrier();
        GSTAMP(3)
    }
#ifdef PV_GRU_STAMPS
    if (a.pair_flags && blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 4; i++) atomicAdd((unsigned long long*)(a.pair_flags + 64) + (ENC ? 0 : 4) + i, acc_t[i]);
#endif
#undef GSTAMP
    __syncthreads();   // the window's outputs (global scratch) are read by the next phase (after the hand-off)
}

template <int TR, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 256 : 512, SPLIT ? 1 : 2) void k_gru_p2(GruArgs a) {
    extern __shared__ float smem[];
    // per direction: hbuf [2][TR][LDH], xbuf [2 (16-row form) or 1][TR][LDXD]
    constexpr int NTHR = SPLIT ? 256 : 512;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR: bases derived from it stay scalar
    const int dir = SPLIT ? (int)(blockIdx.x & 1) : (wv >> 2), wq = wv & 3, tid_dir = tid & 255;
    constexpr int NE = TR / 2;
    float* hbuf = smem + (SPLIT ? 0 : dir) * (2 * TR * LDH + (TR == 16 ? 2 : 1) * TR * LDXD);
    float* xbuf = hbuf + 2 * TR * LDH;
    const int tile = SPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    if (SPLIT && a.drop_part >= 0 && dir == (a.drop_part & 1)) return;   // diagnostic: a partner that never shows up
    int* my_flag = SPLIT ? a.pair_flags + 2 * tile + dir : nullptr;
    const int* their_flag = SPLIT ? a.pair_flags + 2 * tile + (dir ^ 1) : nullptr;
    int handoffs = 0;
    const int64_t b0 = (int64_t)tile * TR;
    float* enc_out = a.enc_out + (size_t)tile * WIN * TR * KPD;
    float* dec_out = a.dec_out + (size_t)tile * WIN * TR * KPD;
    const float* enc_wp = a.enc_wp + ((size_t)(dir * 4 + wq) * ((KPE + HG) / 8)) * 3 * 256;
    const float* dec_wp = a.dec_wp + ((size_t)(dir * 4 + wq) * ((KPD + HG) / 8)) * 3 * 256;
    const float* enc_bias = a.enc_bias + dir * 4 * HG;
    const float* dec_bias = a.dec_bias + dir * 4 * HG;

    for (int i = tid_dir; i < 2 * TR * LDH; i += 256) hbuf[i] = 0.0f;  // hidden = zeros (predict.py:55)
    float hst[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) hst[e] = 0.0f;
    int cur = 0;
    __syncthreads();
    const int unit0 = 32 * wq + lane_unit<TR>(lane);
    if (a.hidden_in) {  // TransducerGRU.forward(x, hidden): hidden [B,2,H], index 0 = forward (simple_model.py:28)
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int row = lane_row<TR>(lane) + elem_row<TR>(e), unit = unit0 + elem_unit<TR>(e);
            int64_t b = b0 + row;
            if (b >= a.B) b = a.B - 1;
            const float h = a.hidden_in[(b * 2 + dir) * HG + unit];
            hst[e] = h;
            hbuf[row * LDH + unit] = h;
        }
        __syncthreads();
    }

    for (int w = 0; w < a.nwin; w++) {
        const int ws = w * JUMP;
        if constexpr (SPLIT) gru_window_ovl<KPE, true>(a, ws, dir, wq, lane, tid_dir, b0, xbuf, hbuf, cur, hst, enc_wp, enc_bias, nullptr, enc_out);
        else gru_window<TR, KPE, true>(a, ws, dir, wq, lane, tid_dir, b0, xbuf, hbuf, cur, hst, enc_wp, enc_bias, nullptr, enc_out);
        if constexpr (SPLIT) pair_handoff(my_flag, their_flag, ++handoffs, tid, a.err, a.spin_limit << 8);   // both halves of the encoder output are there
        // decoder h0 = encoder final state of the same direction: hst / hbuf[cur] simply carry over
        if constexpr (SPLIT) gru_window_ovl<KPD, false>(a, ws, dir, wq, lane, tid_dir, b0, xbuf, hbuf, cur, hst, dec_wp, dec_bias, enc_out, dec_out);
        else gru_window<TR, KPD, false>(a, ws, dir, wq, lane, tid_dir, b0, xbuf, hbuf, cur, hst, dec_wp, dec_bias, enc_out, dec_out);
        if constexpr (SPLIT) pair_handoff(my_flag, their_flag, ++handoffs, tid, a.err, a.spin_limit << 8);   // both halves of the decoder output; the partner is
                                                                                    // done reading the encoder output as well
        // dense1 + softmax + accumulate over the window (predict.py:70-89); TR*100 (row, t) pairs. Split form: the workgroup of
        // direction d owns the columns of parity d (ws is even), for acc, logits and labels alike, so no column is shared.
        for (int p = tid; p < TR * WIN; p += NTHR) {
            const int t = p / TR, row = p - t * TR;
            const int64_t b = b0 + row;
            if (b >= a.B || (SPLIT && (t & 1) != dir)) continue;
            const float* d = dec_out + ((size_t)t * TR + row) * KPD;
            float lg[NCLS];
#pragma unroll
            for (int c = 0; c < NCLS; c++) lg[c] = a.dense_b[c];
            for (int k = 0; k < KPD; k += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(d + k);
#pragma unroll
                for (int c = 0; c < NCLS; c++) {
                    const f32x4 wv4 = *reinterpret_cast<const f32x4*>(a.dense_w + c * KPD + k);
                    lg[c] += v[0] * wv4[0] + v[1] * wv4[1] + v[2] * wv4[2] + v[3] * wv4[3];
                }
            }
            float m = lg[0];
#pragma unroll
            for (int c = 1; c < NCLS; c++) m = fmaxf(m, lg[c]);
            float e[NCLS], sum = 0.0f;
#pragma unroll
            for (int c = 0; c < NCLS; c++) { e[c] = expf(lg[c] - m); sum += e[c]; }
            const float inv = 1.0f / sum;
            float* ac = a.acc + ((size_t)b * a.seq + ws + t) * NCLS;
#pragma unroll
            for (int c = 0; c < NCLS; c++) ac[c] += e[c] * inv;
            if (a.logits && w == a.nwin - 1) {
#pragma unroll
                for (int c = 0; c < NCLS; c++) a.logits[((size_t)b * WIN + t) * NCLS + c] = lg[c];
            }
        }
        __syncthreads();
    }
    if (a.hidden_out) {  // final decoder state (the next window's encoder h0, simple_model.py:41)
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int64_t b = b0 + lane_row<TR>(lane) + elem_row<TR>(e);
            if (b < a.B) a.hidden_out[(b * 2 + dir) * HG + unit0 + elem_unit<TR>(e)] = hst[e];
        }
    }
    // labels = argmax over the 5 classes, first maximum wins (torch.max, predict.py:91)
    for (int p = tid; a.labels && p < TR * a.seq; p += NTHR) {
        const int row = p / a.seq, pos = p - row * a.seq;
        const int64_t b = b0 + row;
        if (b >= a.B || (SPLIT && (pos & 1) != dir)) continue;
        const float* ac = a.acc + ((size_t)b * a.seq + pos) * NCLS;
        int best = 0;
        float bv = ac[0];
#pragma unroll
        for (int c = 1; c < NCLS; c++) if (ac[c] > bv) { bv = ac[c]; best = c; }
        a.labels[(size_t)b * a.seq + pos] = (uint8_t)best;
    }
}


// ---- unit-split form: the 128 hidden units of a (tile, direction) on TWO workgroups (small batches) ----------------------
// Same idea as k_lstm_split (rnn_kernels.hip): the launch is a chain of 3800 dependent steps, so what counts is the MFMA time
// of ONE workgroup per step. Workgroup = (16-row tile, direction, half of the units), 4 waves, one per SIMD; a wave owns 16
// units x {r, z, n} (three 16x16 accumulators for the x-part, three for the h-part). Every step the two halves swap their
// 64 new h values per row: data-tagged 8-byte pairs {h, tag} written through (sc1) into the slot (step parity) of a per-(tile,
// direction) buffer, polled by the twin thread of the partner with L1-bypassing loads AFTER that thread's x-part MFMAs of the
// next step (two thirds of the decoder's MFMAs do not depend on h): no flag, no fence, no store drain. The four workgroups of
// a tile meet only where the model joins directions (decoder input, dense1): counters + agent-scope release / acquire, 38
// times per launch. Weight stream of a wave: [pair of k-blocks][gate][lane][4] = {kb even j0, j1, kb odd j0, j1} with
// W[gate*HG + 64*half + 16*wave + (lane&15)][16p + 8*(kb&1) + 2*(lane>>4) + j], stream order [h | x], through a ring of D
// register sets requested D-1 pairs ahead (a pair is 12 MFMAs = 384 cycles).
constexpr int US_HX_QUADS = 2 * 2 * 2 * 256;   // 16-byte granules per (tile, direction): [2 slots][2 halves][2][256 threads]
constexpr int US_FLAG_STRIDE = 32;
constexpr int US_SC1 = 16;                     // cache policy bit 4 = sc1 (gfx94x / gfx950)

template <int P0, int NP, int NTOTP, int D>
__device__ __forceinline__ void ring_us(f32x4& gr, f32x4& gz, f32x4& gn, const float* __restrict__ A, int lda,
                                        __amdgpu_buffer_rsrc_t wr, f32x4 (&bq)[D][3], int lane) {
    static_assert(NTOTP % D == 0, "ring depth must divide the stream length");
    const float* ap = afrag_ptr<16>(A, lda, lane);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x2 aq[2][2];
    aq[0][0] = *reinterpret_cast<const f32x2*>(ap);
    aq[0][1] = *reinterpret_cast<const f32x2*>(ap + 8);
#pragma unroll
    for (int i = 0; i < NP; i++) {
        {
            const int pv = (P0 + i + D - 1) % NTOTP, slot = (P0 + i + D - 1) % D;
#pragma unroll
            for (int g = 0; g < 3; g++) bq[slot][g] = buf_load4(wr, lane16, (unsigned)((pv * 3 + g) * 1024));
        }
        if (i + 1 < NP) {
            aq[(i + 1) & 1][0] = *reinterpret_cast<const f32x2*>(ap + 16 * (i + 1));
            aq[(i + 1) & 1][1] = *reinterpret_cast<const f32x2*>(ap + 16 * (i + 1) + 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const int slot = (P0 + i) % D;
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float av = aq[i & 1][kh][j];
                    gr = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[slot][0][2 * kh + j], gr, 0, 0, 0);
                    gz = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[slot][1][2 * kh + j], gz, 0, 0, 0);
                    gn = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bq[slot][2][2 * kh + j], gn, 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// counters + fences among the four workgroups of a tile (layer boundaries only)
__device__ __forceinline__ void quad_handoff(int* flags_tile, int me, int value, int tid, int* err, int limit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flags_tile + me * US_FLAG_STRIDE, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 1; k < 4; k++) {
            const int* f = flags_tile + ((me + k) & 3) * US_FLAG_STRIDE;
            int spins = 0;
            while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > limit) { atomicAdd(err, 1); break; }   // bounded: a lost partner ends in POISONED results, never in a hung device
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// one GRU layer over one 100-column window for this workgroup's (direction, half). `step` counts the steps of the launch (tags).
template <int KP, bool ENC, int D>
__device__ __forceinline__ void gru_window_us(const GruArgs& a, int win_start, int dir, int half, int wv, int lane, int tid,
                                              int64_t b0, float* xbuf, float* hbuf, int& cur, float (&hst)[4], unsigned& step,
                                              const float* wp, const float* bias, const float* x_src, float* out_dst,
                                              __amdgpu_buffer_rsrc_t hsr) {
    constexpr int TR = 16;
    constexpr int LDX = KP + 4;
    constexpr int NP_X = KP / 16, NP_H = HG / 16, NTOTP = NP_X + NP_H;
    static_assert(NTOTP % D == 0, "ring depth");
    const int unit = 64 * half + 16 * wv + (lane & 15);
    const int rowg = lane >> 4;
    const float b_r = bias[0 * HG + unit], b_z = bias[1 * HG + unit], b_in = bias[2 * HG + unit], b_hn = bias[3 * HG + unit];
    constexpr int XR = ENC ? 1 : 4;     // decoder: float4 per thread, rows tid/64 + 4u
    constexpr int XE = 2;               // encoder: bytes per thread, rows tid/32 + 8u
    f32x4 xr[XR];
    unsigned xe[ENC ? XE : 1];
    unsigned xoff[ENC ? XE : 1];
    const unsigned xd_g = (unsigned)((tid >> 6) * KPD + (tid & 63) * 4);
    const unsigned xd_l = (unsigned)((tid >> 6) * LDX + (tid & 63) * 4);
    const unsigned xe_l = (unsigned)((tid >> 5) * LDX + (tid & 31));
    const bool xe_valid = (tid & 31) < FEAT;
    if constexpr (ENC) {
#pragma unroll
        for (int u = 0; u < XE; u++) {
            int64_t r = (tid >> 5) + 8 * u;
            if (b0 + r >= a.B) r = a.B - 1 - b0;
            xoff[u] = (unsigned)(r * a.seq * FEAT) + (unsigned)((tid & 31) < FEAT ? (tid & 31) : FEAT - 1);
        }
    }
    const uint8_t* img0 = a.images + ((size_t)b0 * a.seq + win_start) * FEAT;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wp), xsr = make_rsrc(ENC ? (const void*)wp : (const void*)x_src), osr = make_rsrc(out_dst);
    auto x_load = [&](int s) {
        const int t = dir ? (WIN - 1 - s) : s;
        if constexpr (ENC) {
            const uint8_t* src = img0 + t * FEAT;
#pragma unroll
            for (int u = 0; u < XE; u++) xe[u] = src[xoff[u]];
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) xr[u] = buf_load4(xsr, xd_g * 4u, (unsigned)((t * TR + u * 4) * KPD * 4));
        }
    };
    auto x_store = [&](int slot) {
        float* xb = xbuf + slot * TR * LDXD;
        if constexpr (ENC) {
#pragma unroll
            for (int u = 0; u < XE; u++) (xb + u * 8 * LDX)[xe_l] = xe_valid ? (float)xe[u] : 0.0f;
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) *reinterpret_cast<f32x4*>(xb + u * 4 * LDX + xd_l) = xr[u];
        }
    };
    f32x4 bq[D][3];   // the stream order is [h | x]; the prologue runs the x-part of step 0, so the ring starts at pair NP_H
#pragma unroll
    for (int q = 0; q < D - 1; q++)
#pragma unroll
        for (int g = 0; g < 3; g++)
            bq[(NP_H + q) % D][g] = buf_load4(wr, (unsigned)lane * 16u, (unsigned)((((NP_H + q) % NTOTP) * 3 + g) * 1024));
    const unsigned h_l = (unsigned)(4 * rowg * LDH + unit);
    const unsigned o_l = (unsigned)(4 * rowg * KPD + unit);
    auto hx_off = [](int slot, int hf, int g) { return (unsigned)((((slot * 2 + hf) * 2 + g) * 256) * 16); };
    x_load(0);
    x_store(0);
    x_load(1);
    x_store(1);
    __syncthreads();
    f32x4 xr_ = {b_r, b_r, b_r, b_r}, xz_ = {b_z, b_z, b_z, b_z}, xn_ = {b_in, b_in, b_in, b_in};   // x-part (+ biases) of the CURRENT step
    ring_us<NP_H, NP_X, NTOTP, D>(xr_, xz_, xn_, xbuf, LDX, wr, bq, lane);
    for (int s = 0; s < WIN; s++) {
        const int t = dir ? (WIN - 1 - s) : s;
        const int nxt = cur ^ 1;
        if (s + 2 < WIN) x_load(s + 2);
        if (step > 0) {
            // ---- the other half of h (previous step of the launch): the pairs of this thread's twin in the partner workgroup ----
            const unsigned want = a.tag_base + step;
            const int ps = (int)((step - 1) & 1);
            u32x4 fq[2];
            int spins = 0;
            while (true) {   // bounded: a lost partner ends in wrong results, never in a hung device
                bool ok = true;
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    fq[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(hsr, (unsigned)tid * 16u, hx_off(ps, half ^ 1, g), US_SC1));
                    ok = ok && fq[g][1] == want && fq[g][3] == want;
                }
                asm volatile("" ::: "memory");   // the loads are repeated, not hoisted
                if (ok) break;
                if (++spins > a.spin_limit) { atomicAdd(a.err, 1); break; }
            }
            float* dst = hbuf + cur * TR * LDH + 4 * rowg * LDH + 64 * (half ^ 1) + 16 * wv + (lane & 15);
#pragma unroll
            for (int g = 0; g < 2; g++) {
                // (elements go through scalars: hipcc 7.2 compiles __builtin_bit_cast(float, vec[2]) as element 0)
                const unsigned lo = fq[g][0], hi = fq[g][2];
                dst[(2 * g + 0) * LDH] = __builtin_bit_cast(float, lo);
                dst[(2 * g + 1) * LDH] = __builtin_bit_cast(float, hi);
            }
            lds_barrier();
        }
        f32x4 hr = {0.0f, 0.0f, 0.0f, 0.0f}, hz = {0.0f, 0.0f, 0.0f, 0.0f}, hn_ = {b_hn, b_hn, b_hn, b_hn};
        ring_us<0, NP_H, NTOTP, D>(hr, hz, hn_, hbuf + cur * TR * LDH, LDH, wr, bq, lane);
        // ---- cell update; the new h leaves first (the partner waits for it) -------------------------------------------
        float* hnx = hbuf + nxt * TR * LDH;
        float hv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float r = sigmoidf_(xr_[i] + hr[i]);
            const float z = sigmoidf_(xz_[i] + hz[i]);
            const float n = tanhf_(xn_[i] + r * hn_[i]);
            hv[i] = (1.0f - z) * n + z * hst[i];
            hst[i] = hv[i];
        }
        {
            const unsigned tag = a.tag_base + step + 1;
            const int slot = (int)(step & 1);
#pragma unroll
            for (int i = 0; i < 4; i++) {   // 8-byte stores (a 16-byte buffer store with an SGPR soffset has an unpadded data hazard on gfx950)
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 pr = {__builtin_bit_cast(unsigned, hv[i]), tag};
                __builtin_amdgcn_raw_buffer_store_b64(pr, hsr, (unsigned)tid * 16u + 8u * (i & 1), hx_off(slot, half, i >> 1), US_SC1);
            }
        }
        const unsigned ob = (unsigned)((t * TR * KPD + dir * HG) * 4);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            (hnx + i * LDH)[h_l] = hv[i];
            buf_store1(hv[i], osr, o_l * 4u, ob + (unsigned)(i * KPD * 4));
        }
        step++;
        if (s + 1 < WIN) {   // x-part of the NEXT step: the exchange above travels meanwhile
            f32x4 nr = {b_r, b_r, b_r, b_r}, nz = {b_z, b_z, b_z, b_z}, nn = {b_in, b_in, b_in, b_in};
            ring_us<NP_H, NP_X, NTOTP, D>(nr, nz, nn, xbuf + ((s + 1) & 1) * TR * LDXD, LDX, wr, bq, lane);
            xr_ = nr; xz_ = nz; xn_ = nn;
        }
        if (s + 2 < WIN) x_store(s & 1);   // x_{s+2} into the slot x_s left: its last reader was the x-part issued in step s-1
        cur = nxt;
        lds_barrier();
    }
    __syncthreads();   // the window's outputs (global scratch) are read by the next phase (after the hand-off)
}

// Behind every P2 launch: bumps the launch counter of the unit-split form (a replayed hipGraph advances the exchange tags like
// eager launches do) and, while exchange time-outs are pending (a poll or a layer hand-off of a split form gave up: some
// workgroup computed on stale state, and its partners on what it sent afterwards), POISONS what the launch wrote: labels 255,
// accumulated softmax / logits / hidden state NaN. The word stays set until the host acknowledges it
// (pv_rnn_exchange_timeouts, or the host-buffer entry points, which return PV_ERR_STATE).
struct GruFinishArgs {
    unsigned* epoch;      // or NULL
    const int* err;
    uint8_t* labels; int64_t n_labels;
    float* f[3]; int64_t nf[3];   // acc, logits, hidden_out (NULL / 0 when absent)
};
__global__ __launch_bounds__(256) void k_gru_finish(GruFinishArgs a) {
    if (a.epoch && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.epoch, 1u);
#ifdef PV_DBG_NOFINISH
    return;
#endif
    if (a.err[0] == 0) return;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = i0; a.labels && i < a.n_labels; i += stride) a.labels[i] = 255;
    const float nanv = __builtin_nanf("");
    for (int k = 0; k < 3; k++)
        for (int64_t i = i0; a.f[k] && i < a.nf[k]; i += stride) a.f[k][i] = nanv;
}

__global__ __launch_bounds__(256, 1) void k_gru_us(GruArgs a_in) {
    GruArgs a = a_in;
    a.tag_base = a_in.epoch[0] * 4096u;   // 3800 steps per launch; unsigned wrap-around is harmless (tags are compared for equality)
    extern __shared__ float smem[];
    constexpr int TR = 16;
    float* hbuf = smem;                    // [2][16][LDH]: all 128 units
    float* xbuf = hbuf + 2 * TR * LDH;     // [2][16][LDXD]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the four workgroups of a tile sit on ONE XCD (they exchange through its L2)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int sub = q & 3, dir = sub & 1, half = sub >> 1;
    const int tile = (q >> 2) * 8 + xcd;
    const int n_tiles = (int)((a.B + TR - 1) / TR);
    if (tile >= n_tiles) return;           // whole tiles leave together
    if (sub == a.drop_part) return;        // diagnostic: a partner that never shows up (the others' polls time out and say so)
    const int64_t b0 = (int64_t)tile * TR;
    float* enc_out = a.enc_out + (size_t)tile * WIN * TR * KPD;
    float* dec_out = a.dec_out + (size_t)tile * WIN * TR * KPD;
    constexpr int NPE = (KPE + HG) / 16, NPD = (KPD + HG) / 16;
    const float* enc_wp = a.enc_wp + ((size_t)((dir * 2 + half) * 4 + wv) * NPE) * 3 * 256;
    const float* dec_wp = a.dec_wp + ((size_t)((dir * 2 + half) * 4 + wv) * NPD) * 3 * 256;
    const float* enc_bias = a.enc_bias + dir * 4 * HG;
    const float* dec_bias = a.dec_bias + dir * 4 * HG;
    const __amdgpu_buffer_rsrc_t hsr = make_rsrc(a.hx + ((size_t)tile * 2 + dir) * US_HX_QUADS);
    int* flags_tile = a.quad_flags + (size_t)tile * 4 * US_FLAG_STRIDE;
    int handoffs = 0;
    unsigned step = 0;

    for (int i = tid; i < 2 * TR * LDH; i += 256) hbuf[i] = 0.0f;  // hidden = zeros (predict.py:55)
    float hst[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int cur = 0;
    __syncthreads();
    const int unit = 64 * half + 16 * wv + (lane & 15);
    if (a.hidden_in) {  // TransducerGRU.forward(x, hidden): every workgroup needs all 128 units of its direction
        for (int p = tid; p < TR * HG; p += 256) {
            const int row = p / HG, u = p - row * HG;
            int64_t b = b0 + row;
            if (b >= a.B) b = a.B - 1;
            hbuf[row * LDH + u] = a.hidden_in[(b * 2 + dir) * HG + u];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) hst[i] = hbuf[(4 * (lane >> 4) + i) * LDH + unit];
    }

    for (int w = 0; w < a.nwin; w++) {
        const int ws = w * JUMP;
        gru_window_us<KPE, true, NPE>(a, ws, dir, half, wv, lane, tid, b0, xbuf, hbuf, cur, hst, step, enc_wp, enc_bias, nullptr, enc_out, hsr);
        quad_handoff(flags_tile, sub, ++handoffs, tid, a.err, a.spin_limit << 8);   // all four quarters of the encoder output are there
        gru_window_us<KPD, false, 8>(a, ws, dir, half, wv, lane, tid, b0, xbuf, hbuf, cur, hst, step, dec_wp, dec_bias, enc_out, dec_out, hsr);
        quad_handoff(flags_tile, sub, ++handoffs, tid, a.err, a.spin_limit << 8);   // the decoder output; everybody is done reading the encoder output too
        // dense1 + softmax + accumulate (predict.py:70-89): workgroup `sub` owns the columns with (position & 3) == sub, for
        // acc, logits and labels alike (windows overlap, positions do not move), so no column is shared
        for (int p = tid; p < TR * WIN; p += 256) {
            const int t = p / TR, row = p - t * TR;
            const int64_t b = b0 + row;
            if (b >= a.B || ((ws + t) & 3) != sub) continue;
            const float* d = dec_out + ((size_t)t * TR + row) * KPD;
            float lg[NCLS];
#pragma unroll
            for (int c = 0; c < NCLS; c++) lg[c] = a.dense_b[c];
            for (int k = 0; k < KPD; k += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(d + k);
#pragma unroll
                for (int c = 0; c < NCLS; c++) {
                    const f32x4 wv4 = *reinterpret_cast<const f32x4*>(a.dense_w + c * KPD + k);
                    lg[c] += v[0] * wv4[0] + v[1] * wv4[1] + v[2] * wv4[2] + v[3] * wv4[3];
                }
            }
            float m = lg[0];
#pragma unroll
            for (int c = 1; c < NCLS; c++) m = fmaxf(m, lg[c]);
            float e[NCLS], sum = 0.0f;
#pragma unroll
            for (int c = 0; c < NCLS; c++) { e[c] = expf(lg[c] - m); sum += e[c]; }
            const float inv = 1.0f / sum;
            float* ac = a.acc + ((size_t)b * a.seq + ws + t) * NCLS;
#pragma unroll
            for (int c = 0; c < NCLS; c++) ac[c] += e[c] * inv;
            if (a.logits && w == a.nwin - 1) {
#pragma unroll
                for (int c = 0; c < NCLS; c++) a.logits[((size_t)b * WIN + t) * NCLS + c] = lg[c];
            }
        }
        __syncthreads();
    }
    if (a.hidden_out) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int64_t b = b0 + 4 * (lane >> 4) + i;
            if (b < a.B) a.hidden_out[(b * 2 + dir) * HG + unit] = hst[i];
        }
    }
    for (int p = tid; a.labels && p < TR * a.seq; p += 256) {
        const int row = p / a.seq, pos = p - row * a.seq;
        const int64_t b = b0 + row;
        if (b >= a.B || (pos & 3) != sub) continue;
        const float* ac = a.acc + ((size_t)b * a.seq + pos) * NCLS;
        int best = 0;
        float bv = ac[0];
#pragma unroll
        for (int c = 1; c < NCLS; c++) if (ac[c] > bv) { bv = ac[c]; best = c; }
        a.labels[(size_t)b * a.seq + pos] = (uint8_t)best;
    }
}

// unit-split form: [dir][half][wave][pair of k-blocks][gate r,z,n][lane][4], stream order [h | x] (see k_gru_us)
void pack_gru_us(const pv_rnn_dir* dirs, int K, int KP, std::vector<float>& wp) {
    const int np_h = HG / 16, np = (KP + HG) / 16;
    wp.assign((size_t)2 * 2 * 4 * np * 3 * 256, 0.0f);
    for (int d = 0; d < 2; d++) {
        auto wval = [&](int n, int kpos) -> float {   // kpos: position in the [h | x] stream
            if (kpos < HG) return dirs[d].w_hh[(size_t)n * HG + kpos];
            const int k = kpos - HG;
            return k < K ? dirs[d].w_ih[(size_t)n * K + k] : 0.0f;
        };
        (void)np_h;
        for (int hf = 0; hf < 2; hf++)
            for (int w = 0; w < 4; w++)
                for (int pp = 0; pp < np; pp++)
                    for (int g = 0; g < 3; g++)
                        for (int lane = 0; lane < 64; lane++) {
                            float* dst = &wp[(((((size_t)(d * 2 + hf) * 4 + w) * np + pp) * 3 + g) * 64 + lane) * 4];
                            const int n = g * HG + 64 * hf + 16 * w + (lane & 15);
                            for (int kh = 0; kh < 2; kh++)
                                for (int j = 0; j < 2; j++) dst[2 * kh + j] = wval(n, 16 * pp + 8 * kh + 2 * (lane >> 4) + j);
                        }
    }
}
constexpr size_t LDS_US = (size_t)(2 * 16 * LDH + 2 * 16 * LDXD) * sizeof(float);

// Packed weight stream of one wave: [k-block of 8][gate r,z,n][lane][4]; K = [x (padded to KP) | h].
//  32-row form: lane -> gate column 32w + (lane&31), the four values are k = 8kb + 4*(lane>>5) + j, j = 0..3
//  16-row form: lane -> gate columns 32w + (lane&15) (tile 0) and 32w + 16 + (lane&15) (tile 1),
//               the four values are {tile0 j0, tile0 j1, tile1 j0, tile1 j1} with k = 8kb + 2*(lane>>4) + j
// hx_order: the stream starts with the h-part k-blocks ([h | x], the order of the overlapped window form) instead of [x | h]
void pack_gru(const pv_rnn_dir* dirs, int K, int KP, int TR, std::vector<float>& wp, std::vector<float>& bias, bool hx_order = false) {
    const int nkb = (KP + HG) / 8;
    wp.assign((size_t)2 * 4 * nkb * 3 * 256, 0.0f);
    bias.assign((size_t)2 * 4 * HG, 0.0f);
    for (int d = 0; d < 2; d++) {
        for (int u = 0; u < HG; u++) {
            bias[(size_t)d * 4 * HG + 0 * HG + u] = dirs[d].b_ih[0 * HG + u] + dirs[d].b_hh[0 * HG + u];
            bias[(size_t)d * 4 * HG + 1 * HG + u] = dirs[d].b_ih[1 * HG + u] + dirs[d].b_hh[1 * HG + u];
            bias[(size_t)d * 4 * HG + 2 * HG + u] = dirs[d].b_ih[2 * HG + u];
            bias[(size_t)d * 4 * HG + 3 * HG + u] = dirs[d].b_hh[2 * HG + u];
        }
        auto wval = [&](int n, int k) -> float {
            if (hx_order) k = k < HG ? KP + k : k - HG;   // stream position -> position in [x | h]
            if (k < KP) return k < K ? dirs[d].w_ih[(size_t)n * K + k] : 0.0f;
            return dirs[d].w_hh[(size_t)n * HG + (k - KP)];
        };
        for (int w = 0; w < 4; w++)
            for (int kb = 0; kb < nkb; kb++)
                for (int nt = 0; nt < 3; nt++)
                    for (int lane = 0; lane < 64; lane++) {
                        float* dst = &wp[((((size_t)(d * 4 + w) * nkb + kb) * 3 + nt) * 64 + lane) * 4];
                        if (TR == 32) {
                            const int n = nt * HG + 32 * w + (lane & 31);
                            for (int j = 0; j < 4; j++) dst[j] = wval(n, kb * 8 + 4 * (lane >> 5) + j);
                        } else {
                            for (int t = 0; t < 2; t++)
                                for (int j = 0; j < 2; j++)
                                    dst[2 * t + j] = wval(nt * HG + 32 * w + 16 * t + (lane & 15), kb * 8 + 2 * (lane >> 4) + j);
                        }
                    }
    }
}

template <int TR, bool SPLIT> constexpr size_t lds_p2() { return (size_t)(SPLIT ? 1 : 2) * (2 * TR * LDH + (TR == 16 ? 2 : 1) * TR * LDXD) * sizeof(float); }

}  // namespace

struct pv_rnn_p2 {
    float* enc_wp[4] = {nullptr, nullptr, nullptr, nullptr};  // [0] 32-row tile form, [1] 16-row tile form, [2] 16-row form in [h | x] order (split form), [3] unit-split form
    float* dec_wp[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned* us_epoch = nullptr;   // device word
    int* us_err = nullptr;   // exchange time-outs of the unit-split form (device word)
    const void* us_hx_seen = nullptr; size_t us_hx_n = 0;   // the exchange buffer whose tags belong to this epoch sequence
    float* enc_bias = nullptr; float* dec_bias = nullptr;
    float* dense_w = nullptr; float* dense_b = nullptr;
    int dtype = PV_DTYPE_F32;
    pv_p2_bf16_weights bf;   // PV_DTYPE_BF16_INPUT_GEMM: fragment streams / split8 rows of the layer-wise path (rnn_rec_bf16.hip)
    std::vector<void*> owned;
};

static int up2(const float* h, size_t n, float** d, std::vector<void*>& owned) {
    PV_HIP(hipMalloc((void**)d, n * sizeof(float)));
    owned.push_back(*d);
    PV_HIP(hipMemcpy(*d, h, n * sizeof(float), hipMemcpyHostToDevice));
    return PV_OK;
}

void pv_rnn_free_p2(pv_ctx* ctx) {
    if (ctx->p2) {
        for (void* p : ctx->p2->owned) (void)hipFree(p);
        delete ctx->p2;
        ctx->p2 = nullptr;
    }
}

extern "C" int pv_rnn_load_p2(pv_ctx* ctx, const pv_weights_p2* w, int dtype) {
    PV_CHECK(ctx && w, PV_ERR_INVALID, "null argument");
    PV_CHECK(dtype == PV_DTYPE_F32 || dtype == PV_DTYPE_BF16_INPUT_GEMM, PV_ERR_INVALID, "unknown dtype %d", dtype);
    for (int d = 0; d < 2; d++)
        PV_CHECK(w->encoder[d].w_ih && w->encoder[d].w_hh && w->encoder[d].b_ih && w->encoder[d].b_hh &&
                     w->decoder[d].w_ih && w->decoder[d].w_hh && w->decoder[d].b_ih && w->decoder[d].b_hh,
                 PV_ERR_INVALID, "missing GRU tensor");
    PV_CHECK(w->dense_w && w->dense_b, PV_ERR_INVALID, "missing dense1");
    PV_HIP(hipSetDevice(ctx->device));
    if (ctx->p2) {
        PV_HIP(hipStreamSynchronize(ctx->stream));
        pv_rnn_free_p2(ctx);
    }
    pv_rnn_p2* m = new pv_rnn_p2();
    ctx->p2 = m;
    m->dtype = dtype;
    std::vector<float> wp, bias;
    int rc;
    if (dtype == PV_DTYPE_BF16_INPUT_GEMM) {
        // every matrix product on the bf16 MFMA with 3-term split operands: W_hh / encoder W_ih as fragment streams of
        // k_rec_bf16, the decoder's W_ih of both directions as split8 rows for the per-window GEMM
        std::vector<float> eb((size_t)2 * 3 * HG), ehn((size_t)2 * HG), dhn((size_t)2 * HG), wcat((size_t)6 * HG * KPD), bcat((size_t)6 * HG);
        for (int d = 0; d < 2; d++) {
            for (int u = 0; u < HG; u++) {
                for (int g = 0; g < 2; g++) {
                    eb[(size_t)d * 3 * HG + g * HG + u] = w->encoder[d].b_ih[g * HG + u] + w->encoder[d].b_hh[g * HG + u];
                    bcat[(size_t)d * 3 * HG + g * HG + u] = w->decoder[d].b_ih[g * HG + u] + w->decoder[d].b_hh[g * HG + u];
                }
                eb[(size_t)d * 3 * HG + 2 * HG + u] = w->encoder[d].b_ih[2 * HG + u];
                bcat[(size_t)d * 3 * HG + 2 * HG + u] = w->decoder[d].b_ih[2 * HG + u];
                ehn[(size_t)d * HG + u] = w->encoder[d].b_hh[2 * HG + u];
                dhn[(size_t)d * HG + u] = w->decoder[d].b_hh[2 * HG + u];
            }
            memcpy(&wcat[(size_t)d * 3 * HG * KPD], w->decoder[d].w_ih, (size_t)3 * HG * KPD * sizeof(float));
        }
        if ((rc = pv_pack_rec_bf16(w->encoder, 3, FEAT, &m->bf.enc_wp, &m->bf.enc_wx, m->owned))) return rc;
        if ((rc = pv_pack_rec_bf16(w->decoder, 3, 0, &m->bf.dec_wp, nullptr, m->owned))) return rc;
        if ((rc = pv_pack_gru16_bf16(w->encoder, FEAT, &m->bf.enc16_wp, &m->bf.enc16_wx, m->owned))) return rc;
        if ((rc = pv_pack_gru16_bf16(w->decoder, 0, &m->bf.dec16_wp, nullptr, m->owned))) return rc;
        if ((rc = up2(eb.data(), eb.size(), &m->bf.enc_bias, m->owned)) || (rc = up2(ehn.data(), ehn.size(), &m->bf.enc_bias_hn, m->owned)) ||
            (rc = up2(dhn.data(), dhn.size(), &m->bf.dec_bias_hn, m->owned)) || (rc = up2(bcat.data(), bcat.size(), &m->bf.dec_bias_cat, m->owned)))
            return rc;
        if ((rc = pv_upload_split8(wcat.data(), (size_t)6 * HG, KPD, &m->bf.dec_wih_s, m->owned))) return rc;
        if ((rc = pv_pack_p2_dense(w->dense_w, &m->bf.dense_frag, m->owned))) return rc;
        if ((rc = pv_pack_p2_dense16(w->dense_w, &m->bf.dense_frag16, m->owned))) return rc;
        if ((rc = pv_gemm_bf16x3_prepare()) || (rc = pv_rec_bf16_prepare())) return rc;
    }
    for (int f = 0; f < 3; f++) {
        pack_gru(w->encoder, FEAT, KPE, f ? 16 : 32, wp, bias, f == 2);
        if ((rc = up2(wp.data(), wp.size(), &m->enc_wp[f], m->owned))) return rc;
        if (!f && (rc = up2(bias.data(), bias.size(), &m->enc_bias, m->owned))) return rc;
        pack_gru(w->decoder, KPD, KPD, f ? 16 : 32, wp, bias, f == 2);
        if ((rc = up2(wp.data(), wp.size(), &m->dec_wp[f], m->owned))) return rc;
        if (!f && (rc = up2(bias.data(), bias.size(), &m->dec_bias, m->owned))) return rc;
    }
    pack_gru_us(w->encoder, FEAT, KPE, wp);
    if ((rc = up2(wp.data(), wp.size(), &m->enc_wp[3], m->owned))) return rc;
    pack_gru_us(w->decoder, KPD, KPD, wp);
    if ((rc = up2(wp.data(), wp.size(), &m->dec_wp[3], m->owned))) return rc;
    PV_HIP(hipFuncSetAttribute((const void*)k_gru_us, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_US));
    PV_HIP(hipMalloc((void**)&m->us_err, 64));
    m->owned.push_back(m->us_err);
    PV_HIP(hipMemset(m->us_err, 0, 64));
    PV_HIP(hipMalloc((void**)&m->us_epoch, 64));
    m->owned.push_back(m->us_epoch);
    PV_HIP(hipMemset(m->us_epoch, 0, 64));
    if ((rc = up2(w->dense_w, (size_t)NCLS * KPD, &m->dense_w, m->owned)) || (rc = up2(w->dense_b, NCLS, &m->dense_b, m->owned))) return rc;
    m->bf.dense_w = m->dense_w; m->bf.dense_b = m->dense_b;
    PV_HIP(hipFuncSetAttribute((const void*)k_gru_p2<32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p2<32, false>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_gru_p2<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p2<16, false>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_gru_p2<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p2<16, true>()));
    return PV_OK;
}

static int p2_launch(pv_ctx* ctx, const uint8_t* d_images, int64_t B, uint8_t* d_labels, float* d_acc, hipStream_t st,
                     int seq = SEQ, int nwin = NWIN, const float* d_hidden_in = nullptr, float* d_hidden_out = nullptr,
                     float* d_logits = nullptr) {
    pv_rnn_p2* m = ctx->p2;
    if (m->dtype == PV_DTYPE_BF16_INPUT_GEMM) {
        // the layer-wise path: no split form, no exchange (the finishing kernel still honours a pending error word)
        float* acc = d_acc;
        int rcb;
        if (!acc && (rcb = pv_get(ctx, "p2.acc", (size_t)B * seq * NCLS, &acc))) return rcb;
        if ((rcb = pv_p2_bf16_forward(ctx, m->bf, d_images, B, d_labels, acc, st, seq, nwin, d_hidden_in, d_hidden_out, d_logits))) return rcb;
        GruFinishArgs fb;
        fb.epoch = nullptr; fb.err = m->us_err; fb.labels = d_labels; fb.n_labels = d_labels ? B * seq : 0;
        fb.f[0] = acc; fb.nf[0] = B * seq * NCLS;
        fb.f[1] = d_logits; fb.nf[1] = d_logits ? B * WIN * NCLS : 0;
        fb.f[2] = d_hidden_out; fb.nf[2] = d_hidden_out ? B * 2 * HG : 0;
        k_gru_finish<<<(unsigned)std::min<int64_t>((B * seq + 255) / 256, 4 * ctx->num_cu), 256, 0, st>>>(fb);
        PV_HIP(hipGetLastError());
        return PV_OK;
    }
    // tile form: 32-row tiles once they fill the chip, else 16-row tiles (twice the workgroups, half the time per step)
    int tr = ((B + 31) / 32 >= ctx->num_cu) ? 32 : 16;
    if (ctx->opt.gru_rows) tr = ctx->opt.gru_rows;
    const int f = tr == 16 ? 1 : 0;
    const int64_t n_tiles = (B + tr - 1) / tr;
    GruArgs g;
    g.images = d_images;
    g.enc_wp = m->enc_wp[f]; g.dec_wp = m->dec_wp[f]; g.enc_bias = m->enc_bias; g.dec_bias = m->dec_bias;
    g.dense_w = m->dense_w; g.dense_b = m->dense_b;
    int rc;
    if ((rc = pv_get(ctx, "p2.enc_out", (size_t)n_tiles * WIN * tr * KPD, &g.enc_out))) return rc;
    if ((rc = pv_get(ctx, "p2.dec_out", (size_t)n_tiles * WIN * tr * KPD, &g.dec_out))) return rc;
    g.acc = d_acc;
    if (!g.acc)
        if ((rc = pv_get(ctx, "p2.acc", (size_t)B * seq * NCLS, &g.acc))) return rc;
    g.labels = d_labels;
    g.B = B;
    g.seq = seq; g.nwin = nwin; g.hidden_in = d_hidden_in; g.hidden_out = d_hidden_out; g.logits = d_logits;
    if ((rc = pv_zero_async(g.acc, (size_t)B * seq * NCLS * sizeof(float), st))) return rc;
    // the two directions of a tile on two CUs while every (tile, direction) workgroup has a CU of its own: the launch is a
    // chain of dependent steps, and a step then carries one direction's MFMAs per SIMD instead of two
    bool split = tr == 16 && 2 * n_tiles <= ctx->num_cu && ctx->opt.gru_split && !ctx->opt.shared_device;
    g.pair_flags = nullptr;
    g.hx = nullptr; g.quad_flags = nullptr; g.tag_base = 0; g.err = m->us_err; g.epoch = nullptr;
    g.spin_limit = 1 << ctx->opt.exchange_spin_log2; g.drop_part = ctx->opt.debug_drop_part;
    GruFinishArgs fin;
    fin.epoch = nullptr; fin.err = m->us_err; fin.labels = d_labels; fin.n_labels = d_labels ? B * seq : 0;
    fin.f[0] = g.acc; fin.nf[0] = B * seq * NCLS;
    fin.f[1] = d_logits; fin.nf[1] = d_logits ? B * WIN * NCLS : 0;
    fin.f[2] = d_hidden_out; fin.nf[2] = d_hidden_out ? B * 2 * HG : 0;
    const unsigned fin_grid = (unsigned)std::min<int64_t>((B * seq + 255) / 256, 4 * ctx->num_cu);
    // unit-split form: (tile, direction, half of the units) workgroups with a per-step h exchange, while all of them can be
    // resident at once (up to 1024 chunks on 256 CUs); option gru_usplit = 0 keeps the direction-split form, gru_split = 0 or
    // shared_device = 1 the one-workgroup form
    const bool usplit = tr == 16 && 4 * n_tiles <= ctx->num_cu && ctx->opt.gru_usplit && ctx->opt.gru_split && !ctx->opt.shared_device;
    if (usplit) {
        g.enc_wp = m->enc_wp[3]; g.dec_wp = m->dec_wp[3];
        const size_t nfl = (size_t)n_tiles * 4 * US_FLAG_STRIDE;
        if ((rc = pv_get(ctx, "p2.quad_flags", nfl, &g.quad_flags))) return rc;
        if ((rc = pv_zero_async(g.quad_flags, nfl * sizeof(int), st))) return rc;
        const size_t nhx = (size_t)n_tiles * 2 * US_HX_QUADS;
        if ((rc = pv_get(ctx, "p2.hx", nhx, &g.hx))) return rc;
        // tags advance with every launch (4096 per launch, device counter): a buffer this sequence has not written yet is
        // cleared once (tag 0 is never waited for)
        if (m->us_hx_seen != (const void*)g.hx || m->us_hx_n != nhx) {
            m->us_hx_seen = g.hx; m->us_hx_n = nhx;
            if ((rc = pv_zero_async(g.hx, nhx * sizeof(u32x4), st))) return rc;
        }
        g.epoch = m->us_epoch;
        {
            pv_prof_scope ps(ctx, "k_gru_us", st);
            k_gru_us<<<(unsigned)(((n_tiles + 7) / 8) * 32), 256, LDS_US, st>>>(g);
        }
        fin.epoch = m->us_epoch;
        k_gru_finish<<<fin_grid, 256, 0, st>>>(fin);
        PV_HIP(hipGetLastError());
        return PV_OK;
    }
    if (split) {
        g.enc_wp = m->enc_wp[2]; g.dec_wp = m->dec_wp[2];   // [h | x] stream order of the overlapped window form
        if ((rc = pv_get(ctx, "p2.pair_flags", (size_t)2 * n_tiles + 96, &g.pair_flags))) return rc;
        if ((rc = pv_zero_async(g.pair_flags, ((size_t)2 * n_tiles + 96) * sizeof(int), st))) return rc;
    }
    {
        pv_prof_scope ps(ctx, "k_gru_p2", st);
        if (tr == 32) k_gru_p2<32, false><<<(unsigned)n_tiles, 512, lds_p2<32, false>(), st>>>(g);
        else if (split) k_gru_p2<16, true><<<(unsigned)(2 * n_tiles), 256, lds_p2<16, true>(), st>>>(g);
        else k_gru_p2<16, false><<<(unsigned)n_tiles, 512, lds_p2<16, false>(), st>>>(g);
    }
    k_gru_finish<<<fin_grid, 256, 0, st>>>(fin);
    PV_HIP(hipGetLastError());
    return PV_OK;
}

// end of a host-buffer call: wait, and report exchange time-outs of the unit-split form (only possible when its workgroups
// could not all be resident, i.e. on a GPU shared with other work)
static int p2_finish(pv_ctx* ctx, hipStream_t st) {
    int n_timeouts = 0;
    PV_HIP(hipMemcpyAsync(&n_timeouts, ctx->p2->us_err, sizeof(int), hipMemcpyDeviceToHost, st));
    PV_HIP(hipStreamSynchronize(st));
    if (n_timeouts) {
        PV_HIP(hipMemset(ctx->p2->us_err, 0, sizeof(int)));
        pv_set_error("split GRU form: %d exchange polls / hand-offs timed out (GPU shared with other work?); the outputs of this call are poisoned "
                     "(labels 255, NaN). Set option shared_device = 1 (or gru_split = 0) on this context", n_timeouts);
        return PV_ERR_STATE;
    }
    return PV_OK;
}

int pv_p2_take_timeouts(pv_ctx* ctx, int* n) {   // read and clear (the caller has synchronised)
    *n = 0;
    if (!ctx->p2 || !ctx->p2->us_err) return PV_OK;
    PV_HIP(hipMemcpy(n, ctx->p2->us_err, sizeof(int), hipMemcpyDeviceToHost));
    if (*n) PV_HIP(hipMemset(ctx->p2->us_err, 0, sizeof(int)));
    return PV_OK;
}

extern "C" int pv_rnn_forward_p2_dev(pv_ctx* ctx, const uint8_t* d_images, int64_t B, uint8_t* d_labels, float* d_acc,
                                     void* stream) {
    PV_CHECK(ctx && d_images && d_labels, PV_ERR_INVALID, "null argument");
    PV_CHECK(ctx->p2, PV_ERR_STATE, "pv_rnn_load_p2 has not been called on this context");
    PV_CHECK(B >= 0 && B < (1ll << 22), PV_ERR_INVALID, "batch %lld out of range", (long long)B);
    if (B == 0) return PV_OK;
    PV_HIP(hipSetDevice(ctx->device));
    return p2_launch(ctx, d_images, B, d_labels, d_acc, pv_pick_stream(ctx, stream));
}

extern "C" int pv_rnn_forward_p2(pv_ctx* ctx, const uint8_t* images, int64_t B, uint8_t* labels, float* acc) {
    PV_CHECK(ctx && images && labels, PV_ERR_INVALID, "null argument");
    PV_CHECK(ctx->p2, PV_ERR_STATE, "pv_rnn_load_p2 has not been called on this context");
    PV_CHECK(B >= 0 && B < (1ll << 22), PV_ERR_INVALID, "batch %lld out of range", (long long)B);
    if (B == 0) return PV_OK;
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint8_t *d_img, *d_lab;
    float* d_acc;
    int rc;
    if ((rc = pv_get(ctx, "p2.images", (size_t)B * SEQ * FEAT, &d_img))) return rc;
    if ((rc = pv_get(ctx, "p2.labels", (size_t)B * SEQ, &d_lab))) return rc;
    if ((rc = pv_get(ctx, "p2.acc", (size_t)B * SEQ * NCLS, &d_acc))) return rc;
    PV_HIP(hipMemcpyAsync(d_img, images, (size_t)B * SEQ * FEAT, hipMemcpyHostToDevice, st));
    if ((rc = p2_launch(ctx, d_img, B, d_lab, d_acc, st))) return rc;
    PV_HIP(hipMemcpyAsync(labels, d_lab, (size_t)B * SEQ, hipMemcpyDeviceToHost, st));
    if (acc) PV_HIP(hipMemcpyAsync(acc, d_acc, (size_t)B * SEQ * NCLS * sizeof(float), hipMemcpyDeviceToHost, st));
    return p2_finish(ctx, st);
}

// One TransducerGRU.forward(x, hidden) of the polisher model (pepper/modules/python/models/simple_model.py:27-42),
// i.e. the operator the reference's sliding loop calls once per window (predict.py:65): images uint8
// [B,100,10], hidden_in [B,2,128] (NULL = zeros) -> logits [B,100,5], hidden_out [B,2,128]. HOST pointers.
extern "C" int pv_rnn_forward_p2_window(pv_ctx* ctx, const uint8_t* images, const float* hidden_in, int64_t B, float* logits,
                                        float* hidden_out) {
    PV_CHECK(ctx && images && logits, PV_ERR_INVALID, "null argument");
    PV_CHECK(ctx->p2, PV_ERR_STATE, "pv_rnn_load_p2 has not been called on this context");
    PV_CHECK(B >= 0 && B < (1ll << 22), PV_ERR_INVALID, "batch %lld out of range", (long long)B);
    if (B == 0) return PV_OK;
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    uint8_t* d_img;
    float *d_hin = nullptr, *d_hout, *d_log, *d_acc;
    int rc;
    if ((rc = pv_get(ctx, "p2w.images", (size_t)B * WIN * FEAT, &d_img))) return rc;
    if ((rc = pv_get(ctx, "p2w.hout", (size_t)B * 2 * HG, &d_hout))) return rc;
    if ((rc = pv_get(ctx, "p2w.logits", (size_t)B * WIN * NCLS, &d_log))) return rc;
    if ((rc = pv_get(ctx, "p2w.acc", (size_t)B * WIN * NCLS, &d_acc))) return rc;
    PV_HIP(hipMemcpyAsync(d_img, images, (size_t)B * WIN * FEAT, hipMemcpyHostToDevice, st));
    if (hidden_in) {
        if ((rc = pv_get(ctx, "p2w.hin", (size_t)B * 2 * HG, &d_hin))) return rc;
        PV_HIP(hipMemcpyAsync(d_hin, hidden_in, (size_t)B * 2 * HG * sizeof(float), hipMemcpyHostToDevice, st));
    }
    if ((rc = p2_launch(ctx, d_img, B, nullptr, d_acc, st, WIN, 1, d_hin, d_hout, d_log))) return rc;
    PV_HIP(hipMemcpyAsync(logits, d_log, (size_t)B * WIN * NCLS * sizeof(float), hipMemcpyDeviceToHost, st));
    if (hidden_out) PV_HIP(hipMemcpyAsync(hidden_out, d_hout, (size_t)B * 2 * HG * sizeof(float), hipMemcpyDeviceToHost, st));
    return p2_finish(ctx, st);
}

#ifdef PV_GRU_STAMPS
// diagnostic build only: phase cycle sums of workgroup 0 / wave 0 of the last split-form launch (tools/bench_gru.py prints them)
extern "C" int pv_debug_read_gru_stamps(pv_ctx* ctx, unsigned long long* out8) {
    int* d = nullptr;
    if (pv_get(ctx, "p2.pair_flags", 8, &d)) return PV_ERR_HIP;
    PV_HIP(hipDeviceSynchronize());
    PV_HIP(hipMemcpy(out8, d + 64, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PV_OK;
}
#endif
