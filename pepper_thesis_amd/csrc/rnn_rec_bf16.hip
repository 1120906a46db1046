// rnn_rec_bf16.hip — recurrent layers of PV_DTYPE_BF16_INPUT_GEMM on the bf16 MFMA (gfx950), and the layer-wise P2 path.
//
// k_rec_bf16<NG, ENC, MT>: one bidirectional recurrent layer, NG = 4 LSTM (P1: pepper_variant/modules/python/models/
// simple_model.py:23-32,50-54; hidden 256) or NG = 3 GRU (P2: pepper/modules/python/models/simple_model.py:12-21,27-42; hidden
// 128). Workgroup = (batch tile of 32 * MT rows, direction), wave w owns hidden units [32w, 32w + 32) of every gate, so the
// cell update is in-register. Per time step
//     gates = G_t (input projection, pre-computed by k_gemm_bf16x3; biases folded in)          [ENC = false]
//           = bias + x_t . W_ih^T with x_t bytes, exact in bf16: two terms x.w_hi + x.w_lo        [ENC = true]
//           + h_{t-1} . W_hh^T as three terms h_hi.w_hi + h_hi.w_lo + h_lo.w_hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate
// h lives in LDS as "split8" rows (per 8 units: 8 bf16 hi, 8 bf16 lo), double-buffered, one barrier per step; a 16-byte half
// of a group is exactly one A fragment. W_hh (and the LSTM encoder's W_ih) are pre-split on the host into B-fragment order and
// streamed from L2 through a ring of D register sets requested D - 1 (gate, k-step) slots ahead, wrapping into the next step.
// At 4 bytes per weight the stream is what bounds a step (1 MB per LSTM step and workgroup against 12 k MFMA cycles for 32
// rows), hence MT = 2: 64 rows per weight fetch. The projections G arrive as quads [m / 4][column][4] whose four values are
// the accumulator registers 4q .. 4q+3 of a lane: they are loaded straight INTO the accumulators at the end of the previous
// step, so the MFMAs simply continue from them.
#include "rnn_bf16.hpp"
#include "mfma_tiles.hpp"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace {

using namespace pvdev;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * rcpf_(e + 1.0f);
}
__device__ __forceinline__ unsigned split8_off(unsigned k) { return (k >> 3) * 32u + (k & 7u) * 2u; }

template <int NG, bool ENC, int MT = 1> struct RecCfg {
    static constexpr int HID = NG == 4 ? 256 : 128;
    static constexpr int NW = HID / 32, NTHR = 64 * NW;
    static constexpr int KS_H = HID / 16;                       // k-steps (K = 16) of the recurrent product
    static constexpr int XK = ENC ? (NG == 4 ? 32 : 16) : 0;    // padded input features (26 -> 32, 10 -> 16)
    static constexpr int KS_X = XK / 16;
    static constexpr bool XRES = ENC && NG == 3;                // GRU encoder: the x-part fragments (one k-step) stay in registers
    // GRU: W_hh of a direction is 196 KB of fragments = 192 registers per lane of its four waves (one wave per SIMD, 512
    // registers each): the weights are loaded ONCE per launch and stay resident - a true persistent-RNN step with no weight
    // stream at all (streamed, a 32-row step was bound by the 196 KB per step through the CU's 64 B/clk vector-memory path)
    static constexpr bool WRES = NG == 3 && !(ENC && MT == 2);   // (the 64-row encoder form would spill: it streams)
    static constexpr int NSLOT = (XRES ? 0 : KS_X * NG) + KS_H * NG;   // (gate, k-step) slots of the weight stream per step
    // ring depth (NSLOT % D == 0): 8 register sets where they fit; 4 for the 64-row LSTM forms (128 accumulator + 32 state
    // registers of 256) and the GRU encoder (its x-part fragments stay resident)
    static constexpr int D = ((NG == 3 && ENC) || (NG == 4 && MT == 2)) ? 4 : 8;   // (8 sets in the 64-row decoder: 14 spills, no faster)
    static constexpr int NA = NG + ((NG == 3 && ENC) ? 1 : 0);  // accumulators per M-tile (GRU keeps the n gate's x-part apart)
    static constexpr int HS = HID * 4 + 16;                     // LDS row stride of the split8 h tile: (HS / 4) % 64 == 4
    static constexpr int XS = XK * 2 + 16;                      // LDS row stride of the bf16 x tile
    static_assert(NSLOT % D == 0, "ring depth must divide the stream length");
};

struct RecArgs {
    const float* G;
    const unsigned char* wp;
    const unsigned char* wx;
    const float* bias;
    const float* bias_hn;
    const unsigned char* x;
    int64_t x_row_bytes;
    int x_t0, xf, x_signed;
    int64_t B, Bp;
    int T;
    const float* h0;
    float* h_out;
    float* out_f32;
    unsigned char* out_tm;
    unsigned char* out_bm;
    int n_tiles;
    const unsigned char* dw;   // GRU decoder: fragments of the dense layer (pv_pack_p2_dense) ...
    float* dpart;              // ... and where its partial logits go: [T][tiles of 32 rows][2 dirs][4 waves][8 classes][32 rows], or NULL
};

// slots [I0, I0 + NKS * NG) of a stream of NTOT slots: slot = (k-step, gate) = [hi fragments | lo fragments] of 1 KB each.
// At = this lane's A-fragment address of M-tile 0, k-step 0; XP: x-part (plain bf16 rows, exact operand, two terms).
// hook(i) runs behind the MFMAs of slot i (vector / LDS / store instructions of the caller that have nothing to do with the
// product issue there while the matrix pipe works).
template <int NG, int MT, int NA, int D, int NTOT, int I0, int NKS, bool XP, typename Hook>
__device__ __forceinline__ void ring_bf16(f32x16 (&acc)[MT][NA], const unsigned char* __restrict__ At, int m_stride,
                                          __amdgpu_buffer_rsrc_t wr, f32x4 (&bq)[D][2], unsigned lane16, Hook&& hook) {
    constexpr int KSB = XP ? 32 : 64;   // bytes of one k-step inside an A row
    bf16x8 ah[MT], al[MT];
#pragma unroll
    for (int i = 0; i < NKS * NG; i++) {
        {   // request slot i + D - 1 (wrapping into the next step's first slots: same weights every step)
            const int sv = (I0 + i + D - 1) % NTOT, rs = (I0 + i + D - 1) % D;
            bq[rs][0] = buf_load4(wr, lane16, (unsigned)(sv * 2048));
            bq[rs][1] = buf_load4(wr, lane16, (unsigned)(sv * 2048 + 1024));
        }
        const int ks = i / NG, g = i % NG;
        if (g == 0) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                ah[m] = *reinterpret_cast<const bf16x8*>(At + m * m_stride + ks * KSB);
                if (!XP) al[m] = *reinterpret_cast<const bf16x8*>(At + m * m_stride + ks * KSB + 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the prefetches next to their uses
        {
            const int ai = (XP && NG == 3 && g == 2) ? 3 : g;
            const int rs = (I0 + i) % D;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[rs][0]), bl = __builtin_bit_cast(bf16x8, bq[rs][1]);
#pragma unroll
            for (int m = 0; m < MT; m++) {
                acc[m][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh, acc[m][ai], 0, 0, 0);
                acc[m][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl, acc[m][ai], 0, 0, 0);
                if (!XP) acc[m][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh, acc[m][ai], 0, 0, 0);
            }
        }
        hook(i);
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifndef PV_REC_ABL
#define PV_REC_ABL 0   // diagnostic ablations (timing only, results wrong): 1 = no global output stores, 2 = no LDS h writes
#endif
#ifdef PV_REC_STAMPS
// diagnostic build: cycle sums of the phases of a step (workgroup 0, every wave adds), read by pv_debug_rec_stamps
__device__ unsigned long long g_rec_stamps[8];
#define RSTAMP(i) { unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); st_acc[i] += now_ - st_last; st_last = now_; }
#else
#define RSTAMP(i)
#endif

template <int NG, bool ENC, int MT>
__global__ __launch_bounds__(NG == 4 ? 512 : 256, 1) void k_rec_bf16(RecArgs a) {
    typedef RecCfg<NG, ENC, MT> C;
    constexpr int HID = C::HID, NW = C::NW, NTHR = C::NTHR, ROWS = 32 * MT, HS = C::HS, XS = C::XS, D = C::D, NA = C::NA;
    constexpr int NSLOT = C::NSLOT, NXS = C::XRES ? 0 : C::KS_X * NG;   // x slots at the head of the stream
    constexpr int NCOL = 2 * NG * HID;                                  // columns of a G row
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* hbuf = sm;                    // [2][ROWS][HS]
    unsigned char* xbuf = sm + 2 * ROWS * HS;    // [2][ROWS][XS]   (ENC)
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware mapping (blocks b and b + 8 share an XCD): one direction per XCD, so its L2 holds one direction's weights
    const int xcd = blockIdx.x & 7;
    const int dir = xcd & 1;
    const int tile = (blockIdx.x >> 3) * 4 + (xcd >> 1);
    if (tile >= a.n_tiles) return;
    const int64_t b0 = (int64_t)tile * ROWS;
    const int unit = 32 * wv + (lane & 31), rg = lane >> 5;
    const int T = a.T;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.wp + (size_t)(dir * NW + wv) * NSLOT * 2048);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x4 bq[C::WRES ? 1 : D][2];
    bf16x8 wres[C::WRES ? NSLOT : 1][2];
    if constexpr (C::WRES) {
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            wres[k][0] = __builtin_bit_cast(bf16x8, buf_load4(wr, lane16, (unsigned)(k * 2048)));
            wres[k][1] = __builtin_bit_cast(bf16x8, buf_load4(wr, lane16, (unsigned)(k * 2048 + 1024)));
        }
    } else {
#pragma unroll
        for (int k = 0; k < D - 1; k++) {
            bq[k][0] = buf_load4(wr, lane16, (unsigned)(k * 2048));
            bq[k][1] = buf_load4(wr, lane16, (unsigned)(k * 2048 + 1024));
        }
    }
    // GRU encoder: resident x-part fragments [dir][wave][gate][hi, lo][lane][16 B]
    bf16x8 xw[C::XRES ? 3 : 1][2];
    if constexpr (C::XRES) {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.wx + (size_t)(dir * NW + wv) * 3 * 2048);
#pragma unroll
        for (int g = 0; g < 3; g++) {
            xw[g][0] = __builtin_bit_cast(bf16x8, buf_load4(xr, lane16, (unsigned)(g * 2048)));
            xw[g][1] = __builtin_bit_cast(bf16x8, buf_load4(xr, lane16, (unsigned)(g * 2048 + 1024)));
        }
    }
    float bs[NA];   // ENC: gate biases of this lane's unit; GRU (both): index NG - 1 ... see below
    float b_hn = 0.0f;
    if constexpr (ENC) {
#pragma unroll
        for (int g = 0; g < NG; g++) bs[g] = a.bias[dir * NG * HID + g * HID + unit];
    }
    if constexpr (NG == 3) b_hn = a.bias_hn[dir * HID + unit];

    // ---- state: c (LSTM) / h (GRU) of this lane's elements; h_{t-1} as split8 rows in LDS --------------------------------
    float st[MT][16];
    for (int i = tid; i < ROWS * HS / 16; i += NTHR) reinterpret_cast<u32x4*>(hbuf)[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int e = 0; e < 16; e++) st[m][e] = 0.0f;
    const unsigned hl = (unsigned)(4 * rg * HS) + split8_off((unsigned)unit);   // lane part of an h element's LDS offset
    __syncthreads();
    if (NG == 3 && a.h0) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = 32 * m + 8 * (e >> 2) + (e & 3) + 4 * rg;
                const float h = a.h0[((b0 + row) * 2 + dir) * HID + unit];
                st[m][e] = h;
                const __bf16 hi = (__bf16)h;
                const __bf16 lo = (__bf16)(h - (float)hi);
                unsigned char* p = hbuf + (32 * m + 8 * (e >> 2) + (e & 3)) * HS + hl;
                *reinterpret_cast<__bf16*>(p) = hi;
                *reinterpret_cast<__bf16*>(p + 16) = lo;
            }
    }

    // ---- x staging (ENC): bytes -> bf16 rows [ROWS][XK]; thread = (row, group of 4 features) ------------------------------
    constexpr int FG = ENC ? C::XK / 4 : 1;
    const int xrow = tid / FG, xfg = tid % FG;
    const bool x_on = ENC && xrow < ROWS;
    const unsigned char* xsrc = nullptr;
    unsigned xv[4] = {0u, 0u, 0u, 0u};
    if constexpr (ENC) {
        int64_t r = b0 + xrow;
        if (r >= a.B) r = a.B - 1;   // rows beyond B replicate row B-1 (finite values, never handed to the caller)
        xsrc = a.x + r * a.x_row_bytes + (int64_t)a.x_t0 * a.xf;
    }
    auto x_load = [&](int t) {
        if constexpr (ENC) {
            if (x_on) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int f = 4 * xfg + j;
                    xv[j] = f < a.xf ? (unsigned)xsrc[(int64_t)t * a.xf + f] : 0u;
                }
            }
        }
    };
    auto x_store = [&](int slot) {
        if constexpr (ENC) {
            if (x_on) {
                bf16x4 v;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float f = a.x_signed ? (float)(int)(signed char)xv[j] : (float)xv[j];   // |f| <= 255: exact in bf16
                    v[j] = (__bf16)f;
                }
                *reinterpret_cast<bf16x4*>(xbuf + (slot * ROWS + xrow) * XS + xfg * 8) = v;
            }
        }
    };

    // ---- accumulators -------------------------------------------------------------------------------------------------------
    f32x16 acc[MT][NA];
    f32x4 gn[(NG == 3 && !ENC) ? MT : 1][4];   // GRU decoder: the n gate's input projection, kept apart from W_hn h
    const unsigned gl = (unsigned)((rg * NCOL + (lane & 31)) * 16);   // lane part of a G quad's byte offset
    auto acc_init = [&](int t) {
        if constexpr (ENC) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[m][(NG == 3 && g == 2) ? 3 : g][e] = bs[g];
                if constexpr (NG == 3)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[m][2][e] = b_hn;
            }
        } else {
            // quad row (t * Bp + b0) / 4 + 8m + 2q + rg, column dir * NG * HID + g * HID + unit: a per-step resource keeps the
            // offsets inside the tile's 32 * MT rows (33 x 8192 rows x 8 KB would not fit 32 bits)
            const __amdgpu_buffer_rsrc_t gsr = make_rsrc(a.G + ((size_t)t * a.Bp + b0) * NCOL);
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const f32x4 v = buf_load4_nt(gsr, gl, (unsigned)(((8 * m + 2 * q) * NCOL + dir * NG * HID + g * HID + 32 * wv) * 16));
                        if (NG == 3 && g == 2) {
                            gn[m][q] = v;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; j++) acc[m][g][4 * q + j] = v[j];
                        }
                    }
            if constexpr (NG == 3)
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[m][2][e] = b_hn;
        }
    };

    x_load(dir ? T - 1 : 0);
    x_store(0);
    acc_init(dir ? T - 1 : 0);
    __syncthreads();

    const unsigned char* a_h = hbuf + (lane & 31) * HS + rg * 32;    // + cur * ROWS * HS
    const unsigned char* a_x = xbuf + (lane & 31) * XS + rg * 16;    // + slot * ROWS * XS
    const unsigned rowb_tm = 2u * HID * 4u;                          // bytes of a time-major output row
    const unsigned rowb_bm = (unsigned)T * 2u * HID * 4u;            // bytes of a batch-major output row
    const unsigned o_f_l = (unsigned)(4 * rg) * rowb_bm + (unsigned)unit * 4u;
    const __amdgpu_buffer_rsrc_t bmr = make_rsrc(a.out_bm + (size_t)b0 * rowb_bm);
    // The layer's outputs are split8 rows - the very layout of the h tile in LDS. Element-wise they were two 2-byte global
    // stores per value (a third of a GRU step); instead the finished tile (stable for the whole next step: it is that step's
    // A operand) is copied out with 16-byte loads / stores, a row of HID * 4 bytes per 2 * HID / 8 lanes.
    constexpr int CPR = HID / 4, NCP = ROWS * CPR / NTHR;   // 16-byte chunks per row, chunks per thread
    const bool want_out = a.out_tm || a.out_bm;
    // piece k of the copy in two halves - the LDS read, and the stores one hook later - so that neither waits inside the MFMA loop.
    // Thread -> (row tid / CPR + k * RPP, chunk tid % CPR): three lane offsets for the whole launch, the piece is a scalar offset
    static_assert(NTHR % CPR == 0, "a piece is a whole number of rows");
    constexpr int RPP = NTHR / CPR;   // rows per piece
    const unsigned cp_lds = (unsigned)((tid / CPR) * HS + (tid % CPR) * 16);
    const unsigned cp_tm = (unsigned)(tid / CPR) * rowb_tm + (unsigned)((tid % CPR) * 16);
    const unsigned cp_bm = (unsigned)(tid / CPR) * rowb_bm + (unsigned)((tid % CPR) * 16);
    u32x4 tile_v = {0u, 0u, 0u, 0u};
    auto tile_read = [&](int k, const unsigned char* tile) {
        tile_v = *reinterpret_cast<const u32x4*>(tile + k * RPP * HS + cp_lds);
    };
    auto tile_write = [&](int k, int tt) {
        if (a.out_tm) {
            const __amdgpu_buffer_rsrc_t tmr = make_rsrc(a.out_tm + ((size_t)tt * a.Bp + b0) * rowb_tm);
            __builtin_amdgcn_raw_buffer_store_b128(tile_v, tmr, cp_tm, (unsigned)(k * RPP) * rowb_tm + (unsigned)(dir * HID * 4), 2);
        }
        if (a.out_bm)
            __builtin_amdgcn_raw_buffer_store_b128(tile_v, bmr, cp_bm, (unsigned)(k * RPP) * rowb_bm + (unsigned)((tt * 2 + dir) * HID * 4), 2);
    };
    auto tile_out = [&](int tt, const unsigned char* tile) {   // the whole copy at once (after the last step)
        if (!want_out) return;
#pragma unroll
        for (int k = 0; k < NCP; k++) { tile_read(k, tile); tile_write(k, tt); }
    };
    const __amdgpu_buffer_rsrc_t ofr = make_rsrc(reinterpret_cast<unsigned char*>(a.out_f32) + (size_t)b0 * rowb_bm);
    // GRU decoder with the dense layer folded in (P2, pv_p2_bf16_forward): the logits need both directions' h_t, so every
    // (tile, direction) workgroup contributes h_t[:, its 128 units] . W_dense[:, those units]^T as one more MFMA tile (32 rows x
    // 32 "classes", 5 real), wave w over its k-steps 2w and 2w + 1; the partial sums of the 2 x 4 waves go out as quads (105 MB per
    // window at 4096 chunks) instead of the layer's split8 output (420 MB written here and read again by k_p2_dense)
    constexpr bool CAN_DENSE = NG == 3 && !ENC;
    const bool dense = CAN_DENSE && a.dpart != nullptr;
    bf16x8 dfr[CAN_DENSE ? 2 : 1][2];
    if constexpr (CAN_DENSE) {
        if (dense) {
            const __amdgpu_buffer_rsrc_t dr = make_rsrc(a.dw + (size_t)(dir * NW + wv) * 2 * 2048);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                dfr[k][0] = __builtin_bit_cast(bf16x8, buf_load4(dr, lane16, (unsigned)(k * 2048)));
                dfr[k][1] = __builtin_bit_cast(bf16x8, buf_load4(dr, lane16, (unsigned)(k * 2048 + 1024)));
            }
        }
    }
    auto dense_out = [&](int tt, const unsigned char* tile) {   // partial logits of the h tile of step tt
        if constexpr (CAN_DENSE) {
#pragma unroll
            for (int m = 0; m < MT; m++) {
                f32x16 ad;
#pragma unroll
                for (int e = 0; e < 16; e++) ad[e] = 0.0f;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const unsigned char* At = tile + (lane & 31) * HS + rg * 32 + m * 32 * HS + (2 * wv + k) * 64;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(At), al = *reinterpret_cast<const bf16x8*>(At + 16);
                    ad = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, dfr[k][0], ad, 0, 0, 0);
                    ad = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, dfr[k][1], ad, 0, 0, 0);
                    ad = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, dfr[k][0], ad, 0, 0, 0);
                }
                if ((lane & 31) < 8) {   // class lane & 31 (5 real, 3 zero), rows 8q + j + 4rg of this M-tile
                    const size_t tile32 = (size_t)(b0 / 32) + m;
                    float* dst = a.dpart + ((((size_t)tt * (a.Bp / 32) + tile32) * 2 + dir) * 4 + wv) * 256 + (lane & 31) * 32 + 4 * rg;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; j++) v[j] = ad[4 * q + j];
                        *reinterpret_cast<f32x4*>(dst + 8 * q) = v;
                    }
                }
            }
        }
    };
    int cur = 0;
#ifdef PV_REC_STAMPS
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last) :: "memory");
#endif
    for (int s = 0; s < T; s++) {
        const int t = dir ? (T - 1 - s) : s;
        const int tn = dir ? (T - 2 - s) : (s + 1);
        if (s + 1 < T) x_load(tn);
        // the previous step's h tile (this step's A operand: stable until the barrier) leaves for the layer's output in NCP pieces
        // spread over this step's MFMAs (all of them in front of the MFMAs were ~3 k of a 41 k-cycle LSTM step)
        const bool copy = want_out && s > 0;
        const int t_prev = dir ? t + 1 : t - 1;
        const unsigned char* tile_prev = hbuf + cur * ROWS * HS;
        constexpr int HSLOTS = C::KS_H * NG, HSTRIDE = HSLOTS / NCP;   // h-part slots (LSTM 64, GRU 24) per piece
        static_assert(HSTRIDE >= 2, "a piece needs two slots: read, then write");
        auto copy_hook = [&](int i) {
            if (copy) {
                if (i % HSTRIDE == 0 && i / HSTRIDE < NCP) tile_read(i / HSTRIDE, tile_prev);
                if (i % HSTRIDE == 1 && i / HSTRIDE < NCP) tile_write(i / HSTRIDE, t_prev);
            }
        };
        auto no_hook = [](int) {};
        // the two waves of a SIMD (w and w + NW / 2 ... hardware places wave i on SIMD i % 4) would otherwise sit in the same
        // phase - both on the matrix pipe, then both in the cell update: the lower half of the workgroup takes the pipe first
        // and runs its cell update under the upper half's MFMAs
        if (NW == 8) { if (wv < 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
        RSTAMP(0)
        // ---- x-part (ENC) -------------------------------------------------------------------------------------------------
        if constexpr (C::XRES) {
            bf16x8 ax[MT];
#pragma unroll
            for (int m = 0; m < MT; m++) ax[m] = *reinterpret_cast<const bf16x8*>(a_x + ((s & 1) * ROWS + 32 * m) * XS);
#pragma unroll
            for (int g = 0; g < 3; g++)
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const int ai = g == 2 ? 3 : g;
                    acc[m][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[m], xw[g][0], acc[m][ai], 0, 0, 0);
                    acc[m][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[m], xw[g][1], acc[m][ai], 0, 0, 0);
                }
        } else if constexpr (ENC) {
            ring_bf16<NG, MT, NA, D, NSLOT, 0, C::KS_X, true>(acc, a_x + (s & 1) * ROWS * XS, 32 * XS, wr, bq, lane16, no_hook);
        }
        // ---- h-part: h_{t-1} . W_hh^T, three terms ----------------------------------------------------------------------------
        if constexpr (C::WRES) {
            const unsigned char* At = a_h + cur * ROWS * HS;
#pragma unroll
            for (int ks = 0; ks < C::KS_H; ks++) {
                bf16x8 ah[MT], al[MT];
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    ah[m] = *reinterpret_cast<const bf16x8*>(At + m * 32 * HS + ks * 64);
                    al[m] = *reinterpret_cast<const bf16x8*>(At + m * 32 * HS + ks * 64 + 16);
                }
#pragma unroll
                for (int g = 0; g < NG; g++)
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wres[ks * NG + g][0], acc[m][g], 0, 0, 0);
                        acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], wres[ks * NG + g][1], acc[m][g], 0, 0, 0);
                        acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], wres[ks * NG + g][0], acc[m][g], 0, 0, 0);
                        if (m == MT - 1) copy_hook(ks * NG + g);
                    }
            }
        } else {
            ring_bf16<NG, MT, NA, D, NSLOT, NXS, C::KS_H, false>(acc, a_h + cur * ROWS * HS, 32 * HS, wr, bq, lane16, copy_hook);
        }
        if (dense && s > 0) dense_out(t_prev, tile_prev);   // (the previous step's h tile: this step's A operand)
        if (NW == 8) { if (wv < 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
        RSTAMP(1)
        // ---- cell update --------------------------------------------------------------------------------------------------------
        unsigned char* hn = hbuf + (cur ^ 1) * ROWS * HS;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                float h;
                if constexpr (NG == 4) {   // PyTorch LSTM gate order i, f, g, o
                    const float ig = sigmoidf_(acc[m][0][e]);
                    const float fg = sigmoidf_(acc[m][1][e]);
                    const float gg = tanhf_(acc[m][2][e]);
                    const float og = sigmoidf_(acc[m][3][e]);
                    const float c = fg * st[m][e] + ig * gg;
                    st[m][e] = c;
                    h = og * tanhf_(c);
                } else {                   // PyTorch GRU: n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) n + z h
                    const float r = sigmoidf_(acc[m][0][e]);
                    const float z = sigmoidf_(acc[m][1][e]);
                    const float xn = ENC ? acc[m][NA - 1][e] : gn[ENC ? 0 : m][e >> 2][e & 3];
                    const float n = tanhf_(xn + r * acc[m][2][e]);
                    h = (1.0f - z) * n + z * st[m][e];
                    st[m][e] = h;
                }
                const int row = 32 * m + 8 * (e >> 2) + (e & 3);   // + 4 * rg (lane part)
                const __bf16 hi = (__bf16)h;
                const __bf16 lo = (__bf16)(h - (float)hi);
#if PV_REC_ABL != 2
                *reinterpret_cast<__bf16*>(hn + row * HS + hl) = hi;
                *reinterpret_cast<__bf16*>(hn + row * HS + hl + 16) = lo;
#endif
#if PV_REC_ABL == 1
                if (h == 123.456f)
#endif
                {
                if (a.out_f32) buf_store1_nt(h, ofr, o_f_l, (unsigned)row * rowb_bm + (unsigned)((t * 2 + dir) * HID * 4));   // debug taps only
                }
            }
        RSTAMP(2)
        if (s + 1 < T) {
            acc_init(tn);     // the next step's projections travel into the accumulators during the barrier and the ring's wrap
            x_store((s + 1) & 1);
        }
        cur ^= 1;
        RSTAMP(3)
        lds_barrier();        // h_t (and x_{t+1}) complete; LDS only: the weight ring and the output stores stay in flight
        RSTAMP(4)
    }
    tile_out(dir ? 0 : T - 1, hbuf + cur * ROWS * HS);
    if (dense) dense_out(dir ? 0 : T - 1, hbuf + cur * ROWS * HS);
#ifdef PV_REC_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 5; i++) atomicAdd(&g_rec_stamps[i], st_acc[i]);
#endif
    if (NG == 3 && a.h_out) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = 32 * m + 8 * (e >> 2) + (e & 3) + 4 * rg;
                a.h_out[((b0 + row) * 2 + dir) * HID + unit] = st[m][e];
            }
    }
}

// ---- the GRU layers on 16-row tiles (small P2 batches) -----------------------------------------------------------------------
// A launch is a chain of T dependent steps whose duration is the step time of ONE workgroup. While the (16-row tile, direction)
// workgroups all fit on the chip (up to 2048 chunks on 256 CUs) v_mfma_f32_16x16x32_bf16 halves both the MFMA time and the cell
// update of a step against the 32-row form - the same FLOP per cycle on half the rows. Wave w owns units [32w, 32w + 32) as two
// 16-unit tiles; W_hh (3 gates x 2 tiles x 4 k-steps of 32, hi + lo: 192 registers per lane) and, in the encoder, W_ih (one k-step:
// 48 registers) stay resident. Fragment layouts: A lane -> row lane & 15, k = 8 (lane >> 4) + j; B lane -> unit lane & 15, same k;
// D lane -> unit lane & 15, rows 4 (lane >> 4) + register - the quads [m / 4][column][4] the input GEMM writes are one
// accumulator each.
constexpr int G16_HS = 128 * 4 + 16, G16_XS = 32 * 2 + 16;
// NTL = 16-row tiles per workgroup (1, or 2 from 2049 chunks on: the two tiles of a workgroup share the resident weights and the
// step's barrier and run one after the other inside the step - still less per 32 rows than the 32-row form's step)
template <bool ENC, int NTL>
__global__ __launch_bounds__(256, 1) void k_gru16_bf16(RecArgs a) {
    constexpr int HID = 128, NW = 4, ROWS = 16, HS = G16_HS, XS = G16_XS, NCOL = 6 * HID;
    __shared__ __attribute__((aligned(16))) unsigned char hbuf_all[NTL * 2 * ROWS * HS];
    __shared__ __attribute__((aligned(16))) unsigned char xbuf_all[ENC ? NTL * 2 * ROWS * XS : 16];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, dir = xcd & 1;
    const int tile0 = ((blockIdx.x >> 3) * 4 + (xcd >> 1)) * NTL;
    if (tile0 >= a.n_tiles) return;
    const int lr = lane & 15, q = lane >> 4, T = a.T;
    const unsigned lane16 = (unsigned)lane * 16u;
    // resident weights: slot (ks * 3 + g) * 2 + nt of this wave's stream = [hi 1 KB | lo 1 KB]
    bf16x8 wres[24][2];
    {
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.wp + (size_t)(dir * NW + wv) * 24 * 2048);
#pragma unroll
        for (int k = 0; k < 24; k++) {
            wres[k][0] = __builtin_bit_cast(bf16x8, buf_load4(wr, lane16, (unsigned)(k * 2048)));
            wres[k][1] = __builtin_bit_cast(bf16x8, buf_load4(wr, lane16, (unsigned)(k * 2048 + 1024)));
        }
    }
    bf16x8 xw[ENC ? 6 : 1][2];
    if constexpr (ENC) {
        const __amdgpu_buffer_rsrc_t xr = make_rsrc(a.wx + (size_t)(dir * NW + wv) * 6 * 2048);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            xw[k][0] = __builtin_bit_cast(bf16x8, buf_load4(xr, lane16, (unsigned)(k * 2048)));
            xw[k][1] = __builtin_bit_cast(bf16x8, buf_load4(xr, lane16, (unsigned)(k * 2048 + 1024)));
        }
    }
    float b_g[3][2], b_hn[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int unit = 32 * wv + 16 * nt + lr;
        b_hn[nt] = a.bias_hn[dir * HID + unit];
#pragma unroll
        for (int g = 0; g < 3; g++) b_g[g][nt] = ENC ? a.bias[dir * 3 * HID + g * HID + unit] : 0.0f;
    }
    // decoder with the dense layer folded in (see k_rec_bf16): wave w contributes k-step w of its direction's 128 units to the
    // logits of every row, as one more 16 x 16 x 32 product per tile and step; partial sums of the 2 x 4 waves go out as quads
    // [t][16-row tile][direction][wave][8 classes][16 rows]
    const bool dense = !ENC && a.dpart != nullptr;
    bf16x8 dfr[2];
    if (dense) {
        const __amdgpu_buffer_rsrc_t dr = make_rsrc(a.dw + (size_t)(dir * NW + wv) * 2048);
        dfr[0] = __builtin_bit_cast(bf16x8, buf_load4(dr, lane16, 0u));
        dfr[1] = __builtin_bit_cast(bf16x8, buf_load4(dr, lane16, 1024u));
    }
    // state and the h tiles (a workgroup's second tile may lie beyond the batch: it then repeats the first, stores nothing)
    float st[NTL][2][4];
    int64_t b0[NTL];
    bool live[NTL];
#pragma unroll
    for (int u = 0; u < NTL; u++) {
        live[u] = tile0 + u < a.n_tiles;
        b0[u] = (int64_t)(live[u] ? tile0 + u : tile0) * ROWS;
    }
    for (int i = tid; i < NTL * 2 * ROWS * HS / 16; i += 256) reinterpret_cast<u32x4*>(hbuf_all)[i] = u32x4{0u, 0u, 0u, 0u};
    if constexpr (ENC)
        for (int i = tid; i < NTL * 2 * ROWS * XS / 16; i += 256) reinterpret_cast<u32x4*>(xbuf_all)[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < NTL; u++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int j = 0; j < 4; j++) st[u][nt][j] = 0.0f;
    __syncthreads();
    unsigned hl[2];   // lane part of an h element's LDS offset: row 4q (+ j), unit of tile nt
#pragma unroll
    for (int nt = 0; nt < 2; nt++) hl[nt] = (unsigned)(4 * q * HS) + split8_off((unsigned)(32 * wv + 16 * nt + lr));
    if (a.h0) {
#pragma unroll
        for (int u = 0; u < NTL; u++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float h = a.h0[((b0[u] + 4 * q + j) * 2 + dir) * HID + 32 * wv + 16 * nt + lr];
                    st[u][nt][j] = h;
                    const __bf16 hi = (__bf16)h;
                    const __bf16 lo = (__bf16)(h - (float)hi);
                    unsigned char* hb = hbuf_all + u * 2 * ROWS * HS;
                    *reinterpret_cast<__bf16*>(hb + j * HS + hl[nt]) = hi;
                    *reinterpret_cast<__bf16*>(hb + j * HS + hl[nt] + 16) = lo;
                }
    }
    // x staging (encoder): thread -> (row tid / 16, feature tid % 16): one byte per tile and step (features beyond xf stay zero)
    const int xrow = tid >> 4, xf_i = tid & 15;
    const unsigned char* xsrc[NTL];
    unsigned xv[NTL];
    const bool x_on = ENC && xf_i < a.xf;
#pragma unroll
    for (int u = 0; u < NTL; u++) {
        xv[u] = 0u;
        xsrc[u] = nullptr;
        if constexpr (ENC) {
            int64_t r = b0[u] + xrow;
            if (r >= a.B) r = a.B - 1;
            xsrc[u] = a.x + r * a.x_row_bytes + (int64_t)a.x_t0 * a.xf + xf_i;
        }
    }
    auto x_load = [&](int t) {
        if constexpr (ENC) {
            if (x_on) {
#pragma unroll
                for (int u = 0; u < NTL; u++) xv[u] = (unsigned)xsrc[u][(int64_t)t * a.xf];
            }
        }
    };
    auto x_store = [&](int slot) {
        if constexpr (ENC) {
            if (x_on) {
#pragma unroll
                for (int u = 0; u < NTL; u++) {
                    const float f = a.x_signed ? (float)(int)(signed char)xv[u] : (float)xv[u];
                    *reinterpret_cast<__bf16*>(xbuf_all + ((u * 2 + slot) * ROWS + xrow) * XS + xf_i * 2) = (__bf16)f;
                }
            }
        }
    };
    // decoder: the step's input projections (quad row (t * Bp + b0) / 4 + q, column dir * 384 + g * 128 + unit)
    f32x4 gq[NTL][ENC ? 1 : 3][2];
    auto g_load = [&](int t) {
        if constexpr (!ENC) {
#pragma unroll
            for (int u = 0; u < NTL; u++) {
                const __amdgpu_buffer_rsrc_t gsr = make_rsrc(a.G + ((size_t)t * a.Bp + b0[u]) * NCOL);
#pragma unroll
                for (int g = 0; g < 3; g++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
                        gq[u][g][nt] = buf_load4_nt(gsr, (unsigned)((q * NCOL + lr) * 16), (unsigned)((dir * 3 * HID + g * HID + 32 * wv + 16 * nt) * 16));
            }
        }
    };
    // layer output: the finished h tile (split8 rows, 512 bytes each) as 16-byte copies, two per thread and tile
    const unsigned rowb_tm = 2u * HID * 4u, rowb_bm = (unsigned)T * 2u * HID * 4u;
    const bool want_out = a.out_tm || a.out_bm;
    auto tile_out = [&](int tt, int cur_) {
        if (!want_out) return;
#pragma unroll
        for (int u = 0; u < NTL; u++) {
            if (!live[u]) continue;
            const unsigned char* tl = hbuf_all + (u * 2 + cur_) * ROWS * HS;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int c = tid + k * 256, row = c >> 5, col = c & 31;
                const u32x4 v = *reinterpret_cast<const u32x4*>(tl + row * HS + col * 16);
                if (a.out_tm) {
                    const __amdgpu_buffer_rsrc_t tmr = make_rsrc(a.out_tm + ((size_t)tt * a.Bp + b0[u]) * rowb_tm);
                    __builtin_amdgcn_raw_buffer_store_b128(v, tmr, (unsigned)(row * (int)rowb_tm + col * 16), (unsigned)(dir * HID * 4), 2);
                }
                if (a.out_bm) {
                    const __amdgpu_buffer_rsrc_t bmr = make_rsrc(a.out_bm + (size_t)b0[u] * rowb_bm);
                    __builtin_amdgcn_raw_buffer_store_b128(v, bmr, (unsigned)row * rowb_bm + (unsigned)(col * 16), (unsigned)((tt * 2 + dir) * HID * 4), 2);
                }
            }
        }
    };
    auto dense_out = [&](int tt, int cur_) {   // partial logits of the h tiles of step tt (buffer cur_)
#pragma unroll
        for (int u = 0; u < NTL; u++) {
            if (!live[u]) continue;
            const unsigned char* At = hbuf_all + (u * 2 + cur_) * ROWS * HS + lr * HS + q * 32 + wv * 128;
            const bf16x8 dah = *reinterpret_cast<const bf16x8*>(At), dal = *reinterpret_cast<const bf16x8*>(At + 16);
            f32x4 ad = {0.0f, 0.0f, 0.0f, 0.0f};
            ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dah, dfr[0], ad, 0, 0, 0);
            ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dah, dfr[1], ad, 0, 0, 0);
            ad = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dal, dfr[0], ad, 0, 0, 0);
            if (lr < 8)   // class lr (5 real, 3 zero), rows 4q .. 4q + 3
                *reinterpret_cast<f32x4*>(a.dpart + ((((size_t)tt * (a.Bp / 16) + (size_t)(b0[u] / 16)) * 2 + dir) * 4 + wv) * 128 + lr * 16 + 4 * q) = ad;
        }
    };
    x_load(dir ? T - 1 : 0);
    x_store(0);
    g_load(dir ? T - 1 : 0);
    __syncthreads();
    int cur = 0;
    for (int s = 0; s < T; s++) {
        const int t = dir ? (T - 1 - s) : s;
        const int tn = dir ? (T - 2 - s) : (s + 1);
        if (s + 1 < T) x_load(tn);
        if (s > 0) tile_out(dir ? t + 1 : t - 1, cur);
        if (dense && s > 0) dense_out(dir ? t + 1 : t - 1, cur);   // (the previous step's h tiles: this step's A operands)
#pragma unroll
        for (int u = 0; u < NTL; u++) {
            unsigned char* hb = hbuf_all + u * 2 * ROWS * HS;
            f32x4 ar[2], az[2], anx[2], anh[2];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    ar[nt][j] = ENC ? b_g[0][nt] : gq[u][0][nt][j];
                    az[nt][j] = ENC ? b_g[1][nt] : gq[u][1][nt][j];
                    anx[nt][j] = ENC ? b_g[2][nt] : gq[u][ENC ? 0 : 2][nt][j];
                    anh[nt][j] = b_hn[nt];
                }
            }
            if constexpr (ENC) {   // byte input: exact in bf16, two terms
                const bf16x8 ax = *reinterpret_cast<const bf16x8*>(xbuf_all + ((u * 2 + (s & 1)) * ROWS + lr) * XS + q * 16);
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    ar[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, xw[0 * 2 + nt][0], ar[nt], 0, 0, 0);
                    ar[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, xw[0 * 2 + nt][1], ar[nt], 0, 0, 0);
                    az[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, xw[1 * 2 + nt][0], az[nt], 0, 0, 0);
                    az[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, xw[1 * 2 + nt][1], az[nt], 0, 0, 0);
                    anx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, xw[2 * 2 + nt][0], anx[nt], 0, 0, 0);
                    anx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax, xw[2 * 2 + nt][1], anx[nt], 0, 0, 0);
                }
            }
            // h_{t-1} . W_hh^T, three terms; unit tile by unit tile, so that the cell update of the first tile (vector unit) has
            // the second tile's MFMAs (matrix pipe) to run under
            bf16x8 ah[4], al[4];
            {
                const unsigned char* At = hb + cur * ROWS * HS + lr * HS + q * 32;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    ah[ks] = *reinterpret_cast<const bf16x8*>(At + ks * 128);
                    al[ks] = *reinterpret_cast<const bf16x8*>(At + ks * 128 + 16);
                }
            }
            unsigned char* hn = hb + (cur ^ 1) * ROWS * HS;
            auto h_part = [&](int nt) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const int s0 = (ks * 3 + 0) * 2 + nt, s1 = (ks * 3 + 1) * 2 + nt, s2 = (ks * 3 + 2) * 2 + nt;
                    ar[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wres[s0][0], ar[nt], 0, 0, 0);
                    az[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wres[s1][0], az[nt], 0, 0, 0);
                    anh[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wres[s2][0], anh[nt], 0, 0, 0);
                    ar[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wres[s0][1], ar[nt], 0, 0, 0);
                    az[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wres[s1][1], az[nt], 0, 0, 0);
                    anh[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], wres[s2][1], anh[nt], 0, 0, 0);
                    ar[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], wres[s0][0], ar[nt], 0, 0, 0);
                    az[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], wres[s1][0], az[nt], 0, 0, 0);
                    anh[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], wres[s2][0], anh[nt], 0, 0, 0);
                }
            };
            auto cell = [&](int nt) {
#pragma unroll
                for (int j = 0; j < 4; j++) {   // PyTorch GRU: n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) n + z h
                    const float r = sigmoidf_(ar[nt][j]);
                    const float z = sigmoidf_(az[nt][j]);
                    const float n = tanhf_(anx[nt][j] + r * anh[nt][j]);
                    const float h = (1.0f - z) * n + z * st[u][nt][j];
                    st[u][nt][j] = h;
                    const __bf16 hi = (__bf16)h;
                    const __bf16 lo = (__bf16)(h - (float)hi);
                    *reinterpret_cast<__bf16*>(hn + j * HS + hl[nt]) = hi;
                    *reinterpret_cast<__bf16*>(hn + j * HS + hl[nt] + 16) = lo;
                }
            };
            h_part(0);
            h_part(1);
            // the next step's projections are requested behind the last use of this step's (every tile's were copied into its
            // accumulators at the top of its pass): they travel during the cell updates and the barrier
            if (u == NTL - 1 && s + 1 < T) g_load(tn);
            cell(0);
            cell(1);
        }
        if (s + 1 < T) x_store((s + 1) & 1);
        cur ^= 1;
        lds_barrier();
    }
    tile_out(dir ? 0 : T - 1, cur);
    if (dense) dense_out(dir ? 0 : T - 1, cur);
    if (a.h_out) {
#pragma unroll
        for (int u = 0; u < NTL; u++) {
            if (!live[u]) continue;
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int j = 0; j < 4; j++) a.h_out[((b0[u] + 4 * q + j) * 2 + dir) * HID + 32 * wv + 16 * nt + lr] = st[u][nt][j];
        }
    }
}

template <int NG, bool ENC, int MT> constexpr size_t lds_rec() {
    typedef RecCfg<NG, ENC> C;
    return (size_t)2 * 32 * MT * C::HS + (ENC ? (size_t)2 * 32 * MT * C::XS : 0);
}

static inline uint16_t f2bf_bits(float x) {  // round to nearest even
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf_bits2f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}


// ---- the tail of the P1 head in this mode ---------------------------------------------------------------------------------
// k_head_tail runs linear_2..5 (four dependent 512 x 512 layers per 32-row tile) on the fp32 MFMA, every workgroup streaming
// the 4 MB of weights from L2: 0.27-0.30 ms per 8192 windows, 6 % of the bf16x3 chain. Here: 64 rows per workgroup (half the
// weight traffic per row), 3-term split products through the fragment ring of the recurrent layers, the activations as ONE
// split8 tile in LDS that is updated in place (every wave holds its 64 x 64 outputs in accumulators until all waves are done
// reading the tile). Wave w owns output columns [64 w, 64 w + 64) as two 32-column tiles ("gates" of ring_bf16).
constexpr int TL_N = 512, TL_ROWS = 64, TL_HS = TL_N * 4 + 16, TL_KS = TL_N / 16, TL_SLOTS = TL_KS * 2, TL_D = 8;
struct TailBfArgs {
    const float* part; int splits; int64_t part_rows;
    const float* b1;
    const unsigned char* wp;
    const float* b[4];
    const float* wo; const float* bo;
    float* probs; int64_t B;
    unsigned* epoch; const int* err;
};
__device__ __forceinline__ float seluf_(float x) {
    return 1.0507009873554805f * (x > 0.0f ? x : 1.6732632423543772f * (__expf(x) - 1.0f));
}
__global__ __launch_bounds__(512, 1) void k_tail_bf16(TailBfArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* tile = sm;   // [TL_ROWS][TL_HS] split8 rows
    __shared__ float logits[TL_ROWS][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), rg = lane >> 5;
    const int64_t b0 = (int64_t)blockIdx.x * TL_ROWS;
    if (a.epoch && blockIdx.x == 0 && tid == 0) atomicAdd(a.epoch, 1u);   // both LSTM kernels of this call are done (stream order)
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(a.wp + (size_t)wv * 4 * TL_SLOTS * 2048);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x4 bq[TL_D][2];
#pragma unroll
    for (int k = 0; k < TL_D - 1; k++) {
        bq[k][0] = buf_load4(wr, lane16, (unsigned)(k * 2048));
        bq[k][1] = buf_load4(wr, lane16, (unsigned)(k * 2048 + 1024));
    }
    // y = selu(sum of slabs + b1) (simple_model.py:57-59) -> split8 rows; four consecutive columns per thread and pass
    for (int i = tid; i < TL_ROWS * (TL_N / 4); i += 512) {
        const int row = i / (TL_N / 4), n4 = (i % (TL_N / 4)) * 4;
        int64_t b = b0 + row;
        if (b >= a.B) b = a.B - 1;
        f32x4 v = *reinterpret_cast<const f32x4*>(a.b1 + n4);
        for (int s = 0; s < a.splits; s++) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(a.part + ((size_t)s * a.part_rows + b) * TL_N + n4);
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] += p[j];
        }
        bf16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float y = seluf_(v[j]);
            hi[j] = (__bf16)y;
            lo[j] = (__bf16)(y - (float)hi[j]);
        }
        unsigned char* p = tile + row * TL_HS + split8_off((unsigned)n4);
        *reinterpret_cast<bf16x4*>(p) = hi;
        *reinterpret_cast<bf16x4*>(p + 16) = lo;
    }
    __syncthreads();
    const unsigned char* a_h = tile + (lane & 31) * TL_HS + rg * 32;
    auto no_hook = [](int) {};
    auto layer = [&](auto L) {   // linear_(2 + L) + SELU (:61-76), in place
        constexpr int LI = decltype(L)::value;
        f32x16 acc[2][2];
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const float bv = a.b[LI][64 * wv + 32 * g + (lane & 31)];
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[m][g][e] = bv;
        }
        ring_bf16<2, 2, 2, TL_D, 4 * TL_SLOTS, LI * TL_SLOTS, TL_KS, false>(acc, a_h, 32 * TL_HS, wr, bq, lane16, no_hook);
        __syncthreads();   // every wave has read the whole tile
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int g = 0; g < 2; g++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int row = 32 * m + 8 * (e >> 2) + (e & 3) + 4 * rg;
                    const float y = seluf_(acc[m][g][e]);
                    const __bf16 hi = (__bf16)y;
                    const __bf16 lo = (__bf16)(y - (float)hi);
                    unsigned char* p = tile + row * TL_HS + split8_off((unsigned)(64 * wv + 32 * g + (lane & 31)));
                    *reinterpret_cast<__bf16*>(p) = hi;
                    *reinterpret_cast<__bf16*>(p + 16) = lo;
                }
        __syncthreads();
    };
    layer(std::integral_constant<int, 0>());
    layer(std::integral_constant<int, 1>());
    layer(std::integral_constant<int, 2>());
    layer(std::integral_constant<int, 3>());
    // output_layer_type (512 -> 3) + softmax(dim=1) (:77-82): 8 lanes per (row, class) pair, 64 pairs per pass
    {
        const int pair = tid >> 3, sub = tid & 7;
        for (int p = pair; p < TL_ROWS * 3; p += 64) {
            const int row = p / 3, cls = p - row * 3;
            float s = 0.0f;
            for (int k8 = sub; k8 < TL_N / 8; k8 += 8) {   // one group of 8 columns: 8 hi then 8 lo
                const bf16x8 hi = *reinterpret_cast<const bf16x8*>(tile + row * TL_HS + k8 * 32);
                const bf16x8 lo = *reinterpret_cast<const bf16x8*>(tile + row * TL_HS + k8 * 32 + 16);
#pragma unroll
                for (int j = 0; j < 8; j++) s += ((float)hi[j] + (float)lo[j]) * a.wo[cls * TL_N + k8 * 8 + j];
            }
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            if (sub == 0) logits[row][cls] = s + a.bo[cls];
        }
    }
    __syncthreads();
    if (tid < TL_ROWS) {
        const int64_t b = b0 + tid;
        if (b < a.B) {
            const float l0 = logits[tid][0], l1 = logits[tid][1], l2 = logits[tid][2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            const float inv = 1.0f / (e0 + e1 + e2);
            const bool bad = a.err && a.err[0] != 0;   // a poll of a split form gave up: NaN, not stale numbers (see k_head_tail)
            const float nanv = __builtin_nanf("");
            a.probs[b * 3 + 0] = bad ? nanv : e0 * inv;
            a.probs[b * 3 + 1] = bad ? nanv : e1 * inv;
            a.probs[b * 3 + 2] = bad ? nanv : e2 * inv;
        }
    }
}
constexpr size_t LDS_TAIL_BF = (size_t)TL_ROWS * TL_HS;

}  // namespace

// Fragment stream of one (direction, wave): slots [x-part k-steps x gates | h-part k-steps x gates]; a slot is 1 KB of hi
// fragments followed by 1 KB of lo fragments; lane -> weight row gate * HID + 32 * wave + (lane & 31), 8 consecutive K values
// 16 * ks + 8 * (lane >> 5) + j (the B operand of v_mfma_f32_32x32x16_bf16). GRU encoder: the x-part goes to a stream of its own.
int pv_pack_rec_bf16(const pv_rnn_dir* dirs, int cell, int kx, unsigned char** d_wp, unsigned char** d_wx, std::vector<void*>& owned) {
    const int NG = cell, HID = cell == 4 ? 256 : 128, NW = HID / 32, KS_H = HID / 16;
    const int KS_X = kx ? (cell == 4 ? 2 : 1) : 0;
    const bool xres = kx && cell == 3;
    const int nslot = (xres ? 0 : KS_X * NG) + KS_H * NG;
    std::vector<uint16_t> wp((size_t)2 * NW * nslot * 1024), wx(xres ? (size_t)2 * NW * NG * 1024 : 0);
    for (int d = 0; d < 2; d++)
        for (int w = 0; w < NW; w++) {
            auto fill = [&](uint16_t* dst, bool xpart, int ks, int g) {
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int n = g * HID + 32 * w + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                        float v;
                        if (xpart) v = k < kx ? dirs[d].w_ih[(size_t)n * kx + k] : 0.0f;
                        else v = dirs[d].w_hh[(size_t)n * HID + k];
                        const uint16_t hi = f2bf_bits(v);
                        dst[lane * 8 + j] = hi;
                        dst[512 + lane * 8 + j] = f2bf_bits(v - bf_bits2f(hi));
                    }
            };
            uint16_t* base = wp.data() + (size_t)(d * NW + w) * nslot * 1024;
            int slot = 0;
            if (!xres)
                for (int ks = 0; ks < KS_X; ks++)
                    for (int g = 0; g < NG; g++) fill(base + (size_t)(slot++) * 1024, true, ks, g);
            for (int ks = 0; ks < KS_H; ks++)
                for (int g = 0; g < NG; g++) fill(base + (size_t)(slot++) * 1024, false, ks, g);
            if (xres)
                for (int g = 0; g < NG; g++) fill(wx.data() + ((size_t)(d * NW + w) * NG + g) * 1024, true, 0, g);
        }
    PV_HIP(hipMalloc((void**)d_wp, wp.size() * 2));
    owned.push_back(*d_wp);
    PV_HIP(hipMemcpy(*d_wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
    if (d_wx) *d_wx = nullptr;
    if (xres) {
        PV_HIP(hipMalloc((void**)d_wx, wx.size() * 2));
        owned.push_back(*d_wx);
        PV_HIP(hipMemcpy(*d_wx, wx.data(), wx.size() * 2, hipMemcpyHostToDevice));
    }
    return PV_OK;
}

// GRU weights for k_gru16_bf16: per (direction, wave) 24 slots (ks * 3 + g) * 2 + nt of W_hh and, with kx > 0, 6 slots g * 2 + nt of
// W_ih (features beyond kx zero); lane -> unit 32 w + 16 nt + (lane & 15), values k = 32 ks + 8 (lane >> 4) + j
int pv_pack_gru16_bf16(const pv_rnn_dir* dirs, int kx, unsigned char** d_wp, unsigned char** d_wx, std::vector<void*>& owned) {
    const int HID = 128, NW = 4;
    std::vector<uint16_t> wp((size_t)2 * NW * 24 * 1024), wx(kx ? (size_t)2 * NW * 6 * 1024 : 0);
    for (int d = 0; d < 2; d++)
        for (int w = 0; w < NW; w++)
            for (int g = 0; g < 3; g++)
                for (int nt = 0; nt < 2; nt++) {
                    auto fill = [&](uint16_t* dst, bool xpart, int ks) {
                        for (int lane = 0; lane < 64; lane++)
                            for (int j = 0; j < 8; j++) {
                                const int n = g * HID + 32 * w + 16 * nt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
                                const float v = xpart ? (k < kx ? dirs[d].w_ih[(size_t)n * kx + k] : 0.0f) : dirs[d].w_hh[(size_t)n * HID + k];
                                const uint16_t hi = f2bf_bits(v);
                                dst[lane * 8 + j] = hi;
                                dst[512 + lane * 8 + j] = f2bf_bits(v - bf_bits2f(hi));
                            }
                    };
                    for (int ks = 0; ks < 4; ks++) fill(wp.data() + ((size_t)(d * NW + w) * 24 + (ks * 3 + g) * 2 + nt) * 1024, false, ks);
                    if (kx) fill(wx.data() + ((size_t)(d * NW + w) * 6 + g * 2 + nt) * 1024, true, 0);
                }
    PV_HIP(hipMalloc((void**)d_wp, wp.size() * 2));
    owned.push_back(*d_wp);
    PV_HIP(hipMemcpy(*d_wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
    if (d_wx) *d_wx = nullptr;
    if (kx) {
        PV_HIP(hipMalloc((void**)d_wx, wx.size() * 2));
        owned.push_back(*d_wx);
        PV_HIP(hipMemcpy(*d_wx, wx.data(), wx.size() * 2, hipMemcpyHostToDevice));
    }
    return PV_OK;
}

int pv_pack_tail_bf16(const float* const* w, unsigned char** d_wp, std::vector<void*>& owned) {
    // per wave: [layer][k-step][column tile] slots of [hi 1 KB | lo 1 KB]; lane -> column 64 wave + 32 tile + (lane & 31),
    // values k = 16 ks + 8 (lane >> 5) + j
    std::vector<uint16_t> wp((size_t)8 * 4 * TL_SLOTS * 1024);
    for (int wv = 0; wv < 8; wv++)
        for (int l = 0; l < 4; l++)
            for (int ks = 0; ks < TL_KS; ks++)
                for (int g = 0; g < 2; g++) {
                    uint16_t* dst = wp.data() + ((((size_t)wv * 4 + l) * TL_KS + ks) * 2 + g) * 1024;
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int n = 64 * wv + 32 * g + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                            const float v = w[l][(size_t)n * TL_N + k];
                            const uint16_t hi = f2bf_bits(v);
                            dst[lane * 8 + j] = hi;
                            dst[512 + lane * 8 + j] = f2bf_bits(v - bf_bits2f(hi));
                        }
                }
    PV_HIP(hipMalloc((void**)d_wp, wp.size() * 2));
    owned.push_back(*d_wp);
    PV_HIP(hipMemcpy(*d_wp, wp.data(), wp.size() * 2, hipMemcpyHostToDevice));
    return PV_OK;
}

int pv_tail_bf16_prepare() {
    PV_HIP(hipFuncSetAttribute((const void*)k_tail_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TAIL_BF));
    return PV_OK;
}

int pv_tail_bf16_async(pv_ctx* ctx, const pv_tail_desc& d, hipStream_t st) {
    PV_CHECK(d.part && d.wp && d.probs && d.B > 0 && d.splits >= 1, PV_ERR_INVALID, "bad tail launch");
    TailBfArgs a;
    a.part = d.part; a.splits = d.splits; a.part_rows = d.part_rows; a.b1 = d.b1; a.wp = d.wp;
    for (int i = 0; i < 4; i++) a.b[i] = d.b[i];
    a.wo = d.wo; a.bo = d.bo; a.probs = d.probs; a.B = d.B; a.epoch = d.epoch; a.err = d.err;
    pv_prof_scope ps(ctx, "k_tail_bf16", st);
    k_tail_bf16<<<(unsigned)((d.B + TL_ROWS - 1) / TL_ROWS), 512, LDS_TAIL_BF, st>>>(a);
    PV_HIP(hipGetLastError());
    return PV_OK;
}

int pv_rec_bf16_prepare() {
#define PV_REC_ATTR(NG, ENC, MT) \
    PV_HIP(hipFuncSetAttribute((const void*)k_rec_bf16<NG, ENC, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rec<NG, ENC, MT>()))
    PV_REC_ATTR(4, true, 1); PV_REC_ATTR(4, true, 2); PV_REC_ATTR(4, false, 1); PV_REC_ATTR(4, false, 2);
    PV_REC_ATTR(3, true, 1); PV_REC_ATTR(3, true, 2); PV_REC_ATTR(3, false, 1); PV_REC_ATTR(3, false, 2);
#undef PV_REC_ATTR
    return PV_OK;
}

int pv_rec_bf16_async(pv_ctx* ctx, const pv_rec_desc& d, hipStream_t st) {
    PV_CHECK((d.cell == 3 || d.cell == 4) && d.T > 0 && (d.tr16 || ((d.mt == 1 || d.mt == 2) && d.Bp % (32 * d.mt) == 0)), PV_ERR_INVALID, "bad recurrent-layer launch");
    RecArgs a;
    a.G = d.G; a.wp = d.wp; a.wx = d.wx; a.bias = d.bias; a.bias_hn = d.bias_hn; a.x = (const unsigned char*)d.x;
    a.x_row_bytes = d.x_row_bytes; a.x_t0 = d.x_t0; a.xf = d.xf; a.x_signed = d.x_signed; a.B = d.B; a.Bp = d.Bp; a.T = d.T;
    a.h0 = d.h0; a.h_out = d.h_out; a.out_f32 = d.out_f32; a.out_tm = d.out_tm; a.out_bm = d.out_bm;
    a.n_tiles = d.tr16 ? 0 : (int)(d.Bp / (32 * d.mt));
    a.dw = d.dense_w; a.dpart = d.dense_part;
    PV_CHECK(!d.dense_part || (d.cell == 3 && !d.enc && d.dense_w), PV_ERR_INVALID, "the dense layer folds into the GRU decoder only");
    if (d.tr16) {   // GRU on 16-row tiles (its own fragment streams: pv_pack_gru16_bf16)
        PV_CHECK(d.cell == 3 && (d.tr16 == 1 || d.tr16 == 2) && d.Bp % 16 == 0 && !d.out_f32 && !(d.dense_part && d.enc), PV_ERR_INVALID, "bad 16-row GRU launch");
        a.n_tiles = (int)(d.Bp / 16);
        const int ntl = d.tr16;   // 16-row tiles per workgroup
        const int n_wg = (a.n_tiles + ntl - 1) / ntl;
        const unsigned grid16 = (unsigned)(((n_wg + 3) / 4) * 8);
        pv_prof_scope ps16(ctx, d.prof_name, st);
        if (ntl == 2) { if (d.enc) k_gru16_bf16<true, 2><<<grid16, 256, 0, st>>>(a); else k_gru16_bf16<false, 2><<<grid16, 256, 0, st>>>(a); }
        else { if (d.enc) k_gru16_bf16<true, 1><<<grid16, 256, 0, st>>>(a); else k_gru16_bf16<false, 1><<<grid16, 256, 0, st>>>(a); }
        PV_HIP(hipGetLastError());
        return PV_OK;
    }
    const unsigned grid = (unsigned)(((a.n_tiles + 3) / 4) * 8);
    pv_prof_scope ps(ctx, d.prof_name, st);
#define PV_REC_GO(NG, ENC, MT) k_rec_bf16<NG, ENC, MT><<<grid, RecCfg<NG, ENC>::NTHR, lds_rec<NG, ENC, MT>(), st>>>(a)
    if (d.cell == 4) {
        if (d.enc) { if (d.mt == 2) PV_REC_GO(4, true, 2); else PV_REC_GO(4, true, 1); }
        else { if (d.mt == 2) PV_REC_GO(4, false, 2); else PV_REC_GO(4, false, 1); }
    } else {
        if (d.enc) { if (d.mt == 2) PV_REC_GO(3, true, 2); else PV_REC_GO(3, true, 1); }
        else { if (d.mt == 2) PV_REC_GO(3, false, 2); else PV_REC_GO(3, false, 1); }
    }
#undef PV_REC_GO
    PV_HIP(hipGetLastError());
    return PV_OK;
}

// ---- P2 in PV_DTYPE_BF16_INPUT_GEMM: the sliding loop of pepper/modules/python/models/predict.py:47-97 layer by layer -------
// Per 100-column window: encoder layer (k_rec_bf16<3, true>: x-part in the step, output as split8 rows = the A operand of)
// the decoder's input projection as ONE GEMM over the window's 100 x B rows (k_gemm_bf16x3, N = 2 x 384, K = 256), the decoder
// layer on those projections (k_rec_bf16<3, false>), dense1 + softmax + accumulate (k_p2_dense). The hidden state travels
// between the launches in a [Bp][2][128] buffer: encoder h0 = carried state, decoder h0 = encoder final state, next window's
// h0 = decoder final state (simple_model.py:27-42). 19 x 4 launches per call, all stream work (graph-capturable).
namespace {
constexpr int P2_WIN = 100, P2_JUMP = 50, P2_F = 10, P2_H = 128, P2_NC = 5;

// dense1 (256 -> 5) + softmax + accumulate for one window: 8 lanes per (t, b) pair
__global__ __launch_bounds__(256) void k_p2_dense(const unsigned char* __restrict__ dec /*split8 [100][Bp][256]*/, int64_t Bp, int64_t B,
                                                  const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ acc,
                                                  int seq, int ws, float* __restrict__ logits) {
    const int64_t pair = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    if (pair >= (int64_t)P2_WIN * B) return;   // whole groups of 8 lanes leave together
    const int t = (int)(pair / B);
    const int64_t b = pair - (int64_t)t * B;
    // the decoder's output arrives as split8 rows (hi + lo = the fp32 value to 2^-17): lane `sub` takes the 8-unit groups
    // sub, sub + 8, sub + 16, sub + 24 (32 contiguous bytes each)
    const unsigned char* d = dec + ((size_t)t * Bp + b) * (2 * P2_H * 4);
    float lg[P2_NC] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int grp = sub + 8 * i;
        const bf16x8 hi = *reinterpret_cast<const bf16x8*>(d + grp * 32), lo = *reinterpret_cast<const bf16x8*>(d + grp * 32 + 16);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (float)hi[j] + (float)lo[j];
#pragma unroll
        for (int c = 0; c < P2_NC; c++) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + c * 2 * P2_H + grp * 8), w1 = *reinterpret_cast<const f32x4*>(W + c * 2 * P2_H + grp * 8 + 4);
            lg[c] += v[0] * w0[0] + v[1] * w0[1] + v[2] * w0[2] + v[3] * w0[3] + v[4] * w1[0] + v[5] * w1[1] + v[6] * w1[2] + v[7] * w1[3];
        }
    }
#pragma unroll
    for (int c = 0; c < P2_NC; c++) {
        lg[c] += __shfl_xor(lg[c], 4, 64);
        lg[c] += __shfl_xor(lg[c], 2, 64);
        lg[c] += __shfl_xor(lg[c], 1, 64);
        lg[c] += bias[c];
    }
    if (sub != 0) return;
    float m = lg[0];
#pragma unroll
    for (int c = 1; c < P2_NC; c++) m = fmaxf(m, lg[c]);
    float e[P2_NC], sum = 0.0f;
#pragma unroll
    for (int c = 0; c < P2_NC; c++) { e[c] = expf(lg[c] - m); sum += e[c]; }
    const float inv = 1.0f / sum;
    float* ac = acc + ((size_t)b * seq + ws + t) * P2_NC;
#pragma unroll
    for (int c = 0; c < P2_NC; c++) ac[c] += e[c] * inv;
    if (logits) {
#pragma unroll
        for (int c = 0; c < P2_NC; c++) logits[((size_t)b * P2_WIN + t) * P2_NC + c] = lg[c];
    }
}

// labels = argmax of the accumulated softmax, first maximum wins (torch.max, predict.py:91)
__global__ __launch_bounds__(256) void k_p2_argmax(const float* __restrict__ acc, int64_t n, uint8_t* __restrict__ labels) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* ac = acc + i * P2_NC;
    int best = 0;
    float bv = ac[0];
#pragma unroll
    for (int c = 1; c < P2_NC; c++) if (ac[c] > bv) { bv = ac[c]; best = c; }
    labels[i] = (uint8_t)best;
}

// state [Bp][2][128] <- hidden_in [B][2][128] (rows beyond B: zeros) or zeros;   hidden_out [B][2][128] <- state
__global__ __launch_bounds__(256) void k_p2_state_in(float* __restrict__ state, const float* __restrict__ hin, int64_t B, int64_t Bp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= Bp * 2 * P2_H) return;
    state[i] = (hin && i < B * 2 * P2_H) ? hin[i] : 0.0f;
}
__global__ __launch_bounds__(256) void k_p2_state_out(const float* __restrict__ state, float* __restrict__ hout, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < B * 2 * P2_H) hout[i] = state[i];
}

// sum of the decoder kernel's partial logits (2 directions x 4 waves) + bias -> softmax -> accumulate (predict.py:70-89);
// thread = (t, chunk). part: [100][Bp / 32][2][4][8 classes][32 rows]
template <int TR>   // rows of the decoder kernel's tiles (32, or 16: k_gru16_bf16)
__global__ __launch_bounds__(256) void k_p2_combine(const float* __restrict__ part, int64_t Bp, int64_t B, const float* __restrict__ bias,
                                                    float* __restrict__ acc, int seq, int ws, float* __restrict__ logits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)P2_WIN * B) return;
    const int t = (int)(i / B);
    const int64_t b = i - (int64_t)t * B;
    const float* p = part + (((size_t)t * (Bp / TR) + (size_t)(b / TR)) * 8) * (8 * TR) + (b % TR);
    float lg[P2_NC];
#pragma unroll
    for (int c = 0; c < P2_NC; c++) lg[c] = bias[c];
#pragma unroll
    for (int k = 0; k < 8; k++)   // (direction, wave)
#pragma unroll
        for (int c = 0; c < P2_NC; c++) lg[c] += p[k * 8 * TR + c * TR];
    float m = lg[0];
#pragma unroll
    for (int c = 1; c < P2_NC; c++) m = fmaxf(m, lg[c]);
    float e[P2_NC], sum = 0.0f;
#pragma unroll
    for (int c = 0; c < P2_NC; c++) { e[c] = expf(lg[c] - m); sum += e[c]; }
    const float inv = 1.0f / sum;
    float* ac = acc + ((size_t)b * seq + ws + t) * P2_NC;
#pragma unroll
    for (int c = 0; c < P2_NC; c++) ac[c] += e[c] * inv;
    if (logits) {
#pragma unroll
        for (int c = 0; c < P2_NC; c++) logits[((size_t)b * P2_WIN + t) * P2_NC + c] = lg[c];
    }
}
}  // namespace

int pv_pack_p2_dense16(const float* dense_w, unsigned char** d_frag, std::vector<void*>& owned) {
    // k_gru16_bf16: wave w of direction d takes k-step w (32 units); lane -> class lane & 15 (zero beyond the five), values
    // k = 32 w + 8 (lane >> 4) + j
    std::vector<uint16_t> f((size_t)2 * 4 * 1024, 0);
    for (int d = 0; d < 2; d++)
        for (int w = 0; w < 4; w++) {
            uint16_t* dst = f.data() + ((size_t)d * 4 + w) * 1024;
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int n = lane & 15, k = 32 * w + 8 * (lane >> 4) + j;
                    const float v = n < P2_NC ? dense_w[(size_t)n * 2 * P2_H + d * P2_H + k] : 0.0f;
                    const uint16_t hi = f2bf_bits(v);
                    dst[lane * 8 + j] = hi;
                    dst[512 + lane * 8 + j] = f2bf_bits(v - bf_bits2f(hi));
                }
        }
    PV_HIP(hipMalloc((void**)d_frag, f.size() * 2));
    owned.push_back(*d_frag);
    PV_HIP(hipMemcpy(*d_frag, f.data(), f.size() * 2, hipMemcpyHostToDevice));
    return PV_OK;
}

int pv_pack_p2_dense(const float* dense_w, unsigned char** d_frag, std::vector<void*>& owned) {
    // wave w of direction d takes k-steps 2w, 2w + 1 of that direction's 128 units; lane -> class lane & 31 (zero beyond the
    // five), values k = 16 ks + 8 (lane >> 5) + j
    std::vector<uint16_t> f((size_t)2 * 4 * 2 * 1024, 0);
    for (int d = 0; d < 2; d++)
        for (int w = 0; w < 4; w++)
            for (int kk = 0; kk < 2; kk++) {
                uint16_t* dst = f.data() + (((size_t)d * 4 + w) * 2 + kk) * 1024;
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int n = lane & 31, k = 16 * (2 * w + kk) + 8 * (lane >> 5) + j;
                        const float v = n < P2_NC ? dense_w[(size_t)n * 2 * P2_H + d * P2_H + k] : 0.0f;
                        const uint16_t hi = f2bf_bits(v);
                        dst[lane * 8 + j] = hi;
                        dst[512 + lane * 8 + j] = f2bf_bits(v - bf_bits2f(hi));
                    }
            }
    PV_HIP(hipMalloc((void**)d_frag, f.size() * 2));
    owned.push_back(*d_frag);
    PV_HIP(hipMemcpy(*d_frag, f.data(), f.size() * 2, hipMemcpyHostToDevice));
    return PV_OK;
}

int pv_p2_bf16_forward(pv_ctx* ctx, const pv_p2_bf16_weights& w, const uint8_t* d_images, int64_t B, uint8_t* d_labels, float* d_acc,
                       hipStream_t st, int seq, int nwin, const float* d_hidden_in, float* d_hidden_out, float* d_logits) {
    // 64-row tiles (one weight fetch feeds twice the rows) once 32-row (tile, direction) workgroups would need more than two
    // rounds of the chip; below that 32-row tiles keep more CUs busy
    const int mt = ((B + 31) / 32) * 2 > 2 * (int64_t)ctx->num_cu ? 2 : 1;
    // 16-row tiles (k_gru16_bf16: half the MFMA and cell-update time per step) while every (tile, direction) workgroup has a CU
    // of its own (up to 2048 chunks on 256 CUs: 13.2 ms against 17.9 at 2048). Beyond that, two 16-row tiles per workgroup
    // (k_gru16_bf16<.., 2>, one after the other inside the step) measured no better than the 32-row form - 6.0 / 6.3 ms against
    // 6.0 / 6.8 per 19 windows at 2121 chunks - and cannot fold dense1 in: not used.
    const int tr16 = ((B + 15) / 16) * 2 <= (int64_t)ctx->num_cu ? 1 : 0;
    const int rows = tr16 ? 16 : 32 * mt;
    const int64_t Bp = (B + rows - 1) / rows * rows, M = (int64_t)P2_WIN * Bp;
    float *state = nullptr, *G = nullptr;
    unsigned char *enc_s = nullptr, *dec = nullptr;
    int rc;
    if ((rc = pv_get(ctx, "p2b.state", (size_t)Bp * 2 * P2_H, &state))) return rc;
    if ((rc = pv_get(ctx, "p2b.enc_s", (size_t)M * 2 * P2_H * 4, &enc_s))) return rc;
    if ((rc = pv_get(ctx, "p2b.G", (size_t)M * 6 * P2_H, &G))) return rc;
    // dense1: in the 16-row form always, in the 32-row forms from 2048 chunks on, as one more MFMA tile of the decoder's steps (partial logits of 2 directions x 4 waves, 105 MB
    // per window at 4096 chunks, summed by k_p2_combine) instead of the decoder's split8 output (420 MB) and k_p2_dense's pass
    // over it: 24.9 -> 23.9 ms at 4096 chunks. It lengthens every decoder step by ~5 %, which is all a small batch sees (64
    // chunks: 12.3 -> 12.6 ms), so those keep the separate pass.
    const bool fold_dense = tr16 ? true : B >= 2048;   // (the 16-row form gains at every size: 64 chunks 6.22 -> 6.14 ms, 2048 chunks 12.1 -> 11.0)
    float* dpart = nullptr;
    if (fold_dense) {
        if ((rc = pv_get(ctx, "p2b.dpart", (size_t)P2_WIN * (Bp / (tr16 ? 16 : 32)) * 8 * (tr16 ? 128 : 256), &dpart))) return rc;
    } else if ((rc = pv_get(ctx, "p2b.dec_s", (size_t)M * 2 * P2_H * 4, &dec))) {
        return rc;
    }
    if ((rc = pv_zero_async(d_acc, (size_t)B * seq * P2_NC * sizeof(float), st))) return rc;
    k_p2_state_in<<<(unsigned)((Bp * 2 * P2_H + 255) / 256), 256, 0, st>>>(state, d_hidden_in, B, Bp);
    for (int wi = 0; wi < nwin; wi++) {
        const int ws = wi * P2_JUMP;
        pv_rec_desc e = {};
        e.cell = 3; e.enc = 1; e.wp = w.enc_wp; e.wx = w.enc_wx; e.bias = w.enc_bias; e.bias_hn = w.enc_bias_hn;
        e.x = d_images; e.x_row_bytes = (int64_t)seq * P2_F; e.x_t0 = ws; e.xf = P2_F; e.x_signed = 0;
        e.B = B; e.Bp = Bp; e.T = P2_WIN; e.h0 = state; e.h_out = state; e.out_tm = enc_s; e.mt = mt; e.prof_name = "k_rec_bf16_gru_enc";
        if (tr16) { e.tr16 = tr16; e.wp = w.enc16_wp; e.wx = w.enc16_wx; e.prof_name = "k_gru16_bf16_enc"; }
        if ((rc = pv_rec_bf16_async(ctx, e, st))) return rc;
        pv_gemm_desc g = {};
        g.A = enc_s; g.W = w.dec_wih_s; g.bias = w.dec_bias_cat; g.C = G; g.M = M; g.N = 6 * P2_H; g.K = 2 * P2_H; g.splits = 1; g.quads = 1;
        g.prof_name = "k_gemm_bf16x3_gru_dec";
        if ((rc = pv_gemm_bf16x3_async(ctx, g, st))) return rc;
        pv_rec_desc d = {};
        d.cell = 3; d.enc = 0; d.G = G; d.wp = w.dec_wp; d.bias_hn = w.dec_bias_hn; d.B = B; d.Bp = Bp; d.T = P2_WIN;
        d.h0 = state; d.h_out = state; d.mt = mt; d.prof_name = "k_rec_bf16_gru_dec";
        if (tr16) { d.tr16 = tr16; d.wp = w.dec16_wp; d.prof_name = "k_gru16_bf16_dec"; }
        float* lg_out = (d_logits && wi == nwin - 1) ? d_logits : nullptr;
        if (fold_dense) {
            d.dense_w = tr16 ? w.dense_frag16 : w.dense_frag; d.dense_part = dpart;
            if ((rc = pv_rec_bf16_async(ctx, d, st))) return rc;
            pv_prof_scope ps(ctx, "k_p2_combine", st);
            const int64_t nthr = (int64_t)P2_WIN * B;
            if (tr16) k_p2_combine<16><<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(dpart, Bp, B, w.dense_b, d_acc, seq, ws, lg_out);
            else k_p2_combine<32><<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(dpart, Bp, B, w.dense_b, d_acc, seq, ws, lg_out);
        } else {
            d.out_tm = dec;
            if ((rc = pv_rec_bf16_async(ctx, d, st))) return rc;
            pv_prof_scope ps(ctx, "k_p2_dense", st);
            const int64_t nthr = (int64_t)P2_WIN * B * 8;
            k_p2_dense<<<(unsigned)((nthr + 255) / 256), 256, 0, st>>>(dec, Bp, B, w.dense_w, w.dense_b, d_acc, seq, ws, lg_out);
        }
    }
    if (d_hidden_out) k_p2_state_out<<<(unsigned)((B * 2 * P2_H + 255) / 256), 256, 0, st>>>(state, d_hidden_out, B);
    if (d_labels) k_p2_argmax<<<(unsigned)((B * seq + 255) / 256), 256, 0, st>>>(d_acc, B * seq, d_labels);
    PV_HIP(hipGetLastError());
    return PV_OK;
}

#ifdef PV_REC_STAMPS
// diagnostic build only: phase cycle sums {x load issue, MFMA loops, cell update, next-step init, barrier} of workgroup 0 (all
// its waves added up) since the last call; resets them
extern "C" int pv_debug_rec_stamps(unsigned long long* out5) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PV_HIP(hipDeviceSynchronize());
    PV_HIP(hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_rec_stamps), 5 * sizeof(unsigned long long)));
    PV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rec_stamps), z, sizeof z));
    return PV_OK;
}
#endif
