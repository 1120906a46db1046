// rnn_kernels.hip — recurrent-network inference for PEPPER on gfx950 (MI355X), fp32 throughout.
//
// P1 (pepper_variant): images int8 [B,33,26] -> bi-LSTM(256) -> bi-LSTM(256) -> flatten 16896 ->
//     5 x (Linear 512 + SELU) -> Linear 3 -> softmax        (reference: models/simple_model.py:48-82)
//
// Kernel family:
//   k_lstm_layer<KP,INT8,TR>  one launch per LSTM layer; a workgroup (8 waves) = (TR-row batch tile, direction). Per
//        time step the gate pre-activations [TR x 1024] = [x_t | h_{t-1}] . [W_ih | W_hh]^T are ONE concatenated-K
//        product on the f32 MFMA (bitwise an fmaf chain), so the "input-projection GEMM" and the recurrent product
//        share accumulators and no pre-activation tensor ever goes to HBM. A operand (x_t, h_{t-1}) lives in LDS; the
//        B operand (weights) is pre-packed on the host in exact MFMA fragment order and streamed from L2 through a
//        four-deep register ring fed by raw buffer loads three k-blocks ahead; wave w owns hidden units
//        [32w, 32w+32) of all four gates so the cell update is purely in-register. c stays in registers for all 33
//        steps, h is exchanged between the waves through a double-buffered LDS tile. XCD-aware block mapping keeps
//        one direction's weights (<= 3 MB) per XCD L2. Two tile forms (mfma_tiles.hpp): TR = 32 on 32x32x2 MFMAs when
//        the workgroups fill the chip, TR = 16 on 16x16x4 MFMAs below that.
//   k_head_splitk          linear_1 (K = 16896) as a split-K MFMA product over time-step chunks,
//        deterministic partial slabs (no float atomics).
//   k_head_tail<TR>        sum of slabs + bias + SELU, linear_2..5 + SELU (MFMA), output layer, softmax.
//   k_gemm_bf16x3          PV_DTYPE_BF16_INPUT_GEMM: decoder input projection and linear_1 as 3-term bf16 split GEMMs on the
//        bf16 MFMA; the recurrent layers of that mode are k_rec_bf16 (rnn_rec_bf16.hip).
#include "pv_common.hpp"
#include "mfma_tiles.hpp"
#include "rnn_bf16.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

namespace {

using namespace pvdev;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int T_STEPS = 33;
constexpr int F_IN = 26;
constexpr int H = 256;            // LSTM hidden
constexpr int ROWS = 32;          // batch rows per workgroup (one MFMA M-tile)
constexpr int HEAD_N = 512;
constexpr int HEAD_K = T_STEPS * 2 * H;  // 16896
constexpr int HEAD_MAX_SPLITS = 33;      // split-K factor of linear_1 is chosen per launch from {1, 3, 11, 33}

// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence: ~3x fewer VALU instructions in the
// cell update; absolute error of sigmoid/tanh stays ~1e-7 (tests pin 2e-5 on layer outputs).
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // 1 - 2/(e^{2x}+1): saturates cleanly at +-1 (e = inf -> 1, e = 0 -> -1)
    const float e = __expf(2.0f * x);
    return __builtin_fmaf(-2.0f, rcpf_(e + 1.0f), 1.0f);
}
// c' = f * c + i * g with the fused multiply-add spelled out: left to the compiler's contraction, two instantiations of the
// same cell update (k_lstm_layer / k_lstm_split) can round differently, and their outputs are compared bit for bit
__device__ __forceinline__ float lstm_c_(float fg, float c, float ig, float gg) { return __builtin_fmaf(fg, c, ig * gg); }
__device__ __forceinline__ float seluf_(float x) {
    return 1.0507009873554805f * (x > 0.0f ? x : 1.6732632423543772f * (__expf(x) - 1.0f));
}

// Operand format of the bf16x3 GEMMs ("split8"): every 8 consecutive K elements of a row are stored as 32 bytes,
// 8 x bf16 hi followed by 8 x bf16 lo, with x ~= hi + lo, hi = bf16(x), lo = bf16(x - hi) (round to nearest even twice).
// One 16-byte half of a group is exactly one MFMA A/B fragment of v_mfma_f32_32x32x16_bf16, so the GEMM moves operands
// from HBM to LDS by LDS-DMA without touching a register. Byte offset of element k inside a row: (k >> 3) * 32 + (k & 7) * 2
// (+16 for lo).
// ---- weight-fragment rings ------------------------------------------------------------------------------------------
// The B operand (packed weights, one f32x4 per lane per gate tile per k-block of 8) comes from L2, ~1 us away, while a
// k-block is 12-16 MFMAs (0.3-0.45 us): fragments run in a ring of FOUR register sets fed by raw buffer loads THREE
// k-blocks ahead. `sched_barrier` fences keep hipcc from sinking the prefetches back next to their uses. A fragments come
// from LDS one block ahead.

// Recurrent form, acc[nt] += [A1 | A2] . B: k-blocks [0,nkb1) read A1 (x_t), [nkb1,nkb1+nkb2) read A2 (h_{t-1}); the packed
// weight stream is contiguous over both. The ring wraps into the next time step (same weights every step), so `bq` slots
// 0..2 must hold k-blocks 0..2 on entry and do so again on exit: a step never starts with a cold L2 round trip.
// (nkb1 + nkb2) % 4 == 0.
template <int TR, int NT>
__device__ __forceinline__ void mma_dual_ringb(Gate<TR> (&acc)[NT], const float* __restrict__ A1, int lda1, int nkb1,
                                               const float* __restrict__ A2, int lda2, int nkb2,
                                               __amdgpu_buffer_rsrc_t wr, f32x4 (&bq)[4][NT], int lane) {
    typedef typename AFrag<TR>::type afrag;
    constexpr int NJ = TR == 32 ? 4 : 2;
    const float* ap1 = afrag_ptr<TR>(A1, lda1, lane);
    const float* ap2 = afrag_ptr<TR>(A2, lda2, lane) - 8 * nkb1;
    const int nkb = nkb1 + nkb2;
    const unsigned lane16 = (unsigned)lane * 16u;
    afrag aq[2];
#define RB_B(slot, kbv) { _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bq[slot][nt] = buf_load4(wr, lane16, (unsigned)(((kbv) * NT + nt) * 1024)); }
#define RB_A(slot, kbv) { aq[slot] = *reinterpret_cast<const afrag*>(((kbv) < nkb1 ? ap1 : ap2) + 8 * (kbv)); }
#define RB_M(bs, as)                                                                    \
    {                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NJ; j++)                                  \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) gate_mma<TR>(acc[nt], aq[as], bq[bs][nt], j); \
    }
#define RB_F __builtin_amdgcn_sched_barrier(0);
    RB_A(0, 0)
#pragma nounroll
    for (int kb = 0; kb < nkb; kb += 4) {
        const int kw = kb + 4 < nkb ? kb + 4 : 0;
        RB_B(3, kb + 3) RB_A(1, kb + 1) RB_F RB_M(0, 0) RB_F
        RB_B(0, kw) RB_A(0, kb + 2) RB_F RB_M(1, 1) RB_F
        RB_B(1, kw + 1) RB_A(1, kb + 3) RB_F RB_M(2, 0) RB_F
        RB_B(2, kw + 2)
        if (kb + 4 < nkb) RB_A(0, kb + 4)
        RB_F RB_M(3, 1) RB_F
    }
#undef RB_B
#undef RB_A
#undef RB_M
#undef RB_F
}

// Single-operand form over a CONTINUING weight stream: multiplies k-blocks [kpos, kpos + nkb) of the stream behind `wr`
// with A[TR x 8*nkb] in LDS. On entry bq slots 0..2 hold k-blocks kpos..kpos+2; on exit they hold kpos+nkb..kpos+nkb+2,
// i.e. the next call's first blocks, so consecutive calls (time steps of linear_1) never start with a cold L2/HBM round
// trip. Requests beyond the resource's size return zeros. nkb % 4 == 0. Fragment layouts: mfma_tiles.hpp (the K order
// inside a k-block is a permutation of 0..7, which only changes the fp32 summation order).
template <int TR, int NT>
__device__ __forceinline__ void mma_stream_ringb(Gate<TR> (&acc)[NT], const float* __restrict__ A, int lda, int nkb,
                                                 __amdgpu_buffer_rsrc_t wr, int kpos, f32x4 (&bq)[4][NT], int lane) {
    typedef typename AFrag<TR>::type afrag;
    constexpr int NJ = TR == 32 ? 4 : 2;
    const float* ap = afrag_ptr<TR>(A, lda, lane);
    const unsigned lane16 = (unsigned)lane * 16u;
    afrag aq[2];
#define RB_B(slot, kbv) { _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bq[slot][nt] = buf_load4(wr, lane16, (unsigned)(((kpos + (kbv)) * NT + nt) * 1024)); }
#define RB_A(slot, kbv) { aq[slot] = *reinterpret_cast<const afrag*>(ap + 8 * (kbv)); }
#define RB_M(bs, as)                                                                    \
    {                                                                                   \
        _Pragma("unroll") for (int j = 0; j < NJ; j++)                                  \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++) gate_mma<TR>(acc[nt], aq[as], bq[bs][nt], j); \
    }
#define RB_F __builtin_amdgcn_sched_barrier(0);
    RB_A(0, 0)
#pragma nounroll
    for (int kb = 0; kb < nkb; kb += 4) {
        RB_B(3, kb + 3) RB_A(1, kb + 1) RB_F RB_M(0, 0) RB_F
        RB_B(0, kb + 4) RB_A(0, kb + 2) RB_F RB_M(1, 1) RB_F
        RB_B(1, kb + 5) RB_A(1, kb + 3) RB_F RB_M(2, 0) RB_F
        RB_B(2, kb + 6)
        if (kb + 4 < nkb) RB_A(0, kb + 4)
        RB_F RB_M(3, 1) RB_F
    }
#undef RB_B
#undef RB_A
#undef RB_M
#undef RB_F
}
template <int NT>
__device__ __forceinline__ void ring_prime(f32x4 (&bq)[4][NT], __amdgpu_buffer_rsrc_t wr, int kpos, int lane) {
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bq[q][nt] = buf_load4(wr, (unsigned)lane * 16u, (unsigned)(((kpos + q) * NT + nt) * 1024));
}

// x tiles and layer outputs are touched once: streaming (nt) forms, so that the packed weights (3.1 MB per direction in a
// 4 MB L2) stay resident. rocprofv3 FETCH_SIZE of the decoder: 664 MB per launch without the hint (weights re-fetched on
// every time step), 299 MB with it (= the 277 MB x tensor + weights); run time unchanged (the misses were Infinity Cache hits).
#define PV_XLOAD buf_load4_nt
#define PV_OSTORE buf_store1_nt

struct LstmArgs {
    const int8_t* x_i8;   // [B,33,26]    (encoder)
    const float* x_f32;   // [Bp,33,512]  (decoder; padded to whole 32-row tiles)
    const float* wp;      // packed [2 dirs][8 waves][nkb][4 gates][64][4] in the tile form of the launch
    const float* bias;    // [2][1024] b_ih + b_hh
    float* out;           // [Bp,33,512] fp32
    int64_t B;
    int n_tiles;          // tiles of TR rows
};

// One LSTM layer. KP = padded input width (multiple of 32): 32 for the encoder (26 real, from int8), 512 for the decoder.
// TR = batch rows per workgroup (mfma_tiles.hpp). Workgroup = (batch tile, direction), 8 waves (two per SIMD: one wave's
// LDS/L2 waits and cell update hide behind the other's MFMAs); wave w owns hidden units [32w, 32w+32) of the four gates.
template <int KP, bool INT8, int TR>
__global__ __launch_bounds__(512, 2) void k_lstm_layer(LstmArgs a) {
    constexpr int LDX = KP + 4, LDH = H + 4;
    constexpr int NKB_X = KP / 8, NKB_H = H / 8;
    constexpr int NT = 4;         // gates i, f, g, o
    constexpr int NE = TR / 2;    // accumulator elements per lane per gate
    constexpr int NTHR = 512;
    static_assert((NKB_X + NKB_H) % 4 == 0, "mma_dual_ringb works in groups of four k-blocks");
    extern __shared__ float smem[];
    float* xbuf = smem;               // [TR][LDX]
    float* hbuf = smem + TR * LDX;    // [2][TR][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR: bases/offsets derived from it stay scalar
    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch); give every XCD one
    // direction only so that its L2 holds a single direction's packed weights.
    const int xcd = blockIdx.x & 7;
    const int dir = xcd & 1;
    const int tile = (blockIdx.x >> 3) * 4 + (xcd >> 1);
    if (tile >= a.n_tiles) return;
    const int64_t b0 = (int64_t)tile * TR;
    const float* wp = a.wp + ((size_t)(dir * 8 + wv) * (NKB_X + NKB_H)) * NT * 256;
    const float* bias = a.bias + dir * 4 * H;

    for (int i = tid; i < 2 * TR * LDH; i += NTHR) hbuf[i] = 0.0f;
    float cst[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) cst[e] = 0.0f;
    const int unit0 = 32 * wv + lane_unit<TR>(lane);
    float bs[NT][TR == 32 ? 1 : 2];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int t2 = 0; t2 < (TR == 32 ? 1 : 2); t2++) bs[nt][t2] = bias[nt * H + unit0 + 16 * t2];
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wp);
    f32x4 bq[4][NT];
    ring_prime<NT>(bq, wr, 0, lane);

    // x_t staging through registers: the loads for step s+1 are issued before step s's MFMAs and written to LDS after
    // them, so their latency hides behind the matrix work.
    constexpr int V4 = KP / 4;                            // float4 per row (fp32 input)
    constexpr int XR = INT8 ? 1 : TR * V4 / NTHR;         // decoder: float4 per thread, rows tid/128 + 4u
    constexpr int XI = INT8 ? TR * KP / NTHR : 1;         // encoder: floats per thread, rows tid/32 + 16u
    f32x4 xr[XR];
    unsigned xi[XI];                                      // raw bytes: converted when written to LDS (a conversion right behind
                                                          // the load would make hipcc drain the vmcnt queue, ring included, per step)
    int xoff[XI];                                         // encoder: offset of (row's window, feature k) from the tile's first window
                                                          // (negative when a 16-row half tile lies wholly beyond B)
    // the decoder input is padded to whole 32-row tiles (rows beyond B replicate row B-1): no clamp, and the address is
    // (tile resource) + (lane offset, computed once) + (scalar offset of row group u and step t)
    const __amdgpu_buffer_rsrc_t xsr = make_rsrc(INT8 ? (const void*)a.wp : (const void*)(a.x_f32 + (size_t)b0 * T_STEPS * KP));
    const __amdgpu_buffer_rsrc_t osr = make_rsrc(a.out + (size_t)b0 * T_STEPS * 2 * H);
    const unsigned xg_l = (unsigned)(((tid / V4) * T_STEPS * KP + (tid % V4) * 4) * 4);
    const unsigned xl_l = (unsigned)((tid / V4) * LDX + (tid % V4) * 4);
    const unsigned xe_l = (unsigned)((tid >> 5) * LDX + (tid & 31));
    const bool xe_valid = (tid & 31) < F_IN;
    if constexpr (INT8) {
        static_assert(KP == 32, "encoder staging assumes 32 padded features");
#pragma unroll
        for (int u = 0; u < XI; u++) {
            int64_t r = (tid >> 5) + 16 * u;
            if (b0 + r >= a.B) r = a.B - 1 - b0;   // rows beyond B replicate row B-1 (finite values; never stored to the caller)
            xoff[u] = (int)(r * T_STEPS * F_IN) + ((tid & 31) < F_IN ? (tid & 31) : F_IN - 1);   // padding lanes re-read feature 25
        }
    }
    const int8_t* img0 = a.x_i8 + (size_t)b0 * T_STEPS * F_IN;  // uniform
    auto x_load = [&](int t) {
        if constexpr (INT8) {
            const int8_t* src = img0 + t * F_IN;
#pragma unroll
            for (int u = 0; u < XI; u++) xi[u] = (unsigned)(int)src[xoff[u]];
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) xr[u] = PV_XLOAD(xsr, xg_l, (unsigned)(((u * (NTHR / V4)) * T_STEPS + t) * KP * 4));
        }
    };
    auto x_store = [&]() {
        if constexpr (INT8) {
#pragma unroll
            for (int u = 0; u < XI; u++) (xbuf + u * 16 * LDX)[xe_l] = xe_valid ? (float)(int)xi[u] : 0.0f;
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) *reinterpret_cast<f32x4*>(xbuf + u * (NTHR / V4) * LDX + xl_l) = xr[u];
        }
    };
    const unsigned h_l = (unsigned)(lane_row<TR>(lane) * LDH + unit0);                      // lane part of the h tile offset
    const unsigned og_l = (unsigned)((lane_row<TR>(lane) * T_STEPS * 2 * H + unit0) * 4);   // lane part of the output byte offset
    x_load(dir ? T_STEPS - 1 : 0);
    x_store();
    __syncthreads();

    for (int s = 0; s < T_STEPS; s++) {
        const int t = dir ? (T_STEPS - 1 - s) : s;
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < T_STEPS) x_load(dir ? (T_STEPS - 2 - s) : (s + 1));
        // ---- gates = bias + [x_t | h_{t-1}] . [W_ih | W_hh]^T -------------------------------------
        Gate<TR> acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int e = 0; e < NE; e++) gate_set<TR>(acc[nt], e, bs[nt][TR == 32 ? 0 : e >> 2]);
        // h_0 = 0 (simple_model.py:50-54 passes no initial state): the recurrent product of the first step is zero, skip it
        static_assert(NKB_X % 4 == 0, "the x-only first step needs whole groups of four k-blocks");
        mma_dual_ringb<TR, NT>(acc, xbuf, LDX, NKB_X, hbuf + cur * TR * LDH, LDH, s == 0 ? 0 : NKB_H, wr, bq, lane);
        // ---- cell update (PyTorch gate order i,f,g,o) ----------------------------------------------
        float* hn = hbuf + nxt * TR * LDH;
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const float ig = sigmoidf_(gate_get<TR>(acc[0], e));
            const float fg = sigmoidf_(gate_get<TR>(acc[1], e));
            const float gg = tanhf_(gate_get<TR>(acc[2], e));
            const float og = sigmoidf_(gate_get<TR>(acc[3], e));
            const float c = lstm_c_(fg, cst[e], ig, gg);
            cst[e] = c;
            const float h = og * tanhf_(c);
            (hn + elem_row<TR>(e) * LDH + elem_unit<TR>(e))[h_l] = h;
            // outputs are padded to whole tiles: unconditional stores, tile resource + lane offset + scalar offset
            const unsigned o_s = (unsigned)((t * 2 * H + dir * H + elem_row<TR>(e) * T_STEPS * 2 * H + elem_unit<TR>(e)) * 4);
            PV_OSTORE(h, osr, og_l, o_s);
        }
        lds_barrier();  // everyone is done reading xbuf / hbuf[cur]; hbuf[nxt] is complete (LDS only: mfma_tiles.hpp)
        if (s + 1 < T_STEPS) {
            x_store();
            lds_barrier();
        }
    }
}


// ---- unit-split form of the LSTM layer: ONE small batch on the whole chip ----------------------------------------------
// A single 512-window call is 32 tiles of 16 rows x 2 directions = 64 workgroups of the form above on 256 CUs, and a launch
// is a chain of 33 dependent steps whose length is the per-step MFMA time of ONE workgroup. Here the 256 hidden units of a
// (tile, direction) are split over SP_NS = 4 workgroups of 4 waves (one per SIMD); a wave owns 16 units x 4 gates (one
// 16x16 accumulator per gate), so a step costs a quarter of the MFMA cycles. What the split costs is one exchange per step:
// every workgroup needs all 256 values of h_t for the recurrent product of step t+1. It is hidden behind the x-part of that
// step, which does not depend on h_t (two thirds of the decoder's MFMAs), and it needs no flag, fence or store drain:
//   * h leaves a lane as DATA-TAGGED 8-byte pairs {h(row r), tag}, tag = launch base + step + 1, two pairs per 16-byte
//     store, written through (sc1) to slot (step parity) of the exchange buffer; an aligned 8-byte pair lands whole;
//   * after its x-part MFMAs of the next step every thread reads the granules of the same thread of the three partner
//     workgroups with L1-bypassing (sc1) loads until their tags match (bounded), and writes them to the h tile in LDS.
//   A workgroup overwrites slot p two steps later; by then it has consumed every partner's granules of the step in between,
//   which a partner only writes after reading slot p. (PV_SPLIT_FENCES builds the counter + agent-scope release / acquire
//   form of the same exchange: 1.15 instead of 0.9x ms per 512-window call.)
// Used only while every workgroup of the launch can be resident at once (grid <= CUs).
// Weight stream of a wave: [k-block of 8][2][lane][4] = {i j0, i j1, f j0, f j1}, {g j0, g j1, o j0, o j1} with
// W[gate*H + 64*part + 16*wave + (lane&15)][8kb + 2*(lane>>4) + j], K = [x (padded) | h], through a ring of SP_D register
// sets requested SP_D-1 k-blocks ahead (a k-block is only 8 MFMAs = 256 cycles and nothing else hides L2 latency with one
// wave per SIMD); the ring wraps into the next step.
constexpr int SP_D = 12;
// NS = workgroups per (tile, direction): 4 (4 waves each, up to 512 windows on 256 CUs) or 2 (8 waves each, two per SIMD,
// up to 1024 windows). NS * threads per workgroup = 1024 in both, so the exchange buffer has one shape.
constexpr int SP_FLAG_STRIDE = 32;   // ints between two counters: every counter on a 128-byte line of its own (fence form)
constexpr int SP_SC1 = 16;           // cache policy bit 4 = sc1 on gfx94x / gfx950: write-through stores, L1-bypassing loads
constexpr int SP_HX_QUADS = 2 * 2 * 1024;          // 16-byte granules per (tile, direction): [2 slots][NS parts][2][1024 / NS threads]

struct LstmSplitArgs {
    const int8_t* x_i8;   // [B,33,26]    (encoder)
    const float* x_f32;   // [Bp,33,512]  (decoder)
    const float* wp;      // packed [2 dirs][NS parts][16 / NS waves][nkb][2][64][4]
    const float* bias;    // [2][1024]
    float* out;           // [Bp,33,512]
    u32x4* hx;            // [n_tiles*2][SP_HX_QUADS] tagged h granules
    int* flags;           // [n_tiles*2][4 parts][SP_FLAG_STRIDE] (fence form only)
    int64_t B;
    int n_tiles;          // 16-row tiles
    const unsigned* epoch;  // device word, bumped once per forward call by its last kernel (k_head_tail): tags of this launch run
    int layer;              // from (2 * epoch + layer) * 64 + 1, so a replayed hipGraph advances them like eager launches do
    int* err;             // bumped when a bounded poll of the exchange gave up (a partner workgroup never showed up)
    int spin_limit;       // polls give up after this many tries (pv_opts::exchange_spin_log2)
    int drop_part;        // diagnostic (pv_opts::debug_drop_part): the workgroups of this part leave at once; -1 = none
};

template <int KB0, int NKB, int NTOT>
__device__ __forceinline__ void ring_split(f32x4 (&acc)[4], const float* __restrict__ A, int lda, __amdgpu_buffer_rsrc_t wr,
                                           f32x4 (&bq)[SP_D][2], int lane) {
    static_assert(NTOT % SP_D == 0, "ring depth must divide the stream length");
    const float* ap = afrag_ptr<16>(A, lda, lane);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x2 aq[2];
    aq[0] = *reinterpret_cast<const f32x2*>(ap);
#pragma unroll
    for (int i = 0; i < NKB; i++) {
        {   // request block i + SP_D - 1 (wrapping into the next step's first blocks)
            const int kbv = (KB0 + i + SP_D - 1) % NTOT;
            const int slot = (KB0 + i + SP_D - 1) % SP_D;
            bq[slot][0] = buf_load4(wr, lane16, (unsigned)((kbv * 2 + 0) * 1024));
            bq[slot][1] = buf_load4(wr, lane16, (unsigned)((kbv * 2 + 1) * 1024));
        }
        if (i + 1 < NKB) aq[(i + 1) & 1] = *reinterpret_cast<const f32x2*>(ap + 8 * (i + 1));
        __builtin_amdgcn_sched_barrier(0);
        {   // j outer, gate inner: an accumulator is reused only after three other MFMAs
            const int slot = (KB0 + i) % SP_D;
            const f32x2 av = aq[i & 1];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[slot][0][j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[slot][0][2 + j], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[slot][1][j], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bq[slot][1][2 + j], acc[3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int KP, bool INT8, int SP_NS>
__global__ __launch_bounds__(1024 / SP_NS) void k_lstm_split(LstmSplitArgs a) {
    constexpr int TR = 16, LDX = KP + 4, LDH = H + 4;
    constexpr int NKB_X = KP / 8, NKB_H = H / 8, NTOT = NKB_X + NKB_H;
    constexpr int NTHR = 1024 / SP_NS, NW = NTHR / 64, UPP = H / SP_NS;   // threads, waves, units per part
    extern __shared__ float smem[];
    float* xbuf = smem;               // [16][LDX]
    float* hbuf = smem + TR * LDX;    // [2][16][LDH]: all 256 units of h
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the four parts of a (tile, direction) sit on ONE XCD (they exchange through its L2), one direction per XCD
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int part = q % SP_NS;
    const int dir = xcd & 1;
    const int tile = (q / SP_NS) * 4 + (xcd >> 1);
    if (tile >= a.n_tiles) return;   // whole groups leave together: `tile` does not depend on `part`
    if (part == a.drop_part) return; // diagnostic: a partner that never shows up (its twins' polls time out and say so)
    const int td = tile * 2 + dir;
    const int64_t b0 = (int64_t)tile * TR;
    const float* wp = a.wp + ((size_t)((dir * SP_NS + part) * NW + wv) * NTOT) * 2 * 256;
    const float* bias = a.bias + dir * 4 * H;
    const unsigned flag_base = (a.epoch[0] * 2u + (unsigned)a.layer) * 64u;   // unsigned wrap-around is harmless: tags are compared for equality
    const int unit = UPP * part + 16 * wv + (lane & 15);
    const int rowg = lane >> 4;       // this lane holds rows 4*rowg .. 4*rowg+3 of its unit

    for (int i = tid; i < 2 * TR * LDH; i += NTHR) hbuf[i] = 0.0f;
    float cst[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float bs[4];
#pragma unroll
    for (int g = 0; g < 4; g++) bs[g] = bias[g * H + unit];
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wp);
    f32x4 bq[SP_D][2];
#pragma unroll
    for (int k = 0; k < SP_D - 1; k++) {
        bq[k][0] = buf_load4(wr, (unsigned)lane * 16u, (unsigned)((k * 2 + 0) * 1024));
        bq[k][1] = buf_load4(wr, (unsigned)lane * 16u, (unsigned)((k * 2 + 1) * 1024));
    }

    // x_t staging through registers (as in k_lstm_layer): every part loads the whole 16-row x tile
    constexpr int V4 = KP / 4;
    constexpr int XR = INT8 ? 1 : TR * V4 / NTHR;   // decoder: 8 (4) float4 per thread, rows tid/128 + (NTHR/128) u
    constexpr int XI = INT8 ? TR * KP / NTHR : 1;   // encoder: 2 (1) bytes per thread, rows tid/32 + (NTHR/32) u
    constexpr int XRS = NTHR / 32;                  // encoder row stride between a thread's elements
    f32x4 xr[XR];
    unsigned xi[XI];
    int xoff[XI];
    const __amdgpu_buffer_rsrc_t xsr = make_rsrc(INT8 ? (const void*)a.wp : (const void*)(a.x_f32 + (size_t)b0 * T_STEPS * KP));
    const __amdgpu_buffer_rsrc_t osr = make_rsrc(a.out + (size_t)b0 * T_STEPS * 2 * H);
    const unsigned xg_l = (unsigned)(((tid / V4) * T_STEPS * KP + (tid % V4) * 4) * 4);
    const unsigned xl_l = (unsigned)((tid / V4) * LDX + (tid % V4) * 4);
    const unsigned xe_l = (unsigned)((tid >> 5) * LDX + (tid & 31));
    const bool xe_valid = (tid & 31) < F_IN;
    if constexpr (INT8) {
        static_assert(KP == 32, "encoder staging assumes 32 padded features");
#pragma unroll
        for (int u = 0; u < XI; u++) {
            int64_t r = (tid >> 5) + XRS * u;
            if (b0 + r >= a.B) r = a.B - 1 - b0;   // rows beyond B replicate row B-1
            xoff[u] = (int)(r * T_STEPS * F_IN) + ((tid & 31) < F_IN ? (tid & 31) : F_IN - 1);
        }
    }
    const int8_t* img0 = a.x_i8 + (size_t)b0 * T_STEPS * F_IN;
    auto x_load = [&](int t) {
        if constexpr (INT8) {
            const int8_t* src = img0 + t * F_IN;
#pragma unroll
            for (int u = 0; u < XI; u++) xi[u] = (unsigned)(int)src[xoff[u]];
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) xr[u] = PV_XLOAD(xsr, xg_l, (unsigned)(((u * (NTHR / V4)) * T_STEPS + t) * KP * 4));
        }
    };
    auto x_store = [&]() {
        if constexpr (INT8) {
#pragma unroll
            for (int u = 0; u < XI; u++) (xbuf + u * XRS * LDX)[xe_l] = xe_valid ? (float)(int)xi[u] : 0.0f;
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) *reinterpret_cast<f32x4*>(xbuf + u * (NTHR / V4) * LDX + xl_l) = xr[u];
        }
    };
    const unsigned h_l = (unsigned)(4 * rowg * LDH + unit);
    const unsigned og_l = (unsigned)((4 * rowg * T_STEPS * 2 * H + unit) * 4);
    const __amdgpu_buffer_rsrc_t hsr = make_rsrc(a.hx + (size_t)td * SP_HX_QUADS);
    // granule (slot, part, g) of thread tid: 16-byte index ((slot * SP_NS + part) * 2 + g) * NTHR + tid
    auto hx_off = [](int slot, int prt, int g) { return (unsigned)((((slot * SP_NS + prt) * 2 + g) * NTHR) * 16); };
#ifdef PV_SPLIT_FENCES
    int* flags_td = a.flags + (size_t)td * SP_NS * SP_FLAG_STRIDE;
#endif
    x_load(dir ? T_STEPS - 1 : 0);
    x_store();
    __syncthreads();

    for (int s = 0; s < T_STEPS; s++) {
        const int t = dir ? (T_STEPS - 1 - s) : s;
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < T_STEPS) x_load(dir ? (T_STEPS - 2 - s) : (s + 1));
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; g++) acc[g] = f32x4{bs[g], bs[g], bs[g], bs[g]};
        ring_split<0, NKB_X, NTOT>(acc, xbuf, LDX, wr, bq, lane);          // x-part: independent of h_{t-1}
        if (s > 0) {
            // ---- the other three quarters of h_{t-1}: the granules of this thread's twins in the partner workgroups ----
            const unsigned want = flag_base + (unsigned)s;
            const int ps = (s - 1) & 1;
            u32x4 fq[SP_NS - 1][2];
#ifdef PV_SPLIT_FENCES
            if (tid == 0) {
#pragma unroll
                for (int k = 1; k < SP_NS; k++) {
                    const int* f = flags_td + ((part + k) % SP_NS) * SP_FLAG_STRIDE;
                    int spins = 0;
                    while ((unsigned)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > a.spin_limit) { atomicAdd(a.err, 1); break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < SP_NS - 1; u++)
#pragma unroll
                for (int g = 0; g < 2; g++)
                    fq[u][g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(hsr, (unsigned)tid * 16u, hx_off(ps, (part + 1 + u) % SP_NS, g), 0));
#else
            int spins = 0;
            while (true) {   // bounded: a lost partner ends in wrong results, never in a hung device
                bool ok = true;
#pragma unroll
                for (int u = 0; u < SP_NS - 1; u++)
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        fq[u][g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(hsr, (unsigned)tid * 16u, hx_off(ps, (part + 1 + u) % SP_NS, g), SP_SC1));
                        ok = ok && fq[u][g][1] == want && fq[u][g][3] == want;
                    }
                asm volatile("" ::: "memory");   // the loads are repeated, not hoisted
                if (ok) break;
                if (++spins > a.spin_limit) { atomicAdd(a.err, 1); break; }
            }
#endif
            float* hc = hbuf + cur * TR * LDH;
#pragma unroll
            for (int u = 0; u < SP_NS - 1; u++) {
                const int fp = (part + 1 + u) % SP_NS;
                float* dst = hc + 4 * ((tid & 63) >> 4) * LDH + UPP * fp + 16 * (tid >> 6) + (tid & 15);
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    // (elements go through scalars: hipcc 7.2 compiles __builtin_bit_cast(float, vec[2]) as element 0)
                    const unsigned lo = fq[u][g][0], hi = fq[u][g][2];
                    dst[(2 * g + 0) * LDH] = __builtin_bit_cast(float, lo);
                    dst[(2 * g + 1) * LDH] = __builtin_bit_cast(float, hi);
                }
            }
        }
        lds_barrier();   // h tile complete; every wave is past its x-part (x_store below overwrites xbuf)
        ring_split<NKB_X, NKB_H, NTOT>(acc, hbuf + cur * TR * LDH, LDH, wr, bq, lane);   // h-part (h_0 = 0: zeros at s = 0)
        // ---- cell update (gate order i,f,g,o) ---------------------------------------------------------------------
        float* hn = hbuf + nxt * TR * LDH;
        float hv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float ig = sigmoidf_(acc[0][i]);
            const float fg = sigmoidf_(acc[1][i]);
            const float gg = tanhf_(acc[2][i]);
            const float og = sigmoidf_(acc[3][i]);
            const float c = lstm_c_(fg, cst[i], ig, gg);
            cst[i] = c;
            hv[i] = og * tanhf_(c);
        }
        if (s + 1 < T_STEPS) {   // the exchange first: it is what the partners wait for
            const unsigned tag = flag_base + (unsigned)s + 1u;
            // 8-byte stores, one per pair: a 16-byte buffer store with an SGPR soffset has a data hazard on gfx950 that hipcc
            // (ROCm 7.2) does not pad (see k_gemm_bf16x3), and an inline-asm store would hide an entry of the vmcnt queue from
            // the compiler's counted waits on the weight ring
#pragma unroll
            for (int i = 0; i < 4; i++) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 pr = {__builtin_bit_cast(unsigned, hv[i]), tag};
                __builtin_amdgcn_raw_buffer_store_b64(pr, hsr, (unsigned)tid * 16u + 8u * (i & 1), hx_off(cur, part, i >> 1), SP_SC1);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            (hn + i * LDH)[h_l] = hv[i];
            PV_OSTORE(hv[i], osr, og_l, (unsigned)((t * 2 * H + dir * H + i * T_STEPS * 2 * H) * 4));
        }
        if (s + 1 < T_STEPS) x_store();
#ifdef PV_SPLIT_FENCES
        __syncthreads();   // stores have left, LDS tiles (x_{t+1}, own slice of h_t) are complete
        if (s + 1 < T_STEPS && tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flags_td + part * SP_FLAG_STRIDE, (int)(flag_base + (unsigned)s + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#else
        lds_barrier();     // LDS tiles (x_{t+1}, own slice of h_t) are complete; nothing in flight is waited for
#endif
    }
}

struct HeadArgs {
    const float* dec;     // [B,33,512]
    const float* w1p;     // packed linear_1 [4 waves][2112 kb][4][64][4]
    float* part;          // [splits][B][512]
    int64_t B;
    int n_tiles;
    int splits;           // divides 33
    int steps_per_split;
    int per_xcd;          // > 0: XCD-aware order (see k_head_splitk); 0: tile-major order
};

__global__ __launch_bounds__(256, 1) void k_head_splitk(HeadArgs a) {
    constexpr int KC = 2 * H;  // 512 k per time step
    constexpr int LDA = KC + 4;
    extern __shared__ float smem[];
    float* abuf = smem;  // [32][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so XCD x runs blocks x, x+8, ... Give it a
    // CONTIGUOUS range of the split-major list (split, tile): its 32 CUs then stream the same 1/splits slice of the
    // packed linear_1 weights at the same time and the slice passes through that XCD's L2 once, instead of every XCD
    // pulling all 34.6 MB for every round of tiles.
    int tile, split;
    if (a.per_xcd > 0) {
        const int l = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= a.per_xcd || l >= a.n_tiles * a.splits) return;
        split = l / a.n_tiles; tile = l - split * a.n_tiles;
    } else {
        tile = blockIdx.x / a.splits; split = blockIdx.x - tile * a.splits;
    }
    const int64_t b0 = (int64_t)tile * ROWS;
    Gate<32> acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt].v[r] = 0.0f;
    // this wave's slice of the packed linear_1 weights is one contiguous stream over all 33 time steps
    const __amdgpu_buffer_rsrc_t wr = make_rsrc_sized(a.w1p + (size_t)wv * (HEAD_K / 8) * 4 * 256, (unsigned)((HEAD_K / 8) * 4 * 256 * sizeof(float)));
    f32x4 bq[4][4];
    ring_prime<4>(bq, wr, split * a.steps_per_split * (KC / 8), lane);
    // A tile (32 rows x 512 of the decoder output at time step t; the buffer is padded to whole tiles) staged through
    // registers: the loads of step st+1 are issued before step st's MFMAs and written to LDS after them.
    constexpr int V4 = KC / 4, XR = ROWS * V4 / 256;   // 16 float4 per thread, rows tid/128 + 2u
    const __amdgpu_buffer_rsrc_t asr = make_rsrc(a.dec + (size_t)b0 * T_STEPS * KC);
    const unsigned a_g = (unsigned)(((tid >> 7) * T_STEPS * KC + (tid & 127) * 4) * 4);
    const unsigned a_l = (unsigned)((tid >> 7) * LDA + (tid & 127) * 4);
    f32x4 xr[XR];
    auto a_load = [&](int t) {
#pragma unroll
        for (int u = 0; u < XR; u++) xr[u] = buf_load4_nt(asr, a_g, (unsigned)((2 * u * T_STEPS + t) * KC * 4));
    };
    auto a_store = [&]() {
#pragma unroll
        for (int u = 0; u < XR; u++) *reinterpret_cast<f32x4*>(abuf + 2 * u * LDA + a_l) = xr[u];
    };
    a_load(split * a.steps_per_split);
    a_store();
    __syncthreads();
    for (int st = 0; st < a.steps_per_split; st++) {
        const int t = split * a.steps_per_split + st;
        if (st + 1 < a.steps_per_split) a_load(t + 1);
        mma_stream_ringb<32, 4>(acc, abuf, LDA, KC / 8, wr, t * (KC / 8), bq, lane);
        if (st + 1 < a.steps_per_split) {
            lds_barrier();  // every wave is done reading abuf
            a_store();
            lds_barrier();
        }
    }
    float* dst = a.part + (size_t)split * a.B * HEAD_N;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        const int n = 128 * wv + 32 * nt + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t b = b0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (b < a.B) dst[b * HEAD_N + n] = acc[nt].v[r];
        }
    }
}

struct TailArgs {
    const float* part;   // [splits][part_rows][512]
    int splits;
    int64_t part_rows;   // rows per slab (B, or the tile-padded batch in bf16 mode)
    const float* b1;     // [512]
    const float* wp[4];  // packed linear_2..5 [4 waves][64 kb][4][64][4]
    const float* b[4];   // [512]
    const float* wo;     // [3][512]
    const float* bo;     // [3]
    float* probs;        // [B,3]
    int64_t B;
    unsigned* epoch;     // the forward-call counter of the unit-split LSTM form: bumped here, by the call's last kernel
    const int* err;      // exchange time-outs of the unit-split LSTM form since the host last acknowledged them: while it is
                         // non-zero every call's probabilities leave as NaN (both LSTM kernels of this call are complete when
                         // this kernel runs, so the word is stable here)
};

// TR = batch rows per workgroup: 32, or 16 when 32-row tiles would leave CUs idle (the tail is a chain of four dependent
// 512x512 layers per tile). wp[layer] is the packed stream of the matching tile form (pack_linear).
template <int TR>
__global__ __launch_bounds__(256, 1) void k_head_tail(TailArgs a) {
    constexpr int LDY = HEAD_N + 4;
    constexpr int NE = TR / 2;
    extern __shared__ float smem[];
    float* y0 = smem;             // [TR][LDY]
    float* y1 = smem + TR * LDY;  // [TR][LDY]
    __shared__ float logits[TR][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * TR;
    if (a.epoch && blockIdx.x == 0 && tid == 0) atomicAdd(a.epoch, 1u);   // both LSTM kernels of this call are done (stream order)
    // y0 = selu(sum of slabs + b1)   (simple_model.py:57-59)
    for (int i = tid; i < TR * HEAD_N; i += 256) {
        const int row = i / HEAD_N, n = i - row * HEAD_N;
        int64_t b = b0 + row;
        if (b >= a.B) b = a.B - 1;
        float v = a.b1[n];
#pragma unroll 11
        for (int s = 0; s < a.splits; s++) v += a.part[((size_t)s * a.part_rows + b) * HEAD_N + n];
        y0[row * LDY + n] = seluf_(v);
    }
    __syncthreads();
    float* src = y0;
    float* dst = y1;
    const int unit0 = 128 * wv + lane_unit<TR>(lane);  // + 32*nt + elem_unit(e)
    for (int layer = 0; layer < 4; layer++) {  // linear_2..5 + SELU (:61-76)
        Gate<TR> acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int e = 0; e < NE; e++) gate_set<TR>(acc[nt], e, a.b[layer][unit0 + 32 * nt + elem_unit<TR>(e)]);
        {
            const __amdgpu_buffer_rsrc_t wr = make_rsrc_sized(a.wp[layer] + (size_t)wv * (HEAD_N / 8) * 4 * 256,
                                                              (unsigned)((HEAD_N / 8) * 4 * 256 * sizeof(float)));
            f32x4 bq[4][4];
            ring_prime<4>(bq, wr, 0, lane);
            mma_stream_ringb<TR, 4>(acc, src, LDY, HEAD_N / 8, wr, 0, bq, lane);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int e = 0; e < NE; e++)
                dst[(lane_row<TR>(lane) + elem_row<TR>(e)) * LDY + unit0 + 32 * nt + elem_unit<TR>(e)] = seluf_(gate_get<TR>(acc[nt], e));
        __syncthreads();
        float* tmp = src; src = dst; dst = tmp;
    }
    // output_layer_type (512 -> 3) + softmax(dim=1) (:77-82): 8 lanes per (row, class) pair
    {
        const int pair = tid >> 3, sub = tid & 7;  // 32 pairs per pass
        for (int p = pair; p < TR * 3; p += 32) {
            const int row = p / 3, cls = p - row * 3;
            float s = 0.0f;
            for (int k = sub; k < HEAD_N; k += 8) s += src[row * LDY + k] * a.wo[cls * HEAD_N + k];
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            if (sub == 0) logits[row][cls] = s + a.bo[cls];
        }
    }
    __syncthreads();
    if (tid < TR) {
        const int64_t b = b0 + tid;
        if (b < a.B) {
            const float l0 = logits[tid][0], l1 = logits[tid][1], l2 = logits[tid][2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            const float inv = 1.0f / (e0 + e1 + e2);
            // a poll of the unit-split form gave up (a partner workgroup was not resident): the numbers below are computed from
            // stale hidden state. They must not look like results: NaN until the host acknowledges (pv_rnn_exchange_timeouts).
            const bool bad = a.err && a.err[0] != 0;
            const float nanv = __builtin_nanf("");
            a.probs[b * 3 + 0] = bad ? nanv : e0 * inv;
            a.probs[b * 3 + 1] = bad ? nanv : e1 * inv;
            a.probs[b * 3 + 2] = bad ? nanv : e2 * inv;
        }
    }
}

// ---- PV_DTYPE_BF16_INPUT_GEMM: the non-recurrent products on the bf16 MFMA with a 3-term split ------------------
// x = x_hi + x_lo, w = w_hi + w_lo (bf16 each); x.w ~= x_hi.w_hi + x_hi.w_lo + x_lo.w_hi, accumulated in fp32: relative
// error ~2^-17 per term (the dropped lo.lo term), i.e. fp32-class results (softmax error ~7e-6 measured) at 3/16 of the
// f32-MFMA cost. Plain bf16 operands miss the 1e-4 bar (3.5e-3). Used for the decoder input projection
// G[b,t,:] = W_ih . enc_out[b,t,:] + b (both directions, N = 2048, K = 512) and for linear_1 (N = 512, K = 16896);
// the recurrent h-part, the cell update and linear_2..5 stay fp32.
struct GemmArgs {
    const unsigned char* A;  // activations, split8 rows of K*4 bytes, [M][K]
    const unsigned char* W;  // weights, split8 rows, [N][K]
    const float* bias;    // [N] or NULL
    float* C;             // [splits][M][N] row-major, or (c_quads) [M/4][N][4]: four consecutive rows of a column adjacent
    int64_t M;
    int N, K, splits;
    int tiles_m, tiles_n; // 256 x 256 output tiles
    int items;            // tiles_m * tiles_n * splits work items, walked by persistent workgroups
    int c_quads;
    const unsigned* iota; // [1024] = 0 .. 1023: the source of the completion-flag transfers (below)
};

// C = A . W^T (+ bias) with 3-term split-bf16 products (a_hi.w_hi + a_hi.w_lo + a_lo.w_hi) on v_mfma_f32_16x16x32_bf16 (until late in
// round 3: 32x32x16 - same cycles per K step, but the chip sustains more on the 16x16x32 shape and the chain runs 1.4 % faster).
// Persistent workgroups (one per CU, 8 waves as 2 x 4, 128 x 64 outputs per wave = 32 accumulator tiles of 16 x 16) walk
// 256 x 256 output tiles; K in steps of 32. Per K step a workgroup moves 64 KB ({A,W} x {hi,lo} x 256 rows x 64 B) from
// L2/HBM straight into LDS with 64 LDS-DMA wave-instructions (buffer_load_dwordx4 ... lds, 8 per wave, no registers, no
// ds_write): the transfers of step k+1 run during step k's MFMAs into the other half of a double-buffered image. An LDS-DMA
// lands lane-linear (wave base + 16 B per lane), so the XOR swizzle that makes the fragment ds_read_b128 conflict-free
// (16-byte chunk index ^ ((row >> 2) & 3) inside each 64-byte row) is applied on the SOURCE address of every lane. One
// barrier per K step; per step and wave 24 ds_read_b128 feed 96 MFMAs of 16 cycles. Work items are ordered n-tile fastest and dealt to the
// XCDs in groups of one XCD's workgroups, so the CUs of an XCD work on the same few A row tiles at the same time. The K steps
// of a workgroup form ONE stream across its items: the first K step of the next item is requested during the last K step of
// this one like any other (buffer = step parity), so an item ends with its 32 dwordx4 stores per wave (quad layout) and the
// next one's operands are there when they are done; item coordinates advance by a constant step (no division per item).
// (Before: coordinates by division, resources, ten LDS-DMA issues and a barrier BETWEEN two items: 4.6 k + 1.1 k of the
// 13.9 k cycles an item boundary cost, the stores 5.1 k: -DPV_GEMM_STAMPS.)
#ifdef PV_GEMM_STAMPS
// diagnostic build: cycle sums of the K loop's phases (workgroup 0, every wave adds): {wait for the transfers, barrier, MFMA block,
// epilogue + next tile's set-up}, and the K steps counted; read by pv_debug_gemm_stamps
__device__ unsigned long long g_gemm_stamps[8];
#define GSTAMP(i) { unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); gs_acc[i] += now_ - gs_last; gs_last = now_; }
#else
#define GSTAMP(i)
#endif
__global__ __launch_bounds__(512, 2) void k_gemm_bf16x3(GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 32;
    constexpr int ARR = BM * BK * 2;          // bytes of one bf16 operand image (16 KB); buffer = [A_hi | A_lo | W_hi | W_lo]
    extern __shared__ __attribute__((aligned(16))) unsigned char smg[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wc = wv & 3;
    // DMA role: piece p (0, 1) of this wave covers rows 16 * (8p + wv) .. + 15 of an operand image; lane -> row lane >> 2,
    // destination chunk lane & 3, source chunk (lane & 3) ^ ((lane >> 4) & 3) (= ((row >> 2) & 3) swizzle)
    const int d_row = lane >> 2;
    const unsigned d_chunk = (unsigned)((lane & 3) ^ ((lane >> 4) & 3));
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int kslice = g.K / g.splits, nk = kslice / BK;
    const unsigned row_bytes = (unsigned)g.K * 4u;

    // An item's operands: buffer resources of its A row tile / W row tile (+ K slice) and the lane offsets of this wave's pieces.
    // TWO sets: the item being multiplied and the one after it, whose first K step is requested during the last K step of this one.
    struct Item { __amdgpu_buffer_rsrc_t ra, rw; unsigned la[2], lw[2]; int mt, nt, sp; };
    // items of this workgroup: it0, it0 + grid, it0 + 2 grid, ... (group q of per_xcd consecutive items goes to the XCD class
    // q % 8: it = (xcd + 8 q) per_xcd + slot). Coordinates advance by a constant step: no division per tile.
    const int step_n = (int)gridDim.x % g.tiles_n, step_r = (int)gridDim.x / g.tiles_n;
    auto item_setup = [&](Item& I) {
        const int64_t m0 = (int64_t)I.mt * BM;
        const int n0 = I.nt * BN;
        I.ra = make_rsrc(g.A + ((size_t)m0 * g.K + (size_t)I.sp * kslice) * 4);
        I.rw = make_rsrc(g.W + ((size_t)n0 * g.K + (size_t)I.sp * kslice) * 4);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            int64_t r = 16 * (8 * p + wv) + d_row;
            if (m0 + r >= g.M) r = g.M - 1 - m0;             // rows beyond M replicate the last row (never stored)
            I.la[p] = (unsigned)r * row_bytes + d_chunk * 32u;
            int rn = 16 * (8 * p + wv) + d_row;
            if (n0 + rn >= g.N) rn = g.N - 1 - n0;
            I.lw[p] = (unsigned)rn * row_bytes + d_chunk * 32u;
        }
    };
    // piece j = 0..7 of this wave for K step kt of item I into buffer buf: (p = j >> 2) x {A_hi, A_lo, W_hi, W_lo}
    auto dma_piece = [&](const Item& I, int kt, int buf, int j) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        unsigned char* base = smg + buf * 4 * ARR + wv * 1024;
        const unsigned so = (unsigned)(kt * BK * 4);
        const int p = j >> 2, arr = j & 3;
#ifdef PV_GEMM_ABL   // timing experiment (wrong results): bit 0 drops the A pieces, bit 1 the W pieces
        if (((PV_GEMM_ABL) & 1) && arr < 2) return;
        if (((PV_GEMM_ABL) & 2) && arr >= 2) return;
#endif
        if (arr < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(I.ra, (lds_ptr)(base + arr * ARR + p * 8192), 16, I.la[p], so + (arr & 1 ? 16u : 0u), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(I.rw, (lds_ptr)(base + arr * ARR + p * 8192), 16, I.lw[p], so + (arr & 1 ? 16u : 0u), 0, 0);
    };
    // the bias of an item's 256 columns also arrives by LDS-DMA (one 1 KB piece, wave 0) into one of two slots behind the operand
    // buffers (items alternate: the next item's bias travels while this one's is still to be read): the kernel holds no
    // register-destination load at all (one consumed while LDS-DMAs are in flight makes hipcc drain the whole vmcnt queue there)
    float* sbias = reinterpret_cast<float*>(smg + 2 * 4 * ARR);
    auto dma_bias = [&](int nt_, int bslot) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        if (g.bias && wv == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(make_rsrc(g.bias + nt_ * BN), (lds_ptr)(sbias + bslot * BN), 16, (unsigned)lane * 16u, 0, 0, 0);
    };
#ifndef PV_GEMM_VMCNT
    // Completion of a K step's transfers WITHOUT s_waitcnt vmcnt: the counter also holds the epilogue stores of the previous
    // tile (256 KB per workgroup). Loads complete in issue order, so behind the pieces of a K step every wave issues one more
    // LDS-DMA that copies the word iota[seq] into its own flag slot, and polls that slot in LDS until the word is there: the
    // pieces before it have landed, whatever the stores are doing.
    unsigned* sflag = reinterpret_cast<unsigned*>(smg + 2 * 4 * ARR + 2048) + wv * 64;   // 256 B per wave
    sflag[lane] = 0xFFFFFFFFu;
    unsigned seq = 0;
    const __amdgpu_buffer_rsrc_t riota = make_rsrc(g.iota);
    auto dma_flag = [&]() {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        seq = (seq + 1u) & 1023u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(riota, (lds_ptr)sflag, 4, 0u, seq * 4u, 0, 0);
    };
    // (the poll is a ds_read_b32 from inline asm: a C++ load of LDS that may alias an LDS-DMA target makes hipcc wait for
    // vmcnt(0) in front of it - the very wait this replaces - and a volatile one becomes a flat load)
    const unsigned flag_addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)(sflag + lane);
    auto dma_wait = [&]() {
        while (true) {
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(flag_addr) : "memory");
            if (__builtin_amdgcn_ballot_w64(v != seq) == 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    auto dma_flag = [&]() {};
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#endif
    int it = xcd * per_xcd + slot;
    Item cur, nxt;
    cur.mt = cur.nt = cur.sp = 0;
    int r_cur = 0;                                // it / tiles_n of the current item
    unsigned gstep = 0;                           // K steps of this workgroup so far: step s uses operand buffer s & 1, across items
    int tile_no = 0;                              // items so far: bias slot tile_no & 1
    if (it < g.items) {
        cur.nt = it % g.tiles_n;
        r_cur = it / g.tiles_n;
        cur.sp = r_cur % g.splits;
        cur.mt = r_cur / g.splits;
        item_setup(cur);
#pragma unroll
        for (int j = 0; j < 8; j++) dma_piece(cur, 0, 0, j);
        dma_bias(cur.nt, 0);
        dma_flag();
    }
#ifdef PV_GEMM_STAMPS
    unsigned long long gs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gs_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(gs_last) :: "memory");
#endif
    while (it < g.items) {
        // the item after this one (its first K step is requested during this item's last)
        const int it_n = it + (int)gridDim.x;
        const bool has_next = it_n < g.items;
        int r_n = r_cur + step_r;
        nxt.nt = cur.nt + step_n;
        if (nxt.nt >= g.tiles_n) { nxt.nt -= g.tiles_n; r_n++; }
        if (g.splits == 1) { nxt.sp = 0; nxt.mt = r_n; }
        else { nxt.sp = r_n % g.splits; nxt.mt = r_n / g.splits; }
        if (has_next) item_setup(nxt);
        GSTAMP(6)
        f32x4 acc[8][4];   // 16 x 16 tiles of this wave's 128 x 64 outputs
#pragma unroll
        for (int mi = 0; mi < 8; mi++)
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[mi][ni][r] = 0.0f;
        // one K step: wait for its operands, multiply, and request the operands of the step after it - K step s_kt of item S -
        // into the other buffer (issue == false: there is no such step)
        auto k_step = [&](const Item& S, int s_kt, bool issue, bool next_bias) {
            const int buf = (int)(gstep & 1u);
            // this step's operands have landed (every wave waits for its own DMAs, then the barrier); the other buffer is free:
            // every wave is past the step that read it.
            // (a counted wait that lets the previous item's epilogue stores stay in flight is NOT safe here: vmcnt retires loads
            // in order among loads and stores among stores, but a store may retire before an older LDS-DMA load, so "at most 32
            // outstanding" does not imply the DMAs have landed. It gave a sporadic 1e-4 error in one of five full test runs.)
            GSTAMP(3)
            dma_wait();
            GSTAMP(0)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            GSTAMP(1)
            const unsigned char* base = smg + buf * 4 * ARR;
            // the next step's eight pieces are issued between the blocks of this step's MFMAs instead of as a burst behind the
            // barrier: an LDS-DMA costs its wave ~100-150 issue cycles, and the two waves of a SIMD - phase-locked by the barrier
            // - both paid the eight of them before either issued an MFMA (~1200 of a K step's ~4300 cycles). Two per block in the
            // FIRST half of the step: one per block over the whole step left the last pieces ~400 cycles to land before the step
            // ended (the step then waited ~650 cycles for them); four or eight in front of the first blocks stall the pipe again
            // (decoder projection, same box: 1.49 ms one per block, 1.43 two, 1.48 four, 1.46 eight).
            // v_mfma_f32_16x16x32_bf16: a K step is ONE k-step of 32; lane -> row / column lane & 15, 16-byte chunk lane >> 4 of the
            // 64-byte row (under the image's XOR swizzle: conflict-free as the 32 x 32 x 16 reads were). Same MFMA cycles and LDS
            // reads per K step as the 32 x 32 x 16 form, but the chip sustains ~14 % more FLOP/s on this shape under load
            // (tools/dbg/src/mfma_shapes.hip: 2.42 against 2.12 PFLOP/s in a bare loop - the kernel runs power-limited at
            // ~1.8 GHz).
            {
                const unsigned ch16 = (unsigned)((((lane >> 4)) ^ (((lane & 15) >> 2) & 3)) * 16);
                const unsigned fa16 = (unsigned)((128 * wr + (lane & 15)) * 64), fb16 = (unsigned)(2 * ARR + (64 * wc + (lane & 15)) * 64);
                bf16x8 bh[4], bl[4];
#pragma unroll
                for (int ni = 0; ni < 4; ni++) {
                    bh[ni] = *reinterpret_cast<const bf16x8*>(base + fb16 + ni * 16 * 64 + ch16);
                    bl[ni] = *reinterpret_cast<const bf16x8*>(base + ARR + fb16 + ni * 16 * 64 + ch16);
                }
#pragma unroll
                for (int mi = 0; mi < 8; mi++) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(base + fa16 + mi * 16 * 64 + ch16);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(base + ARR + fa16 + mi * 16 * 64 + ch16);
                    if (issue && mi < 4) {   // two pieces in front of each of the first four blocks
                        dma_piece(S, s_kt, buf ^ 1, 2 * mi);
                        dma_piece(S, s_kt, buf ^ 1, 2 * mi + 1);
                        if (mi == 3) {
                            if (next_bias) dma_bias(nxt.nt, (tile_no + 1) & 1);
                            dma_flag();
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < 4; ni++) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ni], acc[mi][ni], 0, 0, 0);
                    }
                }
            }
            GSTAMP(2)
#ifdef PV_GEMM_STAMPS
            gs_acc[4]++;
#endif
            gstep++;
        };
        for (int kt = 0; kt + 1 < nk; kt++) k_step(cur, kt + 1, true, false);
        k_step(nxt, 0, has_next, true);          // the last step requests step 0 of the next item (and its bias)
        // C tile as a sized buffer resource: rows beyond M fall outside it and are dropped by the bounds check; the
        // address of every store is (tile resource) + (lane offset) + (scalar offset of (wave, mi, ni, r))
        const int64_t m0 = (int64_t)cur.mt * BM;
        const int n0 = cur.nt * BN;
        const int64_t rows_left = g.M - m0 < BM ? g.M - m0 : BM;
        float* cbase = g.C + (size_t)cur.sp * g.M * g.N;
        float bv[4];
        // (this item's bias slot is written again during the last K step of the NEXT item: every wave has read it long before)
#pragma unroll
        for (int ni = 0; ni < 4; ni++) bv[ni] = g.bias ? sbias[(tile_no & 1) * BN + 64 * wc + 16 * ni + (lane & 15)] : 0.0f;
        GSTAMP(5)
        if (g.c_quads) {
            // [M/4][N][4]: the four registers of a 16 x 16 accumulator are four consecutive rows of one column: one 16-byte store,
            // lane -> quad row lane >> 4, column lane & 15
            const uint64_t cptr = (uint64_t)(cbase + ((size_t)(m0 >> 2) * g.N + n0) * 4);
            u32x4 rcw;
            rcw[0] = __builtin_amdgcn_readfirstlane((unsigned)cptr);
            rcw[1] = __builtin_amdgcn_readfirstlane((unsigned)(cptr >> 32) & 0xffffu);
            rcw[2] = __builtin_amdgcn_readfirstlane((unsigned)((((rows_left + 3) / 4 - 1) * g.N + BN) * 16));
            rcw[3] = 0x00020000u;
            const unsigned c_l = (unsigned)((((lane >> 4)) * g.N + (lane & 15)) * 16);
#pragma unroll
            for (int mi = 0; mi < 8; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = acc[mi][ni][j] + bv[ni];
                    // (inline asm with its own wait states: see the 32 x 32 form's note on gfx950's late read of store data)
                    const unsigned so = (unsigned)(((32 * wr + 4 * mi) * g.N + 64 * wc + 16 * ni) * 16);
                    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 2"
                                 :: "v"(v), "v"(c_l), "s"(rcw), "s"(so) : "memory");
                }
        } else {
            const __amdgpu_buffer_rsrc_t rc = make_rsrc_sized(cbase + (size_t)m0 * g.N + n0, (unsigned)(((rows_left - 1) * g.N + BN) * 4));
            const unsigned c_l = (unsigned)(((4 * (lane >> 4)) * g.N + (lane & 15)) * 4);
#pragma unroll
            for (int mi = 0; mi < 8; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        buf_store1_nt(acc[mi][ni][r] + bv[ni], rc, c_l, (unsigned)(((128 * wr + 16 * mi + r) * g.N + 64 * wc + 16 * ni) * 4));
        }
        GSTAMP(7)
        cur = nxt;
        r_cur = r_n;
        it = it_n;
        tile_no++;
    }
#ifdef PV_GEMM_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_gemm_stamps[i], gs_acc[i]);
#endif
}

// ---- host-side weight packing -----------------------------------------------------------------------
// LSTM layer, 8 waves per workgroup, wave w owns hidden units [32w, 32w+32) of the gates i,f,g,o (nt).
// Packed stream of one wave: [k-block of 8][gate][lane][4]; K = [x (padded to KP) | h]. Tile forms as in mfma_tiles.hpp:
//  32 rows: lane -> gate column nt*H + 32w + (lane&31); values k = 8kb + 4*(lane>>5) + j, j = 0..3
//  16 rows: lane -> columns nt*H + 32w + 16t + (lane&15), t = 0,1; values {t0 j0, t0 j1, t1 j0, t1 j1}, k = 8kb + 2*(lane>>4) + j
static void pack_lstm(const pv_rnn_dir* dirs, int K, int KP, int TR, std::vector<float>& wp, std::vector<float>& bias) {
    const int nkb = (KP + H) / 8, NT = 4, NW = 8;
    wp.assign((size_t)2 * NW * nkb * NT * 256, 0.0f);
    bias.assign((size_t)2 * 4 * H, 0.0f);
    for (int d = 0; d < 2; d++) {
        for (int n = 0; n < 4 * H; n++) bias[(size_t)d * 4 * H + n] = dirs[d].b_ih[n] + dirs[d].b_hh[n];
        auto wval = [&](int n, int k) -> float {
            if (k < KP) return k < K ? dirs[d].w_ih[(size_t)n * K + k] : 0.0f;
            return dirs[d].w_hh[(size_t)n * H + (k - KP)];
        };
        for (int w = 0; w < NW; w++)
            for (int kb = 0; kb < nkb; kb++)
                for (int nt = 0; nt < NT; nt++)
                    for (int lane = 0; lane < 64; lane++) {
                        float* dst = &wp[((((size_t)(d * NW + w) * nkb + kb) * NT + nt) * 64 + lane) * 4];
                        if (TR == 32) {
                            for (int j = 0; j < 4; j++) dst[j] = wval(nt * H + 32 * w + (lane & 31), kb * 8 + 4 * (lane >> 5) + j);
                        } else {
                            for (int t = 0; t < 2; t++)
                                for (int j = 0; j < 2; j++)
                                    dst[2 * t + j] = wval(nt * H + 32 * w + 16 * t + (lane & 15), kb * 8 + 2 * (lane >> 4) + j);
                        }
                    }
    }
}

// unit-split form (k_lstm_split): [dir][part][wave][k-block][2][lane][4], see the kernel
static void pack_lstm_split(const pv_rnn_dir* dirs, int K, int KP, int SP_NS, std::vector<float>& wp) {
    const int nkb = (KP + H) / 8, NW = 16 / SP_NS, UPP = H / SP_NS;
    wp.assign((size_t)2 * 16 * nkb * 2 * 256, 0.0f);
    for (int d = 0; d < 2; d++) {
        auto wval = [&](int n, int k) -> float {
            if (k < KP) return k < K ? dirs[d].w_ih[(size_t)n * K + k] : 0.0f;
            return dirs[d].w_hh[(size_t)n * H + (k - KP)];
        };
        for (int p = 0; p < SP_NS; p++)
            for (int w = 0; w < NW; w++)
                for (int kb = 0; kb < nkb; kb++)
                    for (int half = 0; half < 2; half++)
                        for (int lane = 0; lane < 64; lane++) {
                            float* dst = &wp[(((((size_t)(d * SP_NS + p) * NW + w) * nkb + kb) * 2 + half) * 64 + lane) * 4];
                            const int u = UPP * p + 16 * w + (lane & 15);
                            for (int gg = 0; gg < 2; gg++)
                                for (int j = 0; j < 2; j++)
                                    dst[2 * gg + j] = wval((2 * half + gg) * H + u, kb * 8 + 2 * (lane >> 4) + j);
                        }
    }
}

// Linear [512, K]: wave w owns columns [128w, 128w+128) as four 32-column "gate tiles" nt; tile forms as in pack_lstm
static void pack_linear(const float* W, int K, int TR, std::vector<float>& wp) {
    const int nkb = K / 8;
    wp.assign((size_t)4 * nkb * 4 * 256, 0.0f);
    for (int w = 0; w < 4; w++)
        for (int kb = 0; kb < nkb; kb++)
            for (int nt = 0; nt < 4; nt++)
                for (int lane = 0; lane < 64; lane++) {
                    float* dst = &wp[((((size_t)w * nkb + kb) * 4 + nt) * 64 + lane) * 4];
                    if (TR == 32) {
                        const int n = 128 * w + 32 * nt + (lane & 31);
                        for (int j = 0; j < 4; j++) dst[j] = W[(size_t)n * K + kb * 8 + 4 * (lane >> 5) + j];
                    } else {
                        for (int t = 0; t < 2; t++)
                            for (int j = 0; j < 2; j++)
                                dst[2 * t + j] = W[(size_t)(128 * w + 32 * nt + 16 * t + (lane & 15)) * K + kb * 8 + 2 * (lane >> 4) + j];
                    }
                }
}

}  // namespace

static constexpr int64_t P1_BF16_MAX_BATCH = 16384;
template <int KP> constexpr size_t lds_lstm_split() { return (size_t)(16 * (KP + 4) + 2 * 16 * (H + 4)) * sizeof(float); }
static constexpr int SP_MAX_TILES = 64;   // 16-row tiles the exchange buffers are sized for (1024 windows)
template <int KP, int TR> constexpr size_t lds_lstm() { return (size_t)(TR * (KP + 4) + 2 * TR * (H + 4)) * sizeof(float); }
static constexpr size_t LDS_GEMM = (size_t)2 * 4 * 256 * 32 * 2 + 2048 + 2048;   // 2 buffers x {A_hi, A_lo, W_hi, W_lo} x 256 rows x 32 bf16 = 128 KB, + two bias slots + completion flags
static constexpr size_t LDS_SPLITK = (size_t)ROWS * (2 * H + 4) * sizeof(float);
template <int TR> constexpr size_t lds_tail() { return (size_t)2 * TR * (HEAD_N + 4) * sizeof(float); }

struct pv_rnn_p1 {
    float* enc_wp[2] = {nullptr, nullptr};  // [0] 32-row tile form, [1] 16-row tile form (mfma_tiles.hpp)
    float* dec_wp[2] = {nullptr, nullptr};
    float* enc_bias = nullptr; float* dec_bias = nullptr;
    float* enc_wps[2] = {nullptr, nullptr};                // unit-split form (k_lstm_split): [0] four parts, [1] two parts
    float* dec_wps[2] = {nullptr, nullptr};
    u32x4* sp_hx = nullptr; int* sp_flags = nullptr;       // its exchange buffer (tagged granules) and counters (fence form)
    int* sp_err = nullptr;                                 // exchange time-outs (device word, read by the host-buffer entry points)
    unsigned* sp_epoch = nullptr;                          // device word: forward calls so far (tags of the exchange derive from it)
    float* w1p = nullptr; float* b1 = nullptr;
    float* wlp[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};  // [tile form][layer]
    float* bl[4] = {nullptr, nullptr, nullptr, nullptr};
    float* wo = nullptr; float* bo = nullptr;
    int dtype = PV_DTYPE_F32;
    // PV_DTYPE_BF16_INPUT_GEMM: bf16 hi/lo splits of the decoder W_ih (both directions, [2048,512]) and linear_1 ([512,16896])
    unsigned char* dec_wih_s = nullptr; float* dec_bias_cat = nullptr;   // split8 rows
    unsigned char* w1_s = nullptr;
    unsigned char* enc_rb = nullptr; unsigned char* dec_rb = nullptr;    // bf16 fragment streams of k_rec_bf16 (encoder: W_ih | W_hh; decoder: W_hh)
    unsigned char* tail_wb = nullptr;                                    // bf16 fragment stream of k_tail_bf16 (linear_2..5)
    std::vector<void*> owned;
};

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
// w [N, K] row-major -> split8 rows (per 8 elements: 8 x bf16 hi, 8 x bf16 lo with w ~= hi + lo), K % 8 == 0
static int dev_upload_split(const float* w, size_t N, size_t K, unsigned char** d_split, std::vector<void*>& owned) {
    const size_t n = N * K;
    std::vector<uint16_t> sw(2 * n);
    for (size_t r = 0; r < N; r++)
        for (size_t k = 0; k < K; k++) {
            const float v = w[r * K + k];
            const uint16_t hi = f2bf_bits(v);
            const uint16_t lo = f2bf_bits(v - bf_bits2f(hi));
            const size_t o = r * K * 2 + (k >> 3) * 16 + (k & 7);   // in 2-byte units
            sw[o] = hi;
            sw[o + 8] = lo;
        }
    PV_HIP(hipMalloc((void**)d_split, n * 4));
    owned.push_back(*d_split);
    PV_HIP(hipMemcpy(*d_split, sw.data(), n * 4, hipMemcpyHostToDevice));
    return PV_OK;
}

static int dev_upload(const std::vector<float>& h, float** d, std::vector<void*>& owned) {
    PV_HIP(hipMalloc((void**)d, h.size() * sizeof(float)));
    owned.push_back(*d);
    PV_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return PV_OK;
}
static int dev_upload(const float* h, size_t n, float** d, std::vector<void*>& owned) {
    PV_HIP(hipMalloc((void**)d, n * sizeof(float)));
    owned.push_back(*d);
    PV_HIP(hipMemcpy(*d, h, n * sizeof(float), hipMemcpyHostToDevice));
    return PV_OK;
}

void pv_rnn_free(pv_ctx* ctx) {
    if (ctx->p1) {
        for (void* p : ctx->p1->owned) (void)hipFree(p);
        delete ctx->p1;
        ctx->p1 = nullptr;
    }
    pv_rnn_free_p2(ctx);
}

extern "C" int pv_rnn_load_p1(pv_ctx* ctx, const pv_weights_p1* w, int dtype) {
    PV_CHECK(ctx && w, PV_ERR_INVALID, "null argument");
    PV_CHECK(dtype == PV_DTYPE_F32 || dtype == PV_DTYPE_BF16_INPUT_GEMM, PV_ERR_INVALID, "unknown dtype %d", dtype);
    for (int d = 0; d < 2; d++) {
        PV_CHECK(w->encoder[d].w_ih && w->encoder[d].w_hh && w->encoder[d].b_ih && w->encoder[d].b_hh &&
                     w->decoder[d].w_ih && w->decoder[d].w_hh && w->decoder[d].b_ih && w->decoder[d].b_hh,
                 PV_ERR_INVALID, "missing LSTM tensor");
    }
    for (int i = 0; i < 5; i++) PV_CHECK(w->linear_w[i] && w->linear_b[i], PV_ERR_INVALID, "missing linear_%d", i + 1);
    PV_CHECK(w->out_w && w->out_b, PV_ERR_INVALID, "missing output layer");
    PV_HIP(hipSetDevice(ctx->device));
    if (ctx->p1) {
        PV_HIP(hipStreamSynchronize(ctx->stream));
        for (void* p : ctx->p1->owned) (void)hipFree(p);
        delete ctx->p1;
        ctx->p1 = nullptr;
    }
    pv_rnn_p1* m = new pv_rnn_p1();
    ctx->p1 = m;
    m->dtype = dtype;
    std::vector<float> wp, bias;
    int rc;
    for (int f = 0; f < 2; f++) {
        pack_lstm(w->encoder, F_IN, 32, f ? 16 : 32, wp, bias);
        if ((rc = dev_upload(wp, &m->enc_wp[f], m->owned))) return rc;
        if (!f && (rc = dev_upload(bias, &m->enc_bias, m->owned))) return rc;
        pack_lstm(w->decoder, 2 * H, 2 * H, f ? 16 : 32, wp, bias);
        if ((rc = dev_upload(wp, &m->dec_wp[f], m->owned))) return rc;
        if (!f && (rc = dev_upload(bias, &m->dec_bias, m->owned))) return rc;
    }
    for (int f = 0; f < 2; f++) {
        pack_lstm_split(w->encoder, F_IN, 32, f ? 2 : 4, wp);
        if ((rc = dev_upload(wp, &m->enc_wps[f], m->owned))) return rc;
        pack_lstm_split(w->decoder, 2 * H, 2 * H, f ? 2 : 4, wp);
        if ((rc = dev_upload(wp, &m->dec_wps[f], m->owned))) return rc;
    }
    {
        const size_t hx_bytes = (size_t)SP_MAX_TILES * 2 * SP_HX_QUADS * sizeof(u32x4);
        const size_t fl_bytes = (size_t)SP_MAX_TILES * 2 * 4 * SP_FLAG_STRIDE * sizeof(int);
        PV_HIP(hipMalloc((void**)&m->sp_hx, hx_bytes));
        m->owned.push_back(m->sp_hx);
        PV_HIP(hipMalloc((void**)&m->sp_flags, fl_bytes));
        m->owned.push_back(m->sp_flags);
        PV_HIP(hipMemset(m->sp_flags, 0, fl_bytes));
        PV_HIP(hipMemset(m->sp_hx, 0, hx_bytes));   // tags start at 0: no launch ever waits for tag 0
        PV_HIP(hipMalloc((void**)&m->sp_err, 64));
        m->owned.push_back(m->sp_err);
        PV_HIP(hipMemset(m->sp_err, 0, 64));
        PV_HIP(hipMalloc((void**)&m->sp_epoch, 64));
        m->owned.push_back(m->sp_epoch);
        PV_HIP(hipMemset(m->sp_epoch, 0, 64));
    }
    pack_linear(w->linear_w[0], HEAD_K, 32, wp);
    if ((rc = dev_upload(wp, &m->w1p, m->owned)) || (rc = dev_upload(w->linear_b[0], HEAD_N, &m->b1, m->owned))) return rc;
    for (int i = 0; i < 4; i++) {
        for (int f = 0; f < 2; f++) {
            pack_linear(w->linear_w[i + 1], HEAD_N, f ? 16 : 32, wp);
            if ((rc = dev_upload(wp, &m->wlp[f][i], m->owned))) return rc;
        }
        if ((rc = dev_upload(w->linear_b[i + 1], HEAD_N, &m->bl[i], m->owned))) return rc;
    }
    if ((rc = dev_upload(w->out_w, 3 * HEAD_N, &m->wo, m->owned)) || (rc = dev_upload(w->out_b, 3, &m->bo, m->owned))) return rc;
    if (dtype == PV_DTYPE_BF16_INPUT_GEMM) {
        std::vector<float> wcat((size_t)2048 * 512), bcat(2048);
        for (int d = 0; d < 2; d++) {
            memcpy(&wcat[(size_t)d * 1024 * 512], w->decoder[d].w_ih, (size_t)1024 * 512 * sizeof(float));
            for (int n = 0; n < 1024; n++) bcat[d * 1024 + n] = w->decoder[d].b_ih[n] + w->decoder[d].b_hh[n];
        }
        if ((rc = dev_upload_split(wcat.data(), 2048, 512, &m->dec_wih_s, m->owned))) return rc;
        if ((rc = dev_upload(bcat, &m->dec_bias_cat, m->owned))) return rc;
        if ((rc = dev_upload_split(w->linear_w[0], HEAD_N, HEAD_K, &m->w1_s, m->owned))) return rc;
        if ((rc = pv_pack_rec_bf16(w->encoder, 4, F_IN, &m->enc_rb, nullptr, m->owned))) return rc;
        if ((rc = pv_pack_rec_bf16(w->decoder, 4, 0, &m->dec_rb, nullptr, m->owned))) return rc;
        const float* tail_w[4] = {w->linear_w[1], w->linear_w[2], w->linear_w[3], w->linear_w[4]};
        if ((rc = pv_pack_tail_bf16(tail_w, &m->tail_wb, m->owned))) return rc;
        if ((rc = pv_gemm_bf16x3_prepare()) || (rc = pv_rec_bf16_prepare()) || (rc = pv_tail_bf16_prepare())) return rc;
    }
    // opt in to > 64 KB of dynamic LDS (exact sizes; static LDS counts against the 160 KB too)
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<32, true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm<32, 32>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<32, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm<32, 16>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<512, false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm<512, 32>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<512, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm<512, 16>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_split<32, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm_split<32>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_split<512, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm_split<512>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_split<32, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm_split<32>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_split<512, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_lstm_split<512>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_head_splitk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_SPLITK));
    PV_HIP(hipFuncSetAttribute((const void*)k_head_tail<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tail<32>()));
    PV_HIP(hipFuncSetAttribute((const void*)k_head_tail<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tail<16>()));
    return PV_OK;
}

// tail of the head: 16-row tiles unless 32-row tiles already fill the chip
static void launch_tail(pv_ctx* ctx, pv_rnn_p1* m, TailArgs& t, int n_tiles32, hipStream_t st) {
    int tr = n_tiles32 >= ctx->num_cu ? 32 : 16;
    if (ctx->opt.tail_rows) tr = ctx->opt.tail_rows;
    for (int i = 0; i < 4; i++) { t.wp[i] = m->wlp[tr == 16 ? 1 : 0][i]; t.b[i] = m->bl[i]; }
    pv_prof_scope ps(ctx, "k_head_tail", st);
    if (tr == 32) k_head_tail<32><<<(unsigned)n_tiles32, 256, lds_tail<32>(), st>>>(t);
    else k_head_tail<16><<<(unsigned)(2 * n_tiles32), 256, lds_tail<16>(), st>>>(t);
}

static int p1_forward_launch(pv_ctx* ctx, const int8_t* d_images, int64_t B, float* d_probs, float* enc_out,
                             float* dec_out, float* part, hipStream_t st, bool taps = false) {
    pv_rnn_p1* m = ctx->p1;
    const int n_tiles = (int)((B + ROWS - 1) / ROWS);   // 32-row tiles: the granularity of every buffer and of the head
    // LSTM tile form: 32-row tiles once (tile, direction) workgroups fill the chip, else 16-row tiles: twice the workgroups,
    // half the MFMA cycles per time step (the bf16x3 mode has its own layer kernel, below)
    int tr = (int64_t)n_tiles * 2 >= ctx->num_cu ? 32 : 16;
    if (ctx->opt.lstm_rows) tr = ctx->opt.lstm_rows;
    const int f = tr == 16 ? 1 : 0;
    const int n_lt = n_tiles * (ROWS / tr);             // whole 32-row tiles are covered in either form
    const unsigned lstm_grid = (unsigned)(((n_lt + 3) / 4) * 8);
    if (m->dtype == PV_DTYPE_BF16_INPUT_GEMM && B >= ctx->opt.p1_bf16_min_batch) {   // (a small call is faster on the fp32 kernels below)
        // every matrix product on the bf16 MFMA: encoder layer (x-part in the step) -> decoder input projection as ONE GEMM
        // -> decoder layer on the projections -> linear_1 as a split-K GEMM -> fp32 tail. 64-row tiles (one weight fetch of
        // the recurrent stream feeds twice the rows) once 32-row (tile, direction) workgroups would not fit the chip at once.
        const int mt = (int64_t)n_tiles * 2 > ctx->num_cu ? 2 : 1;
        const int64_t Bp = (B + 32 * mt - 1) / (32 * mt) * (32 * mt), M = Bp * T_STEPS;
        const size_t nel = (size_t)Bp * T_STEPS * 2 * H;
        unsigned char *enc_split = nullptr, *dec_split = nullptr;
        float* G = nullptr;
        int rcb;
        if ((rcb = pv_get(ctx, "p1.enc_split", nel * 4, &enc_split)) || (rcb = pv_get(ctx, "p1.dec_split", nel * 4, &dec_split)) ||
            (rcb = pv_get(ctx, "p1.G", (size_t)M * 2048, &G))) return rcb;
        pv_rec_desc re = {};
        re.cell = 4; re.enc = 1; re.wp = m->enc_rb; re.bias = m->enc_bias; re.x = d_images; re.x_row_bytes = PV_WINDOW_BYTES; re.x_t0 = 0;
        re.xf = F_IN; re.x_signed = 1; re.B = B; re.Bp = Bp; re.T = T_STEPS; re.out_tm = enc_split; re.out_f32 = taps ? enc_out : nullptr;
        re.mt = mt; re.prof_name = "k_rec_bf16_lstm_enc";
        if ((rcb = pv_rec_bf16_async(ctx, re, st))) return rcb;
        pv_gemm_desc ga = {};
        ga.A = enc_split; ga.W = m->dec_wih_s; ga.bias = m->dec_bias_cat; ga.C = G; ga.M = M; ga.N = 2048; ga.K = 2 * H; ga.splits = 1;
        ga.quads = 1; ga.prof_name = "k_gemm_bf16x3_dec";
        if ((rcb = pv_gemm_bf16x3_async(ctx, ga, st))) return rcb;
        pv_rec_desc rd = {};
        rd.cell = 4; rd.enc = 0; rd.G = G; rd.wp = m->dec_rb; rd.B = B; rd.Bp = Bp; rd.T = T_STEPS; rd.out_bm = dec_split;
        rd.out_f32 = taps ? dec_out : nullptr; rd.mt = mt; rd.prof_name = "k_rec_bf16_lstm_dec";
        if ((rcb = pv_rec_bf16_async(ctx, rd, st))) return rcb;
        // linear_1 as a split-K bf16x3 GEMM into slabs [splits][Bp][512]: the smallest split factor (a divisor of the 528
        // K steps of 32) whose work items fill the chip
        static const int divs[] = {1, 2, 3, 4, 6, 8, 11, 12, 16, 22, 24, 33};
        const int tiles = (int)((Bp + 255) / 256) * (HEAD_N / 256);
        int gs = 33;
        for (int dv : divs) if (tiles * dv >= ctx->num_cu) { gs = dv; break; }
        pv_gemm_desc gl = {};
        gl.A = dec_split; gl.W = m->w1_s; gl.bias = nullptr; gl.C = part; gl.M = Bp; gl.N = HEAD_N; gl.K = HEAD_K; gl.splits = gs; gl.quads = 0;
        gl.prof_name = "k_gemm_bf16x3_lin1";
        if ((rcb = pv_gemm_bf16x3_async(ctx, gl, st))) return rcb;
        // sum of the slabs + linear_2..5 + output layer + softmax. Large batches: the four 512 x 512 layers as 3-term split products
        // too (k_tail_bf16, 64 rows per workgroup: 0.11 ms per 8192 windows against 0.27 for k_head_tail); a small batch is a few
        // workgroups each pulling the 4 MB of weights through one CU (0.30 ms for 64-512 windows), where k_head_tail's 16-row
        // tiles spread the same fetch over four times the CUs (0.14 ms): it keeps those
        if ((B + 63) / 64 < ctx->num_cu / 4) {
            TailArgs tf;
            tf.part = part; tf.b1 = m->b1; tf.splits = gs; tf.part_rows = Bp;
            tf.wo = m->wo; tf.bo = m->bo; tf.probs = d_probs; tf.B = B; tf.epoch = m->sp_epoch; tf.err = m->sp_err;
            launch_tail(ctx, m, tf, n_tiles, st);
            PV_HIP(hipGetLastError());
            return PV_OK;
        }
        pv_tail_desc tb = {};
        tb.part = part; tb.b1 = m->b1; tb.splits = gs; tb.part_rows = Bp; tb.wp = m->tail_wb;
        for (int i = 0; i < 4; i++) tb.b[i] = m->bl[i];
        tb.wo = m->wo; tb.bo = m->bo; tb.probs = d_probs; tb.B = B; tb.epoch = m->sp_epoch; tb.err = m->sp_err;
        return pv_tail_bf16_async(ctx, tb, st);
    }
    LstmArgs e;
    e.x_i8 = d_images; e.x_f32 = nullptr; e.wp = m->enc_wp[f]; e.bias = m->enc_bias; e.out = enc_out; e.B = B; e.n_tiles = n_lt;
    // unit-split form: one small fp32 batch whose (16-row tile, direction, part of the hidden units) workgroups all fit on
    // the chip at once: four parts up to 512 windows on 256 CUs, two parts up to 1024. Options lstm_split = 0, an explicit
    // lstm_rows, or shared_device = 1 (other work on this GPU: residency is not given) keep the one-workgroup form
    const int n_t16 = n_tiles * 2;
    const int sp_ns = (int64_t)n_t16 * 2 * 4 <= ctx->num_cu ? 4 : 2;
    bool split = n_t16 <= SP_MAX_TILES && (int64_t)n_t16 * 2 * sp_ns <= ctx->num_cu;
    if (!ctx->opt.lstm_split || ctx->opt.lstm_rows || ctx->opt.shared_device) split = false;
    LstmSplitArgs se;
    const unsigned split_grid = (unsigned)(((n_t16 + 3) / 4) * 8 * sp_ns);
    if (split) {
        se.x_i8 = d_images; se.x_f32 = nullptr; se.wp = m->enc_wps[sp_ns == 4 ? 0 : 1]; se.bias = m->enc_bias; se.out = enc_out;
        se.hx = m->sp_hx; se.flags = m->sp_flags; se.B = B; se.n_tiles = n_t16; se.epoch = m->sp_epoch; se.layer = 0;
        se.err = m->sp_err; se.spin_limit = 1 << ctx->opt.exchange_spin_log2; se.drop_part = ctx->opt.debug_drop_part;
        pv_prof_scope ps(ctx, "k_lstm_split_enc", st);
        if (sp_ns == 4) k_lstm_split<32, true, 4><<<split_grid, 256, lds_lstm_split<32>(), st>>>(se);
        else k_lstm_split<32, true, 2><<<split_grid, 512, lds_lstm_split<32>(), st>>>(se);
    } else {
        pv_prof_scope ps(ctx, "k_lstm_layer_enc", st);
        if (tr == 32) k_lstm_layer<32, true, 32><<<lstm_grid, 512, lds_lstm<32, 32>(), st>>>(e);
        else k_lstm_layer<32, true, 16><<<lstm_grid, 512, lds_lstm<32, 16>(), st>>>(e);
    }
    LstmArgs d = e;
    d.x_i8 = nullptr; d.x_f32 = enc_out; d.wp = m->dec_wp[f]; d.bias = m->dec_bias; d.out = dec_out;
    if (split) {
        LstmSplitArgs sd = se;
        sd.x_i8 = nullptr; sd.x_f32 = enc_out; sd.wp = m->dec_wps[sp_ns == 4 ? 0 : 1]; sd.bias = m->dec_bias; sd.out = dec_out;
        sd.layer = 1;
        pv_prof_scope ps(ctx, "k_lstm_split_dec", st);
        if (sp_ns == 4) k_lstm_split<512, false, 4><<<split_grid, 256, lds_lstm_split<512>(), st>>>(sd);
        else k_lstm_split<512, false, 2><<<split_grid, 512, lds_lstm_split<512>(), st>>>(sd);
    } else {
        pv_prof_scope ps(ctx, "k_lstm_layer_dec", st);
        if (tr == 32) k_lstm_layer<512, false, 32><<<lstm_grid, 512, lds_lstm<512, 32>(), st>>>(d);
        else k_lstm_layer<512, false, 16><<<lstm_grid, 512, lds_lstm<512, 16>(), st>>>(d);
    }
    HeadArgs h;
    // split-K factor: 11 slabs of 3 time steps; 33 single-step slabs only for batches too small to fill the chip
    int splits = ((int64_t)n_tiles * 11 >= ctx->num_cu) ? 11 : 33;
    if (ctx->opt.head_splits) splits = ctx->opt.head_splits;
    h.dec = dec_out; h.w1p = m->w1p; h.part = part; h.B = B; h.n_tiles = n_tiles; h.splits = splits; h.steps_per_split = T_STEPS / splits;
    const int head_map = ctx->opt.head_map;
    const int total_wg = n_tiles * splits;
    h.per_xcd = head_map ? (total_wg + 7) / 8 : 0;
    const unsigned head_grid = head_map ? (unsigned)(h.per_xcd * 8) : (unsigned)total_wg;
    { pv_prof_scope ps(ctx, "k_head_splitk", st); k_head_splitk<<<head_grid, 256, LDS_SPLITK, st>>>(h); }
    TailArgs t;
    t.part = part; t.b1 = m->b1; t.splits = splits; t.part_rows = B;
    t.wo = m->wo; t.bo = m->bo; t.probs = d_probs; t.B = B; t.epoch = m->sp_epoch; t.err = m->sp_err;
    launch_tail(ctx, m, t, n_tiles, st);
    PV_HIP(hipGetLastError());
    return PV_OK;
}

static int p1_workspace(pv_ctx* ctx, int64_t B, float** enc, float** dec, float** part) {
    int rc;
    const size_t Bp = (size_t)((B + 2 * ROWS - 1) / (2 * ROWS)) * 2 * ROWS;  // layer kernels store whole tiles (32 rows; 64 in the bf16x3 mode)
    if ((rc = pv_get(ctx, "p1.enc_out", Bp * T_STEPS * 2 * H, enc))) return rc;
    if ((rc = pv_get(ctx, "p1.dec_out", Bp * T_STEPS * 2 * H, dec))) return rc;
    if ((rc = pv_get(ctx, "p1.part", (size_t)HEAD_MAX_SPLITS * Bp * HEAD_N, part))) return rc;
    return PV_OK;
}

extern "C" int pv_rnn_forward_p1_dev(pv_ctx* ctx, const int8_t* d_images, int64_t B, float* d_probs, void* stream) {
    PV_CHECK(ctx && d_images && d_probs, PV_ERR_INVALID, "null argument");
    PV_CHECK(ctx->p1, PV_ERR_STATE, "pv_rnn_load_p1 has not been called on this context");
    PV_CHECK(B >= 0 && B < (1ll << 24), PV_ERR_INVALID, "batch %lld out of range", (long long)B);
    if (B == 0) return PV_OK;
    PV_HIP(hipSetDevice(ctx->device));
    // the bf16x3 mode materialises the decoder's input projections (33 x 8 KB per window: 4.4 GB at 16384 windows): larger
    // batches run as chunks of P1_BF16_MAX_BATCH windows on the same stream
    const int64_t chunk = ctx->p1->dtype == PV_DTYPE_BF16_INPUT_GEMM ? P1_BF16_MAX_BATCH : B;
    float *enc, *dec, *part;
    int rc = p1_workspace(ctx, std::min(B, chunk), &enc, &dec, &part);
    if (rc) return rc;
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = std::min(chunk, B - b0);
        if ((rc = p1_forward_launch(ctx, d_images + b0 * PV_WINDOW_BYTES, nb, d_probs + b0 * 3, enc, dec, part, pv_pick_stream(ctx, stream)))) return rc;
    }
    return PV_OK;
}

extern "C" int pv_rnn_forward_p1_debug(pv_ctx* ctx, const int8_t* images, int64_t B, float* probs, float* enc_out,
                                       float* dec_out) {
    PV_CHECK(ctx && images && probs, PV_ERR_INVALID, "null argument");
    PV_CHECK(ctx->p1, PV_ERR_STATE, "pv_rnn_load_p1 has not been called on this context");
    PV_CHECK(B >= 0 && B < (1ll << 24), PV_ERR_INVALID, "batch %lld out of range", (long long)B);
    if (B == 0) return PV_OK;
    if (ctx->p1->dtype == PV_DTYPE_BF16_INPUT_GEMM && B > P1_BF16_MAX_BATCH) {
        // host-buffer form in the bf16x3 mode: chunks (the taps are per-window, so chunking does not change them)
        for (int64_t b0 = 0; b0 < B; b0 += P1_BF16_MAX_BATCH) {
            const int64_t nb = std::min<int64_t>(P1_BF16_MAX_BATCH, B - b0);
            const size_t tap = (size_t)b0 * T_STEPS * 2 * H;
            int rcc = pv_rnn_forward_p1_debug(ctx, images + b0 * PV_WINDOW_BYTES, nb, probs + b0 * 3, enc_out ? enc_out + tap : nullptr,
                                              dec_out ? dec_out + tap : nullptr);
            if (rcc) return rcc;
        }
        return PV_OK;
    }
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    float *enc, *dec, *part, *d_probs;
    int8_t* d_img;
    int rc = p1_workspace(ctx, B, &enc, &dec, &part);
    if (rc) return rc;
    if ((rc = pv_get(ctx, "p1.images", (size_t)B * PV_WINDOW_BYTES, &d_img))) return rc;
    if ((rc = pv_get(ctx, "p1.probs", (size_t)B * 3, &d_probs))) return rc;
    PV_HIP(hipMemcpyAsync(d_img, images, (size_t)B * PV_WINDOW_BYTES, hipMemcpyHostToDevice, st));
    if ((rc = p1_forward_launch(ctx, d_img, B, d_probs, enc, dec, part, st, enc_out || dec_out))) return rc;
    PV_HIP(hipMemcpyAsync(probs, d_probs, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    const size_t nb = (size_t)B * T_STEPS * 2 * H * sizeof(float);
    if (enc_out) PV_HIP(hipMemcpyAsync(enc_out, enc, nb, hipMemcpyDeviceToHost, st));
    if (dec_out) PV_HIP(hipMemcpyAsync(dec_out, dec, nb, hipMemcpyDeviceToHost, st));
    int n_timeouts = 0;
    PV_HIP(hipMemcpyAsync(&n_timeouts, ctx->p1->sp_err, sizeof(int), hipMemcpyDeviceToHost, st));
    PV_HIP(hipStreamSynchronize(st));
    if (n_timeouts) {   // only possible when the launch's workgroups could not all be resident (a GPU shared with other work)
        PV_HIP(hipMemset(ctx->p1->sp_err, 0, sizeof(int)));
        pv_set_error("unit-split LSTM form: %d exchange polls timed out (GPU shared with other work?); the probabilities of this call are NaN. "
                     "Set option shared_device = 1 (or lstm_split = 0) on this context", n_timeouts);
        return PV_ERR_STATE;
    }
    return PV_OK;
}

int pv_p2_take_timeouts(pv_ctx* ctx, int* n);   // rnn_gru.hip

extern "C" int pv_rnn_exchange_timeouts(pv_ctx* ctx) {
    PV_CHECK(ctx, PV_ERR_INVALID, "null argument");
    PV_HIP(hipSetDevice(ctx->device));
    PV_HIP(hipStreamSynchronize(ctx->stream));
    int n1 = 0, n2 = 0;
    if (ctx->p1 && ctx->p1->sp_err) {
        PV_HIP(hipMemcpy(&n1, ctx->p1->sp_err, sizeof(int), hipMemcpyDeviceToHost));
        if (n1) PV_HIP(hipMemset(ctx->p1->sp_err, 0, sizeof(int)));
    }
    int rc = pv_p2_take_timeouts(ctx, &n2);
    if (rc) return rc;
    return n1 + n2;
}

extern "C" int pv_rnn_forward_p1(pv_ctx* ctx, const int8_t* images, int64_t B, float* probs) {
    return pv_rnn_forward_p1_debug(ctx, images, B, probs, nullptr, nullptr);
}

// iota[1024] per device (allocated once, never freed: 4 KB), the source of k_gemm_bf16x3's completion-flag transfers
static const unsigned* gemm_iota() {
    static std::mutex mu;
    static std::map<int, unsigned*> per_dev;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_dev.find(dev);
    if (it != per_dev.end()) return it->second;
    unsigned h[1024];
    for (unsigned i = 0; i < 1024; i++) h[i] = i;
    unsigned* d = nullptr;
    if (hipMalloc((void**)&d, sizeof h) != hipSuccess || hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    per_dev[dev] = d;
    return d;
}

static void gemm_launch(pv_ctx* ctx, GemmArgs& g, hipStream_t st) {
    g.tiles_m = (int)((g.M + 255) / 256); g.tiles_n = g.N / 256; g.items = g.tiles_m * g.tiles_n * g.splits;
    const unsigned grid = (unsigned)(std::min((g.items + 7) / 8 * 8, (ctx->num_cu + 7) / 8 * 8));
    k_gemm_bf16x3<<<grid, 512, LDS_GEMM, st>>>(g);
}

int pv_gemm_bf16x3_prepare() {
    PV_HIP(hipFuncSetAttribute((const void*)k_gemm_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_GEMM));
    PV_CHECK(gemm_iota() != nullptr, PV_ERR_HIP, "allocation of the GEMM's flag table failed");
    return PV_OK;
}

int pv_upload_split8(const float* w, size_t N, size_t K, unsigned char** d_split, std::vector<void*>& owned) {
    return dev_upload_split(w, N, K, d_split, owned);
}

int pv_gemm_bf16x3_async(pv_ctx* ctx, const pv_gemm_desc& d, hipStream_t st) {
    PV_CHECK(d.A && d.W && d.C && d.M > 0 && d.M % 4 == 0 && d.N % 256 == 0 && d.splits >= 1 && d.K % (32 * d.splits) == 0 &&
                 (!d.quads || d.splits == 1), PV_ERR_INVALID, "bad GEMM shape");
    GemmArgs g;
    g.A = d.A; g.W = d.W; g.bias = d.bias; g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K; g.splits = d.splits;
    g.c_quads = d.quads;
    g.iota = gemm_iota();
    PV_CHECK(g.iota, PV_ERR_HIP, "GEMM flag table missing");
    pv_prof_scope ps(ctx, d.prof_name ? d.prof_name : "k_gemm_bf16x3", st);
    gemm_launch(ctx, g, st);
    PV_HIP(hipGetLastError());
    return PV_OK;
}

// Diagnostic entry (tests, tuning): C = A . W^T + bias through k_gemm_bf16x3 alone. HOST pointers, fp32 row-major A [M,K],
// W [N,K], bias [N] or NULL, C [splits][M][N] row-major (quads = 0) or [M/4][N][4] (quads = 1, splits must be 1).
// M % 4 == 0, N % 256 == 0, K % (32 * splits) == 0. Returns the kernel time in ms through *ms when non-NULL.
extern "C" int pv_debug_gemm_bf16x3(pv_ctx* ctx, const float* A, const float* W, const float* bias, int64_t M, int N, int K,
                                    int splits, int quads, float* C, float* ms) {
    PV_CHECK(ctx && A && W && C, PV_ERR_INVALID, "null argument");
    PV_CHECK(M > 0 && M % 4 == 0 && N % 256 == 0 && splits >= 1 && K % (32 * splits) == 0 && (!quads || splits == 1), PV_ERR_INVALID, "bad GEMM shape");
    PV_HIP(hipSetDevice(ctx->device));
    std::vector<void*> owned;
    unsigned char *dA = nullptr, *dW = nullptr;
    float *dB = nullptr, *dC = nullptr;
    int rc;
    auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
    if ((rc = dev_upload_split(A, (size_t)M, (size_t)K, &dA, owned)) || (rc = dev_upload_split(W, (size_t)N, (size_t)K, &dW, owned))) { cleanup(); return rc; }
    if (bias && (rc = dev_upload(bias, (size_t)N, &dB, owned))) { cleanup(); return rc; }
    const size_t nc = (size_t)splits * M * N;
    if (hipMalloc((void**)&dC, nc * sizeof(float)) != hipSuccess) { cleanup(); pv_set_error("hipMalloc failed"); return PV_ERR_HIP; }
    owned.push_back(dC);
    (void)hipMemset(dC, 0xff, nc * sizeof(float));
    if (pv_gemm_bf16x3_prepare() != PV_OK) { cleanup(); return PV_ERR_HIP; }
    GemmArgs g;
    g.A = dA; g.W = dW; g.bias = dB; g.C = dC; g.M = M; g.N = N; g.K = K; g.splits = splits; g.c_quads = quads;
    g.iota = gemm_iota();
    if (!g.iota) { cleanup(); pv_set_error("GEMM flag table missing"); return PV_ERR_HIP; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    gemm_launch(ctx, g, ctx->stream);   // warm
    (void)hipEventRecord(e0, ctx->stream);
    gemm_launch(ctx, g, ctx->stream);
    (void)hipEventRecord(e1, ctx->stream);
    hipError_t er = hipStreamSynchronize(ctx->stream);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ms) *ms = t;
    if (er == hipSuccess) er = hipMemcpy(C, dC, nc * sizeof(float), hipMemcpyDeviceToHost);
    cleanup();
    if (er != hipSuccess) { pv_set_error("GEMM failed: %s", hipGetErrorString(er)); return PV_ERR_HIP; }
    return PV_OK;
}

// ---- P2 (bi-GRU polisher model): see rnn_gru.hip ----------------------------------------------------


#ifdef PV_GEMM_STAMPS
extern "C" int pv_debug_gemm_stamps(unsigned long long* out5) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PV_HIP(hipDeviceSynchronize());
    PV_HIP(hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_gemm_stamps), 8 * sizeof(unsigned long long)));
    PV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z, sizeof z));
    return PV_OK;
}
#endif
