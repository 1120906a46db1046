// rnn_kernels.hip — recurrent-network inference for PEPPER on gfx950 (MI355X), fp32 throughout.
//
// P1 (pepper_variant): images int8 [B,33,26] -> bi-LSTM(256) -> bi-LSTM(256) -> flatten 16896 ->
//     5 x (Linear 512 + SELU) -> Linear 3 -> softmax        (reference: models/simple_model.py:48-82)
//
// Kernel family (one 256-thread workgroup = 4 waves, one wave per SIMD, owns a 32-row batch tile):
//   k_lstm_layer<KP,INT8>  one launch per LSTM layer; a workgroup = (batch tile, direction). Per time
//        step the gate pre-activations [32 x 1024] = [x_t | h_{t-1}] . [W_ih | W_hh]^T are ONE
//        concatenated-K product on the f32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain), so
//        the "input-projection GEMM" and the recurrent product share accumulators and no
//        pre-activation tensor ever goes to HBM. A operand (x_t, h_{t-1}) lives in LDS; the B operand
//        (weights) is pre-packed on the host in exact MFMA fragment order and streamed from L2 with
//        one coalesced 16-B load per lane per 4 MFMAs; wave w owns hidden units [64w,64w+64) for all
//        four gates so the cell update is purely in-register (C/D layout puts i,f,g,o of one
//        (row,unit) in the same lane and register index). c stays in registers for all 33 steps, h is
//        exchanged between the 4 waves through a double-buffered LDS tile. XCD-aware block mapping
//        keeps one direction's weights (<= 3 MB) per XCD L2.
//   k_head_splitk          linear_1 (K = 16896) as a split-K MFMA product over time-step chunks,
//        deterministic partial slabs (no float atomics).
//   k_head_tail            sum of slabs + bias + SELU, linear_2..5 + SELU (MFMA), output layer, softmax.
#include "pv_common.hpp"

#include <cmath>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int T_STEPS = 33;
constexpr int F_IN = 26;
constexpr int H = 256;            // LSTM hidden
constexpr int ROWS = 32;          // batch rows per workgroup (one MFMA M-tile)
constexpr int HEAD_N = 512;
constexpr int HEAD_K = T_STEPS * 2 * H;  // 16896
constexpr int HEAD_MAX_SPLITS = 33;      // split-K factor of linear_1 is chosen per launch from {1, 3, 11, 33}

// Streaming accesses (x_t tiles, layer outputs, the head's A operand) carry the non-temporal hint: they are touched
// once, and without the hint they push the packed weights (3.1 MB per direction in a 4 MB L2) out of the XCD's L2
// every time step (rocprofv3 FETCH_SIZE showed the weights re-fetched on each of the 33 steps).
#ifdef PV_NO_NT
#define PV_LD_STREAM(p) (*(p))
#define PV_ST_STREAM(v, p) (*(p) = (v))
#else
#define PV_LD_STREAM(p) __builtin_nontemporal_load(p)
#define PV_ST_STREAM(v, p) __builtin_nontemporal_store((v), (p))
#endif

// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the IEEE division sequence: ~3x fewer VALU instructions in the
// cell update; absolute error of sigmoid/tanh stays ~1e-7 (tests pin 2e-5 on layer outputs).
__device__ __forceinline__ float rcpf_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return rcpf_(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // 1 - 2/(e^{2x}+1): saturates cleanly at +-1 (e = inf -> 1, e = 0 -> -1)
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * rcpf_(e + 1.0f);
}
__device__ __forceinline__ float seluf_(float x) {
    return 1.0507009873554805f * (x > 0.0f ? x : 1.6732632423543772f * (__expf(x) - 1.0f));
}

// acc[nt] += A[32 x 8*nkb] . B  for NT column tiles of 32. A: LDS, row-major, `lda` floats per row
// (lda % 4 == 0, lda % 64 == 4 keeps ds_read_b128 conflict-free); lane reads 16 B at
// A[lane&31][8*kb + 4*(lane>>5)]. Bp: this wave's packed stream, NT*256 floats per k-block:
// Bp[(kb*NT + nt)*256 + lane*4 + j] = W[n(nt,lane&31)][8*kb + 4*(lane>>5) + j].
// MFMA j of a k-block multiplies k = 8kb + 4*(lane>>5) + j: lanes 0-31 carry k..k+3 of the low half,
// lanes 32-63 of the high half; the K order inside a block is a permutation of 0..7, which only
// changes the (fp32) summation order.
template <int NT>
__device__ __forceinline__ void mma_panel(f32x16 (&acc)[NT], const float* __restrict__ A, int lda,
                                          const float* __restrict__ Bp, int nkb, int lane) {
    const float* ap = A + (lane & 31) * lda + 4 * (lane >> 5);
    const f32x4* bp = reinterpret_cast<const f32x4*>(Bp) + lane;
    f32x4 b0[NT], b1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) b0[nt] = bp[nt * 64];
    int kb = 0;
#pragma nounroll
    for (; kb + 1 < nkb; kb += 2) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b1[nt] = bp[((kb + 1) * NT + nt) * 64];
        f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * kb);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[nt][j], acc[nt], 0, 0, 0);
        if (kb + 2 < nkb) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b0[nt] = bp[((kb + 2) * NT + nt) * 64];
        }
        a = *reinterpret_cast<const f32x4*>(ap + 8 * (kb + 1));
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b1[nt][j], acc[nt], 0, 0, 0);
    }
    if (kb < nkb) {
        f32x4 a = *reinterpret_cast<const f32x4*>(ap + 8 * kb);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b0[nt][j], acc[nt], 0, 0, 0);
    }
}

struct LstmArgs {
    const int8_t* x_i8;   // [B,33,26]    (encoder)
    const float* x_f32;   // [B,33,512]   (decoder)
    const float* wp;      // packed [2 dirs][NW waves][nkb][32/NW tiles][64][4]
    const float* bias;    // [2][1024] b_ih + b_hh
    float* out;           // [B,33,512]
    float* out_cm;        // optional chunk-major copy [512/32][Bp*33][32] (A operand of the bf16x3 decoder GEMM)
    int64_t cm_rows;      // Bp*33
    float* out_packed;    // optional [n_tiles][33][64 kb][64 lanes][4]: A-fragment order of the next layer's x operand
    int64_t B;
    int n_tiles;
    unsigned long long* stamps;  // diagnostic builds (-DPV_STAMPS) only
    int ablate;  // diagnostics only (PV_ABLATE): 1 = trivial cell update, 2 = stage x_t only at step 0, 4 = no global h store
};

// One software-pipelined K loop over TWO LDS operands ([x_t | h_{t-1}]): k-blocks [0,nkb1) read A1,
// [nkb1,nkb1+nkb2) read A2; the packed weight stream is contiguous over both. B fragments of block 0
// are resident in registers (`bres`, identical every time step), blocks kb+1 (B from L2, A from LDS)
// are fetched while block kb multiplies.
template <int NT>
__device__ __forceinline__ void mma_dual(f32x16 (&acc)[NT], const float* __restrict__ A1, int lda1, int nkb1,
                                         const float* __restrict__ A2, int lda2, int nkb2,
                                         const float* __restrict__ Bp, const f32x4 (&bres)[NT], int lane) {
    const float* ap1 = A1 + (lane & 31) * lda1 + 4 * (lane >> 5);
    const float* ap2 = A2 + (lane & 31) * lda2 + 4 * (lane >> 5) - 8 * nkb1;
    const f32x4* bp = reinterpret_cast<const f32x4*>(Bp) + lane;
    const int nkb = nkb1 + nkb2;
    f32x4 b0[NT], b1[NT], a0, a1;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) b0[nt] = bres[nt];
    a0 = *reinterpret_cast<const f32x4*>((0 < nkb1 ? ap1 : ap2));
    int kb = 0;
#pragma nounroll
    for (; kb + 1 < nkb; kb += 2) {
#pragma unroll
#ifndef ABL_NOBLOAD
        for (int nt = 0; nt < NT; nt++) b1[nt] = bp[((kb + 1) * NT + nt) * 64];
#else
        for (int nt = 0; nt < NT; nt++) { b1[nt] = b0[nt]; asm volatile("" : "+v"(b1[nt])); }
#endif
        a1 = *reinterpret_cast<const f32x4*>((kb + 1 < nkb1 ? ap1 : ap2) + 8 * (kb + 1));
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[nt][j], acc[nt], 0, 0, 0);
        if (kb + 2 < nkb) {
#pragma unroll
#ifndef ABL_NOBLOAD
            for (int nt = 0; nt < NT; nt++) b0[nt] = bp[((kb + 2) * NT + nt) * 64];
#else
            for (int nt = 0; nt < NT; nt++) { b0[nt] = b1[nt]; asm volatile("" : "+v"(b0[nt])); }
#endif
            a0 = *reinterpret_cast<const f32x4*>((kb + 2 < nkb1 ? ap1 : ap2) + 8 * (kb + 2));
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[nt][j], acc[nt], 0, 0, 0);
    }
    if (kb < nkb) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[nt][j], acc[nt], 0, 0, 0);
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// Same contract as mma_dual, but the B fragments are raw buffer loads (SGPR resource + lane offset + scalar offset: no
// address VGPRs) into a ring of FOUR register sets requested THREE k-blocks ahead, fenced so that hipcc keeps the
// distance; the ring wraps into the next time step (same weights every step), so `bq` slots 0..2 must hold k-blocks 0..2
// on entry and do so again on exit. Requires (nkb1 + nkb2) % 4 == 0.
template <int NT>
__device__ __forceinline__ void mma_dual_ringb(f32x16 (&acc)[NT], const float* __restrict__ A1, int lda1, int nkb1,
                                               const float* __restrict__ A2, int lda2, int nkb2,
                                               __amdgpu_buffer_rsrc_t wr, f32x4 (&bq)[4][NT], int lane) {
    const float* ap1 = A1 + (lane & 31) * lda1 + 4 * (lane >> 5);
    const float* ap2 = A2 + (lane & 31) * lda2 + 4 * (lane >> 5) - 8 * nkb1;
    const int nkb = nkb1 + nkb2;
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x4 aq[2];
#define RB_B(slot, kbv) { _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bq[slot][nt] = buf_load4(wr, lane16, (unsigned)(((kbv) * NT + nt) * 1024)); }
#define RB_A(slot, kbv) { aq[slot] = *reinterpret_cast<const f32x4*>(((kbv) < nkb1 ? ap1 : ap2) + 8 * (kbv)); }
#define RB_M(bs, as)                                                                    \
    {                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                   \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                           \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[as][j], bq[bs][nt][j], acc[nt], 0, 0, 0); \
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
// with A[32 x 8*nkb] in LDS. On entry bq slots 0..2 hold k-blocks kpos..kpos+2; on exit they hold kpos+nkb..kpos+nkb+2,
// i.e. the next call's first blocks, so consecutive calls (time steps of linear_1, layers of the head) never start with a
// cold L2/HBM round trip. Requests beyond the resource's size return zeros (raw buffer bounds check). nkb % 4 == 0.
template <int NT>
__device__ __forceinline__ void mma_stream_ringb(f32x16 (&acc)[NT], const float* __restrict__ A, int lda, int nkb,
                                                 __amdgpu_buffer_rsrc_t wr, int kpos, f32x4 (&bq)[4][NT], int lane) {
    const float* ap = A + (lane & 31) * lda + 4 * (lane >> 5);
    const unsigned lane16 = (unsigned)lane * 16u;
    f32x4 aq[2];
#define RB_B(slot, kbv) { _Pragma("unroll") for (int nt = 0; nt < NT; nt++) bq[slot][nt] = buf_load4(wr, lane16, (unsigned)(((kpos + (kbv)) * NT + nt) * 1024)); }
#define RB_A(slot, kbv) { aq[slot] = *reinterpret_cast<const f32x4*>(ap + 8 * (kbv)); }
#define RB_M(bs, as)                                                                    \
    {                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                   \
            _Pragma("unroll") for (int nt = 0; nt < NT; nt++)                           \
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[as][j], bq[bs][nt][j], acc[nt], 0, 0, 0); \
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
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_sized(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// KP = padded input width (multiple of 8): 32 for the encoder (26 real), 512 for the decoder.
// NW = waves per workgroup: 8 (two per SIMD: one wave's LDS/L2 waits and cell update hide behind the
// other's MFMAs) or 4 (one per SIMD).
template <int KP, bool INT8, int NW, bool PACKED>
__global__ __launch_bounds__(NW * 64, NW / 4) void k_lstm_layer(LstmArgs a) {
    constexpr int LDX = KP + 4, LDH = H + 4;
    constexpr int NKB_X = KP / 8, NKB_H = H / 8;
    constexpr int NT = 32 / NW;   // 32-column gate tiles per wave
    constexpr int S2 = NT / 4;    // 32-unit sub-tiles per wave
    constexpr int UW = H / NW;    // hidden units per wave
    constexpr int NTHR = NW * 64;
    extern __shared__ float smem[];
    float* xbuf = smem;                 // [32][LDX]
    float* hbuf = smem + ROWS * LDX;    // [2][32][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR: bases/offsets derived from it stay scalar
    // XCD-aware mapping: blocks b and b+8 share an XCD (round-robin dispatch); give every XCD one
    // direction only so that its L2 holds a single direction's packed weights.
    const int xcd = blockIdx.x & 7;
    const int dir = xcd & 1;
    const int tile = (blockIdx.x >> 3) * 4 + (xcd >> 1);
    if (tile >= a.n_tiles) return;
    const int64_t b0 = (int64_t)tile * ROWS;
    const float* wp = a.wp + ((size_t)(dir * NW + wv) * (NKB_X + NKB_H)) * NT * 256;
    const float* bias = a.bias + dir * 4 * H;

    for (int i = tid; i < 2 * ROWS * LDH; i += NTHR) hbuf[i] = 0.0f;
    f32x16 cst[S2];
#pragma unroll
    for (int s2 = 0; s2 < S2; s2++)
#pragma unroll
        for (int r = 0; r < 16; r++) cst[s2][r] = 0.0f;
    float bs[NT];
    f32x4 bres[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        bs[nt] = bias[(nt / S2) * H + UW * wv + 32 * (nt % S2) + (lane & 31)];
        bres[nt] = reinterpret_cast<const f32x4*>(wp)[nt * 64 + lane];
    }

    // x_t staging through registers: the global loads for step s+1 are issued before step s's MFMAs
    // and written to LDS after them, so their latency hides behind the matrix work.
    constexpr int V4 = KP / 4;                                  // float4 per row (fp32 input)
    constexpr int XR = INT8 ? 1 : (ROWS * V4 + NTHR - 1) / NTHR;  // float4 registers per thread
    constexpr int XI = (ROWS * KP + NTHR - 1) / NTHR;           // int8 elements per thread (encoder)
    f32x4 xr[XR];
    float xi[INT8 ? XI : 1];
    const __amdgpu_buffer_rsrc_t xsr = make_rsrc(INT8 ? (const void*)a.wp : (const void*)(a.x_f32 + (size_t)b0 * T_STEPS * KP));
    const __amdgpu_buffer_rsrc_t osr = make_rsrc(a.out + (size_t)b0 * T_STEPS * 2 * H);
    const unsigned xg_l = (unsigned)(((tid / V4) * T_STEPS * KP + (tid % V4) * 4) * 4);  // byte offset of this thread's float4 in row group 0, step 0
    const unsigned og_l = (unsigned)((4 * (lane >> 5) * T_STEPS * 2 * H + (lane & 31)) * 4);
    auto x_load = [&](int t) {
        if constexpr (INT8) {
#pragma unroll
            for (int u = 0; u < XI; u++) {
                const int i = tid + u * NTHR;
                const int row = i / KP, k = i - row * KP;
                int64_t b = b0 + row;
                if (b >= a.B) b = a.B - 1;
                xi[u] = (i < ROWS * KP && k < F_IN) ? (float)a.x_i8[(b * T_STEPS + t) * F_IN + k] : 0.0f;
            }
        } else {
            // the layer input is padded to whole 32-row tiles (rows beyond B replicate row B-1): no clamp, and the
            // address is (tile resource) + (lane offset, computed once) + (scalar offset of row group u and step t)
            static_assert((ROWS * V4) % NTHR == 0, "x staging assumes whole row groups per pass");
#pragma unroll
            for (int u = 0; u < XR; u++)
                xr[u] = buf_load4(xsr, xg_l, (unsigned)(((u * (NTHR / V4)) * T_STEPS + t) * KP * 4));
        }
    };
    auto x_store = [&]() {
        if constexpr (INT8) {
#pragma unroll
            for (int u = 0; u < XI; u++) {
                const int i = tid + u * NTHR;
                const int row = i / KP, k = i - row * KP;
                if (i < ROWS * KP) xbuf[row * LDX + k] = xi[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < XR; u++) {
                const int i = tid + u * NTHR;
                const int row = i / V4, c4 = i - row * V4;
                if (i < ROWS * V4) *reinterpret_cast<f32x4*>(xbuf + row * LDX + c4 * 4) = xr[u];
            }
        }
    };
    // NW == 8: weight fragments run in a 4-deep register ring fed by raw buffer loads three k-blocks ahead (mma_dual_ringb)
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wp);
    f32x4 bq[NW == 8 ? 4 : 1][NT];
    if constexpr (NW == 8) ring_prime<NT>(bq, wr, 0, lane);
    x_load(dir ? T_STEPS - 1 : 0);
    x_store();
    __syncthreads();

#ifdef PV_STAMPS
    unsigned long long st_m = 0, st_c = 0, st_b = 0, st_s = 0, t0s, t1s;
#define STAMPL(v) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    for (int s = 0; s < T_STEPS; s++) {
        const int t = dir ? (T_STEPS - 1 - s) : s;
        const int cur = s & 1, nxt = cur ^ 1;
#ifdef PV_STAMPS
        STAMPL(t0s)
#endif
        if (s + 1 < T_STEPS && !(a.ablate & 2)) x_load(dir ? (T_STEPS - 2 - s) : (s + 1));
        // ---- gates = bias + [x_t | h_{t-1}] . [W_ih | W_hh]^T -------------------------------------
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = bs[nt];
        if constexpr (NW == 8) {
            mma_dual_ringb<NT>(acc, xbuf, LDX, NKB_X, hbuf + cur * ROWS * LDH, LDH, NKB_H, wr, bq, lane);
        } else {
            mma_dual<NT>(acc, xbuf, LDX, NKB_X, hbuf + cur * ROWS * LDH, LDH, NKB_H, wp, bres, lane);
        }
#ifdef PV_STAMPS
        STAMPL(t1s) st_m += t1s - t0s; t0s = t1s;
#endif
        // ---- cell update (PyTorch gate order i,f,g,o; nt = gate*S2 + sub-tile) ---------------------
        float* hn = hbuf + nxt * ROWS * LDH;
#pragma unroll
        for (int s2 = 0; s2 < S2; s2++) {
            const int unit = UW * wv + 32 * s2 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (a.ablate & 1) {
                    const float h = acc[0 * S2 + s2][r] * 1e-3f + acc[1 * S2 + s2][r] * 1e-3f + acc[2 * S2 + s2][r] * 1e-3f + acc[3 * S2 + s2][r] * 1e-3f;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    hn[row * LDH + unit] = h;
                    if (!(a.ablate & 4)) a.out[((size_t)(b0 + row) * T_STEPS + t) * (2 * H) + dir * H + unit] = h;
                    continue;
                }
                const float ig = sigmoidf_(acc[0 * S2 + s2][r]);
                const float fg = sigmoidf_(acc[1 * S2 + s2][r]);
                const float gg = tanhf_(acc[2 * S2 + s2][r]);
                const float og = sigmoidf_(acc[3 * S2 + s2][r]);
                const float c = fg * cst[s2][r] + ig * gg;
                cst[s2][r] = c;
                const float h = og * tanhf_(c);
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                hn[row * LDH + unit] = h;
                // outputs are padded to whole 32-row tiles: unconditional stores, uniform base + 32-bit lane offset
                if constexpr (PACKED) {  // fragment order: [kb = col/8][lane = (col%8/4)*32 + row][col%4]
                    const unsigned col = dir * H + unit;
                    float* pk = a.out_packed + ((size_t)tile * T_STEPS + t) * 64 * 256;
                    pk[(col >> 3) * 256 + ((((col >> 2) & 1) * 32 + row) << 2) + (col & 3)] = h;
                } else {
                    const int rr = (r & 3) + 8 * (r >> 2);  // compile-time part of the row
                    buf_store1(h, osr, og_l, (unsigned)((t * 2 * H + dir * H + UW * wv + 32 * s2 + rr * T_STEPS * 2 * H) * 4));
                    if (a.out_cm) {
                        const unsigned col = dir * H + unit;
                        a.out_cm[((size_t)(col >> 5) * a.cm_rows + (size_t)(b0 + row) * T_STEPS + t) * 32 + (col & 31)] = h;
                    }
                }
            }
        }
#ifdef PV_STAMPS
        STAMPL(t1s) st_c += t1s - t0s; t0s = t1s;
#endif
        __syncthreads();  // everyone is done reading xbuf / hbuf[cur]; hbuf[nxt] is complete
#ifdef PV_STAMPS
        STAMPL(t1s) st_b += t1s - t0s; t0s = t1s;
#endif
        if (s + 1 < T_STEPS && !(a.ablate & 2)) {
            x_store();
            __syncthreads();
        }
#ifdef PV_STAMPS
        STAMPL(t1s) st_s += t1s - t0s;
#endif
    }
#ifdef PV_STAMPS
    if (lane == 0 && a.stamps) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wv) * 4;
        o[0] = st_m; o[1] = st_c; o[2] = st_b; o[3] = st_s;
    }
#endif
}


// ---- decoder LSTM, staggered ------------------------------------------------------------------------------
// Same math and weight packing as k_lstm_layer<512,false,8>, different choreography:
//  * the x operand (encoder output) is read straight from HBM/L2 in A-fragment order (the encoder writes a
//    packed copy), one coalesced 1-KB load per wave per k-block, prefetched like the weights: no x tile in
//    LDS, no staging barriers;
//  * the only per-step synchronisation is the h exchange, done with a monotonic LDS counter instead of
//    s_barrier: a wave may run its x-part MFMAs of step s+1 as soon as its own cell update of step s is
//    done, and waits only before the h-part;
//  * waves 4-7 (the SIMD partners of waves 0-3) start half an x-part late, so one partner's cell update
//    (VALU) overlaps the other's x-part (MFMA) instead of both idling the matrix pipe together.
struct DecArgs {
    const float* xp;      // packed encoder output [n_tiles][33][64][64][4]
    const float* wp;      // packed [2 dirs][8 waves][96 kb][4 tiles][64][4]
    const float* bias;    // [2][1024]
    float* out;           // [B,33,512]
    int64_t B;
    int n_tiles;
    int stagger;          // s_sleep(127) iterations for waves 4-7
    unsigned long long* stamps;  // diagnostic builds (-DPV_STAMPS) only: [grid][8 waves][4] phase cycle sums
};

__global__ __launch_bounds__(512, 2) void k_lstm_dec_stagger(DecArgs a) {
    constexpr int LDH = H + 4;
    constexpr int NKB_X = 64, NKB_H = H / 8, NT = 4, UW = 32;
    extern __shared__ float smem[];
    float* hbuf = smem;  // [2][32][LDH]
    __shared__ int s_hdone;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int xcd = blockIdx.x & 7;
    const int dir = xcd & 1;
    const int tile = (blockIdx.x >> 3) * 4 + (xcd >> 1);
    if (tile >= a.n_tiles) return;
    const int64_t b0 = (int64_t)tile * ROWS;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp + ((size_t)(dir * 8 + wv) * (NKB_X + NKB_H)) * NT * 256) + lane;
    const f32x4* xp = reinterpret_cast<const f32x4*>(a.xp + (size_t)tile * T_STEPS * NKB_X * 256) + lane;
    const float* bias = a.bias + dir * 4 * H;
    for (int i = tid; i < 2 * ROWS * LDH; i += 512) hbuf[i] = 0.0f;
    if (tid == 0) s_hdone = 0;
    f32x16 cst;
#pragma unroll
    for (int r = 0; r < 16; r++) cst[r] = 0.0f;
    float bs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) bs[nt] = bias[nt * H + UW * wv + (lane & 31)];
    __syncthreads();
    if (wv >= 4)
        for (int i = 0; i < a.stagger; i++) __builtin_amdgcn_s_sleep(127);
    const float* ah_base = hbuf + (lane & 31) * LDH + 4 * (lane >> 5);

#ifdef PV_STAMPS
    unsigned long long st_x = 0, st_w = 0, st_h = 0, st_c = 0, t0s, t1s;
#define STAMP(v) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(v)
#endif
    for (int s = 0; s < T_STEPS; s++) {
        const int t = dir ? (T_STEPS - 1 - s) : s;
        const int cur = s & 1, nxt = cur ^ 1;
#ifdef PV_STAMPS
        STAMP(t0s)
#endif
        const f32x4* xs = xp + (size_t)t * NKB_X * 64;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = bs[nt];
        // ---- x part: both operands stream from L2/HBM, one k-block ahead (two register sets, compiler-scheduled)
        f32x4 b0v[NT], b1v[NT], a0, a1;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b0v[nt] = wp[nt * 64];
        a0 = xs[0];
#pragma nounroll
        for (int kb = 0; kb < NKB_X; kb += 2) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b1v[nt] = wp[((kb + 1) * NT + nt) * 64];
            a1 = xs[(kb + 1) * 64];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0v[nt][j], acc[nt], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b0v[nt] = wp[((kb + 2) * NT + nt) * 64];  // kb+2 == NKB_X: first h block
            if (kb + 2 < NKB_X) a0 = xs[(kb + 2) * 64];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1v[nt][j], acc[nt], 0, 0, 0);
        }
#ifdef PV_STAMPS
        STAMP(t1s) st_x += t1s - t0s; t0s = t1s;
#endif
        // ---- wait for h_{s-1} of every wave ------------------------------------------------------------------
        if (s > 0) {
            const int need = 8 * s;
            while (__hip_atomic_load(&s_hdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");  // LDS has no cache: program order is enough, and a fence would also drain vmcnt
        }
#ifdef PV_STAMPS
        STAMP(t1s) st_w += t1s - t0s; t0s = t1s;
#endif
        // ---- h part: A from LDS, B keeps streaming -------------------------------------------------------------
        const float* ah = ah_base + cur * ROWS * LDH;
        a0 = *reinterpret_cast<const f32x4*>(ah);
#pragma nounroll
        for (int kb = 0; kb < NKB_H; kb += 2) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b1v[nt] = wp[((NKB_X + kb + 1) * NT + nt) * 64];
            a1 = *reinterpret_cast<const f32x4*>(ah + 8 * (kb + 1));
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0v[nt][j], acc[nt], 0, 0, 0);
            if (kb + 2 < NKB_H) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) b0v[nt] = wp[((NKB_X + kb + 2) * NT + nt) * 64];
                a0 = *reinterpret_cast<const f32x4*>(ah + 8 * (kb + 2));
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1v[nt][j], acc[nt], 0, 0, 0);
        }
#ifdef PV_STAMPS
        STAMP(t1s) st_h += t1s - t0s; t0s = t1s;
#endif
        // ---- cell update ---------------------------------------------------------------------------------------
        float* hn = hbuf + nxt * ROWS * LDH;
        const int unit = UW * wv + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float ig = sigmoidf_(acc[0][r]);
            const float fg = sigmoidf_(acc[1][r]);
            const float gg = tanhf_(acc[2][r]);
            const float og = sigmoidf_(acc[3][r]);
            const float c = fg * cst[r] + ig * gg;
            cst[r] = c;
            const float h = og * tanhf_(c);
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            hn[row * LDH + unit] = h;
            float* ob = a.out + ((size_t)b0 * T_STEPS + t) * (2 * H) + dir * H;  // rows padded to the tile: no bounds branch
            ob[(unsigned)row * (T_STEPS * 2 * H) + (unsigned)unit] = h;
        }
        // publish: this wave's slice of h_s is in LDS. Only the LDS queue has to drain (it is in-order per wave);
        // a workgroup release fence would also wait for the h stores and the prefetched operands (vmcnt(0)).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(&s_hdone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef PV_STAMPS
        STAMP(t1s) st_c += t1s - t0s;
#endif
    }
#ifdef PV_STAMPS
    if (lane == 0 && a.stamps) {
        unsigned long long* o = a.stamps + ((size_t)blockIdx.x * 8 + wv) * 4;
        o[0] = st_x; o[1] = st_w; o[2] = st_h; o[3] = st_c;
    }
#endif
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
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
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
        for (int u = 0; u < XR; u++) xr[u] = buf_load4(asr, a_g, (unsigned)((2 * u * T_STEPS + t) * KC * 4));
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
        mma_stream_ringb<4>(acc, abuf, LDA, KC / 8, wr, t * (KC / 8), bq, lane);
        if (st + 1 < a.steps_per_split) {
            __syncthreads();  // every wave is done reading abuf
            a_store();
            __syncthreads();
        }
    }
    float* dst = a.part + (size_t)split * a.B * HEAD_N;
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
        const int n = 128 * wv + 32 * nt + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int64_t b = b0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (b < a.B) dst[b * HEAD_N + n] = acc[nt][r];
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
};

__global__ __launch_bounds__(256, 1) void k_head_tail(TailArgs a) {
    constexpr int LDY = HEAD_N + 4;
    extern __shared__ float smem[];
    float* y0 = smem;               // [32][LDY]
    float* y1 = smem + ROWS * LDY;  // [32][LDY]
    __shared__ float logits[ROWS][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b0 = (int64_t)blockIdx.x * ROWS;
    // y0 = selu(sum of slabs + b1)   (simple_model.py:57-59)
    for (int i = tid; i < ROWS * HEAD_N; i += 256) {
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
    for (int layer = 0; layer < 4; layer++) {  // linear_2..5 + SELU (:61-76)
        f32x16 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const float bv = a.b[layer][128 * wv + 32 * nt + (lane & 31)];
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = bv;
        }
        {
            const __amdgpu_buffer_rsrc_t wr = make_rsrc_sized(a.wp[layer] + (size_t)wv * (HEAD_N / 8) * 4 * 256,
                                                              (unsigned)((HEAD_N / 8) * 4 * 256 * sizeof(float)));
            f32x4 bq[4][4];
            ring_prime<4>(bq, wr, 0, lane);
            mma_stream_ringb<4>(acc, src, LDY, HEAD_N / 8, wr, 0, bq, lane);
        }
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            const int n = 128 * wv + 32 * nt + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[row * LDY + n] = seluf_(acc[nt][r]);
            }
        }
        __syncthreads();
        float* tmp = src; src = dst; dst = tmp;
    }
    // output_layer_type (512 -> 3) + softmax(dim=1) (:77-82): 8 lanes per (row, class) pair
    {
        const int pair = tid >> 3, sub = tid & 7;  // 32 pairs per pass
        for (int p = pair; p < ROWS * 3; p += 32) {
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
    if (tid < ROWS) {
        const int64_t b = b0 + tid;
        if (b < a.B) {
            const float l0 = logits[tid][0], l1 = logits[tid][1], l2 = logits[tid][2];
            const float m = fmaxf(l0, fmaxf(l1, l2));
            const float e0 = expf(l0 - m), e1 = expf(l1 - m), e2 = expf(l2 - m);
            const float inv = 1.0f / (e0 + e1 + e2);
            a.probs[b * 3 + 0] = e0 * inv;
            a.probs[b * 3 + 1] = e1 * inv;
            a.probs[b * 3 + 2] = e2 * inv;
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
    const float* Ac;      // activations fp32, CHUNK-MAJOR [K/32][M][32]: one BK slice of a row tile is 16 KB contiguous
    const __bf16* Wh;     // weights bf16 hi, chunk-major [K/32][N][32]
    const __bf16* Wl;     // weights bf16 lo, chunk-major
    const float* bias;    // [N] or NULL
    float* C;             // [splits][M, N]
    int64_t M;
    int N, K, splits;
};

// C tile 128 x 128 per workgroup (4 waves as 2 x 2, 64 x 64 each), BK = 32, double-buffered LDS, register staging.
__global__ __launch_bounds__(256, 2) void k_gemm_bf16x3(GemmArgs g) {
    constexpr int BM = 128, BN = 128, BK = 32, LD = 40;  // LD: bf16 elements per LDS row (80 B: conflict-free b128 reads)
    extern __shared__ __bf16 sm16[];
    __bf16* sAh = sm16;                    // [2][BM*LD]
    __bf16* sAl = sAh + 2 * BM * LD;
    __bf16* sBh = sAl + 2 * BM * LD;       // [2][BN*LD]
    __bf16* sBl = sBh + 2 * BN * LD;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wr = wv >> 1, wc = wv & 1;
    const int n_tiles_n = g.N / BN;
    const int64_t mt = blockIdx.x / n_tiles_n;
    const int nt_ = blockIdx.x - (int)(mt * n_tiles_n);
    const int64_t m0 = mt * BM;
    const int n0 = nt_ * BN;
    const int kslice = g.K / g.splits;
    const int kbeg = blockIdx.y * kslice;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.0f;
    f32x4 ra[4];  // 16 fp32 activations
    f32x4 rb[4];  // 4 x 8 bf16 weights (hi, hi, lo, lo segments)
    auto g_load = [&](int k0) {
        const int kc = k0 >> 5;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
            int64_t m = m0 + row;
            if (m >= g.M) m = g.M - 1;
            ra[u] = *reinterpret_cast<const f32x4*>(g.Ac + ((size_t)kc * g.M + m) * 32 + c4 * 4);
            const int arr = idx >> 9, sidx = idx & 511, brow = sidx >> 2, c8 = sidx & 3;
            rb[u] = *reinterpret_cast<const f32x4*>((arr ? g.Wl : g.Wh) + ((size_t)kc * g.N + n0 + brow) * 32 + c8 * 8);
        }
    };
    auto s_store = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int idx = tid + u * 256, row = idx >> 3, c4 = idx & 7;
            bf16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                hi[j] = (__bf16)ra[u][j];
                lo[j] = (__bf16)(ra[u][j] - (float)hi[j]);
            }
            *reinterpret_cast<bf16x4*>(sAh + buf * BM * LD + row * LD + c4 * 4) = hi;
            *reinterpret_cast<bf16x4*>(sAl + buf * BM * LD + row * LD + c4 * 4) = lo;
            const int arr = idx >> 9, sidx = idx & 511, brow = sidx >> 2, c8 = sidx & 3;
            *reinterpret_cast<f32x4*>((arr ? sBl : sBh) + buf * BN * LD + brow * LD + c8 * 8) = rb[u];
        }
    };
    const int nk = kslice / BK;
    g_load(kbeg);
    s_store(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) g_load(kbeg + (kt + 1) * BK);
        const __bf16* pAh = sAh + buf * BM * LD + (wr * 64 + (lane & 31)) * LD + 8 * (lane >> 5);
        const __bf16* pAl = sAl + buf * BM * LD + (wr * 64 + (lane & 31)) * LD + 8 * (lane >> 5);
        const __bf16* pBh = sBh + buf * BN * LD + (wc * 64 + (lane & 31)) * LD + 8 * (lane >> 5);
        const __bf16* pBl = sBl + buf * BN * LD + (wc * 64 + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                ah[q] = *reinterpret_cast<const bf16x8*>(pAh + q * 32 * LD + ks * 16);
                al[q] = *reinterpret_cast<const bf16x8*>(pAl + q * 32 * LD + ks * 16);
                bh[q] = *reinterpret_cast<const bf16x8*>(pBh + q * 32 * LD + ks * 16);
                bl[q] = *reinterpret_cast<const bf16x8*>(pBl + q * 32 * LD + ks * 16);
            }
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) s_store(buf ^ 1);
        __syncthreads();
    }
    float* C = g.C + (size_t)blockIdx.y * g.M * g.N;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int n = n0 + wc * 64 + b * 32 + (lane & 31);
            const float bv = g.bias ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int64_t m = m0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.M) C[m * g.N + n] = acc[a][b][r] + bv;
            }
        }
}

// decoder recurrence on pre-computed input projections G (fp32 [Bp*33, 2048], bias included): per step only the
// h part runs on the f32 MFMA; G is fetched into registers at the start of the step and added in the cell update.
struct RecArgs {
    const float* G;       // [Bp, 33, 2][1024]  (row (b,t), columns dir*1024 + gate*256 + unit)
    const float* wp;      // packed decoder weights [2 dirs][8 waves][96 kb][4][64][4] (the h part starts at k-block 64)
    float* out;           // [Bp, 33, 512]
    float* out_cm;        // chunk-major copy [16896/32][Bp][32] (A operand of the linear_1 GEMM)
    int64_t cm_rows;      // Bp
    int n_tiles;
};

__global__ __launch_bounds__(512, 2) void k_lstm_rec_g(RecArgs a) {
    constexpr int LDH = H + 4, NKB_X = 64, NKB_H = H / 8, NT = 4, UW = 32;
    extern __shared__ float smem[];
    float* hbuf = smem;  // [2][32][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7;
    const int dir = xcd & 1;
    const int tile = (blockIdx.x >> 3) * 4 + (xcd >> 1);
    if (tile >= a.n_tiles) return;
    const int64_t b0 = (int64_t)tile * ROWS;
    const float* wph = a.wp + ((size_t)(dir * 8 + wv) * (NKB_X + NKB_H) + NKB_X) * NT * 256;
    for (int i = tid; i < 2 * ROWS * LDH; i += 512) hbuf[i] = 0.0f;
    f32x16 cst;
#pragma unroll
    for (int r = 0; r < 16; r++) cst[r] = 0.0f;
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(wph);
    f32x4 bq[4][NT];
    ring_prime<NT>(bq, wr, 0, lane);
    const int unit = UW * wv + (lane & 31);
    // raw buffer accesses (tile resource + lane offset + scalar offset): no per-lane 64-bit addresses next to the 64 gx values
    const __amdgpu_buffer_rsrc_t gsr = make_rsrc(a.G + (size_t)b0 * T_STEPS * 2048);
    const __amdgpu_buffer_rsrc_t osr = make_rsrc(a.out + (size_t)b0 * T_STEPS * 2 * H);
    const unsigned gl_l = (unsigned)((4 * (lane >> 5) * T_STEPS * 2048 + (lane & 31)) * 4);
    const unsigned og_l = (unsigned)((4 * (lane >> 5) * T_STEPS * 2 * H + (lane & 31)) * 4);
    const unsigned cm_l = (unsigned)((4 * (lane >> 5) * 32 + (lane & 31)) * 4);
    __syncthreads();
    for (int s = 0; s < T_STEPS; s++) {
        const int t = dir ? (T_STEPS - 1 - s) : s;
        const int cur = s & 1, nxt = cur ^ 1;
        // input projections of this step: 64 values per lane, in flight during the h-part MFMAs
        float gx[NT][16];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = (r & 3) + 8 * (r >> 2);  // compile-time part of the row; the lane part is in gl_l
                gx[nt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    gsr, gl_l, (unsigned)((t * 2048 + dir * 1024 + UW * wv + rr * T_STEPS * 2048 + nt * H) * 4), 0));
            }
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[nt][r] = 0.0f;
        mma_dual_ringb<NT>(acc, hbuf + cur * ROWS * LDH, LDH, NKB_H, hbuf, LDH, 0, wr, bq, lane);
        float* hn = hbuf + nxt * ROWS * LDH;
        const __amdgpu_buffer_rsrc_t cmr = make_rsrc(a.out_cm + ((size_t)(t * 16 + dir * 8 + wv) * a.cm_rows + (size_t)b0) * 32);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float ig = sigmoidf_(acc[0][r] + gx[0][r]);
            const float fg = sigmoidf_(acc[1][r] + gx[1][r]);
            const float gg = tanhf_(acc[2][r] + gx[2][r]);
            const float og = sigmoidf_(acc[3][r] + gx[3][r]);
            const float c = fg * cst[r] + ig * gg;
            cst[r] = c;
            const float h = og * tanhf_(c);
            const int rr = (r & 3) + 8 * (r >> 2);
            const int row = rr + 4 * (lane >> 5);
            hn[row * LDH + unit] = h;
            buf_store1(h, osr, og_l, (unsigned)((t * 2 * H + dir * H + UW * wv + rr * T_STEPS * 2 * H) * 4));
            // chunk-major copy: flattened [t][512] column = K index of linear_1; this wave's 32 units are one 32-wide chunk
            buf_store1(h, cmr, cm_l, (unsigned)(rr * 32 * 4));
        }
        __syncthreads();
    }
}

// ---- host-side weight packing -----------------------------------------------------------------------
// LSTM layer, NW waves per workgroup: NT = 32/NW tiles per wave, S2 = NT/4 sub-tiles;
// gate column of (wave w, tile nt, lane) = (nt/S2)*H + (H/NW)*w + 32*(nt%S2) + (lane&31)
static void pack_lstm(const pv_rnn_dir* dirs, int K, int KP, int NW, std::vector<float>& wp, std::vector<float>& bias) {
    const int nkb = (KP + H) / 8, NT = 32 / NW, S2 = NT / 4, UW = H / NW;
    wp.assign((size_t)2 * NW * nkb * NT * 256, 0.0f);
    bias.assign((size_t)2 * 4 * H, 0.0f);
    for (int d = 0; d < 2; d++) {
        for (int n = 0; n < 4 * H; n++) bias[(size_t)d * 4 * H + n] = dirs[d].b_ih[n] + dirs[d].b_hh[n];
        for (int w = 0; w < NW; w++)
            for (int kb = 0; kb < nkb; kb++)
                for (int nt = 0; nt < NT; nt++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int n = (nt / S2) * H + UW * w + 32 * (nt % S2) + (lane & 31);
                        float* dst = &wp[((((size_t)(d * NW + w) * nkb + kb) * NT + nt) * 64 + lane) * 4];
                        for (int j = 0; j < 4; j++) {
                            const int k = kb * 8 + 4 * (lane >> 5) + j;
                            float v = 0.0f;
                            if (k < KP) { if (k < K) v = dirs[d].w_ih[(size_t)n * K + k]; }
                            else v = dirs[d].w_hh[(size_t)n * H + (k - KP)];
                            dst[j] = v;
                        }
                    }
    }
}
// Linear [512, K]: column of (wave w, tile nt, lane) = 128w + 32nt + (lane&31)
static void pack_linear(const float* W, int K, std::vector<float>& wp) {
    const int nkb = K / 8;
    wp.assign((size_t)4 * nkb * 4 * 256, 0.0f);
    for (int w = 0; w < 4; w++)
        for (int kb = 0; kb < nkb; kb++)
            for (int nt = 0; nt < 4; nt++)
                for (int lane = 0; lane < 64; lane++) {
                    const int n = 128 * w + 32 * nt + (lane & 31);
                    float* dst = &wp[((((size_t)w * nkb + kb) * 4 + nt) * 64 + lane) * 4];
                    for (int j = 0; j < 4; j++) dst[j] = W[(size_t)n * K + kb * 8 + 4 * (lane >> 5) + j];
                }
}

}  // namespace

static constexpr size_t LDS_ENC = (size_t)(ROWS * (32 + 4) + 2 * ROWS * (H + 4)) * sizeof(float);
static constexpr size_t LDS_DEC = (size_t)(ROWS * (2 * H + 4) + 2 * ROWS * (H + 4)) * sizeof(float);
static constexpr size_t LDS_DEC_STAGGER = (size_t)(2 * ROWS * (H + 4)) * sizeof(float);
static constexpr size_t LDS_GEMM = (size_t)4 * 2 * 128 * 40 * 2;  // 4 operand arrays x 2 buffers x 128 rows x 40 bf16
static constexpr size_t LDS_SPLITK = (size_t)ROWS * (2 * H + 4) * sizeof(float);
static constexpr size_t LDS_TAIL = (size_t)2 * ROWS * (HEAD_N + 4) * sizeof(float);

struct pv_rnn_p1 {
    float* enc_wp = nullptr; float* enc_bias = nullptr;
    float* dec_wp = nullptr; float* dec_bias = nullptr;
    float* w1p = nullptr; float* b1 = nullptr;
    float* wlp[4] = {nullptr, nullptr, nullptr, nullptr};
    float* bl[4] = {nullptr, nullptr, nullptr, nullptr};
    float* wo = nullptr; float* bo = nullptr;
    int dtype = PV_DTYPE_F32;
    // PV_DTYPE_BF16_INPUT_GEMM: bf16 hi/lo splits of the decoder W_ih (both directions, [2048,512]) and linear_1 ([512,16896])
    __bf16* dec_wih_h = nullptr; __bf16* dec_wih_l = nullptr; float* dec_bias_cat = nullptr;
    __bf16* w1_h = nullptr; __bf16* w1_l = nullptr;
    int nw = 8;  // waves per LSTM workgroup (PV_LSTM_WAVES=4|8)
    int dec_stagger = -1;  // PV_DEC_STAGGER: -1 (default) = barrier decoder k_lstm_layer<512>; >= 0 = experimental flag-synchronised
                           // decoder k_lstm_dec_stagger with that many sleep units of stagger (measured slower: DESIGN.md section 6)
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
// w [N, K] row-major -> (hi, lo) bf16 with w ~= hi + lo, stored chunk-major [K/32][N][32]
static int dev_upload_split(const float* w, size_t N, size_t K, __bf16** d_hi, __bf16** d_lo, std::vector<void*>& owned) {
    const size_t n = N * K;
    std::vector<uint16_t> hi(n), lo(n);
    for (size_t r = 0; r < N; r++)
        for (size_t k = 0; k < K; k++) {
            const float v = w[r * K + k];
            const size_t o = ((k >> 5) * N + r) * 32 + (k & 31);
            hi[o] = f2bf_bits(v);
            lo[o] = f2bf_bits(v - bf_bits2f(hi[o]));
        }
    PV_HIP(hipMalloc((void**)d_hi, n * 2));
    owned.push_back(*d_hi);
    PV_HIP(hipMalloc((void**)d_lo, n * 2));
    owned.push_back(*d_lo);
    PV_HIP(hipMemcpy(*d_hi, hi.data(), n * 2, hipMemcpyHostToDevice));
    PV_HIP(hipMemcpy(*d_lo, lo.data(), n * 2, hipMemcpyHostToDevice));
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
    if (const char* e = getenv("PV_LSTM_WAVES")) m->nw = (atoi(e) == 4) ? 4 : 8;
    if (const char* e = getenv("PV_DEC_STAGGER")) m->dec_stagger = atoi(e);
    if (m->nw != 8) m->dec_stagger = -1;
    pack_lstm(w->encoder, F_IN, 32, m->nw, wp, bias);
    if ((rc = dev_upload(wp, &m->enc_wp, m->owned)) || (rc = dev_upload(bias, &m->enc_bias, m->owned))) return rc;
    pack_lstm(w->decoder, 2 * H, 2 * H, m->nw, wp, bias);
    if ((rc = dev_upload(wp, &m->dec_wp, m->owned)) || (rc = dev_upload(bias, &m->dec_bias, m->owned))) return rc;
    pack_linear(w->linear_w[0], HEAD_K, wp);
    if ((rc = dev_upload(wp, &m->w1p, m->owned)) || (rc = dev_upload(w->linear_b[0], HEAD_N, &m->b1, m->owned))) return rc;
    for (int i = 0; i < 4; i++) {
        pack_linear(w->linear_w[i + 1], HEAD_N, wp);
        if ((rc = dev_upload(wp, &m->wlp[i], m->owned)) || (rc = dev_upload(w->linear_b[i + 1], HEAD_N, &m->bl[i], m->owned)))
            return rc;
    }
    if ((rc = dev_upload(w->out_w, 3 * HEAD_N, &m->wo, m->owned)) || (rc = dev_upload(w->out_b, 3, &m->bo, m->owned))) return rc;
    if (dtype == PV_DTYPE_BF16_INPUT_GEMM) {
        std::vector<float> wcat((size_t)2048 * 512), bcat(2048);
        for (int d = 0; d < 2; d++) {
            memcpy(&wcat[(size_t)d * 1024 * 512], w->decoder[d].w_ih, (size_t)1024 * 512 * sizeof(float));
            for (int n = 0; n < 1024; n++) bcat[d * 1024 + n] = w->decoder[d].b_ih[n] + w->decoder[d].b_hh[n];
        }
        if ((rc = dev_upload_split(wcat.data(), 2048, 512, &m->dec_wih_h, &m->dec_wih_l, m->owned))) return rc;
        if ((rc = dev_upload(bcat, &m->dec_bias_cat, m->owned))) return rc;
        if ((rc = dev_upload_split(w->linear_w[0], HEAD_N, HEAD_K, &m->w1_h, &m->w1_l, m->owned))) return rc;
        PV_HIP(hipFuncSetAttribute((const void*)k_gemm_bf16x3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_GEMM));
        PV_HIP(hipFuncSetAttribute((const void*)k_lstm_rec_g, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DEC_STAGGER));
    }
    // opt in to > 64 KB of dynamic LDS (exact sizes; static LDS counts against the 160 KB too)
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<32, true, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_ENC));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<512, false, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DEC));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<32, true, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_ENC));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<32, true, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_ENC));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_layer<512, false, 8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DEC));
    PV_HIP(hipFuncSetAttribute((const void*)k_lstm_dec_stagger, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_DEC_STAGGER));
    PV_HIP(hipFuncSetAttribute((const void*)k_head_splitk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_SPLITK));
    PV_HIP(hipFuncSetAttribute((const void*)k_head_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TAIL));
    return PV_OK;
}

static int p1_forward_launch(pv_ctx* ctx, const int8_t* d_images, int64_t B, float* d_probs, float* enc_out,
                             float* dec_out, float* part, hipStream_t st, float* enc_packed = nullptr) {
    pv_rnn_p1* m = ctx->p1;
    const int n_tiles = (int)((B + ROWS - 1) / ROWS);
    const unsigned lstm_grid = (unsigned)(((n_tiles + 3) / 4) * 8);
    LstmArgs e;
    e.ablate = getenv("PV_ABLATE") ? atoi(getenv("PV_ABLATE")) : 0;
    e.stamps = nullptr;
#ifdef PV_STAMPS
    if (getenv("PV_STAMP_ENC")) {
        unsigned long long* sp = nullptr;
        if (pv_get(ctx, "p1.stamps", (size_t)lstm_grid * 8 * 4, &sp) == PV_OK) e.stamps = sp;
    }
#endif
    const bool stag = m->dec_stagger >= 0 && enc_packed != nullptr && m->dtype == PV_DTYPE_F32;
    e.x_i8 = d_images; e.x_f32 = nullptr; e.wp = m->enc_wp; e.bias = m->enc_bias; e.out = enc_out; e.B = B; e.n_tiles = n_tiles;
    e.out_packed = stag ? enc_packed : nullptr;
    e.out_cm = nullptr; e.cm_rows = 0;
    struct { float *enc_cm, *dec_cm; } bf = {nullptr, nullptr};
    if (m->dtype == PV_DTYPE_BF16_INPUT_GEMM) {
        const size_t nel = (size_t)n_tiles * ROWS * T_STEPS * 2 * H;
        int rcb;
        if ((rcb = pv_get(ctx, "p1.enc_cm", nel, &bf.enc_cm)) || (rcb = pv_get(ctx, "p1.dec_cm", nel, &bf.dec_cm))) return rcb;
        e.out_cm = bf.enc_cm; e.cm_rows = (int64_t)n_tiles * ROWS * T_STEPS;
    }
    {
        pv_prof_scope ps(ctx, "k_lstm_layer_enc", st);
        if (m->nw == 8 && stag) k_lstm_layer<32, true, 8, true><<<lstm_grid, 512, LDS_ENC, st>>>(e);
        else if (m->nw == 8) k_lstm_layer<32, true, 8, false><<<lstm_grid, 512, LDS_ENC, st>>>(e);
        else k_lstm_layer<32, true, 4, false><<<lstm_grid, 256, LDS_ENC, st>>>(e);
    }
    if (m->dtype == PV_DTYPE_BF16_INPUT_GEMM) {
        // decoder: G = enc_out . W_ih^T + b on the bf16 MFMA (3-term split), then the fp32 recurrence on G
        const int64_t Bp = (int64_t)n_tiles * ROWS, M = Bp * T_STEPS;
        float* G = nullptr;
        int rc2 = pv_get(ctx, "p1.G", (size_t)M * 2048, &G);
        if (rc2) return rc2;
        GemmArgs ga;
        ga.Ac = bf.enc_cm; ga.Wh = m->dec_wih_h; ga.Wl = m->dec_wih_l; ga.bias = m->dec_bias_cat; ga.C = G;
        ga.M = M; ga.N = 2048; ga.K = 2 * H; ga.splits = 1;
        {
            pv_prof_scope ps(ctx, "k_gemm_bf16x3_dec", st);
            k_gemm_bf16x3<<<dim3((unsigned)(((M + 127) / 128) * (2048 / 128)), 1), 256, LDS_GEMM, st>>>(ga);
        }
        RecArgs ra;
        ra.G = G; ra.wp = m->dec_wp; ra.out = dec_out; ra.out_cm = bf.dec_cm; ra.cm_rows = Bp; ra.n_tiles = n_tiles;
        {
            pv_prof_scope ps(ctx, "k_lstm_rec_g", st);
            k_lstm_rec_g<<<lstm_grid, 512, LDS_DEC_STAGGER, st>>>(ra);
        }
        // linear_1 as a split-K bf16x3 GEMM into slabs [splits][Bp][512]
        const int64_t mtiles = (Bp + 127) / 128;
        int gs = 1;
        while (gs < 8 && mtiles * 4 * gs < ctx->num_cu && (HEAD_K / 32) % (gs * 2) == 0) gs *= 2;
        GemmArgs gl;
        gl.Ac = bf.dec_cm; gl.Wh = m->w1_h; gl.Wl = m->w1_l; gl.bias = nullptr; gl.C = part;
        gl.M = Bp; gl.N = HEAD_N; gl.K = HEAD_K; gl.splits = gs;
        {
            pv_prof_scope ps(ctx, "k_gemm_bf16x3_lin1", st);
            k_gemm_bf16x3<<<dim3((unsigned)(mtiles * 4), (unsigned)gs), 256, LDS_GEMM, st>>>(gl);
        }
        TailArgs tb;
        tb.part = part; tb.b1 = m->b1; tb.splits = gs; tb.part_rows = Bp;
        for (int i = 0; i < 4; i++) { tb.wp[i] = m->wlp[i]; tb.b[i] = m->bl[i]; }
        tb.wo = m->wo; tb.bo = m->bo; tb.probs = d_probs; tb.B = B;
        { pv_prof_scope ps(ctx, "k_head_tail", st); k_head_tail<<<(unsigned)n_tiles, 256, LDS_TAIL, st>>>(tb); }
        PV_HIP(hipGetLastError());
        return PV_OK;
    }
    LstmArgs d = e;
    d.stamps = nullptr;
    d.x_i8 = nullptr; d.x_f32 = enc_out; d.wp = m->dec_wp; d.bias = m->dec_bias; d.out = dec_out;
    d.out_packed = nullptr;
#ifdef PV_STAMPS
    {
        unsigned long long* sp = nullptr;
        if (!getenv("PV_STAMP_ENC") && pv_get(ctx, "p1.stamps", (size_t)lstm_grid * 8 * 4, &sp) == PV_OK) d.stamps = sp;
    }
#endif
    if (stag) {
        DecArgs da;
        da.xp = enc_packed; da.wp = m->dec_wp; da.bias = m->dec_bias; da.out = dec_out; da.B = B; da.n_tiles = n_tiles;
        da.stagger = m->dec_stagger;
        da.stamps = nullptr;
#ifdef PV_STAMPS
        {
            unsigned long long* sp = nullptr;
            if (pv_get(ctx, "p1.stamps", (size_t)lstm_grid * 8 * 4, &sp) == PV_OK) da.stamps = sp;
        }
#endif
        pv_prof_scope ps(ctx, "k_lstm_layer_dec", st);
        k_lstm_dec_stagger<<<lstm_grid, 512, LDS_DEC_STAGGER, st>>>(da);
    } else {
        pv_prof_scope ps(ctx, "k_lstm_layer_dec", st);
        if (m->nw == 8) k_lstm_layer<512, false, 8, false><<<lstm_grid, 512, LDS_DEC, st>>>(d);
        else k_lstm_layer<512, false, 4, false><<<lstm_grid, 256, LDS_DEC, st>>>(d);
    }
    HeadArgs h;
    // split-K factor: 11 slabs of 3 time steps; 33 single-step slabs only for batches too small to fill the chip
    int splits = ((int64_t)n_tiles * 11 >= ctx->num_cu) ? 11 : 33;
    if (const char* e = getenv("PV_HEAD_SPLITS")) { const int v = atoi(e); if (v == 1 || v == 3 || v == 11 || v == 33) splits = v; }
    h.dec = dec_out; h.w1p = m->w1p; h.part = part; h.B = B; h.n_tiles = n_tiles; h.splits = splits; h.steps_per_split = T_STEPS / splits;
    static const int head_map = getenv("PV_HEAD_MAP") ? atoi(getenv("PV_HEAD_MAP")) : 1;
    const int total_wg = n_tiles * splits;
    h.per_xcd = head_map ? (total_wg + 7) / 8 : 0;
    const unsigned head_grid = head_map ? (unsigned)(h.per_xcd * 8) : (unsigned)total_wg;
    { pv_prof_scope ps(ctx, "k_head_splitk", st); k_head_splitk<<<head_grid, 256, LDS_SPLITK, st>>>(h); }
    TailArgs t;
    t.part = part; t.b1 = m->b1; t.splits = splits; t.part_rows = B;
    for (int i = 0; i < 4; i++) { t.wp[i] = m->wlp[i]; t.b[i] = m->bl[i]; }
    t.wo = m->wo; t.bo = m->bo; t.probs = d_probs; t.B = B;
    { pv_prof_scope ps(ctx, "k_head_tail", st); k_head_tail<<<(unsigned)n_tiles, 256, LDS_TAIL, st>>>(t); }
    PV_HIP(hipGetLastError());
    return PV_OK;
}

static int p1_workspace(pv_ctx* ctx, int64_t B, float** enc, float** dec, float** part, float** enc_packed = nullptr) {
    int rc;
    if (enc_packed) {
        const size_t n_tiles = (size_t)((B + ROWS - 1) / ROWS);
        if ((rc = pv_get(ctx, "p1.enc_packed", n_tiles * T_STEPS * 64 * 256, enc_packed))) return rc;
    }
    const size_t Bp = (size_t)((B + ROWS - 1) / ROWS) * ROWS;  // LSTM kernels store whole 32-row tiles
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
    float *enc, *dec, *part;
    float* encp = nullptr;
    int rc = p1_workspace(ctx, B, &enc, &dec, &part, ctx->p1->dec_stagger >= 0 ? &encp : nullptr);
    if (rc) return rc;
    return p1_forward_launch(ctx, d_images, B, d_probs, enc, dec, part, pv_pick_stream(ctx, stream), encp);
}

extern "C" int pv_rnn_forward_p1_debug(pv_ctx* ctx, const int8_t* images, int64_t B, float* probs, float* enc_out,
                                       float* dec_out) {
    PV_CHECK(ctx && images && probs, PV_ERR_INVALID, "null argument");
    PV_CHECK(ctx->p1, PV_ERR_STATE, "pv_rnn_load_p1 has not been called on this context");
    PV_CHECK(B >= 0 && B < (1ll << 24), PV_ERR_INVALID, "batch %lld out of range", (long long)B);
    if (B == 0) return PV_OK;
    PV_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    float *enc, *dec, *part, *d_probs;
    int8_t* d_img;
    // the encoder tap needs the row-major encoder output, which only the barrier decoder path produces
    float* encp = nullptr;
    const bool want_packed = ctx->p1->dec_stagger >= 0 && !enc_out;
    int rc = p1_workspace(ctx, B, &enc, &dec, &part, want_packed ? &encp : nullptr);
    if (rc) return rc;
    if ((rc = pv_get(ctx, "p1.images", (size_t)B * PV_WINDOW_BYTES, &d_img))) return rc;
    if ((rc = pv_get(ctx, "p1.probs", (size_t)B * 3, &d_probs))) return rc;
    PV_HIP(hipMemcpyAsync(d_img, images, (size_t)B * PV_WINDOW_BYTES, hipMemcpyHostToDevice, st));
    if ((rc = p1_forward_launch(ctx, d_img, B, d_probs, enc, dec, part, st, encp))) return rc;
    PV_HIP(hipMemcpyAsync(probs, d_probs, (size_t)B * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    const size_t nb = (size_t)B * T_STEPS * 2 * H * sizeof(float);
    if (enc_out) PV_HIP(hipMemcpyAsync(enc_out, enc, nb, hipMemcpyDeviceToHost, st));
    if (dec_out) PV_HIP(hipMemcpyAsync(dec_out, dec, nb, hipMemcpyDeviceToHost, st));
    PV_HIP(hipStreamSynchronize(st));
    return PV_OK;
}

extern "C" int pv_rnn_forward_p1(pv_ctx* ctx, const int8_t* images, int64_t B, float* probs) {
    return pv_rnn_forward_p1_debug(ctx, images, B, probs, nullptr, nullptr);
}

// ---- P2 (bi-GRU polisher model): see rnn_gru.hip ----------------------------------------------------

#ifdef PV_STAMPS
// diagnostic builds only: copy the phase stamps of the last staggered-decoder launch to the host
extern "C" int pv_debug_read_stamps(pv_ctx* ctx, unsigned long long* out, int64_t n) {
    unsigned long long* sp = nullptr;
    if (pv_get(ctx, "p1.stamps", (size_t)n, &sp)) return PV_ERR_HIP;
    PV_HIP(hipDeviceSynchronize());
    PV_HIP(hipMemcpy(out, sp, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PV_OK;
}
#endif
