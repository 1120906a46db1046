// mfma_tiles.hpp — device helpers shared by the recurrent kernels (rnn_kernels.hip, rnn_gru.hip), gfx950 only.
//
//  * raw buffer accesses: a 128-bit resource in SGPRs + one 32-bit lane offset + a scalar offset. Unlike global_load /
//    global_store with per-lane 64-bit addresses they need no address VGPRs at all, which is what keeps kernels that hold
//    64 accumulator + 48-64 ring + 32 staging registers per lane inside the register file;
//  * the two batch-tile forms of a "gate accumulator": 32 hidden units of one gate for TR batch rows.
//      TR = 32: one 32x32 tile, v_mfma_f32_32x32x2_f32
//      TR = 16: two 16x16 tiles, v_mfma_f32_16x16x4_f32 — the same FLOP per cycle, half the cycles per recurrent step and
//               twice the workgroups: the form for batches that do not fill the chip (a launch is a chain of dependent
//               steps, so its duration is set by the per-step MFMA time of ONE workgroup until the workgroups fill all CUs).
#pragma once
#include <hip/hip_runtime.h>

namespace pvdev {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// with a size: requests beyond it return zeros / are dropped (raw buffer bounds check)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_sized(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// streaming forms (aux bit 1 = nt): data touched once; the hint lets L2 evict it first, so that operands that ARE reused
// (the packed weights, 3.1 MB per direction in a 4 MB L2) stay resident
__device__ __forceinline__ f32x4 buf_load4_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2));
}
__device__ __forceinline__ void buf_store1_nt(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 2);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

// Workgroup barrier for LDS hand-offs only. `__syncthreads()` is a workgroup-scope fence + s_barrier, and for the fence hipcc
// drains EVERYTHING the wave has in flight (`s_waitcnt vmcnt(0) lgkmcnt(0)`): the weight fragments requested several k-blocks
// ahead and the step's output stores, i.e. an L2 round trip per recurrent step with no MFMA issued. The per-step barriers of
// the recurrent kernels only publish LDS tiles (h, x), so they wait for this wave's LDS operations and nothing else; global
// data is ordered where it is handed over (end of a layer / window: a full __syncthreads() or an agent-scope hand-off).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int TR> struct Gate;
template <> struct Gate<32> { f32x16 v; };      // lane -> unit lane&31, rows (e&3) + 8*(e>>2) + 4*(lane>>5), e = 0..15
template <> struct Gate<16> { f32x4 v[2]; };    // lane -> unit 16*t + (lane&15), rows 4*(lane>>4) + i, e = 4t + i
template <int TR> struct AFrag;
template <> struct AFrag<32> { typedef f32x4 type; };  // A[row = lane&31][8kb + 4*(lane>>5) + j], j = 0..3
template <> struct AFrag<16> { typedef f32x2 type; };  // A[row = lane&15][8kb + 2*(lane>>4) + j], j = 0..1

template <int TR> __device__ __forceinline__ float gate_get(const Gate<TR>& g, int e) {
    if constexpr (TR == 32) return g.v[e];
    else return g.v[e >> 2][e & 3];
}
template <int TR> __device__ __forceinline__ void gate_set(Gate<TR>& g, int e, float x) {
    if constexpr (TR == 32) g.v[e] = x;
    else g.v[e >> 2][e & 3] = x;
}
// One k-step (j) of one gate. B fragment (one f32x4 per lane per gate per k-block of 8):
//   TR = 32: W[unit lane&31][8kb + 4*(lane>>5) + j], j = 0..3
//   TR = 16: {tile0 j0, tile0 j1, tile1 j0, tile1 j1} with W[unit 16t + (lane&15)][8kb + 2*(lane>>4) + j]
template <int TR> __device__ __forceinline__ void gate_mma(Gate<TR>& g, const typename AFrag<TR>::type& a, const f32x4& b, int j) {
    if constexpr (TR == 32) {
        g.v = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], g.v, 0, 0, 0);
    } else {
        g.v[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], g.v[0], 0, 0, 0);
        g.v[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[2 + j], g.v[1], 0, 0, 0);
    }
}
// element e of a gate accumulator -> (row, unit) = (lane part) + (compile-time part)
template <int TR> __device__ __forceinline__ int lane_row(int lane) { return TR == 32 ? 4 * (lane >> 5) : 4 * (lane >> 4); }
template <int TR> __device__ __forceinline__ int lane_unit(int lane) { return TR == 32 ? (lane & 31) : (lane & 15); }
template <int TR> __device__ __forceinline__ constexpr int elem_row(int e) { return TR == 32 ? (e & 3) + 8 * (e >> 2) : (e & 3); }
template <int TR> __device__ __forceinline__ constexpr int elem_unit(int e) { return TR == 32 ? 0 : 16 * (e >> 2); }
// the A-fragment address of a lane in an LDS tile with `lda` floats per row
template <int TR> __device__ __forceinline__ const float* afrag_ptr(const float* A, int lda, int lane) {
    return TR == 32 ? A + (lane & 31) * lda + 4 * (lane >> 5) : A + (lane & 15) * lda + 2 * (lane >> 4);
}

}  // namespace pvdev
