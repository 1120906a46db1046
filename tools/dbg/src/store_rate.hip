// build: hipcc --offload-arch=gfx950 -O3 -o tools/dbg/src/store_rate tools/dbg/src/store_rate.hip ; run: ./tools/dbg/src/store_rate
// Diagnostic: how fast can ONE CU issue 16-byte-per-lane stores? (k_gemm_bf16x3's epilogue: 256 KB per tile)
// 256 workgroups x 512 threads, every wave stores NST x 1 KB; cycles per workgroup by s_memtime -> bytes per clock and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NST = 64;
template <int MODE>
__global__ __launch_bounds__(512) void k_store(float* out, unsigned long long* cyc, size_t wg_stride_f) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* base = out + (size_t)blockIdx.x * wg_stride_f + (size_t)wv * NST * 256;
    f32x4 v = {(float)tid, 1.f, 2.f, 3.f};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) {
#pragma unroll 8
        for (int i = 0; i < NST; i++) { *reinterpret_cast<f32x4*>(base + i * 256 + lane * 4) = v; v[0] += 1.f; }
    } else if (MODE == 1) {
#pragma unroll 8
        for (int i = 0; i < NST; i++) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + i * 256 + lane * 4)); v[0] += 1.f; }
    } else if (MODE == 2) {   // dword stores, same bytes
#pragma unroll 8
        for (int i = 0; i < NST * 4; i++) { base[i * 64 + lane] = v[0]; v[0] += 1.f; }
    } else if (MODE == 3) {   // two 512-byte halves 4 KB apart (the quad layout's lane split)
#pragma unroll 8
        for (int i = 0; i < NST; i++) { *reinterpret_cast<f32x4*>(base + (i >> 1) * 512 + (i & 1) * 128 + (lane >> 5) * 256 + (lane & 31) * 4) = v; v[0] += 1.f; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, float* out, unsigned long long* cyc, int grid) {
    const size_t stride = (size_t)8 * NST * 256;
    k_store<MODE><<<grid, 512>>>(out, cyc, stride);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k_store<MODE><<<grid, 512>>>(out, cyc, stride);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x;
    const double bytes = 8.0 * NST * 1024;
    printf("%-28s grid %4d: %.1f cycles per workgroup -> %.1f B/clk/CU (issue + drain), kernel %.3f ms = %.2f TB/s\n", name, grid, s / grid, bytes / (s / grid), ms, bytes * grid / ms / 1e9);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)2048 * 8 * NST * 1024); hipMalloc(&cyc, 2048 * 8);
    for (int grid : {1, 8, 64, 256, 2048}) {
        run<0>("dwordx4", out, cyc, grid);
        run<1>("dwordx4 nontemporal", out, cyc, grid);
        run<2>("dword", out, cyc, grid);
        run<3>("dwordx4 split halves", out, cyc, grid);
    }
    return 0;
}
