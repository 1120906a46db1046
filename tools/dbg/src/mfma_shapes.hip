// build: hipcc --offload-arch=gfx950 -O3 -o tools/dbg/src/mfma_shapes tools/dbg/src/mfma_shapes.hip ; run: ./tools/dbg/src/mfma_shapes
// Diagnostic: sustained bf16 MFMA rate of the two shapes (operands in registers, 2 waves per SIMD, every CU), a second or so each:
// does the chip hold a higher clock on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
    bf16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    float r = 0.f;
    if (SHAPE == 32) {
        f32x16 c[4];
        for (int i = 0; i < 4; i++) for (int e = 0; e < 16; e++) c[i][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 4; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) r += c[i][0];
    } else {
        f32x4 c[16];
        for (int i = 0; i < 16; i++) for (int e = 0; e < 4; e++) c[i][e] = 0.f;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int i = 0; i < 16; i++) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; i++) r += c[i][0];
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int SHAPE> void run(const char* name, float* out, unsigned long long* cyc, int iters) {
    const double flop_per_iter_per_wave = SHAPE == 32 ? 16.0 * 2 * 32 * 32 * 16 : 32.0 * 2 * 16 * 16 * 32;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        k<SHAPE><<<256, 512>>>(out, iters, cyc);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c0; (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        const double tf = flop_per_iter_per_wave * iters * 8 * 256 / (ms * 1e-3) / 1e12;
        printf("%s: %.1f ms, %.0f TFLOP/s, %.3f GHz (cycles of workgroup 0 / time)\n", name, ms, tf, c0 / (ms * 1e-3) / 1e9);
    }
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 400000;
    run<32>("32x32x16", out, cyc, iters);
    run<16>("16x16x32", out, cyc, iters);
    run<32>("32x32x16", out, cyc, iters);
    run<16>("16x16x32", out, cyc, iters);
    return 0;
}
