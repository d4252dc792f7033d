// Micro-benchmark: sustained rate of v_mfma_f32_32x32x16_f16 with no memory traffic (not part of the product).
// 8 waves per CU (2 per SIMD), 4 independent accumulators per wave, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, long long* cycles, int noisy) {
    f16x8 a = {(_Float16)(threadIdx.x & 3), 1, 0, 1, 0, 1, 0, 1}, b = {1, 0, (_Float16)(threadIdx.x & 1), 0, 1, 0, 1, 0};
    if (noisy) {  // operands with realistic bit activity (pseudo-random mantissas, mixed signs)
        unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            a[j] = (_Float16)(((int)(s >> 9) & 0x3fff) - 8192) * (_Float16)(1.0f / 8192);
            s = s * 1664525u + 1013904223u;
            b[j] = (_Float16)(((int)(s >> 9) & 0x3fff) - 8192) * (_Float16)(1.0f / 8192);
        }
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

// the same with v_mfma_f32_16x16x32_f16 (half the cycles, half the FLOPs per instruction): MI355X_MICROARCH.md "DVFS give-back" (7) reports
// that under load the chip holds a higher clock on this shape
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma16_loop(float* out, int iters, long long* cycles, int noisy) {
    f16x8 a = {(_Float16)(threadIdx.x & 3), 1, 0, 1, 0, 1, 0, 1}, b = {1, 0, (_Float16)(threadIdx.x & 1), 0, 1, 0, 1, 0};
    if (noisy) {
        unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            a[j] = (_Float16)(((int)(s >> 9) & 0x3fff) - 8192) * (_Float16)(1.0f / 8192);
            s = s * 1664525u + 1013904223u;
            b[j] = (_Float16)(((int)(s >> 9) & 0x3fff) - 8192) * (_Float16)(1.0f / 8192);
        }
    }
    f32x4 c[8];
    for (int k = 0; k < 8; ++k) c[k] = f32x4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[j], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][1] + c[k][2] + c[k][3];
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    for (int noisy = 0; noisy < 2; ++noisy) for (int iters : {20000, 100000}) {
        hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, out, iters, cyc, noisy);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(512), 0, 0, out, iters, cyc, noisy);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double flops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8 * grid;
        printf("noisy %d iters %6d: %8.3f ms  %7.1f TFLOP/s  (s_memtime ticks %lld; MFMA issue slots at 32 cycles each: %.0f cycles/SIMD -> %.2f GHz)\n", noisy, iters, ms,
               flops / (ms * 1e-3) / 1e12, c, 2.0 * 16 * iters * 32, 2.0 * 16 * iters * 32 / (ms * 1e-3) / 1e9);
    }
    for (int noisy = 0; noisy < 2; ++noisy) for (int iters : {20000, 100000}) {
        hipLaunchKernelGGL(mfma16_loop, dim3(grid), dim3(512), 0, 0, out, iters, cyc, noisy);
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma16_loop, dim3(grid), dim3(512), 0, 0, out, iters, cyc, noisy);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * 16 * 16 * 32 * 32.0 * iters * 8 * grid;
        printf("16x16x32 noisy %d iters %6d: %8.3f ms  %7.1f TFLOP/s  (32 MFMAs of 16 cycles per iteration and wave, two waves per SIMD -> %.2f GHz)\n", noisy, iters, ms,
               flops / (ms * 1e-3) / 1e12, 2.0 * 32 * iters * 16 / (ms * 1e-3) / 1e9);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
