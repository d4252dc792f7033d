// Micro-benchmark (not part of the product): do the two waves of a SIMD share the matrix pipe without loss?  One workgroup per CU with
// WAVES waves (4 = one per SIMD, 8 = two per SIMD); every wave loops over { BURST x v_mfma_f32_32x32x16_f16 rotating over NACC
// accumulators; optional s_barrier } and reports shader cycles per MFMA *of the SIMD* (its own MFMAs x waves per SIMD).  Background:
// phase 3 of ffn_f16.hip (four accumulators per wave, 24 MFMAs per wave between barriers) runs at twice its MFMA time even with all its
// LDS traffic removed (profiles/r04_ai_ffn_noreads.txt).
//     hipcc --offload-arch=gfx950 -O3 -o mfma_pair mfma_pair.hip && ./mfma_pair
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int WAVES, int BURST, int BARRIER, int SETPRIO>
__global__ __launch_bounds__(WAVES * 64) void pair_kernel(unsigned long long* cycles, float* sink, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * ((threadIdx.x * 7 + j * 13) % 97)); b[j] = (_Float16)(0.02f * ((threadIdx.x * 11 + j * 5) % 89)); }
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (SETPRIO) asm volatile("s_setprio 1" ::: "memory");
#pragma unroll
        for (int g = 0; g < BURST; ++g) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g % NACC]) : "v"(a), "v"(b));
        if (SETPRIO) asm volatile("s_setprio 0" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int q = 0; q < NACC; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int WAVES, int BURST, int BARRIER, int SETPRIO = 0>
static double run(int iters) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * WAVES * 8);
    hipMalloc(&sink, 4096);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((pair_kernel<NACC, WAVES, BURST, BARRIER, SETPRIO>), dim3(256), dim3(WAVES * 64), 0, 0, d, sink, iters);
    std::vector<unsigned long long> h(256 * WAVES);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    hipFree(sink);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / ((double)BURST * iters * (WAVES / 4));  // cycles per MFMA of the SIMD
}

int main() {
    const int it = 2000;
    printf("shader cycles per MFMA of a SIMD (32 = the matrix pipe never idles); rows: accumulators per wave; columns: MFMAs per wave between barriers 12 24 48 | no barrier 24\n");
    printf("one wave per SIMD,  2 acc: %5.1f %5.1f %5.1f | %5.1f\n", run<2, 4, 12, 1>(it), run<2, 4, 24, 1>(it), run<2, 4, 48, 1>(it), run<2, 4, 24, 0>(it));
    printf("one wave per SIMD,  4 acc: %5.1f %5.1f %5.1f | %5.1f\n", run<4, 4, 12, 1>(it), run<4, 4, 24, 1>(it), run<4, 4, 48, 1>(it), run<4, 4, 24, 0>(it));
    printf("one wave per SIMD,  8 acc: %5.1f %5.1f %5.1f | %5.1f\n", run<8, 4, 12, 1>(it), run<8, 4, 24, 1>(it), run<8, 4, 48, 1>(it), run<8, 4, 24, 0>(it));
    printf("two waves per SIMD, 2 acc: %5.1f %5.1f %5.1f | %5.1f\n", run<2, 8, 12, 1>(it), run<2, 8, 24, 1>(it), run<2, 8, 48, 1>(it), run<2, 8, 24, 0>(it));
    printf("two waves per SIMD, 4 acc: %5.1f %5.1f %5.1f | %5.1f\n", run<4, 8, 12, 1>(it), run<4, 8, 24, 1>(it), run<4, 8, 48, 1>(it), run<4, 8, 24, 0>(it));
    printf("two waves per SIMD, 8 acc: %5.1f %5.1f %5.1f | %5.1f\n", run<8, 8, 12, 1>(it), run<8, 8, 24, 1>(it), run<8, 8, 48, 1>(it), run<8, 8, 24, 0>(it));
    printf("two waves per SIMD, 4 acc, s_setprio 1 around the burst: %5.1f %5.1f %5.1f | %5.1f\n", run<4, 8, 12, 1, 1>(it), run<4, 8, 24, 1, 1>(it), run<4, 8, 48, 1, 1>(it), run<4, 8, 24, 0, 1>(it));
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
