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

// Ping-pong: waves 0-3 (one per SIMD) run a BURST of MFMAs while waves 4-7 (their SIMD partners) issue NREAD ds_read_b128, NWRITE ds_write_b128
// and NDMA LDS-DMA requests and wait for them; a barrier; roles swap; a barrier.  Cycles per SLOT (one burst of a SIMD): BURST x 32 = the pipe never idles.
template <int BURST, int NREAD, int NWRITE, int NDMA, int SETPRIO, int BIGLDS = 0, int DEP = 0, int XBAR = 0, int REAL = 0, int UNR = 1>
__global__ __launch_bounds__(512) void pingpong_kernel(unsigned long long* cycles, float* sink, int iters, const float* src) {
    __shared__ float lds[BIGLDS ? 36 * 1024 : 8 * 4096];  // BIGLDS: 144 KB, one workgroup per CU by LDS as in ffn_f16.hip
    for (int i = threadIdx.x; i < 8 * 4096; i += 512) lds[i] = i * 0.001f;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * ((threadIdx.x * 7 + j * 13) % 97)); b[j] = (_Float16)(0.02f * ((threadIdx.x * 11 + j * 5) % 89)); }
    const int wave = threadIdx.x >> 6, grp = __builtin_amdgcn_readfirstlane(wave >> 2);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t rsrc;
    {
        const unsigned long long ad = (unsigned long long)src;
        rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)ad);
        rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(ad >> 32) & 0xffff);
        rsrc[2] = 0x7fffffffu;
        rsrc[3] = 0x00020000u;
    }
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + wave * 4096);
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lbase + 8192);
    const unsigned laddr = lbase + (threadIdx.x & 63) * 16;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f ld[8];
    for (int i = 0; i < 8; ++i) ld[i] = v4f{0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    auto burst = [&]() {
        if (SETPRIO) asm volatile("s_setprio 2" ::: "memory");
#pragma unroll
        for (int g = 0; g < BURST; ++g) {
            if (DEP) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(ld[g & 7]), "v"(ld[(g + 3) & 7]));  // operands = what the memory phase read
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
        }
        if (SETPRIO) asm volatile("s_setprio 0" ::: "memory");
        if (XBAR) {  // the DMA requests of the real kernel: issued behind the burst, waited for one burst later (in flight across two barriers)
#pragma unroll
            for (int i = 0; i < NDMA; ++i)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"((unsigned)(((blockIdx.x & 15) * 8 + (i & 7)) * 1024 + (threadIdx.x & 63) * 16)), "s"(rsrc), "s"(m0v + i * 1024) : "memory");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        }
    };
    auto mem = [&](int it) {
#pragma unroll
        for (int i = 0; i < (XBAR ? 0 : NDMA); ++i)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"((unsigned)(((blockIdx.x & 15) * 8 + ((it + i) & 7)) * 1024 + (threadIdx.x & 63) * 16)), "s"(rsrc), "s"(m0v + i * 1024) : "memory");
        if (REAL) {  // the kernel's addressing: 64-byte rows, the lane's 16-byte piece XOR-swizzled, a different 16 KB region of a 128 KB ring every slot
            const unsigned i32 = threadIdx.x & 31, hh = (threadIdx.x >> 5) & 1;
            const unsigned ra = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (i32 * 4 + ((2 * hh) ^ ((i32 >> 2) & 3))) * 16 + (it & 7) * 16384;
#pragma unroll
            for (int i = 0; i < NREAD; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[i & 7]) : "v"(ra), "n"((i & 7) * 2048 + (i >> 3) * 16) : "memory");
        } else
#pragma unroll
        for (int i = 0; i < NREAD; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[i & 7]) : "v"(laddr), "n"((i & 7) * 1024) : "memory");
#pragma unroll
        for (int i = 0; i < ((REAL && (wave & 3) != (it & 3)) ? 0 : NWRITE); ++i) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(laddr), "v"(ld[i & 7]), "n"(8192 + 4096) : "memory");  // REAL: one writer of four per slot
        if (XBAR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    if (UNR == 1) {
        for (int it = 0; it < iters; ++it) {
            if (grp == 0) { burst(); __builtin_amdgcn_s_barrier(); mem(it); __builtin_amdgcn_s_barrier(); }
            else { mem(it); __builtin_amdgcn_s_barrier(); burst(); __builtin_amdgcn_s_barrier(); }
        }
    } else {  // straight-line code as in the kernel: UNR slot pairs per group, one copy per group, looped
        for (int it0 = 0; it0 < iters; it0 += UNR) {
            if (grp == 0) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) { burst(); __builtin_amdgcn_s_barrier(); mem(it0 + u); __builtin_amdgcn_s_barrier(); }
            } else {
#pragma unroll
                for (int u = 0; u < UNR; ++u) { mem(it0 + u); __builtin_amdgcn_s_barrier(); burst(); __builtin_amdgcn_s_barrier(); }
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 8; ++i) s += ld[i].x + ld[i].w;
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int BURST, int NREAD, int NWRITE, int NDMA, int SETPRIO = 0, int BIGLDS = 0, int DEP = 0, int XBAR = 0, int REAL = 0, int UNR = 1>
static double run_pp(int iters) {
    unsigned long long* d;
    float* sink;
    static float* src = nullptr;
    if (!src) { hipMalloc(&src, 16 * 8 * 1024 + 4096); hipMemset(src, 0, 16 * 8 * 1024 + 4096); }
    hipMalloc(&d, 256 * 8 * 8);
    hipMalloc(&sink, 4096);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((pingpong_kernel<BURST, NREAD, NWRITE, NDMA, SETPRIO, BIGLDS, DEP, XBAR, REAL, UNR>), dim3(256), dim3(512), 0, 0, d, sink, iters, src);
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    hipFree(sink);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / (2.0 * iters);  // cycles per slot
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
    printf("ping-pong (waves 0-3 multiply while waves 4-7 load, then swap; a barrier per slot): cycles per slot, ideal = 32 x burst\n");
    printf("  burst 12 / 24 / 48, partner idle:               %6.1f %6.1f %6.1f (ideal 384 768 1536)\n", run_pp<12, 0, 0, 0>(it), run_pp<24, 0, 0, 0>(it), run_pp<48, 0, 0, 0>(it));
    printf("  burst 12, partner 4 reads / 2 reads:            %6.1f %6.1f\n", run_pp<12, 4, 0, 0>(it), run_pp<12, 2, 0, 0>(it));
    printf("  burst 12, partner 8 reads:                      %6.1f (ideal 384)\n", run_pp<12, 8, 0, 0>(it));
    printf("  burst 12, partner 8 reads + 2 DMA:              %6.1f\n", run_pp<12, 8, 0, 2>(it));
    printf("  burst 12, partner 8 reads + 8 writes + 4 DMA:   %6.1f\n", run_pp<12, 8, 8, 4>(it));
    printf("  burst 12, same, s_setprio 2 around the burst:   %6.1f\n", run_pp<12, 8, 8, 4, 1>(it));
    printf("  burst 24, partner 12 reads + 5 DMA:             %6.1f (ideal 768)\n", run_pp<24, 12, 0, 5>(it));
    printf("  burst 24, partner 16 reads + 8 writes + 4 DMA:  %6.1f\n", run_pp<24, 16, 8, 4>(it));
    printf("  burst 24, partner 16 reads (+ 8 writes), no DMA:  %6.1f %6.1f; with a 144 KB LDS allocation: %6.1f %6.1f\n", run_pp<24, 16, 0, 0>(it), run_pp<24, 16, 8, 0>(it), run_pp<24, 16, 0, 0, 0, 1>(it), run_pp<24, 16, 8, 0, 0, 1>(it));
    printf("  burst 24, partner 16 reads: MFMA operands = the registers those reads load: %6.1f; + 4 DMA requests behind the burst, waited one burst later: %6.1f; both + 8 writes: %6.1f\n",
           run_pp<24, 16, 0, 0, 0, 1, 1, 0>(it), run_pp<24, 16, 0, 4, 0, 1, 1, 1>(it), run_pp<24, 16, 8, 4, 0, 1, 1, 1>(it));
    printf("  the same three, 100 x longer (~0.2 s each: sustained power): %6.1f %6.1f %6.1f; pure MFMA, two waves per SIMD, barrier every 24, 100 x longer: %5.2f cycles per MFMA\n",
           run_pp<24, 16, 0, 0, 0, 1, 1, 0>(100 * it), run_pp<24, 16, 0, 4, 0, 1, 1, 1>(100 * it), run_pp<24, 16, 8, 4, 0, 1, 1, 1>(100 * it), run<4, 8, 24, 1>(100 * it));
    printf("  ... + the kernel's LDS addressing (swizzled 64-byte rows, a different 16 KB of a 128 KB ring every slot) and one writer wave of four: %6.1f\n", run_pp<24, 16, 8, 4, 0, 1, 1, 1, 1>(it));
    printf("  burst 24, partner idle, as straight-line code of 16 slot pairs per group (the kernel's form) instead of a loop: %6.1f; with 16 reads: %6.1f\n", run_pp<24, 0, 0, 0, 0, 1, 0, 0, 0, 16>(it), run_pp<24, 16, 0, 0, 0, 1, 1, 0, 0, 16>(it));
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
