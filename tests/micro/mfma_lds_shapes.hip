// Micro-benchmark (not part of the product): the inner loop of the 3x3 convolution kernels -- fragment reads from LDS
// (ds_read_b128, conflict-free, a new address every step) under split-fp16 MFMAs -- in the two MFMA shapes, to decide
// whether converting conv3x3_dma_kernel / conv1ab_mfma_kernel from v_mfma_f32_32x32x16_f16 to v_mfma_f32_16x16x32_f16
// pays under load ("Measured for the next round").  Register-only loops (mfma_peak.hip) say +15-20 %.
//
//   A  32x32x16, one tap per step as today: 8 fragment reads (W hi/lo of two cout tiles, x hi/lo of two pixel tiles),
//      12 MFMAs of 32 cycles into 2 x 2 accumulators of 16 registers (64 couts x 64 pixels per wave).
//   B  16x16x32, a tap PAIR per step: K = 32 = [hi | lo] halves of the 16 channels of a chunk.  Per tap 4 + 4 reads
//      (Fw = [W_hi | W_lo] of four cout tiles, Fx = [x_lo | x_hi] of four pixel tiles) and 16 MFMAs Fw x Fx = both cross terms
//      (hi.lo + lo.hi) at once; then v_permlane32_swap turns (Fw(t), Fw(t+1)) into [W_hi(t+1) | W_hi(t)]-style pairs in place (and
//      the same for x), and 16 more MFMAs add hi.hi of both taps: 48 MFMAs of 16 cycles per tap pair = the same matrix-pipe
//      cycles as A's 24 x 32, from the same 16 fragment reads, + 32 swaps.
// Both double-buffer the fragments by hand with the pinned "MFMA first, one filler per MFMA" schedule of the product kernels; B's swaps
// are fillers too (27 of 32 under MFMAs whose operands they do not touch).
// 8 waves per workgroup, one workgroup per CU (LDS size), operands = pseudo-random halves in [-1, 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define PIN() __builtin_amdgcn_sched_barrier(0)
constexpr int LDS_PIECES = 6144;  // 96 KB of 16-byte pieces: one workgroup per CU, like the convolution kernels

__device__ __forceinline__ void fill_lds(u32x4* s) {
    unsigned r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = threadIdx.x; i < LDS_PIECES; i += 512) {
        f16x8 v;
        for (int j = 0; j < 8; ++j) {
            r = r * 1664525u + 1013904223u;
            v[j] = (_Float16)(((int)(r >> 9) & 0x3fff) - 8192) * (_Float16)(1.0f / 8192);
        }
        s[i] = __builtin_bit_cast(u32x4, v);
    }
    __syncthreads();
}

__global__ __launch_bounds__(512, 2) void loop_a(float* out, int steps) {
    __shared__ u32x4 s[LDS_PIECES];
    fill_lds(s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f16x8* sf = reinterpret_cast<const f16x8*>(s);
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8 fr[2][8];
    auto read1 = [&](int set, int step, int i) {
        const int base = ((step * 8 + wave) * 64) % (LDS_PIECES - 8 * 64);
        fr[set][i] = sf[base + i * 64 + lane];
    };
    // the schedule of conv3x3_dma_kernel: MFMA first, one fragment read of the next step per MFMA, pinned
    auto mul = [&](int S, int nstep) {
        int f = 0;
        auto fill = [&]() {
            if (f < 8) read1(1 - S, nstep, f);
            ++f;
            PIN();
        };
        // [ah0, bh0, bh1, ah1, bl0, bl1, al0, al1] as in conv3x3_dma_kernel
        PIN();
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][0], fr[S][1], acc[0][0], 0, 0, 0); fill();
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][0], fr[S][2], acc[0][1], 0, 0, 0); fill();
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][3], fr[S][1], acc[1][0], 0, 0, 0); fill();
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][3], fr[S][2], acc[1][1], 0, 0, 0); fill();
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][0], fr[S][4], acc[0][0], 0, 0, 0); fill();
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][0], fr[S][5], acc[0][1], 0, 0, 0); fill();
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][3], fr[S][4], acc[1][0], 0, 0, 0); fill();
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][3], fr[S][5], acc[1][1], 0, 0, 0); fill();
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][6], fr[S][1], acc[0][0], 0, 0, 0); fill();
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][6], fr[S][2], acc[0][1], 0, 0, 0); fill();
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][7], fr[S][1], acc[1][0], 0, 0, 0); fill();
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[S][7], fr[S][2], acc[1][1], 0, 0, 0); fill();
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) read1(0, 0, i);
    for (int st = 0; st < steps; st += 2) {
        mul(0, st + 1);
        mul(1, st + 2);
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 12345.f) out[threadIdx.x] = sum;
}

// a[lanes 32-63] <-> b[lanes 0-31] of register e of two fragments (their sources are LDS reads: no VALU-write hazard to cover)
__device__ __forceinline__ void swap1(f16x8& a, f16x8& b, int e) {
    u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[e]), "+v"(y[e]));
    a = __builtin_bit_cast(f16x8, x);
    b = __builtin_bit_cast(f16x8, y);
}

template <bool SWAPS>
__global__ __launch_bounds__(512, 2) void loop_b(float* out, int steps) {
    __shared__ u32x4 s[LDS_PIECES];
    fill_lds(s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f16x8* sf = reinterpret_cast<const f16x8*>(s);
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f16x8 fw[2][2][4], fx[2][2][4];  // [register set][tap of the pair][tile]
    auto read1 = [&](int set, int step, int r) {  // r = 0..15: tap r >> 3, W tiles then x tiles
        const int base = ((step * 8 + wave) * 128) % (LDS_PIECES - 16 * 64);
        const int t = r >> 3, i = r & 3;
        if (r & 4) fx[set][t][i] = sf[base + r * 64 + lane];
        else fw[set][t][i] = sf[base + r * 64 + lane];
    };
    // 48 MFMAs of 16 cycles per step; fillers: the 16 fragment reads of the next step (one per MFMA of the first 16), and the
    // swaps of tile i of both operands (8 permlane swaps) spread under the second tap's cross MFMAs
    auto mul = [&](int S, int nstep) {
        PIN();
        // cross terms of tap 0: the 16 fragment reads of the next step ride under them
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[S][0][i], fx[S][0][j], acc[i][j], 0, 0, 0);
                read1(1 - S, nstep, 4 * i + j);
                PIN();
            }
        // cross terms of tap 1, row-major; W row i - 1 is free once row i - 1 is done: its four swaps ride under row i, and column 0
        // of x is free after the first MFMA of the last row
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[S][1][i], fx[S][1][j], acc[i][j], 0, 0, 0);
                if (SWAPS && i > 0) swap1(fw[S][0][i - 1], fw[S][1][i - 1], j);
                if (SWAPS && i == 3 && j > 0) swap1(fx[S][1][0], fx[S][0][0], j - 1);
                PIN();
            }
        if (SWAPS) {
            swap1(fx[S][1][0], fx[S][0][0], 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) swap1(fw[S][0][3], fw[S][1][3], e);
            PIN();
        }
        // hi.hi of both taps, column-major: the swaps of x column j + 1 ride under column j
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[S][0][i], fx[S][0][j], acc[i][j], 0, 0, 0);
                if (SWAPS && j < 3) swap1(fx[S][1][j + 1], fx[S][0][j + 1], i);
                PIN();
            }
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) read1(0, 0, r);
    for (int st = 0; st < steps; st += 2) {
        mul(0, st + 1);
        mul(1, st + 2);
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.f) out[threadIdx.x] = sum;
}

template <class K>
static void run(const char* name, K kern, int steps, double flop_per_step_wave, double pipe_cycles_per_step) {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, steps);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, out, steps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = flop_per_step_wave * steps * 8 * grid;
    printf("%-34s steps %6d: %8.3f ms  %7.1f TFLOP/s executed  (matrix pipe %.0f cycles per step and wave, two waves per SIMD -> %.2f GHz if never idle)\n", name, steps,
           ms, flops / (ms * 1e-3) / 1e12, pipe_cycles_per_step, 2.0 * pipe_cycles_per_step * steps / (ms * 1e-3) / 1e9);
    hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run("A 32x32x16 (tap per step)", loop_a, 40000, 12 * 2.0 * 32 * 32 * 16, 12 * 32.0);
        run("B 16x16x32 (tap pair, swaps)", loop_b<true>, 20000, 48 * 2.0 * 16 * 16 * 32, 48 * 16.0);
        run("B 16x16x32 (tap pair, no swaps)", loop_b<false>, 20000, 48 * 2.0 * 16 * 16 * 32, 48 * 16.0);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
