// SuperPoint detector head in ONE kernel (MG_PREC_F16X3): convPb (1x1, 256 -> 65) + soft-max over the 65 logits + drop the dustbin
// + depth-to-space into the (B, 8 hc, 8 wc) score map (superpoint.py:224-230; superpoint_open.py:134-140 with its BatchNorm affine).
//
// As three launches the 65 logits were computed by the generic 256-feature GEMM tile (a quarter of its MFMAs useful), written
// (118 MB at 64 VGA images) and read back by a one-wave-per-cell soft-max kernel.  Here:
//   * the whole weight matrix (65 rows padded to 96, K = 256, SH8: 96 KB) is loaded into LDS ONCE per persistent workgroup;
//   * a wave owns 32 cells: their SH8 rows arrive by LDS-DMA in a private three-slot ring (two 1-KB pieces per 16-deep stage),
//     so the main loop needs NO barrier -- a wave waits only for its own DMA;
//   * 9 MFMAs per stage (3 feature tiles x {hi.hi, hi.lo, lo.hi}); afterwards lane (cell, h) holds, per 32-feature tile, features
//     {(r&3) + 8 (r>>2) + 4h}: register group g of tile ft is row 4 ft + g of the cell's 8x8 block, columns 4h..4h+3 -- one
//     16-byte store per group, and the 32 cells x 2 halves of a wave write 1 KB of one score-map row;
//   * soft-max statistics: in-lane over 48 registers, one shuffle across the lane halves.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int HT = 512;      // 8 waves
constexpr int HCELLS = 256;  // cells per workgroup pass (32 per wave)
constexpr int HF = 96;       // 65 logits padded to three MFMA tiles
constexpr int HK = 256, HNK = HK / 16;
constexpr int HRS = 3;       // ring slots per wave (2 KB each: 32 cells x 64 B)

__device__ __forceinline__ int swz(int row, int q) { return row * 4 + (q ^ ((row >> 2) & 3)); }

__global__ __launch_bounds__(HT, 2) void heads_f16x3_kernel(const HeadsF16Args p) {
    // [weights: 16 k-steps x 96 rows x 64 B = 96 KB][8 waves x 3 slots x 2 KB = 48 KB][bias | scale | shift : 3 x 96 floats]
    __shared__ u32x4 s_mem[HNK * HF * 4 + 8 * HRS * 32 * 4 + 3 * HF / 4];
    u32x4* const s_w = s_mem;
    u32x4* const s_tok = s_mem + HNK * HF * 4;
    float* const s_par = reinterpret_cast<float*>(s_tok + 8 * HRS * 32 * 4);

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int lane_hi = swz(i32, 2 * h), lane_lo = swz(i32, 2 * h + 1);
    const long long ncell = (long long)p.B * p.hc * p.wc;
    const u32x4* Wg = reinterpret_cast<const u32x4*>(p.W);
    const u32x4* A = reinterpret_cast<const u32x4*>(p.in);

    // weights -> LDS once: stage kt = rows 0..95 of the tiled array [K/16][Npad][64 B]; 96 rows x 4 pieces = 6 wave-pieces per stage
    for (int piece = wu; piece < HNK * 6; piece += 8) {
        const int kt = piece / 6, sl = (piece % 6) * 64 + lane, row = sl >> 2, q = (sl & 3) ^ ((row >> 2) & 3);
        __builtin_amdgcn_global_load_lds(Wg + ((long long)kt * p.Npad + row) * 4 + q, s_w + kt * HF * 4 + (piece % 6) * 64, 16, 0, 0);
    }
    for (int i = t; i < HF; i += HT) {
        const bool ok = i < 65;
        s_par[i] = ok && p.bias ? p.bias[i] : 0.f;
        s_par[HF + i] = ok && p.post_scale ? p.post_scale[i] : 1.f;
        s_par[2 * HF + i] = ok && p.post_shift ? p.post_shift[i] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    u32x4* const ring = s_tok + wu * HRS * 32 * 4;  // this wave's private ring
    const int ntile = (int)((ncell + HCELLS - 1) / HCELLS);
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long c0 = (long long)tile * HCELLS + wu * 32;  // first cell of this wave
        // DMA sources of this lane: piece j (0/1) of a stage = rows 16 j .. 16 j + 15 of the wave's 32 cells
        long long src_off[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int sl = j * 64 + lane, row = sl >> 2, q = (sl & 3) ^ ((row >> 2) & 3);
            src_off[j] = min(c0 + row, ncell - 1) * 64 + q;  // 256 elements x 4 B = 64 pieces per row
        }
        auto dma = [&](int kt) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds(A + src_off[j] + kt * 4, ring + (kt % HRS) * 32 * 4 + j * 64, 16, 0, 0);
        };
        f32x16 acc[3];
#pragma unroll
        for (int ft = 0; ft < 3; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ft][r] = 0.f;
#pragma unroll
        for (int s0 = 0; s0 < HRS; ++s0) dma(s0);
#pragma unroll
        for (int kt = 0; kt < HNK; ++kt) {
            // my two pieces of stage kt have landed: all but the (up to two) younger stages
            if (kt + 2 <= HNK - 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (kt + 1 <= HNK - 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const f16x8* st = reinterpret_cast<const f16x8*>(ring + (kt % HRS) * 32 * 4);
            const f16x8* sw = reinterpret_cast<const f16x8*>(s_w + kt * HF * 4);
            const f16x8 bh = st[lane_hi], bl = st[lane_lo];
            f16x8 ah[3], al[3];
#pragma unroll
            for (int ft = 0; ft < 3; ++ft) {
                ah[ft] = sw[lane_hi + ft * 32 * 4];
                al[ft] = sw[lane_lo + ft * 32 * 4];
            }
#pragma unroll
            for (int ft = 0; ft < 3; ++ft) acc[ft] = mfma16(ah[ft], bh, acc[ft]);
#pragma unroll
            for (int ft = 0; ft < 3; ++ft) acc[ft] = mfma16(ah[ft], bl, acc[ft]);
#pragma unroll
            for (int ft = 0; ft < 3; ++ft) acc[ft] = mfma16(al[ft], bh, acc[ft]);
            // the fragments of stage kt are in registers (the MFMAs above consumed them): its slot may be refilled
            if (kt + HRS < HNK) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma(kt + HRS);
            }
        }
        // epilogue: + bias (and the open variant's BatchNorm affine), soft-max over features 0..64, depth-to-space
        float m = -__builtin_inff();
#pragma unroll
        for (int ft = 0; ft < 3; ++ft)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int f0 = ft * 32 + 8 * g + 4 * h;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(s_par + f0);
                const f32x4 sv = *reinterpret_cast<const f32x4*>(s_par + HF + f0);
                const f32x4 tv = *reinterpret_cast<const f32x4*>(s_par + 2 * HF + f0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fma_winv(acc[ft][4 * g + e], p.winv, bv[e]);
                    v = __fadd_rn(__fmul_rn(v, sv[e]), tv[e]);
                    if (f0 + e > 64) v = -__builtin_inff();  // padding features
                    acc[ft][4 * g + e] = v;
                    m = fmaxf(m, v);
                }
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.f;
#pragma unroll
        for (int ft = 0; ft < 3; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(acc[ft][r] - m);
                acc[ft][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        const long long cell = c0 + i32;
        if (cell < ncell) {
            const int b = (int)(cell / (p.hc * p.wc));
            const int rem = (int)(cell - (long long)b * p.hc * p.wc);
            const int cy = rem / p.wc, cx = rem - cy * p.wc;
            float* dst = p.heat + ((long long)b * p.hc * 8 + cy * 8) * (p.wc * 8) + cx * 8 + 4 * h;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 o = {acc[ft][4 * g] / sum, acc[ft][4 * g + 1] / sum, acc[ft][4 * g + 2] / sum, acc[ft][4 * g + 3] / sum};
                    *reinterpret_cast<f32x4*>(dst + (long long)(4 * ft + g) * (p.wc * 8)) = o;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my ring reads are done before the next tile's DMA lands in it
    }
}

}  // namespace

void launch_heads_f16x3(const HeadsF16Args& a, hipStream_t s) {
    const long long ncell = (long long)a.B * a.hc * a.wc;
    const int ntile = (int)((ncell + HCELLS - 1) / HCELLS);
    hipLaunchKernelGGL(heads_f16x3_kernel, dim3(ntile < 256 ? ntile : 256), dim3(HT), 0, s, a);
}
