// SuperPoint 3x3 convolutions on the f16 MFMA with split-fp16 operands (common.h: "f16x3").
//
// Same implicit-GEMM structure as conv.hip, re-cut for the 16x faster matrix pipe:
//   * activations and weights are stored SH8 (hi/lo half groups), so every fragment is one 16-byte
//     LDS read and no conversion happens in the main loop;
//   * Cin advances in chunks of 16 (one 32x32x16 MFMA k-step); the halo tile is [pixel][64 B] with the
//     pixel pitch chosen == 8 (mod 16) and the four 16-byte pieces of a pixel XOR-swizzled by
//     (pixel>>2)&3 -- every ds_read_b128 of the pixel operand and of the [tap][cout][64 B] weight
//     slice is bank-conflict-free without padding;
//   * weights are the A operand (rows = output channels) and pixels the B operand (lanes), so after
//     bias/ReLU/affine a 2x2 max-pool is two quad DPP max-ops, one v_permlane32_swap per register pair
//     leaves 16 consecutive channels of one pixel in a lane, and the lane stores 64 contiguous bytes
//     (SH8 for the next MFMA consumer, or plain fp32 NHWC).
// A workgroup is 8 waves (16x32 or 32x16 pixels x 64 channels), one per CU with two waves per SIMD:
// one 83 KB LDS stage, the next chunk prefetched into registers while the current one is multiplied.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int CT = 512;  // threads per workgroup of the 8-wave arrangements
#define PIN() __builtin_amdgcn_sched_barrier(0)
constexpr int CONV_DMA_MAXC = 512;  // conv3x3_dma_kernel keeps the bias of the whole layer in LDS

// Phase ablations used for the numbers in DESIGN.md: compile this file with -DCONV_ABL=<n> into a second library and
// run bench.py with MG_LIB_PATH pointing at it (same box, same clocks).  7 = epilogue without its stores (results are
// wrong by design).  Earlier variants measured conv1ab's phases: conv1a for the first chunk only, no weight re-staging,
// no MFMA phase.

// Work-item order (item_of / tile_of in the kernels): workgroups are dealt round-robin over the 8 XCDs, each with a
// private L2; ids are re-numbered so that one XCD works through a CONTIGUOUS run of (image, tile, cout slice) items with the slice index fastest: the cout slices of a
// pixel tile (which read the same input tile) and spatially adjacent tiles (which share halo rows / columns) are
// then in flight on the same L2 at the same time instead of being fetched from HBM once per XCD.

// One 16-channel chunk of a 3x3 convolution for a wave's 64 couts x 64 pixels: 9 taps x (8 fragment reads, 12
// MFMAs).  Software-pipelined by hand: the 8 ds_read_b128 of tap k+1 are issued BEFORE the MFMAs of tap k (two
// fragment sets), and the scheduler is fenced so it cannot sink them back next to their use -- the compiler's own
// order left 100-200 cycles of LDS latency exposed in front of every tap (measured: multiply phase at 47 % of
// the MFMA rate).  Per accumulator the products keep the order hi.hi, hi.lo, lo.hi; the four accumulators are
// interleaved so consecutive MFMAs are independent.
struct TapFrag {
    f16x8 bh0, bh1, bl0, bl1, ah0, ah1, al0, al1;
};
template <int PW>
__device__ __forceinline__ void load_tap(TapFrag& f, const f16x8* si, const f16x8* sw, int P0, int h, int a_hi, int a_lo, int tap) {
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int Pa = P0 + ky * PW + kx, Pb = Pa + 2 * PW;
    const int sa = (Pa >> 2) & 3, sb = (Pb >> 2) & 3;
    f.bh0 = si[Pa * 4 + ((2 * h) ^ sa)];
    f.bl0 = si[Pa * 4 + ((2 * h + 1) ^ sa)];
    f.bh1 = si[Pb * 4 + ((2 * h) ^ sb)];
    f.bl1 = si[Pb * 4 + ((2 * h + 1) ^ sb)];
    const int ro = tap * 64 * 4;
    f.ah0 = sw[ro + a_hi];
    f.al0 = sw[ro + a_lo];
    f.ah1 = sw[ro + 32 * 4 + a_hi];
    f.al1 = sw[ro + 32 * 4 + a_lo];
}
__device__ __forceinline__ void mma_tap(const TapFrag& f, f32x16 (&acc)[2][2]) {
    acc[0][0] = mfma16(f.ah0, f.bh0, acc[0][0]);
    acc[0][1] = mfma16(f.ah0, f.bh1, acc[0][1]);
    acc[1][0] = mfma16(f.ah1, f.bh0, acc[1][0]);
    acc[1][1] = mfma16(f.ah1, f.bh1, acc[1][1]);
    acc[0][0] = mfma16(f.ah0, f.bl0, acc[0][0]);
    acc[0][1] = mfma16(f.ah0, f.bl1, acc[0][1]);
    acc[1][0] = mfma16(f.ah1, f.bl0, acc[1][0]);
    acc[1][1] = mfma16(f.ah1, f.bl1, acc[1][1]);
    acc[0][0] = mfma16(f.al0, f.bh0, acc[0][0]);
    acc[0][1] = mfma16(f.al0, f.bh1, acc[0][1]);
    acc[1][0] = mfma16(f.al1, f.bh0, acc[1][0]);
    acc[1][1] = mfma16(f.al1, f.bh1, acc[1][1]);
}
template <int PW>
__device__ __forceinline__ void multiply_chunk_pipelined(const f16x8* si, const f16x8* sw, int P0, int h, int a_hi, int a_lo, f32x16 (&acc)[2][2]) {
    TapFrag f0, f1;
    load_tap<PW>(f0, si, sw, P0, h, a_hi, a_lo, 0);
#pragma unroll
    for (int tap = 0; tap < 9; tap += 2) {
        if (tap + 1 < 9) load_tap<PW>(f1, si, sw, P0, h, a_hi, a_lo, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_tap(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < 9) {
            if (tap + 2 < 9) load_tap<PW>(f0, si, sw, P0, h, a_hi, a_lo, tap + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(f1, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// (v[i], v[i+8]) half-wave exchanges of one accumulator tile in one asm block (one hazard gap instead of eight)
__device__ __forceinline__ void half_swap8(float (&v)[16]) {
    asm volatile(
        "v_nop\n\tv_nop\n\t"
        "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\t"
        "v_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]),
          "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
}

// (non-temporal stores were tried here: no measurable difference)
#define STORE_OUT(ptr, val) (*(ptr) = (val))
// Hot epilogue (every layer of the MagicLeap layout): ReLU on accumulators that were initialised with the bias,
// optional 2x2 max-pool inside the lane quad, SH8 output.  32-bit offsets inside the image, no per-channel loads.
template <bool POOL>
__device__ __forceinline__ void epilogue_sh8(f32x16 (&acc)[2][2], u32x4* __restrict__ out_img, int H, int W, int Cout, int yw, int xw, int n0,
                                             int lane) {
    const int i32 = lane & 31, h = lane >> 5, ql = i32 & 3, blk = i32 >> 2;
    const int c4 = Cout >> 2;
#if defined(CONV_ABL) && CONV_ABL == 7  // timing experiment 7: the whole epilogue runs but no store is executed
    const int Ho = Cout > 100000 ? 1 : 0, Wo = POOL ? W >> 1 : W;
#else
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
#endif
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int ch0 = n0 + mt * 32 + 16 * h;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = fmaxf(acc[mt][nt][r], 0.f);
                if (POOL) {
                    x = fmaxf(x, dpp_quad_xor1(x));
                    x = fmaxf(x, dpp_quad_xor2(x));
                }
                v[r] = x;
            }
            half_swap8(v);
            // (lower | upper half-wave): v[0..3],v[8..11] = channels +0..7 | +16..23 ; v[4..7],v[12..15] = +8..15 | +24..31
            f16x8 hi0, lo0, hi1, lo1;
            split8(f32x8{v[0], v[1], v[2], v[3], v[8], v[9], v[10], v[11]}, hi0, lo0);
            split8(f32x8{v[4], v[5], v[6], v[7], v[12], v[13], v[14], v[15]}, hi1, lo1);
            u32x4 q0 = __builtin_bit_cast(u32x4, hi0), q1 = __builtin_bit_cast(u32x4, lo0);
            u32x4 q2 = __builtin_bit_cast(u32x4, hi1), q3 = __builtin_bit_cast(u32x4, lo1);
            if (POOL) {
                const int oy = (yw >> 1) + nt, ox = (xw >> 1) + blk;
                if (oy < Ho && ox < Wo && ch0 < Cout) STORE_OUT(&out_img[(oy * Wo + ox) * c4 + (ch0 >> 2) + ql], quad_pick(q0, q1, q2, q3, ql));
            } else {
                quad_transpose(q0, q1, q2, q3, ql);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int oy = yw + 2 * nt + (sidx >> 1), ox = xw + 2 * blk + (sidx & 1);
                    if (oy < Ho && ox < Wo && ch0 < Cout)
                        STORE_OUT(&out_img[(oy * Wo + ox) * c4 + (ch0 >> 2) + ql], sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3);
                }
            }
        }
    }
}


// Epilogue of one 64-cout x (4 x 16)-pixel wave tile.  `sb` = the 64 biases of this cout slice in LDS (nullptr when
// the accumulators were initialised with the bias).  Hot path: ReLU + SH8 (+ pool); generic path below it.
template <bool GENERIC>
__device__ __forceinline__ void conv_epilogue(const ConvF16Args& p, f32x16 (&acc)[2][2], const float* sb, int b, int yw, int xw, int n0, int lane) {
    const int i32 = lane & 31, h = lane >> 5;
    if (sb) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + mt * 32 + 8 * j + 4 * h);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt][4 * j] = fma_winv(acc[mt][nt][4 * j], p.winv, bv[0]); acc[mt][nt][4 * j + 1] = fma_winv(acc[mt][nt][4 * j + 1], p.winv, bv[1]);
                    acc[mt][nt][4 * j + 2] = fma_winv(acc[mt][nt][4 * j + 2], p.winv, bv[2]); acc[mt][nt][4 * j + 3] = fma_winv(acc[mt][nt][4 * j + 3], p.winv, bv[3]);
                }
            }
    }
    if (!GENERIC) {  // the launcher picks this instantiation when relu && !out_f32 && !post_scale
        if (p.pool) {
            u32x4* oimg = reinterpret_cast<u32x4*>(p.out) + (long long)b * (p.H >> 1) * (p.W >> 1) * (p.Cout >> 2);
            epilogue_sh8<true>(acc, oimg, p.H, p.W, p.Cout, yw, xw, n0, lane);
        } else {
            u32x4* oimg = reinterpret_cast<u32x4*>(p.out) + (long long)b * p.H * p.W * (p.Cout >> 2);
            epilogue_sh8<false>(acc, oimg, p.H, p.W, p.Cout, yw, xw, n0, lane);
        }
        return;
    }
    // generic path: BatchNorm affine after the ReLU (open variant), fp32 output, linear layers
    const int Ho = p.pool ? p.H / 2 : p.H, Wo = p.pool ? p.W / 2 : p.W;
    const int blk = i32 >> 2;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        // register r of this lane is output channel cb + 8*(r>>2) + 4*h + (r&3)
        const int cb = n0 + mt * 32;
        float ps[16], pt[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = cb + 8 * (r >> 2) + 4 * h + (r & 3);
            const bool ok = ch < p.Cout;
            ps[r] = (p.post_scale && ok) ? p.post_scale[ch] : 1.f;
            pt[r] = (p.post_shift && ok) ? p.post_shift[ch] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[mt][nt][r];
                if (p.relu) x = fmaxf(x, 0.f);
                if (p.post_scale) x = __fadd_rn(__fmul_rn(x, ps[r]), pt[r]);
                if (p.pool) {
                    x = fmaxf(x, dpp_quad_xor1(x));
                    x = fmaxf(x, dpp_quad_xor2(x));
                }
                v[r] = x;
            }
            // gather 16 consecutive channels per lane: lanes 0-31 get cb+0..15, lanes 32-63 get cb+16..31
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                half_swap(v[e], v[8 + e]);
                half_swap(v[4 + e], v[12 + e]);
            }
            // now (lower | upper half): v[0..3],v[8..11] = channels +0..7 | +16..23 ; v[4..7],v[12..15] = +8..15 | +24..31
            const f32x8 g0 = {v[0], v[1], v[2], v[3], v[8], v[9], v[10], v[11]};
            const f32x8 g1 = {v[4], v[5], v[6], v[7], v[12], v[13], v[14], v[15]};
            // 64 output bytes of this lane as four 16-byte pieces (SH8: hi|lo|hi|lo, fp32: 4 x float4)
            u32x4 q0, q1, q2, q3;
            if (p.out_f32) {
                q0 = __builtin_bit_cast(u32x4, f32x4{g0[0], g0[1], g0[2], g0[3]});
                q1 = __builtin_bit_cast(u32x4, f32x4{g0[4], g0[5], g0[6], g0[7]});
                q2 = __builtin_bit_cast(u32x4, f32x4{g1[0], g1[1], g1[2], g1[3]});
                q3 = __builtin_bit_cast(u32x4, f32x4{g1[4], g1[5], g1[6], g1[7]});
            } else {
                f16x8 hi0, lo0, hi1, lo1;
                split8(g0, hi0, lo0);
                split8(g1, hi1, lo1);
                q0 = __builtin_bit_cast(u32x4, hi0); q1 = __builtin_bit_cast(u32x4, lo0);
                q2 = __builtin_bit_cast(u32x4, hi1); q3 = __builtin_bit_cast(u32x4, lo1);
            }
            const int ch0 = cb + 16 * h;
            const int ql = i32 & 3;  // lane within its 2x2 pixel quad
            u32x4* obase = reinterpret_cast<u32x4*>(p.out);
            if (p.pool) {
                // the four lanes of a quad hold the same pooled values: each writes one 16-byte piece
                const int oy = (yw + 2 * nt) >> 1, ox = ((xw) >> 1) + blk;
                if (oy < Ho && ox < Wo && ch0 < p.Cout) {
                    const long long pix = ((long long)b * Ho + oy) * Wo + ox;
                    obase[(pix * p.Cout + ch0) / 4 + ql] = quad_pick(q0, q1, q2, q3, ql);
                }
            } else {
                // quad transpose: store s writes pixel s of every quad, 4 lanes x 16 bytes
                quad_transpose(q0, q1, q2, q3, ql);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int oy = yw + 2 * nt + (sidx >> 1), ox = xw + 2 * blk + (sidx & 1);
                    if (oy < Ho && ox < Wo && ch0 < p.Cout) {
                        const long long pix = ((long long)b * Ho + oy) * Wo + ox;
                        obase[(pix * p.Cout + ch0) / 4 + ql] = sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3;
                    }
                }
            }
        }
    }
}

// Persistent: a workgroup walks a strided list of (image, tile, cout slice) items and treats their 16-channel
// chunks as ONE stream of stages -- the inputs of stage s+2 and the weights of stage s+1 are requested while stage
// s is multiplied, across item boundaries.  The epilogue's stores are fire-and-forget: the next item's first two
// chunks are already in flight, nothing waits for a workgroup to retire (measured: the stores of a one-tile
// workgroup cost 20-25 % of the kernel, the time its slot sat idle until they were acknowledged).
template <int WY, int WX, int KYU, bool GENERIC>
__global__ __launch_bounds__(64 * WY * WX, 2) void conv3x3_f16x3_kernel(const ConvF16Args p) {
    constexpr int CT = 64 * WY * WX;
    constexpr int TH = 4 * WY, TW = 16 * WX;
    constexpr int PW = ((TW + 2 + 7) / 16) * 16 + 8;  // pixel pitch == 8 (mod 16), >= TW + 2
    constexpr int NPIX = (TH + 2) * (TW + 2);
    constexpr int NIN = (NPIX * 4 + CT - 1) / CT;
    constexpr int NWT = (9 * 64 * 4 + CT - 1) / CT;
    __shared__ u32x4 s_in[(TH + 2) * PW * 4];
    __shared__ u32x4 s_w[9 * 64 * 4];
    __shared__ float s_bias[2][64];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y, nslices = (p.Cout + 63) / 64;
    const int total = tiles * nslices * p.B;
    const int cin4 = p.Cin >> 2;  // u32x4 per pixel
    const u32x4* wt = reinterpret_cast<const u32x4*>(p.Wt);

    struct Item { int b, y0, x0, n0; };
    auto item_of = [&](int id) {
        // XCD-contiguous order (xcd_item): ids 8 apart are neighbours on the same L2, cout slice fastest
        int logical = id;
        if ((total & 7) == 0) logical = (id & 7) * (total >> 3) + (id >> 3);
        const int slice = logical % nslices, r = logical / nslices;
        const int tile = r % tiles;
        return Item{r / tiles, (tile / tiles_x) * TH, (tile % tiles_x) * TW, slice * 64};
    };

    // Staging registers: the next stage (possibly the next item's first) is in flight while this one is multiplied
    u32x4 rin[NIN], rwt[NWT];
    const int nchunks = p.Cin >> 4;
    // staging geometry is recomputed from the thread index (a few integer ops) instead of living in registers
    // through the epilogue: element idx = t + CT*i -> halo pixel (py, px), 16-byte piece q
    // uniform 64-bit base (SGPRs) + unsigned 32-bit lane offset: one address VGPR per load
    auto load_in = [&](u32x4 (&r)[NIN], const Item& it, int c) {
        const u32x4* inb = reinterpret_cast<const u32x4*>(p.in) + (((long long)it.b * p.H + (it.y0 - 1)) * p.W + (it.x0 - 1)) * cin4 + c * 4;
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int idx = t + CT * i, pix = idx >> 2, q = idx & 3;
            const int py = pix / (TW + 2), px = pix - py * (TW + 2);
            const int gy = it.y0 + py - 1, gx = it.x0 + px - 1;
            const bool ok = idx < NPIX * 4 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const unsigned off = (unsigned)((py * p.W + px) * cin4 + q);
            r[i] = ok ? inb[off] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto load_w = [&](int n0, int c) {
        const u32x4* wb = wt + ((long long)c * 9 * p.CoutPad + n0) * 4;
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int idx = t + CT * i, row = idx >> 2, q = idx & 3, tap = row >> 6, n = row & 63;
            const unsigned off = (unsigned)((tap * p.CoutPad + n) * 4 + q);
            rwt[i] = (idx < 9 * 64 * 4) ? wb[off] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_chunk = [&](const u32x4 (&r)[NIN]) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int idx = t + CT * i, pix = idx >> 2, q = idx & 3;
            const int py = pix / (TW + 2), px = pix - py * (TW + 2);
            const int P = py * PW + px;
            if (idx < NPIX * 4) s_in[P * 4 + (q ^ ((P >> 2) & 3))] = r[i];
        }
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int idx = t + CT * i, row = idx >> 2, q = idx & 3;
            if (idx < 9 * 64 * 4) s_w[row * 4 + (q ^ ((row >> 2) & 3))] = rwt[i];
        }
    };
    auto load_bias = [&](int n0) { return (t < 64 && n0 + t < p.Cout) ? p.bias[n0 + t] : 0.f; };

    const int wy0 = (wave / WX) * 4, wx0 = (wave % WX) * 16;
    const int i32 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];  // [cout tile][pixel tile]
    // pixel operand: lane n -> pixel (dy,dx) = ((n>>1)&1, n&1) of 2x2 block n>>2 (blocks along x)
    const int P0 = (wy0 + ((i32 >> 1) & 1)) * PW + wx0 + 2 * (i32 >> 2) + (i32 & 1);
    // weight operand: row = tap*64 + mt*32 + i32; swizzle term depends on i32 only
    const int wsw = (i32 >> 2) & 3;
    const int a_hi = i32 * 4 + ((2 * h) ^ wsw), a_lo = i32 * 4 + ((2 * h + 1) ^ wsw);

    // KYU == 0 selects the hand-pipelined tap loop (32 more VGPRs for the second fragment set: the 8-wave tiles
    // have room for it, the 4-wave tile with its larger per-thread staging share does not)
    auto multiply = [&]() {
        const f16x8* si = reinterpret_cast<const f16x8*>(s_in);
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w);
        if constexpr (KYU == 0) {
            multiply_chunk_pipelined<PW>(si, sw, P0, h, a_hi, a_lo, acc);
        } else {
#pragma unroll KYU
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    TapFrag f;
                    load_tap<PW>(f, si, sw, P0, h, a_hi, a_lo, ky * 3 + kx);
                    mma_tap(f, acc);
                }
            }
        }
    };

    int id = blockIdx.x;
    if (id >= total) return;
    Item cur = item_of(id);
    float bias_cur = load_bias(cur.n0), bias_nxt = 0.f;
    load_w(cur.n0, 0);
    load_in(rin, cur, 0);
    for (int k = 0;; ++k) {
        const bool more = id + (int)gridDim.x < total;
        const Item nxt = item_of(more ? id + (int)gridDim.x : id);
        if (t < 64) s_bias[k & 1][t] = bias_cur;  // read by this item's epilogue (barriers in between)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();  // everyone is done reading the previous stage
            store_chunk(rin);
            __syncthreads();
            if (c + 1 < nchunks) {
                load_w(cur.n0, c + 1);
                load_in(rin, cur, c + 1);
            } else if (more) {  // first stage of the next item: in flight under this multiply and the epilogue
                bias_nxt = load_bias(nxt.n0);
                load_w(nxt.n0, 0);
                load_in(rin, nxt, 0);
            }
            multiply();
        }
        conv_epilogue<GENERIC>(p, acc, s_bias[k & 1], cur.b, cur.y0 + wy0, cur.x0 + wx0, cur.n0, lane);
        if (!more) break;
        cur = nxt;
        bias_cur = bias_nxt;
        id += gridDim.x;
    }
}

// ---- round 2: the same convolution with LDS-DMA staging -----------------------------------------------------------
// The kernel above stops all eight waves twice per 16-channel chunk (barrier, ds_write of the staged registers,
// barrier).  Here nothing passes through registers: a STAGE is one filter row of one chunk (3 taps x 12 MFMAs per wave),
//   * the halo tile of a chunk (46 / 52 KB) is double-buffered and fetched by global_load_lds one chunk ahead
//     (all of it is issued during the first stage of the previous chunk);
//   * the 3 x 64 x 64 B weight rows of a stage (12 KB) go through a four-slot ring, fetched three stages ahead;
//   * one raw s_barrier per stage, nothing between it and the first MFMA (its fragments were read under the previous
//     tap), one filler per MFMA: the eight fragment reads of the next tap into the other register set, then the DMA
//     pieces this wave owes the ring.
// vmcnt retires in order, so "everything issued up to the stage before last has landed" is one counted s_waitcnt; to
// keep that count exact every DMA and every output store is executed unconditionally, through buffer instructions: a
// halo lane outside the image gets an offset past num_records and the bounds check returns zeros into LDS, an output
// lane outside the image is dropped the same way (no exec-mask branches that could skip a VMEM instruction).  Every
// wave issues the same number of pieces: halo pieces of LP <= 64 lanes, weight pieces of 48 lanes (two per wave and stage: with
// the twelve full pieces on waves 0-3 those four arrived late at every barrier), so the counts are compile-time constants.
// lane id recomputed on the spot (volatile: never hoisted or kept live): for the once-per-tile code of a kernel at the register limit
__device__ __forceinline__ int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void wait_vm_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins exist in the device pass only (the host pass needs just the kernel stubs)
struct BufStore {  // out-of-range lanes get an offset past num_records: the store is dropped, the instruction still issues
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ void operator()(int idx, bool ok, u32x4 v) const {
        __builtin_amdgcn_raw_buffer_store_b128(v, r, ok ? idx * 16 : (int)0x80000000, 0, 0);
    }
};

// ReLU (+ 2x2 max-pool) + SH8 stores of one wave tile through a store functor (idx = 16-byte piece inside the image)
template <bool POOL, class Store>
__device__ __forceinline__ void epilogue_sh8_st(f32x16 (&acc)[2][2], const Store& st, int H, int W, int Cout, int yw, int xw, int n0, int lane) {
    const int i32 = lane & 31, h = lane >> 5, ql = i32 & 3, blk = i32 >> 2;
    const int c4 = Cout >> 2;
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int ch0 = n0 + mt * 32 + 16 * h;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = fmaxf(acc[mt][nt][r], 0.f);
                if (POOL) {
                    x = fmaxf(x, dpp_quad_xor1(x));
                    x = fmaxf(x, dpp_quad_xor2(x));
                }
                v[r] = x;
            }
            half_swap8(v);
            f16x8 hi0, lo0, hi1, lo1;
            split8(f32x8{v[0], v[1], v[2], v[3], v[8], v[9], v[10], v[11]}, hi0, lo0);
            split8(f32x8{v[4], v[5], v[6], v[7], v[12], v[13], v[14], v[15]}, hi1, lo1);
            u32x4 q0 = __builtin_bit_cast(u32x4, hi0), q1 = __builtin_bit_cast(u32x4, lo0);
            u32x4 q2 = __builtin_bit_cast(u32x4, hi1), q3 = __builtin_bit_cast(u32x4, lo1);
            if (POOL) {
                const int oy = (yw >> 1) + nt, ox = (xw >> 1) + blk;
                st((oy * Wo + ox) * c4 + (ch0 >> 2) + ql, oy < Ho && ox < Wo && ch0 < Cout, quad_pick(q0, q1, q2, q3, ql));
            } else {
                quad_transpose(q0, q1, q2, q3, ql);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int oy = yw + 2 * nt + (sidx >> 1), ox = xw + 2 * blk + (sidx & 1);
                    st((oy * Wo + ox) * c4 + (ch0 >> 2) + ql, oy < Ho && ox < Wo && ch0 < Cout, sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3);
                }
            }
        }
    }
}

// Bias + ReLU + 2x2 max-pool + SH8 stores of a POOLED wave tile, starting from the raw accumulators.  x -> relu(x * winv + b) is
// monotone (winv > 0, and so is the rounding of the fma), hence max over the 2x2 block commutes with it: pooling runs FIRST, on the
// raw sums, and everything after it on the quarter of the data a lane actually stores -- after the pool the four lanes of a quad hold
// the same 16 channels of one output pixel and each stores ONE 16-byte piece of them (quad lane ql: hi or lo halves (ql & 1) of channel
// group ql >> 1), so a lane selects its 8 channels, adds their bias, clamps, and converts only those.  Bit-identical to "bias, ReLU,
// pool, split everything, pick a piece" (tests/test_gpu_conv_dma.py compares with the register-staged kernels, which still do that).
// Per (mt, nt) block: 32 v_max_f32_dpp + 8 swaps + 8 selects + 16 + 24 + 4  ~  94 VALU instead of 142.
template <class Store>
__device__ __forceinline__ void epilogue_pool_sh8_st(f32x16 (&acc)[2][2], const Store& st, const float* sb, float winv, int H, int W, int Cout, int yw, int xw, int n0,
                                                     int lane) {
    const int i32 = lane & 31, h = lane >> 5, ql = i32 & 3, blk = i32 >> 2;
    const int c4 = Cout >> 2;
    const int Ho = H >> 1, Wo = W >> 1;
    const bool upper = (ql & 2) != 0, want_lo = (ql & 1) != 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int ch0 = n0 + mt * 32 + 16 * h;
        const float* bp = sb + mt * 32 + 16 * h + 8 * (ql >> 1);  // the 8 channels this lane stores
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[mt][nt][r];
                x = fmaxf(x, dpp_quad_xor1(x));
                x = fmaxf(x, dpp_quad_xor2(x));
                v[r] = x;
            }
            half_swap8(v);  // 16 consecutive channels of the pooled pixel: v[0..3], v[8..11] = +0..7;  v[4..7], v[12..15] = +8..15
            // (as opaque register values: a select between ELEMENTS OF v makes hipcc index the array dynamically -- through scratch, whose
            // loads and stores are VMEM instructions the kernels' counted vmcnt waits do not expect)
            f32x4 a0 = {v[0], v[1], v[2], v[3]}, a1 = {v[4], v[5], v[6], v[7]}, c0 = {v[8], v[9], v[10], v[11]}, c1 = {v[12], v[13], v[14], v[15]};
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1));
            const f32x4 ga = upper ? a1 : a0, gc = upper ? c1 : c0;
            f32x8 g = {ga[0], ga[1], ga[2], ga[3], gc[0], gc[1], gc[2], gc[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g[j] = fmaxf(fma_winv(g[j], winv, b0[j]), 0.f);
                g[4 + j] = fmaxf(fma_winv(g[4 + j], winv, b1[j]), 0.f);
            }
            f16x8 hi, lo;
            split8(g, hi, lo);
            const u32x4 qh = __builtin_bit_cast(u32x4, hi), qlo = __builtin_bit_cast(u32x4, lo);
            const u32x4 piece = {want_lo ? qlo[0] : qh[0], want_lo ? qlo[1] : qh[1], want_lo ? qlo[2] : qh[2], want_lo ? qlo[3] : qh[3]};
            const int oy = (yw >> 1) + nt, ox = (xw >> 1) + blk;
            st((oy * Wo + ox) * c4 + (ch0 >> 2) + ql, oy < Ho && ox < Wo && ch0 < Cout, piece);
        }
    }
}

#endif
template <int WY, int WX, bool POOL>
__global__ __launch_bounds__(512, 2) void conv3x3_dma_kernel(const ConvF16Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WY * WX == 8, "eight waves");
    constexpr int TH = 4 * WY, TW = 16 * WX;
    constexpr int PW = ((TW + 2 + 7) / 16) * 16 + 8;            // pixel pitch == 8 (mod 16)
    constexpr int NSLOT = ((TH + 1) * PW + TW + 2) * 4;         // 16-byte slots of a halo tile (last row without its pitch padding)
    constexpr int NIW = (NSLOT + 511) / 512;                    // halo DMA pieces per wave
    constexpr int LP = (NSLOT + 8 * NIW - 1) / (8 * NIW);       // lanes (16-byte slots) per halo piece
    constexpr int IN_SZ = 8 * NIW * LP;
    constexpr int WS = 3 * 64 * 4, WR = 4;                      // weight stage: 3 taps x 64 couts x 4 pieces = 768 slots of 16 B
    constexpr int NWA = 2, WLP = WS / (8 * NWA);                // ... moved as two 48-lane pieces per wave (every wave issues the same VMEM count)
    constexpr int E = POOL ? 4 : 16;                            // output stores per wave and item
    static_assert(LP <= 64 && NIW + NWA <= 12 && NIW + NWA + E < 64, "filler slots / vmcnt range");
    // ONE LDS array (see gemm_f16.hip): [2 halo tiles][4 weight stages][bias of the layer]
    __shared__ u32x4 s_all[2 * IN_SZ + WR * WS + CONV_DMA_MAXC / 4];
    u32x4* const s_in = s_all;
    u32x4* const s_w = s_all + 2 * IN_SZ;
    float* const s_bias = reinterpret_cast<float*>(s_all + 2 * IN_SZ + WR * WS);

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y, nslices = (p.Cout + 63) / 64;
    const int total = tiles * nslices * p.B;
    const int cin4 = p.Cin >> 2, nchunks = p.Cin >> 4;
    const u32x4* in = reinterpret_cast<const u32x4*>(p.in);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), 0, 0x7fffffff, 0x00020000);

    struct Item { int b, y0, x0, n0; };
    auto item_of = [&](int id) {
        int logical = id;
        if ((total & 7) == 0) logical = (id & 7) * (total >> 3) + (id >> 3);
        const int slice = logical % nslices, r = logical / nslices;
        const int tile = r % tiles;
        return Item{r / tiles, (tile / tiles_x) * TH, (tile % tiles_x) * TW, slice * 64};
    };

    for (int i = t; i < CONV_DMA_MAXC; i += 512) s_bias[i] = i < p.Cout ? p.bias[i] : 0.f;

    // DMA geometry of this lane: halo piece j = wu + 8 i covers LDS slots [LP j, LP j + LP); slot g -> halo pixel P = g >> 2,
    // stored piece g & 3 = source piece ^ ((P >> 2) & 3) (the DMA writes LDS linearly, so the swizzle goes on the source)
    int in_off[NIW], in_yx[NIW], vo[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int g = (wu + 8 * i) * LP + lane, P = g >> 2, py = P / PW, px = P - py * PW;
        const bool valid = g < NSLOT && px < TW + 2;
        in_off[i] = ((py * p.W + px) * cin4 + ((g & 3) ^ ((P >> 2) & 3))) * 16;  // bytes from the tile's halo origin
        in_yx[i] = valid ? (py << 16 | px) : -1;
    }
    auto set_vo = [&](const Item& it) {  // byte offsets of this lane's halo slots for item `it` (outside the image: out of range -> zeros)
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int py = in_yx[i] >> 16, px = in_yx[i] & 0xffff;
            const bool ok = in_yx[i] >= 0 && (unsigned)(it.y0 + py - 1) < (unsigned)p.H && (unsigned)(it.x0 + px - 1) < (unsigned)p.W;
            vo[i] = ok ? in_off[i] : (int)0x80000000;
        }
    };
    int w_off[NWA];  // weight pieces NWA wu .. NWA wu + NWA - 1 of a stage: LDS slots [WLP j, WLP j + WLP)
#pragma unroll
    for (int i = 0; i < NWA; ++i) {
        const int g = (wu * NWA + i) * WLP + min(lane, WLP - 1), row = g >> 2, tl = row >> 6, n = row & 63;
        w_off[i] = ((tl * p.CoutPad + n) * 4 + ((g & 3) ^ ((n >> 2) & 3))) * 16;
    }
    auto in_rsrc = [&](const Item& it, int c) {  // base = halo origin of the tile (may lie before the image: those lanes are out of range)
        const u32x4* base = in + (((long long)it.b * p.H + (it.y0 - 1)) * p.W + (it.x0 - 1)) * cin4 + c * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(base), 0, 0x7fffffff, 0x00020000);
    };
    auto dma_in = [&](const __amdgpu_buffer_rsrc_t& rs, int buf, int i) {
        if (lane < LP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, s_in + buf * IN_SZ + (wu + 8 * i) * LP, 16, vo[i], 0, 0, 0);
    };
    auto dma_w = [&](int n0, int c, int ky, int slot, int i) {
        const int soff = (((c * 9 + 3 * ky) * p.CoutPad + n0) * 4) * 16;
        if (lane < WLP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, s_w + slot * WS + (wu * NWA + i) * WLP, 16, w_off[i], soff, 0, 0);
    };

    const int wy0 = (wu / WX) * 4, wx0 = (wu % WX) * 16;
    const int i32 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
    // pixel operand: lane n -> pixel (dy,dx) = ((n>>1)&1, n&1) of 2x2 block n>>2 (blocks along x).  PW = 8 (mod 16): the swizzle
    // term of pixel P0 + ky PW + kx is that of P0 + kx with bit 1 flipped for odd ky, and the second pixel tile (+2 PW) has the same
    const int P0 = (wy0 + ((i32 >> 1) & 1)) * PW + wx0 + 2 * (i32 >> 2) + (i32 & 1);
    int b_hi[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) b_hi[kx] = (P0 + kx) * 4 + ((2 * h) ^ (((P0 + kx) >> 2) & 3));
    const int a_hi = i32 * 4 + ((2 * h) ^ ((i32 >> 2) & 3));

    f16x8 fr[2][8];  // [register set][ah0, bh0, bh1, ah1, bl0, bl1, al0, al1] in the order the MFMAs need them
    auto read_frag = [&](int set, int buf, int slot, int ky, int kx, int i) {
        const f16x8* si = reinterpret_cast<const f16x8*>(s_in + buf * IN_SZ) + ky * PW * 4;
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w + slot * WS) + kx * 64 * 4;
        const int bx = b_hi[kx] ^ ((ky & 1) << 1);
        switch (i) {
            case 0: fr[set][0] = sw[a_hi]; break;
            case 1: fr[set][1] = si[bx]; break;
            case 2: fr[set][2] = si[bx + 2 * PW * 4]; break;
            case 3: fr[set][3] = sw[a_hi + 32 * 4]; break;
            case 4: fr[set][4] = si[bx ^ 1]; break;
            case 5: fr[set][5] = si[(bx ^ 1) + 2 * PW * 4]; break;
            case 6: fr[set][6] = sw[a_hi ^ 1]; break;
            default: fr[set][7] = sw[(a_hi ^ 1) + 32 * 4]; break;
        }
    };

    int id = blockIdx.x;
    if (id >= total) return;
    Item cur = item_of(id);
    // prologue: halo tile of chunk 0 -> buffer 0, weight stages 0..2 -> slots 0..2
    set_vo(cur);
    {
        const __amdgpu_buffer_rsrc_t rs = in_rsrc(cur, 0);
#pragma unroll
        for (int i = 0; i < NIW; ++i) dma_in(rs, 0, i);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < NWA; ++i) dma_w(cur.n0, 0, ky, ky, i);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frag(0, 0, 0, 0, 0, i);

    int sidx = 0;                  // stage counter of the stream (ring slot = sidx & 3)
    bool e_now = false, e_prev = false;  // an epilogue (E stores) was issued just before the current / the previous stage
    for (;;) {
        const bool more = id + (int)gridDim.x < total;
        const Item nxt = item_of(more ? id + (int)gridDim.x : id);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        // one chunk = three stages; PAR = chunk parity = halo buffer, and tap T of the chunk uses register set (T + PAR) & 1
        auto chunk = [&](auto par_c, int c) {
            constexpr int PAR = decltype(par_c)::value;
            const bool last = c + 1 == nchunks;
            const bool has_next = !last || more;      // the chunk after this one exists
            const Item& nit = last ? nxt : cur;
            const int nc = last ? 0 : c + 1;
            if (last) set_vo(nxt);                    // the halo tile fetched during this chunk belongs to the next item
            const __amdgpu_buffer_rsrc_t rs = in_rsrc(nit, nc);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // my DMA pieces up to the stage before last have landed (VMEM issued since: the previous stage's pieces and,
                // around an item boundary, the E output stores), then everyone's
                if (ky == 0) {        // previous stage = last stage of the previous chunk (its NWA weight pieces)
                    if (e_now) wait_vm_c<NWA + E>(); else wait_vm_c<NWA>();
                } else if (!has_next) {
                    wait_vm_c<0>();
                } else if (ky == 1) { // previous stage issued the weight pieces and the whole next halo tile
                    if (e_prev) wait_vm_c<NWA + NIW + E>(); else wait_vm_c<NWA + NIW>();
                } else {
                    wait_vm_c<NWA>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                e_prev = e_now;
                e_now = false;
                const int slot = sidx & 3, slot1 = (sidx + 1) & 3, slot3 = (sidx + 3) & 3;
                int dslot = 0;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int T = 3 * ky + kx, S = (T + PAR) & 1;
                    int fslot = 0;
                    auto filler = [&]() {
                        const int f = fslot++;
                        if (f < 8) {  // fragment read f of the next tap -> the other register set
                            if (T < 8) read_frag(1 - S, PAR, kx == 2 ? slot1 : slot, (T + 1) / 3, (T + 1) % 3, f);
                            else read_frag(1 - S, 1 - PAR, slot1, 0, 0, f);
                        } else {      // DMA: the weight rows three stages ahead, then (first stage of a chunk) the next halo tile
                            const int d = dslot++;
                            if (has_next) {
                                if (d < NWA) dma_w(nit.n0, nc, ky, slot3, d);
                                else if (ky == 0 && d - NWA < NIW) dma_in(rs, 1 - PAR, d - NWA);
                            }
                        }
                        PIN();
                    };
                    PIN();
                    acc[0][0] = mfma16(fr[S][0], fr[S][1], acc[0][0]); filler();
                    acc[0][1] = mfma16(fr[S][0], fr[S][2], acc[0][1]); filler();
                    acc[1][0] = mfma16(fr[S][3], fr[S][1], acc[1][0]); filler();
                    acc[1][1] = mfma16(fr[S][3], fr[S][2], acc[1][1]); filler();
                    acc[0][0] = mfma16(fr[S][0], fr[S][4], acc[0][0]); filler();
                    acc[0][1] = mfma16(fr[S][0], fr[S][5], acc[0][1]); filler();
                    acc[1][0] = mfma16(fr[S][3], fr[S][4], acc[1][0]); filler();
                    acc[1][1] = mfma16(fr[S][3], fr[S][5], acc[1][1]); filler();
                    acc[0][0] = mfma16(fr[S][6], fr[S][1], acc[0][0]); filler();
                    acc[0][1] = mfma16(fr[S][6], fr[S][2], acc[0][1]); filler();
                    acc[1][0] = mfma16(fr[S][7], fr[S][1], acc[1][0]); filler();
                    acc[1][1] = mfma16(fr[S][7], fr[S][2], acc[1][1]); filler();
                }
                ++sidx;
            }
        };
#pragma unroll 1
        for (int c = 0; c < nchunks; c += 2) {
            chunk(std::integral_constant<int, 0>{}, c);
            chunk(std::integral_constant<int, 1>{}, c + 1);
        }
        // epilogue: bias (LDS table of the whole layer), ReLU, pool, SH8 -- E unconditional buffer stores per wave
        {
            const float* sb = s_bias + cur.n0;
            const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
            const int img_bytes = Ho * Wo * p.Cout * 4;
            BufStore st{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (long long)cur.b * img_bytes, 0, img_bytes, 0x00020000)};
            if (POOL) {
                epilogue_pool_sh8_st(acc, st, sb, p.winv, p.H, p.W, p.Cout, cur.y0 + wy0, cur.x0 + wx0, cur.n0, lane);
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + mt * 32 + 8 * j + 4 * h);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            acc[mt][nt][4 * j] = fma_winv(acc[mt][nt][4 * j], p.winv, bv[0]); acc[mt][nt][4 * j + 1] = fma_winv(acc[mt][nt][4 * j + 1], p.winv, bv[1]);
                            acc[mt][nt][4 * j + 2] = fma_winv(acc[mt][nt][4 * j + 2], p.winv, bv[2]); acc[mt][nt][4 * j + 3] = fma_winv(acc[mt][nt][4 * j + 3], p.winv, bv[3]);
                        }
                    }
                epilogue_sh8_st<POOL>(acc, st, p.H, p.W, p.Cout, cur.y0 + wy0, cur.x0 + wx0, cur.n0, lane);
            }
            e_now = true;
        }
        if (!more) break;
        cur = nxt;
        id += gridDim.x;
    }
#endif
}

// ---- round 3 (end): the same convolution on v_mfma_f32_16x16x32_f16 with TAP PAIRS as the contraction index ------------------
// tests/micro/mfma_lds_shapes.hip: under this loop's LDS traffic the 16x16x32 shape sustains 13 % more than 32x32x16 (the chip holds a
// higher clock on it), provided K = 32 is filled without extra instructions.  Here K = 32 = [16 channels of tap T | 16 channels of tap
// T + 1]: lanes 0-31 of both operands address the first tap of a pair and lanes 32-63 the second (k-group g = lane >> 4: bit 0 = channel
// half, bit 1 = tap of the pair), which is nothing but a per-lane LDS address.  Nine taps per chunk are odd, so pairs run through the
// stream of taps of TWO chunks (18 taps = 9 pairs; every layer has an even number of 16-channel chunks): the fifth pair is
// [tap 8 of chunk c | tap 0 of chunk c + 1].
//   * a DOUBLE STAGE = two filter rows = 6 taps = 3 pairs x 48 MFMAs per wave, one barrier each (half as many as the kernel above);
//     3 double stages = 2 chunks.  The data of double stage D + 1 (two weight stages into the other half of the four-slot ring, and the
//     next halo tile where one falls due) is requested under the first pair of D and awaited (vmcnt 0, or E when output stores were
//     issued last) before the barrier of D + 1: no counted waits beyond that.
//   * a wave's 64 couts x 64 pixels are 4 x 4 tiles of 16 x 16; row m of cout tile i is channel 16 (m >> 2) + 4 i + (m & 3), so the
//     accumulator rows 4 g + r of the four tiles are 16 CONSECUTIVE channels 16 g + 4 i + r of the lane's pixel -- the epilogue needs
//     no permlane swaps.  Pixel tile j: 2 rows x 8 columns at (+2 (j >> 1), +8 (j & 1)), lane n = 2x2 block n >> 2, pixel n & 3 of it.
//   * weight rows in LDS are swizzled by the Gray code of (cout >> 4) & 3 (see gray2 below: ds_read_b128's lane groups), pixels as above.
// Not bit-identical to the kernels above (the taps are summed in another order); tests/test_gpu_conv_dma.py holds it to them within
// the fp32 rounding of the sums and to float64 like every other stage.
template <class Store>
__device__ __forceinline__ void epilogue16_sh8_st(f32x4 (&acc)[4][4], const Store& st, const float* sb, float winv, int H, int W, int Cout, int yw, int xw, int n0,
                                                  int lane) {
    const int n = lane & 15, g = lane >> 4, ql = n & 3, blk = n >> 2;
    const int c4 = Cout >> 2, ch0 = n0 + 16 * g;
    const float* bp = sb + 16 * g;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4), b2 = *reinterpret_cast<const f32x4*>(bp + 8),
                b3 = *reinterpret_cast<const f32x4*>(bp + 12);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x8 v0, v1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v0[r] = fmaxf(fma_winv(acc[0][j][r], winv, b0[r]), 0.f);
            v0[4 + r] = fmaxf(fma_winv(acc[1][j][r], winv, b1[r]), 0.f);
            v1[r] = fmaxf(fma_winv(acc[2][j][r], winv, b2[r]), 0.f);
            v1[4 + r] = fmaxf(fma_winv(acc[3][j][r], winv, b3[r]), 0.f);
        }
        f16x8 hi0, lo0, hi1, lo1;
        split8(v0, hi0, lo0);
        split8(v1, hi1, lo1);
        u32x4 q0 = __builtin_bit_cast(u32x4, hi0), q1 = __builtin_bit_cast(u32x4, lo0);
        u32x4 q2 = __builtin_bit_cast(u32x4, hi1), q3 = __builtin_bit_cast(u32x4, lo1);
        quad_transpose(q0, q1, q2, q3, ql);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int oy = yw + 2 * (j >> 1) + (sidx >> 1), ox = xw + 8 * (j & 1) + 2 * blk + (sidx & 1);
            st((oy * W + ox) * c4 + (ch0 >> 2) + ql, oy < H && ox < W && ch0 < Cout, sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3);
        }
    }
}
// pooled variant: pool the raw sums first (see epilogue_pool_sh8_st), then each quad lane finishes and stores ONE piece
template <class Store>
__device__ __forceinline__ void epilogue16_pool_sh8_st(f32x4 (&acc)[4][4], const Store& st, const float* sb, float winv, int H, int W, int Cout, int yw, int xw,
                                                       int n0, int lane) {
    const int n = lane & 15, g = lane >> 4, ql = n & 3, blk = n >> 2;
    const int c4 = Cout >> 2, ch0 = n0 + 16 * g;
    const int Ho = H >> 1, Wo = W >> 1;
    const bool upper = (ql & 2) != 0, want_lo = (ql & 1) != 0;
    const float* bp = sb + 16 * g + 8 * (ql >> 1);  // the 8 channels this lane stores
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[i][j][r];
                x = fmaxf(x, dpp_quad_xor1(x));
                x = fmaxf(x, dpp_quad_xor2(x));
                a[i][r] = x;
            }
        asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));  // opaque values: no dynamic indexing (scratch) for the selects
        const f32x4 ga = upper ? a[2] : a[0], gc = upper ? a[3] : a[1];
        f32x8 gv = {ga[0], ga[1], ga[2], ga[3], gc[0], gc[1], gc[2], gc[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gv[r] = fmaxf(fma_winv(gv[r], winv, b0[r]), 0.f);
            gv[4 + r] = fmaxf(fma_winv(gv[4 + r], winv, b1[r]), 0.f);
        }
        f16x8 hi, lo;
        split8(gv, hi, lo);
        const u32x4 qh = __builtin_bit_cast(u32x4, hi), qlo = __builtin_bit_cast(u32x4, lo);
        const u32x4 piece = {want_lo ? qlo[0] : qh[0], want_lo ? qlo[1] : qh[1], want_lo ? qlo[2] : qh[2], want_lo ? qlo[3] : qh[3]};
        const int oy = (yw >> 1) + (j >> 1), ox = (xw >> 1) + 4 * (j & 1) + blk;
        st((oy * Wo + ox) * c4 + (ch0 >> 2) + ql, oy < Ho && ox < Wo && ch0 < Cout, piece);
    }
}

// Weight rows of conv3x3_pair_kernel are swizzled by the Gray code of (cout >> 4) & 3.  ds_read_b128 is served in four NON-contiguous
// 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, the same + 32: MI355X_MICROARCH.md, LDS): a group holds the cout quarters
// {0, 3} of one channel half (piece base 0) and {1, 2} of the other (piece base 2), or the reverse, so the four (base ^ swizzle) values
// must differ: with q -> q they collide pairwise (measured: SQ_LDS_BANK_CONFLICT = a third of all LDS cycles, every weight read
// 2-way), with q -> q ^ (q >> 1) = {0, 1, 3, 2} they do not.
__device__ __forceinline__ int gray2(int q) { return (q ^ (q >> 1)) & 3; }
__device__ __forceinline__ f32x4 mfma16x32(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <int WY, int WX, bool POOL>
__global__ __launch_bounds__(512, 2) void conv3x3_pair_kernel(const ConvF16Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WY * WX == 8, "eight waves");
    constexpr int TH = 4 * WY, TW = 16 * WX;
    constexpr int PW = ((TW + 2 + 7) / 16) * 16 + 8;            // pixel pitch == 8 (mod 16)
    constexpr int NSLOT = ((TH + 1) * PW + TW + 2) * 4;         // 16-byte slots of a halo tile (last row without its pitch padding)
    constexpr int NIW = (NSLOT + 511) / 512;                    // halo DMA pieces per wave
    constexpr int LP = (NSLOT + 8 * NIW - 1) / (8 * NIW);       // lanes (16-byte slots) per halo piece
    constexpr int IN_SZ = 8 * NIW * LP;
    constexpr int WS = 3 * 64 * 4, WR = 4;                      // weight stage: 3 taps x 64 couts x 4 pieces = 768 slots of 16 B
    constexpr int NWA = 2, WLP = WS / (8 * NWA);                // ... moved as two 48-lane pieces per wave
    constexpr int E = POOL ? 4 : 16;                            // output stores per wave and item
    static_assert(LP <= 64 && E < 64, "vmcnt range");
    __shared__ u32x4 s_all[2 * IN_SZ + WR * WS + CONV_DMA_MAXC / 4];
    u32x4* const s_in = s_all;
    u32x4* const s_w = s_all + 2 * IN_SZ;
    float* const s_bias = reinterpret_cast<float*>(s_all + 2 * IN_SZ + WR * WS);

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y, nslices = (p.Cout + 63) / 64;
    const int total = tiles * nslices * p.B;
    const int cin4 = p.Cin >> 2, nchunks = p.Cin >> 4;
    const u32x4* in = reinterpret_cast<const u32x4*>(p.in);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), 0, 0x7fffffff, 0x00020000);

    struct Item { int b, y0, x0, n0; };
    auto item_of = [&](int id) {
        int logical = id;
        if ((total & 7) == 0) logical = (id & 7) * (total >> 3) + (id >> 3);
        const int slice = logical % nslices, r = logical / nslices;
        const int tile = r % tiles;
        return Item{r / tiles, (tile / tiles_x) * TH, (tile % tiles_x) * TW, slice * 64};
    };

    for (int i = t; i < CONV_DMA_MAXC; i += 512) s_bias[i] = i < p.Cout ? p.bias[i] : 0.f;

    // DMA geometry (as in conv3x3_dma_kernel; the weight rows are swizzled by cout >> 4 here)
    int in_off[NIW], in_yx[NIW], vo[NIW];
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int gs = (wu + 8 * i) * LP + lane, P = gs >> 2, py = P / PW, px = P - py * PW;
        const bool valid = gs < NSLOT && px < TW + 2;
        in_off[i] = ((py * p.W + px) * cin4 + ((gs & 3) ^ ((P >> 2) & 3))) * 16;
        in_yx[i] = valid ? (py << 16 | px) : -1;
    }
    auto set_vo = [&](const Item& it) {
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
            const int py = in_yx[i] >> 16, px = in_yx[i] & 0xffff;
            const bool ok = in_yx[i] >= 0 && (unsigned)(it.y0 + py - 1) < (unsigned)p.H && (unsigned)(it.x0 + px - 1) < (unsigned)p.W;
            vo[i] = ok ? in_off[i] : (int)0x80000000;
        }
    };
    int w_off[NWA];
#pragma unroll
    for (int i = 0; i < NWA; ++i) {
        const int gs = (wu * NWA + i) * WLP + min(lane, WLP - 1), row = gs >> 2, tl = row >> 6, nn = row & 63;
        w_off[i] = ((tl * p.CoutPad + nn) * 4 + ((gs & 3) ^ gray2(nn >> 4))) * 16;
    }
    auto in_rsrc = [&](const Item& it, int c) {
        const u32x4* base = in + (((long long)it.b * p.H + (it.y0 - 1)) * p.W + (it.x0 - 1)) * cin4 + c * 4;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(base), 0, 0x7fffffff, 0x00020000);
    };
    auto dma_in = [&](const __amdgpu_buffer_rsrc_t& rs, int buf, int i) {
        if (lane < LP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, s_in + buf * IN_SZ + (wu + 8 * i) * LP, 16, vo[i], 0, 0, 0);
    };
    auto dma_w = [&](int n0, int c, int ky, int slot, int i) {
        const int soff = (((c * 9 + 3 * ky) * p.CoutPad + n0) * 4) * 16;
        if (lane < WLP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, s_w + slot * WS + (wu * NWA + i) * WLP, 16, w_off[i], soff, 0, 0);
    };

    // fragment geometry: byte offsets inside a halo tile / inside one tap's 64 x 64 B weight block
    const int wy0 = (wu / WX) * 4, wx0 = (wu % WX) * 16;
    const int n = lane & 15, g = lane >> 4, ch = g & 1;
    const bool second = lane >= 32;
    const int P00 = (wy0 + ((n >> 1) & 1)) * PW + wx0 + 2 * (n >> 2) + (n & 1);
    int bk[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) bk[kx] = ((P00 + kx) * 4 + ((2 * ch) ^ (((P00 + kx) >> 2) & 3))) * 16;
    const int a0 = ((16 * (n >> 2) + (n & 3)) * 4 + ((2 * ch) ^ gray2(n >> 2))) * 16;
    const char* const lds = reinterpret_cast<const char*>(s_all);
    constexpr int W_BASE = 2 * IN_SZ * 16;  // byte offset of the weight ring

    f32x4 acc[4][4];
    // Fragments of a pair: weights hi / lo of four cout tiles, pixels hi / lo of four pixel tiles.  Products run lo.hi, hi.lo, hi.hi, so
    // A_lo is dead after the first and B_lo after the second: the next pair's A_lo / B_lo are read into the same registers under the
    // second / third product, A_hi (first needed by the second product) under the pair's own first product, and only B_hi, live
    // through the whole pair, has two register sets -- 80 fragment registers instead of 128.
    f16x8 fah[4], fal[4], fbl[4], fbh[2][4];
    int ax[2], bx[2][2];  // byte addresses of the pair being read: A hi / lo;  B [hi / lo][j & 1]

    // tap T = 0..17 of a chunk pair: chunk parity T / 9 (= halo buffer), filter row, column, stage of its double stage
    auto pair_addr = [&](auto kc, int wring) {  // addresses of pair K of the period; wring = byte offset of the double stage's two ring slots
        constexpr int K = decltype(kc)::value;
        constexpr int Ta = 2 * K, Tb = 2 * K + 1;
        constexpr int cpa = Ta / 9, kya = (Ta % 9) / 3, kxa = Ta % 3, sta = (Ta / 3) & 1;
        constexpr int cpb = Tb / 9, kyb = (Tb % 9) / 3, kxb = Tb % 3, stb = (Tb / 3) & 1;
        const int fa = (bk[kxa] ^ ((kya & 1) << 5)) + (cpa * IN_SZ + kya * PW * 4) * 16;
        const int fb = (bk[kxb] ^ ((kyb & 1) << 5)) + (cpb * IN_SZ + kyb * PW * 4) * 16;
        const int sel = second ? fb : fa;
        bx[0][0] = sel;
        bx[1][0] = sel ^ 16;
        bx[0][1] = (sel ^ 32) + 8 * 4 * 16;
        bx[1][1] = bx[0][1] ^ 16;
        const int wa = W_BASE + (sta * WS + kxa * 256) * 16, wb = W_BASE + (stb * WS + kxb * 256) * 16;
        ax[0] = a0 + wring + (second ? wb : wa);
        ax[1] = ax[0] ^ 16;
    };
    auto rd_a = [&](int lo, int i) { return *reinterpret_cast<const f16x8*>(lds + ax[lo] + i * 256); };
    auto rd_b = [&](int lo, int j) { return *reinterpret_cast<const f16x8*>(lds + bx[lo][j & 1] + (j >> 1) * (2 * PW * 4 * 16)); };

    int id = blockIdx.x;
    if (id >= total) return;
    Item cur = item_of(id);
    // prologue: halo tile of chunk 0 -> buffer 0, weight stages (0, 0), (0, 1) -> ring slots 0, 1
    set_vo(cur);
    {
        const __amdgpu_buffer_rsrc_t rs = in_rsrc(cur, 0);
#pragma unroll
        for (int i = 0; i < NIW; ++i) dma_in(rs, 0, i);
#pragma unroll
        for (int ky = 0; ky < 2; ++ky)
#pragma unroll
            for (int i = 0; i < NWA; ++i) dma_w(cur.n0, 0, ky, ky, i);
    }
    int D = 0;            // double stages of the stream so far (ring half = D & 1)
    bool e_now = false;   // E output stores were issued after the last DMA request
    for (;;) {
        const bool more = id + (int)gridDim.x < total;
        const Item nxt = item_of(more ? id + (int)gridDim.x : id);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // double stage Q = 0, 1, 2 of the chunk pair (c, c + 1)
        auto dstage = [&](auto qc, int c) {
            constexpr int Q = decltype(qc)::value;
            const bool last = c + 2 == nchunks;
            // what double stage D + 1 needs: Q = 0 -> weight stages (c, 2), (c + 1, 0) and the halo tile of chunk c + 1 (buffer 1);
            // Q = 1 -> (c + 1, 1), (c + 1, 2);  Q = 2 -> (c + 2, 0), (c + 2, 1) and the halo tile of chunk c + 2 (buffer 0) -- of the
            // next item when this was the item's last chunk pair
            const bool has_next = Q < 2 || !last || more;
            const Item& nit = (Q == 2 && last) ? nxt : cur;
            const int c0 = Q == 0 ? c : Q == 1 ? c + 1 : (last ? 0 : c + 2), ky0 = Q == 0 ? 2 : Q == 1 ? 1 : 0;
            const int c1 = Q == 0 ? c + 1 : Q == 1 ? c + 1 : (last ? 0 : c + 2), ky1 = Q == 0 ? 0 : Q == 1 ? 2 : 1;
            if (Q == 2 && last) set_vo(nxt);
            const __amdgpu_buffer_rsrc_t rs = in_rsrc(nit, Q == 0 ? c + 1 : (last ? 0 : c + 2));
            const int wring = (D & 1) * (2 * WS * 16), nslot = 2 * ((D + 1) & 1);

            if (e_now) wait_vm_c<E>(); else wait_vm_c<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            e_now = false;
            pair_addr(std::integral_constant<int, 3 * Q>{}, wring);
#pragma unroll
            for (int r = 0; r < 4; ++r) { fal[r] = rd_a(1, r); fbh[0][r] = rd_b(0, r); }
#pragma unroll
            for (int r = 0; r < 4; ++r) fbl[r] = rd_b(1, r);
            PIN();
            constexpr int ND = 2 * NWA + NIW;  // DMA requests of a double stage (all under its first pair)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int S = k & 1;
                const int d0 = k == 0 ? 4 : 0, p0 = k == 0 ? 4 + ND : 4;  // slots: [A_hi x 4][DMA][addresses of the next pair][B_hi' x 4] ... [16+: A_lo'] ... [32+: B_lo']
                const int l0 = p0 + 5 > 16 ? p0 + 5 : 16;
                static_assert(4 + ND + 5 + 4 <= 32, "filler slots");
                int f = 0;
                auto filler = [&]() {
                    const int s = f++;
                    if (s < 4) {
                        fah[s] = rd_a(0, s);  // this pair's A_hi: first used by the second product
                    } else if (k == 0 && s - d0 < ND) {
                        const int d = s - d0;
                        if (d < 2 * NWA) {
                            if (has_next) dma_w(nit.n0, (d < NWA) ? c0 : c1, (d < NWA) ? ky0 : ky1, nslot + (d >= NWA), d % NWA);
                        } else {
                            if (Q != 1 && has_next) dma_in(rs, Q == 0 ? 1 : 0, d - 2 * NWA);
                        }
                    } else if (k < 2) {
                        if (s == p0) {
                            if (k == 0) pair_addr(std::integral_constant<int, 3 * Q + 1>{}, wring);
                            else pair_addr(std::integral_constant<int, 3 * Q + 2>{}, wring);
                        } else if (s > p0 && s <= p0 + 4) {
                            fbh[1 - S][s - p0 - 1] = rd_b(0, s - p0 - 1);
                        } else if (s >= l0 && s < l0 + 4) {
                            fal[s - l0] = rd_a(1, s - l0);      // A_lo is dead after the first product
                        } else if (s >= 32 && s < 36) {
                            fbl[s - 32] = rd_b(1, s - 32);      // B_lo is dead after the second product
                        }
                    }
                    PIN();
                };
                // lo.hi, hi.lo, hi.hi: consecutive MFMAs on different accumulators
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[i][j] = mfma16x32(fal[i], fbh[S][j], acc[i][j]); filler(); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[i][j] = mfma16x32(fah[i], fbl[j], acc[i][j]); filler(); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[i][j] = mfma16x32(fah[i], fbh[S][j], acc[i][j]); filler(); }
            }
            ++D;
        };
#pragma unroll 1
        for (int c = 0; c < nchunks; c += 2) {
            dstage(std::integral_constant<int, 0>{}, c);
            dstage(std::integral_constant<int, 1>{}, c);
            dstage(std::integral_constant<int, 2>{}, c);
        }
        {
            const float* sb = s_bias + cur.n0;
            const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
            const int img_bytes = Ho * Wo * p.Cout * 4;
            BufStore st{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (long long)cur.b * img_bytes, 0, img_bytes, 0x00020000)};
            if (POOL) epilogue16_pool_sh8_st(acc, st, sb, p.winv, p.H, p.W, p.Cout, cur.y0 + wy0, cur.x0 + wx0, cur.n0, lane);
            else epilogue16_sh8_st(acc, st, sb, p.winv, p.H, p.W, p.Cout, cur.y0 + wy0, cur.x0 + wx0, cur.n0, lane);
            e_now = true;
        }
        if (!more) break;
        cur = nxt;
        id += gridDim.x;
    }
#endif
}

// conv1a + conv1b + 2x2 max-pool in one kernel (superpoint.py:209-211): the 64-channel full-resolution
// map of conv1a (78.6 MB per VGA image) never exists in HBM.  Per 16-channel chunk the conv1a outputs of
// the 18x34 halo tile are recomputed from a 20x36 gray tile in LDS (VALU, weights in SGPRs: every wave
// handles one 8-channel group so the filter taps are wave-uniform) and written as SH8 pieces straight
// into the double-buffered conv1b input tile.  Waves 0-3 produce the next chunk and then multiply the
// current one, waves 4-7 do it in the opposite order: SIMD partners (w, w+4) are always in opposite
// phases, so the VALU work of one runs under the MFMAs of the other.
template <bool GENERIC>
__global__ __launch_bounds__(CT, 2) void conv1ab_f16x3_kernel(const ConvF16Args p, const Conv1aArgs c1) {
    constexpr int WY = 4, WX = 2;
    constexpr int TH = 4 * WY, TW = 16 * WX;
    constexpr int PW = ((TW + 2 + 7) / 16) * 16 + 8;
    constexpr int NPIX = (TH + 2) * (TW + 2);
    constexpr int NWT = (9 * 64 * 4 + CT - 1) / CT;
    constexpr int GH = TH + 4, GW = TW + 4, GP = GW + 1;
    constexpr int NG = (GH * GW + CT - 1) / CT;
    __shared__ u32x4 s_in[2][(TH + 2) * PW * 4];
    __shared__ u32x4 s_w[9 * 64 * 4];
    __shared__ float s_g[GH * GP];
    __shared__ __attribute__((aligned(16))) float s_c1[12 * 64];  // conv1a: [9 taps][64 ch] | bias | post_scale | post_shift
    __shared__ __attribute__((aligned(16))) float s_bias[64];     // conv1b bias (Cout = 64: one cout slice)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y, total = tiles * p.B;
    const u32x4* wt = reinterpret_cast<const u32x4*>(p.Wt);

    // Persistent: the workgroup walks tiles in the XCD-contiguous order of xcd_item and treats their 16-channel chunks
    // as one stream -- while the last chunk of a tile is multiplied, conv1a already produces the first chunk of the
    // NEXT tile (its gray tile was fetched two chunks earlier), so only the epilogue separates two tiles' MFMA phases.
    struct Tile { int b, y0, x0; };
    auto tile_of = [&](int id) {
        int logical = id;
        if ((total & 7) == 0) logical = (id & 7) * (total >> 3) + (id >> 3);
        const int tl = logical % tiles;
        return Tile{logical / tiles, (tl / tiles_x) * TH, (tl % tiles_x) * TW};
    };
    // gray tile (RGB -> gray as superpoint.py:204-206), zero outside the image: global -> registers -> s_g
    float rg[NG];
    auto load_gray = [&](const Tile& tl) {
        const long long HW = (long long)c1.H * c1.W;
        const float* img = c1.image + (long long)tl.b * c1.C * HW;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = t + CT * k;
            const int gy = i / GW, gx = i - gy * GW;
            const int yy = tl.y0 + gy - 2, xx = tl.x0 + gx - 2;
            float v = 0.f;
            if (i < GH * GW && yy >= 0 && yy < c1.H && xx >= 0 && xx < c1.W) {
                const long long o = (long long)yy * c1.W + xx;
                if (c1.C == 3)
                    v = __fadd_rn(__fadd_rn(__fmul_rn(img[o], 0.299f), __fmul_rn(img[HW + o], 0.587f)), __fmul_rn(img[2 * HW + o], 0.114f));
                else
                    v = img[o];
            }
            rg[k] = v;
        }
    };
    auto store_gray = [&]() {
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int i = t + CT * k;
            const int gy = i / GW, gx = i - gy * GW;
            if (i < GH * GW) s_g[gy * GP + gx] = rg[k];
        }
    };

    for (int i = t; i < 12 * 64; i += CT) {
        const int k = i >> 6, ch = i & 63;
        float v;
        if (k < 9) v = c1.w[ch * 9 + k];
        else if (k == 9) v = c1.bias[ch];
        else if (k == 10) v = c1.post_scale ? c1.post_scale[ch] : 1.f;
        else v = c1.post_shift ? c1.post_shift[ch] : 0.f;
        s_c1[i] = v;
    }
    if (t < 64) s_bias[t] = t < p.Cout ? p.bias[t] : 0.f;

    u32x4 rwt[NWT];
    auto load_w = [&](int c) {
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int idx = t + CT * i, row = idx >> 2, q = idx & 3, tap = row >> 6, n = row & 63;
            rwt[i] = (idx < 9 * 64 * 4) ? wt[((long long)(c * 9 + tap) * p.CoutPad + n) * 4 + q] : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < NWT; ++i) {
            const int idx = t + CT * i, row = idx >> 2, q = idx & 3;
            if (idx < 9 * 64 * 4) s_w[row * 4 + (q ^ ((row >> 2) & 3))] = rwt[i];
        }
    };
    // conv1a for chunk c (channels 16c .. 16c+15) over the halo tile -> s_in[buf].  A wave owns one 8-channel
    // group (uniform) and three halo pixels per lane; the filter taps come from the LDS table s_c1 ([tap][channel],
    // every lane reads the same address = broadcast) -- as scalar loads they were ~50 dependent round trips per
    // chunk and made this phase 4x longer than the MFMA phase it is meant to hide under.
    auto produce = [&](int c, int buf, int y0, int x0) {
        const int gg = wu & 1;              // 8-channel group of this wave (uniform)
        const int ch0 = c * 16 + gg * 8;
        const int hp0 = (wu >> 1) * 64 + lane;
        int gofs[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hp = min(hp0 + 256 * i, NPIX - 1);
            const int py = hp / (TW + 2), px = hp - py * (TW + 2);
            gofs[i] = py * GP + px;
        }
        f32x8 o8[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) o8[i] = f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; ++k) {  // same tap order as the reference accumulation
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(&s_c1[k * 64 + ch0]);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(&s_c1[k * 64 + ch0 + 4]);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float g = s_g[gofs[i] + (k / 3) * GP + (k % 3)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o8[i][j] = fmaf(w0[j], g, o8[i][j]);
                    o8[i][4 + j] = fmaf(w1[j], g, o8[i][4 + j]);
                }
            }
        }
        {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&s_c1[9 * 64 + ch0]), b1 = *reinterpret_cast<const f32x4*>(&s_c1[9 * 64 + ch0 + 4]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o8[i][j] = fmaxf(o8[i][j] + b0[j], 0.f);
                    o8[i][4 + j] = fmaxf(o8[i][4 + j] + b1[j], 0.f);
                }
        }
        if (c1.post_scale) {  // BatchNorm affine after the ReLU (open variant)
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(&s_c1[10 * 64 + ch0]), s1 = *reinterpret_cast<const f32x4*>(&s_c1[10 * 64 + ch0 + 4]);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(&s_c1[11 * 64 + ch0]), t1 = *reinterpret_cast<const f32x4*>(&s_c1[11 * 64 + ch0 + 4]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o8[i][j] = __fadd_rn(__fmul_rn(o8[i][j], s0[j]), t0[j]);
                    o8[i][4 + j] = __fadd_rn(__fmul_rn(o8[i][4 + j], s1[j]), t1[j]);
                }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int hp = hp0 + 256 * i;
            if (hp >= NPIX) continue;
            const int py = hp / (TW + 2), px = hp - py * (TW + 2);
            const int yy = y0 + py - 1, xx = x0 + px - 1;
            const bool inside = yy >= 0 && yy < c1.H && xx >= 0 && xx < c1.W;
            f16x8 hi, lo;
            split8(inside ? o8[i] : f32x8{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, hi, lo);
            const int P = py * PW + px, sw4 = (P >> 2) & 3;
            s_in[buf][P * 4 + ((2 * gg) ^ sw4)] = __builtin_bit_cast(u32x4, hi);
            s_in[buf][P * 4 + ((2 * gg + 1) ^ sw4)] = __builtin_bit_cast(u32x4, lo);
        }
    };

    const int wy0 = (wave / WX) * 4, wx0 = (wave % WX) * 16;
    const int i32 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
    const int P0 = (wy0 + ((i32 >> 1) & 1)) * PW + wx0 + 2 * (i32 >> 2) + (i32 & 1);
    const int wsw = (i32 >> 2) & 3;
    const int a_hi = i32 * 4 + ((2 * h) ^ wsw), a_lo = i32 * 4 + ((2 * h + 1) ^ wsw);

    auto multiply = [&](int buf) {
        multiply_chunk_pipelined<PW>(reinterpret_cast<const f16x8*>(s_in[buf]), reinterpret_cast<const f16x8*>(s_w), P0, h, a_hi, a_lo, acc);
    };

    int id = blockIdx.x;
    if (id >= total) return;
    Tile cur = tile_of(id);
    load_gray(cur);
    load_w(0);
    store_gray();
    __syncthreads();          // gray tile, tap table and bias table ready
    produce(0, 0, cur.y0, cur.x0);
    store_w();
    __syncthreads();
    for (;;) {
        const bool more = id + (int)gridDim.x < total;
        const Tile nxt = tile_of(more ? id + (int)gridDim.x : id);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int c = 0; c < 4; ++c) {  // conv1a has 64 output channels = 4 chunks; chunk parity == buffer
            const int buf = c & 1;
            const bool feed = c < 3 || more;  // the stage after this one exists
            if (feed) load_w((c + 1) & 3);
            if (c == 1 && more) load_gray(nxt);
            const int py0 = c < 3 ? cur.y0 : nxt.y0, px0 = c < 3 ? cur.x0 : nxt.x0;
            if (wu < 4 && feed) produce((c + 1) & 3, buf ^ 1, py0, px0);
            multiply(buf);
            if (wu >= 4 && feed) produce((c + 1) & 3, buf ^ 1, py0, px0);
            __syncthreads();      // everyone is done with s_w, s_in[buf] and (after chunk 2's produce) s_g
            if (feed) store_w();
            if (c == 2 && more) store_gray();
            __syncthreads();
        }
        conv_epilogue<GENERIC>(p, acc, s_bias, cur.b, cur.y0 + wy0, cur.x0 + wx0, 0, lane);
        if (!more) break;
        cur = nxt;
        id += gridDim.x;
    }
}

// conv1a + conv1b + pool on the stage machinery of conv3x3_dma_kernel.  Differences to the register-staged kernel above:
//   * the weight rows of conv1b go through the same four-slot LDS-DMA ring, one barrier per filter row -- the two barriers
//     and the register -> LDS weight copy per chunk are gone; all eight waves multiply all the time;
//   * round 3: conv1a itself runs on the matrix pipe.  It is a K = 9 contraction (superpoint.py:209): with the split operands
//     the three products hi.hi + hi.lo + lo.hi of all nine taps are 27 of the 32 k-slots of ONE v_mfma_f32_16x16x32_f16 --
//     A = the chunk's 16 channels ([W_hi taps 0-7 | W_hi taps 0-7 | W_lo taps 0-7 | W_hi8, W_hi8, W_lo8, 0...]), B = 16 halo
//     pixels ([g_hi | g_lo | g_hi | g_hi8, g_lo8, g_hi8, 0...]), C = the bias.  The B fragment of a pixel group (its 3x3 gray
//     window as fp16 halves) does not depend on the chunk: a wave builds the fragments of its five pixel groups ONCE per image
//     tile from the packed (hi | lo << 16) gray tile and keeps them in 20 registers; per chunk it issues five 16-cycle MFMAs and
//     five finishes (ReLU / zero outside the image as one v_med3, split, two 8-byte LDS writes: a lane owns 4 channels of one
//     pixel).  Round 2 computed the taps with packed fp32 FMAs between the MFMAs: 6.3 VALU instructions per MFMA, 17 % LDS bank
//     conflicts from the tap-table and gray-pair reads, 55 % MFMA-busy (profiles/r02_z_f16x3_sq_counters.json);
//   * the gray tile of the next image tile is fetched during chunk 0 (six unconditional buffer loads per thread: a
//     one-channel image is read three times so that the VMEM count does not depend on C), converted, split and stored into
//     the second gray buffer during chunk 1; the next tile's fragments are built in the third stage of chunk 2 (the last
//     produce step of this tile is over by then) and used by the produce steps of chunk 3.
// conv1b's accumulation order is that of conv1ab_f16x3_kernel; conv1a differs from it by the arithmetic (split-fp16 products on
// the matrix pipe instead of fp32 FMAs): same error class as every other layer of this mode (tests/test_gpu_conv_dma.py).
__device__ __forceinline__ f32x4 mfma16x16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(512, 2) void conv1ab_mfma_kernel(const ConvF16Args p, const Conv1aArgs c1) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WX = 2, TH = 16, TW = 32;
    constexpr int PW = ((TW + 2 + 7) / 16) * 16 + 8;
    constexpr int NPIX = (TH + 2) * (TW + 2);
    constexpr int IN_SZ = (TH + 2) * PW * 4;
    constexpr int WS = 3 * 64 * 4, WR = 4, NWA = 2, WLP = WS / (8 * NWA);  // two 48-lane weight pieces per wave and stage
    constexpr int E = 4;                              // pooled output: four stores per wave and tile
    constexpr int NGL = 6;                            // gray-tile loads per thread and tile (2 pixels x 3 channels)
    constexpr int GH = TH + 4, GW = TW + 4, GP = GW + 1, GSZ = ((GH * GP + 3) / 4) * 4;
    static_assert(GH * GW <= 512 + 256, "second gray pixel only on threads 0..255");
    constexpr int NG = (NPIX + 15) / 16, NF = (NG + 7) / 8;  // 39 groups of 16 halo pixels, five per wave
    constexpr int RAW = 3 * 512 + 3 * 256 + 64;       // raw image values: first pixel of every thread, second pixel of waves 0-3, a dump slot
    // ONE LDS array: [2 halo tiles][4 weight stages][2 gray tiles][conv1a A fragments][conv1a bias][conv1b bias][raw image values of the next gray tile]
    __shared__ u32x4 s_all[2 * IN_SZ + WR * WS + 2 * GSZ / 4 + 4 * 64 + 64 / 4 + 64 / 4 + RAW / 4];
    static_assert(sizeof(s_all) <= 160 * 1024, "LDS");
    u32x4* const s_in = s_all;
    u32x4* const s_w = s_all + 2 * IN_SZ;
    unsigned* const s_g = reinterpret_cast<unsigned*>(s_all + 2 * IN_SZ + WR * WS);  // packed fp16 halves of the gray value: hi | lo << 16
    u32x4* const s_a = reinterpret_cast<u32x4*>(s_g + 2 * GSZ);                      // [chunk][lane]: A operand of the conv1a MFMA
    float* const s_b1 = reinterpret_cast<float*>(s_a + 4 * 64);                      // conv1a bias
    float* const s_bias = s_b1 + 64;
    float* const s_raw = s_bias + 64;                 // written by LDS-DMA, read back by the same thread

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int tiles = tiles_x * tiles_y, total = tiles * p.B;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), 0, 0x7fffffff, 0x00020000);

    struct Tile { int b, y0, x0; };
    auto tile_of = [&](int id) {
        int logical = id;
        if ((total & 7) == 0) logical = (id & 7) * (total >> 3) + (id >> 3);
        const int tl = logical % tiles;
        return Tile{logical / tiles, (tl / tiles_x) * TH, (tl % tiles_x) * TW};
    };

    // A operand of the conv1a MFMA, per chunk: lane l = row (channel) l & 15, k-block q = l >> 4 (8 halves):
    //   q = 0, 1: W_hi of taps 0-7;  q = 2: W_lo of taps 0-7;  q = 3: W_hi8, W_hi8, W_lo8, 0, 0, 0, 0, 0
    if (t < 4 * 64) {
        const int ch = (t >> 6) * 16 + (t & 15), q = (t >> 4) & 3;
        f16x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = q < 3 ? e : 8;
            const float w = c1.w[ch * 9 + tap];
            const _Float16 wh = (_Float16)w, wl = (_Float16)(w - (float)wh);
            a[e] = q < 2 ? wh : q == 2 ? wl : e < 2 ? wh : e == 2 ? wl : (_Float16)0.f;
        }
        s_a[t] = __builtin_bit_cast(u32x4, a);
    }
    if (t < 64) {
        s_b1[t] = c1.bias[t];
        s_bias[t] = t < p.Cout ? p.bias[t] : 0.f;
    }

    // gray tile (RGB -> gray as superpoint.py:204-206), zero outside the image: thread t owns tile pixels t and t + 512
    // (the raw values travel global -> LDS by DMA and are converted a chunk later: held in registers across a chunk they were spilled)
    const int HWb = c1.H * c1.W * 4;
    auto gray_rsrc = [&](const Tile& tl) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c1.image) + (long long)tl.b * c1.C * c1.H * c1.W, 0, c1.C * HWb, 0x00020000);
    };
    auto gray_load = [&](const __amdgpu_buffer_rsrc_t& rs, const Tile& tl, int j) {  // load j of NGL: pixel j / 3, channel j % 3
        const int ch = c1.C == 3 ? j % 3 : 0;
        const int i = wu * 64 + fresh_lane() + 512 * (j / 3), gy = i / GW, gx = i - gy * GW;  // recomputed per load: nothing kept live
        const int yy = tl.y0 + gy - 2, xx = tl.x0 + gx - 2;
        const bool ok = i < GH * GW && (unsigned)yy < (unsigned)c1.H && (unsigned)xx < (unsigned)c1.W;
        // first pixel: [channel][512 threads]; second pixel (tile pixels 512..719): waves 0-3 only, the others write zeros into a dump slot
        float* const dst = j < 3 ? s_raw + j * 512 + wu * 64 : wu < 4 ? s_raw + 3 * 512 + (j - 3) * 256 + wu * 64 : s_raw + 3 * 512 + 3 * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 4, ok ? (yy * c1.W + xx) * 4 : (int)0x80000000, ch * HWb, 0, 0);
    };
    auto gray_store = [&](int gb) {
        const int tt = wu * 64 + fresh_lane();  // addresses recomputed here, not hoisted out of the tile loop (see p_tap)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tt + 512 * k, gy = i / GW, gx = i - gy * GW;
            if (k == 1 && tt >= 256) break;  // wave-uniform: tile pixels 512.. belong to waves 0-3
            const float* const src = k == 0 ? s_raw + tt : s_raw + 3 * 512 + tt;
            const int cs = k == 0 ? 512 : 256;
            const float r0 = src[0], r1 = src[cs], r2 = src[2 * cs];
            const float v = c1.C == 3 ? __fadd_rn(__fadd_rn(__fmul_rn(r0, 0.299f), __fmul_rn(r1, 0.587f)), __fmul_rn(r2, 0.114f)) : r0;
            const _Float16 vh = (_Float16)v, vl = (_Float16)(v - (float)vh);
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, vh) | ((unsigned)__builtin_bit_cast(unsigned short, vl) << 16);
            if (i < GH * GW) s_g[gb * GSZ + gy * GP + gx] = pk;
        }
    };

    // conv1a on the matrix pipe.  Per image tile and wave: NF = 5 groups of 16 halo pixels (group g = wu + 8 i covers the linear halo
    // indices 16 g .. 16 g + 15); lane l = pixel l & 15 of the group, k-block q = l >> 4.
    f16x8 pf[NF];      // B fragments (the pixel's 3x3 gray window as fp16 halves, module header), rebuilt once per tile
    int wadr[NF];      // byte offset of the lane's 8-byte hi piece inside a halo buffer (its lo piece: ^ 16); a padding slot for pixels past the tile
    float lim[NF];     // +inf inside the image, 0 outside: relu + zero padding of conv1b's input as one v_med3_f32(x, 0, lim)
    unsigned gw[9];    // the nine packed gray values of the group under construction (between the two halves of a build step)
    const int q4 = lane >> 4;
    auto build_a = [&](int i, int gb, const Tile& tl) {  // first half: request the window, derive the write address and the inside flag
        const int hp = 16 * (wu + 8 * i) + (fresh_lane() & 15);
        const int hc = min(hp, NPIX - 1), py = hc / (TW + 2), px = hc - py * (TW + 2);
        const unsigned* g = s_g + gb * GSZ + py * GP + px;
#pragma unroll
        for (int e = 0; e < 9; ++e) gw[e] = g[(e / 3) * GP + (e % 3)];
        const int yy = tl.y0 + py - 1, xx = tl.x0 + px - 1;
        lim[i] = (unsigned)yy < (unsigned)c1.H && (unsigned)xx < (unsigned)c1.W ? __builtin_inff() : 0.f;
        // pixels past the tile (last group only) are parked in the pitch padding of halo row 0 (columns TW + 2 .. PW - 1 are never read)
        const int P = hp < NPIX ? py * PW + px : TW + 2 + (hp & 3);
        wadr[i] = (P * 4 + ((q4 & 2) ^ ((P >> 2) & 3))) * 16 + (q4 & 1) * 8;
    };
    auto build_b = [&](int i) {  // second half: pack the halves.  q = 0, 2: hi halves of taps 0-7; q = 1: lo halves; q = 3: hi8, lo8, hi8, 0...
        const unsigned sel = q4 == 1 ? 0x07060302u : 0x05040100u;
        u32x4 f;
#pragma unroll
        for (int m = 0; m < 4; ++m) f[m] = __builtin_amdgcn_perm(gw[2 * m + 1], gw[2 * m], sel);
        if (q4 == 3) f = u32x4{gw[8], gw[8] & 0xffffu, 0u, 0u};
        pf[i] = __builtin_bit_cast(f16x8, f);
    };
    // Per chunk: request the A fragment and the bias, then MFMA i (16 channels x 16 pixels, all three split products of all nine taps)
    // and, two steps later, its finish.  Nine steps, placed every third filler slot of the first two stages of the previous chunk.
    f16x8 afr;
    f32x4 bia, da, db;
    auto fin = [&](const f32x4& d, int i, int buf) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(d[r], 0.f, lim[i]);
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hi[r] = (_Float16)v[r];
            lo[r] = (_Float16)(v[r] - (float)hi[r]);
        }
        // Written as asm: for a C++ store without alias-scope metadata hipcc's waitcnt pass assumes it may alias the in-flight LDS-DMA
        // writes (the weight ring) and puts `s_waitcnt vmcnt(0)` in front of it -- a full drain of the DMA queue once per finish.
        // The halo buffers and the ring are disjoint; the writes are ordered before their readers by the `lgkmcnt(0)` + barrier that
        // opens every stage.  (s_in is the start of the workgroup's LDS: offset = byte address.)
        const int a_hi = wadr[i], a_lo = wadr[i] ^ 16;
        const unsigned long long d_hi = __builtin_bit_cast(unsigned long long, hi), d_lo = __builtin_bit_cast(unsigned long long, lo);
        if (buf == 0) {
            asm volatile("ds_write_b64 %0, %1" ::"v"(a_hi), "v"(d_hi) : "memory");
            asm volatile("ds_write_b64 %0, %1" ::"v"(a_lo), "v"(d_lo) : "memory");
        } else {
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_hi), "v"(d_hi), "n"(IN_SZ * 16) : "memory");
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a_lo), "v"(d_lo), "n"(IN_SZ * 16) : "memory");
        }
    };
    auto p_step = [&](int pos, int chn, int buf) {
        switch (pos) {
            case 0:
                afr = __builtin_bit_cast(f16x8, s_a[chn * 64 + fresh_lane()]);
                bia = *reinterpret_cast<const f32x4*>(s_b1 + chn * 16 + 4 * (fresh_lane() >> 4));
                break;
            case 1: da = mfma16x16(afr, pf[0], bia); break;
            case 2: db = mfma16x16(afr, pf[1], bia); break;
            case 3: fin(da, 0, buf); da = mfma16x16(afr, pf[2], bia); break;
            case 4: fin(db, 1, buf); db = mfma16x16(afr, pf[3], bia); break;
            case 5: fin(da, 2, buf); da = mfma16x16(afr, pf[4], bia); break;
            case 6: fin(db, 3, buf); break;
            case 7: fin(da, 4, buf); break;
            default: break;
        }
    };
    static_assert(NF == 5, "p_step is written out for five pixel groups per wave");

    int w_off[NWA];  // weight pieces NWA wu .. NWA wu + NWA - 1 of a stage: LDS slots [WLP j, WLP j + WLP)
#pragma unroll
    for (int i = 0; i < NWA; ++i) {
        const int g = (wu * NWA + i) * WLP + min(lane, WLP - 1), row = g >> 2, tl = row >> 6, n = row & 63;
        w_off[i] = ((tl * p.CoutPad + n) * 4 + ((g & 3) ^ ((n >> 2) & 3))) * 16;
    }
    auto dma_w = [&](int c, int ky, int slot, int i) {
        const int soff = (((c * 9 + 3 * ky) * p.CoutPad) * 4) * 16;
        if (lane < WLP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, s_w + slot * WS + (wu * NWA + i) * WLP, 16, w_off[i], soff, 0, 0);
    };

    const int wy0 = (wu / WX) * 4, wx0 = (wu % WX) * 16;
    const int i32 = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
    const int P0 = (wy0 + ((i32 >> 1) & 1)) * PW + wx0 + 2 * (i32 >> 2) + (i32 & 1);
    int b_hi[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) b_hi[kx] = (P0 + kx) * 4 + ((2 * h) ^ (((P0 + kx) >> 2) & 3));
    const int a_hi = i32 * 4 + ((2 * h) ^ ((i32 >> 2) & 3));

    f16x8 fr[2][8];
    auto read_frag = [&](int set, int buf, int slot, int ky, int kx, int i) {
        const f16x8* si = reinterpret_cast<const f16x8*>(s_in + buf * IN_SZ) + ky * PW * 4;
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w + slot * WS) + kx * 64 * 4;
        const int bx = b_hi[kx] ^ ((ky & 1) << 1);
        switch (i) {
            case 0: fr[set][0] = sw[a_hi]; break;
            case 1: fr[set][1] = si[bx]; break;
            case 2: fr[set][2] = si[bx + 2 * PW * 4]; break;
            case 3: fr[set][3] = sw[a_hi + 32 * 4]; break;
            case 4: fr[set][4] = si[bx ^ 1]; break;
            case 5: fr[set][5] = si[(bx ^ 1) + 2 * PW * 4]; break;
            case 6: fr[set][6] = sw[a_hi ^ 1]; break;
            default: fr[set][7] = sw[(a_hi ^ 1) + 32 * 4]; break;
        }
    };

    int id = blockIdx.x;
    if (id >= total) return;
    Tile cur = tile_of(id);
    // prologue: gray tile -> s_g[0]; conv1a chunk 0 -> halo buffer 0; weight stages 0..2 -> slots 0..2
    {
        const __amdgpu_buffer_rsrc_t rs = gray_rsrc(cur);
#pragma unroll
        for (int j = 0; j < NGL; ++j) gray_load(rs, cur, j);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < NWA; ++i) dma_w(0, ky, ky, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread reads back what its own wave's DMA wrote
        gray_store(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // gray tile, A fragments, bias tables
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        build_a(i, 0, cur);
        build_b(i);
    }
#pragma unroll
    for (int pos = 0; pos <= 7; ++pos) p_step(pos, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) read_frag(0, 0, 0, 0, 0, i);

    int sidx = 0, gb = 0;
    bool e_now = false, e_prev = false;
    for (;;) {
        const bool more = id + (int)gridDim.x < total;
        const Tile nxt = tile_of(more ? id + (int)gridDim.x : id);
        const __amdgpu_buffer_rsrc_t rs_g = gray_rsrc(nxt);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        auto chunk = [&](auto c_c) {
            constexpr int C = decltype(c_c)::value, PAR = C & 1;
            const bool has_next = C < 3 || more;        // the chunk after this one exists
            constexpr int NC = (C + 1) & 3;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // my VMEM instructions up to the stage before last are done (issued since: the previous stage's and, around a tile
                // boundary, the E output stores), then everyone's
                if (ky == 0) {
                    if (C == 1) {   // previous stage (0,2): weight pieces + the gray loads of the next tile
                        if (more) wait_vm_c<NWA + NGL>(); else wait_vm_c<NWA>();
                    } else if (e_now) wait_vm_c<NWA + E>();
                    else wait_vm_c<NWA>();
                } else if (!has_next) {
                    wait_vm_c<0>();
                } else if (ky == 1 && e_prev) {
                    wait_vm_c<NWA + E>();
                } else {
                    wait_vm_c<NWA>();
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                e_prev = e_now;
                e_now = false;
                const int slot = sidx & 3, slot1 = (sidx + 1) & 3, slot3 = (sidx + 3) & 3;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int T = 3 * ky + kx, S = (T + PAR) & 1;
                    int fslot = 0;
                    // fillers of tap kx: slots 0-7 the fragment reads of the next tap; slots 2, 5, 8, 11 are the 24 conv1a slots of the first two
                    // stages: every third one runs a step of the next chunk's conv1a (request, five MFMAs, five finishes); in the third stage of
                    // chunk 2 the same slots build the next tile's five B fragments (two halves each); slots 9, 10 the VMEM work: weight pieces of
                    // the stage three ahead (class A: taps 0, 0, 1) and, in the third stage of chunk 0, the six gray-tile loads; the gray store in
                    // the third stage of chunk 1
                    auto filler = [&]() {
                        const int f = fslot++;
                        if (f < 8) {
                            if (T < 8) read_frag(1 - S, PAR, kx == 2 ? slot1 : slot, (T + 1) / 3, (T + 1) % 3, f);
                            else read_frag(1 - S, 1 - PAR, slot1, 0, 0, f);
                        }
                        if (f == 2 || f == 5 || f == 8 || f == 11) {
                            const int sl = kx * 4 + (f - 2) / 3;  // 0..11 inside the stage
                            if (ky < 2) {
                                const int s24 = ky * 12 + sl;
                                if (has_next && s24 % 3 == 0) p_step(s24 / 3, NC, 1 - PAR);
                            } else if (C == 2 && more && sl < 2 * NF) {
                                if (sl & 1) build_b(sl >> 1);
                                else build_a(sl >> 1, gb ^ 1, nxt);
                            }
                        }
                        if (f == 9 || f == 10) {
                            const int d = kx * 2 + (f - 9);  // 0..5
                            if (d < NWA && has_next) dma_w(NC, ky, slot3, d);
                            if (C == 0 && ky == 2 && more) gray_load(rs_g, nxt, d);
                            if (C == 1 && ky == 2 && d == 0 && more) gray_store(gb ^ 1);
                        }
                        PIN();
                    };
                    PIN();
                    acc[0][0] = mfma16(fr[S][0], fr[S][1], acc[0][0]); filler();
                    acc[0][1] = mfma16(fr[S][0], fr[S][2], acc[0][1]); filler();
                    acc[1][0] = mfma16(fr[S][3], fr[S][1], acc[1][0]); filler();
                    acc[1][1] = mfma16(fr[S][3], fr[S][2], acc[1][1]); filler();
                    acc[0][0] = mfma16(fr[S][0], fr[S][4], acc[0][0]); filler();
                    acc[0][1] = mfma16(fr[S][0], fr[S][5], acc[0][1]); filler();
                    acc[1][0] = mfma16(fr[S][3], fr[S][4], acc[1][0]); filler();
                    acc[1][1] = mfma16(fr[S][3], fr[S][5], acc[1][1]); filler();
                    acc[0][0] = mfma16(fr[S][6], fr[S][1], acc[0][0]); filler();
                    acc[0][1] = mfma16(fr[S][6], fr[S][2], acc[0][1]); filler();
                    acc[1][0] = mfma16(fr[S][7], fr[S][1], acc[1][0]); filler();
                    acc[1][1] = mfma16(fr[S][7], fr[S][2], acc[1][1]); filler();
                }
                ++sidx;
            }
        };
        chunk(std::integral_constant<int, 0>{});
        chunk(std::integral_constant<int, 1>{});
        chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{});
        {
            const int ln = fresh_lane();
            const int img_bytes = (p.H >> 1) * (p.W >> 1) * p.Cout * 4;
            BufStore st{__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (long long)cur.b * img_bytes, 0, img_bytes, 0x00020000)};
            epilogue_pool_sh8_st(acc, st, s_bias, p.winv, p.H, p.W, p.Cout, cur.y0 + wy0, cur.x0 + wx0, 0, ln);
            e_now = true;
        }
        if (!more) break;
        cur = nxt;
        gb ^= 1;
        id += gridDim.x;
    }
#endif
}

// conv1a writing SH8: 8 lanes per pixel, 8 output channels each (one hi + one lo 16-byte piece).
__global__ __launch_bounds__(256) void conv1a_sh8_kernel(const Conv1aArgs p) {
    __shared__ float s_w[64 * 9 + 64 * 3];
    for (int i = threadIdx.x; i < 64 * 9; i += 256) s_w[i] = p.w[i];
    for (int i = threadIdx.x; i < 64; i += 256) {
        s_w[576 + i] = p.bias[i];
        s_w[640 + i] = p.post_scale ? p.post_scale[i] : 1.f;
        s_w[704 + i] = p.post_shift ? p.post_shift[i] : 0.f;
    }
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid >> 3;
    const int cg = (int)(gid & 7);
    const long long HW = (long long)p.H * p.W;
    if (pix >= (long long)p.B * HW) return;
    const int b = (int)(pix / HW);
    const int rem = (int)(pix - (long long)b * HW);
    const int y = rem / p.W, x = rem - y * p.W;
    const float* img = p.image + (long long)b * p.C * HW;
    float g[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            float v = 0.f;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
                const long long o = (long long)yy * p.W + xx;
                if (p.C == 3)
                    v = __fadd_rn(__fadd_rn(__fmul_rn(img[o], 0.299f), __fmul_rn(img[HW + o], 0.587f)), __fmul_rn(img[2 * HW + o], 0.114f));
                else
                    v = img[o];
            }
            g[ky * 3 + kx] = v;
        }
    f32x8 o8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) a = fmaf(s_w[c * 9 + k], g[k], a);
        a += s_w[576 + c];
        a = fmaxf(a, 0.f);
        if (p.post_scale) a = __fadd_rn(__fmul_rn(a, s_w[640 + c]), s_w[704 + c]);
        o8[j] = a;
    }
    f16x8 hi, lo;
    split8(o8, hi, lo);
    f16x8* o = reinterpret_cast<f16x8*>(p.out) + pix * 16 + cg * 2;
    o[0] = hi;
    o[1] = lo;
}

}  // namespace

static int cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

void launch_conv3x3_f16x3(const ConvF16Args& a, hipStream_t s) {
    auto waste = [&](int th, int tw) {
        long long ty = (a.H + th - 1) / th, tx = (a.W + tw - 1) / tw;
        return ty * th * tx * tw;
    };
    const long long w42 = waste(16, 32), w81 = waste(32, 16);
    const int nslices = (a.Cout + 63) / 64;
    const long long w22 = waste(8, 32);
    // fewest wasted tile pixels wins; on a tie the 4-wave 8x32 tile (two workgroups per CU, whose
    // staging / epilogue phases overlap each other's MFMA phases) is preferred -- measured 7-11 % faster
    // than one 8-wave workgroup per CU on the 240x320 and 120x160 maps.
    const bool generic = !(a.relu && !a.out_f32 && !a.post_scale);  // BatchNorm affine / fp32 output / no ReLU
    // Hot layout (ReLU, SH8 out): the LDS-DMA kernel, 16x32 or 32x16 pixel tiles, one 8-wave workgroup per CU.
    // a.staged (debug request "sp.conv_staged", tests/test_gpu_conv_dma.py) keeps the register-staged kernel below.
    const bool staged_only = a.staged == 1;
    const long long out_img_bytes = (long long)(a.pool ? a.H / 2 : a.H) * (a.pool ? a.W / 2 : a.W) * a.Cout * 4;
    if (!generic && !staged_only && a.Cin % 32 == 0 && a.Cout % 4 == 0 && a.Cout <= CONV_DMA_MAXC && out_img_bytes < (1ll << 31) &&
        (!a.pool || (a.H % 2 == 0 && a.W % 2 == 0))) {
        const bool wide = w42 <= w81;
        const long long items = (wide ? (long long)((a.H + 15) / 16) * ((a.W + 31) / 32) : (long long)((a.H + 31) / 32) * ((a.W + 15) / 16)) * nslices * a.B;
        const dim3 grid((unsigned)std::min<long long>(items, cu_count()));
        // a.staged == 2 (debug request "sp.conv_dma32"): the 32x32x16 LDS-DMA kernel instead of the tap-pair kernel (A/B runs, tests)
        if (a.staged != 2) {
            if (wide) {
                if (a.pool) hipLaunchKernelGGL((conv3x3_pair_kernel<4, 2, true>), grid, dim3(CT), 0, s, a);
                else hipLaunchKernelGGL((conv3x3_pair_kernel<4, 2, false>), grid, dim3(CT), 0, s, a);
            } else {
                if (a.pool) hipLaunchKernelGGL((conv3x3_pair_kernel<8, 1, true>), grid, dim3(CT), 0, s, a);
                else hipLaunchKernelGGL((conv3x3_pair_kernel<8, 1, false>), grid, dim3(CT), 0, s, a);
            }
            return;
        }
        if (wide) {
            if (a.pool) hipLaunchKernelGGL((conv3x3_dma_kernel<4, 2, true>), grid, dim3(CT), 0, s, a);
            else hipLaunchKernelGGL((conv3x3_dma_kernel<4, 2, false>), grid, dim3(CT), 0, s, a);
        } else {
            if (a.pool) hipLaunchKernelGGL((conv3x3_dma_kernel<8, 1, true>), grid, dim3(CT), 0, s, a);
            else hipLaunchKernelGGL((conv3x3_dma_kernel<8, 1, false>), grid, dim3(CT), 0, s, a);
        }
        return;
    }
    // persistent grid: as many workgroups as fit on the chip at once (LDS: two 4-wave, one 8-wave per CU)
    auto grid_for = [&](long long items, int per_cu) { return dim3((unsigned)std::min<long long>(items, (long long)cu_count() * per_cu)); };
    if (w22 <= w42 && w22 <= w81) {
        const long long items = (long long)((a.H + 7) / 8) * ((a.W + 31) / 32) * nslices * a.B;
        if (generic) hipLaunchKernelGGL((conv3x3_f16x3_kernel<2, 2, 1, true>), grid_for(items, 2), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_f16x3_kernel<2, 2, 1, false>), grid_for(items, 2), dim3(256), 0, s, a);
    } else if (w42 <= w81) {
        const long long items = (long long)((a.H + 15) / 16) * ((a.W + 31) / 32) * nslices * a.B;
        if (generic) hipLaunchKernelGGL((conv3x3_f16x3_kernel<4, 2, 1, true>), grid_for(items, 1), dim3(CT), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_f16x3_kernel<4, 2, 1, false>), grid_for(items, 1), dim3(CT), 0, s, a);
    } else {
        const long long items = (long long)((a.H + 31) / 32) * ((a.W + 15) / 16) * nslices * a.B;
        if (generic) hipLaunchKernelGGL((conv3x3_f16x3_kernel<8, 1, 1, true>), grid_for(items, 1), dim3(CT), 0, s, a);
        else hipLaunchKernelGGL((conv3x3_f16x3_kernel<8, 1, 1, false>), grid_for(items, 1), dim3(CT), 0, s, a);
    }
}

void launch_conv1ab_f16x3(const ConvF16Args& a, const Conv1aArgs& c1, hipStream_t s) {
    const long long items = (long long)((a.H + 15) / 16) * ((a.W + 31) / 32) * a.B;  // Cout = 64: one cout slice
    dim3 grid((unsigned)std::min<long long>(items, cu_count()));                     // persistent, one workgroup per CU (135 KB of LDS)
    const long long out_img_bytes = (long long)(a.H / 2) * (a.W / 2) * a.Cout * 4;
    // a.staged (debug requests "sp.conv_staged" / "sp.conv1ab_staged"): the register-staged kernel (conv1a in fp32 on the VALU)
    if (a.staged != 1 && !a.post_scale && !c1.post_scale && a.Cout == 64 && a.H % 2 == 0 && a.W % 2 == 0 && out_img_bytes < (1ll << 31) &&
        (long long)c1.C * c1.H * c1.W * 4 < (1ll << 31)) {
        hipLaunchKernelGGL(conv1ab_mfma_kernel, grid, dim3(CT), 0, s, a, c1);
        return;
    }
    if (a.post_scale) hipLaunchKernelGGL(conv1ab_f16x3_kernel<true>, grid, dim3(CT), 0, s, a, c1);
    else hipLaunchKernelGGL(conv1ab_f16x3_kernel<false>, grid, dim3(CT), 0, s, a, c1);
}

void launch_conv1a_sh8(const Conv1aArgs& a, hipStream_t s) {
    const long long threads = (long long)a.B * a.H * a.W * 8;
    hipLaunchKernelGGL(conv1a_sh8_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
}
