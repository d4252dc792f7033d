// The whole feed-forward half of a LightGlue block in ONE kernel (MG_PREC_F16X3):
//
//     x <- x + W3 . gelu(LayerNorm(W0f . [x | ctx] + b0)) + b3          (lightglue.py:135-139,156-157 / :185-189,213-214)
//
// (W0f = ffn.0 with the attention out-projection folded in, api.hip: prep_ffn0_folded).  As three launches (GEMM, ln_gelu,
// GEMM) the 512-wide hidden row went through HBM twice as fp32 / SH8 (4 x 134 MB per block at 65 536 tokens, plus a
// VALU-only pass); here it never leaves the CU.  SuperGlue's / GlueStick's  x + W3 . relu(W0f . [x | ctx] + b0) + b3
// (BatchNorm folded at upload, superglue.py:139-141) is the ACT_RELU instantiation of the same kernel.
//
// One workgroup (8 waves) owns a tile of 128 tokens:
//   phase 1  H[512 x 128] = W0f . A^T : wave w accumulates hidden features [64w, 64w+64) x 128 tokens (8 MFMA tiles, 128
//            accumulator registers); K = 512 in stages of 16, token rows (HBM) and weight rows (L2) arrive by LDS-DMA
//            into rings of their own depth, issued by different waves (vmcnt retires in order), as in gemm_f16.hip.
//   phase 2  bias, LayerNorm over the 512 features of a token (two passes: sum, then sum of squared deviations; the
//            8 waves' partial sums meet in LDS and are added in wave order), GELU(erf) -- all on the accumulator registers.
//   phase 3  OUT[256 x 128] = W3 . H'^T : the hidden features are now the contraction index, and they are spread over the
//            waves.  They go through LDS one k-step (16 features x 128 tokens, 8 KB) at a time, in a ring of 8 slots: a
//            wave's 64 x 128 block is four k-steps ("quarters"); k-step j of the contraction is quarter j/8 of wave j%8, so
//            every wave writes its first quarter before the loop and one more quarter every eighth iteration, seven
//            iterations ahead of its use, under the other waves' MFMAs -- and a quarter's 32 registers are dead once written,
//            which keeps the loop at 96 + 64 accumulator registers.  Wave (wf, wt) accumulates output features
//            [64 wf, +64) x tokens [64 wt, +64).  An accumulator register block of a lane is used as it stands:
//            registers 8s..8s+7 of lane (token j, half h) become the 16-byte fragment of half h -- which fixes the order of
//            the contraction index: position 8h + 4a + e of k-step j is hidden feature 64 (j%8) + 16 (j/8) + 8a + 4h + e;
//            W3 is tiled in that order at upload (api.hip: kperm).
//   epilogue as in gemm_f16.hip: + b3, + residual, fp32 row and its SH8 copy, quad-transposed full-line stores.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int FT = 512;                  // threads: 8 waves
constexpr int TMF = 128;                 // tokens per workgroup
constexpr int HID = 512, DOUT = 256;     // hidden width, output width
constexpr int TS1 = 5, WS1 = 3;          // phase-1 rings: token stages (8 KB) / weight stages (32 KB)
constexpr int WS3 = 4;                   // phase-3 weight ring (16 KB stages)

constexpr int TOK_STAGE = TMF * 4;       // u32x4 per stage
constexpr int W1_STAGE = HID * 4;
constexpr int W3_STAGE = DOUT * 4;
constexpr int HR = 8;                    // H' ring: k-steps of 128 tokens x 64 B
constexpr int RING1 = TS1 * TOK_STAGE + WS1 * W1_STAGE;  // u32x4
constexpr int RING3 = HR * TOK_STAGE + WS3 * W3_STAGE;
constexpr int RING = RING1 > RING3 ? RING1 : RING3;
constexpr int NPAR = 3 * HID + DOUT;     // b0 | gamma | beta | b3
constexpr int NSTAT = 2 * 8 * TMF;       // partial sums / partial squared deviations per wave and token

__device__ __forceinline__ int swz(int row, int q) { return row * 4 + (q ^ ((row >> 2) & 3)); }  // 64-byte rows

template <int N>
__device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}
// wait until all but `stages` younger stages (of `per` pieces each) of this wave's DMA stream have landed
template <int per>
__device__ __forceinline__ void wait_stages(int ahead) {
    if (ahead >= 3) { if constexpr (per * 3 <= 16) wait_vm<per * 3>(); else wait_vm<16>(); }
    else if (ahead == 2) wait_vm<per * 2>();
    else if (ahead == 1) wait_vm<per>();
    else wait_vm<0>();
}

// LDS writes of this wave are complete, then the workgroup barrier (the raw s_barrier does not wait for them; DMA stays in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

enum { ACT_LN_GELU = 0, ACT_RELU = 1 };

// erf(x), fp32 (max error 9e-8 against fp64 over [-6, 6]): the two minimax branches of ROCm's ocml erff (|x| < 1: odd
// polynomial; else 1 - exp(-poly)), both evaluated and selected -- no divergent branch inside the 128-element register loop.
__device__ __forceinline__ float erf_select(float x) {
    const float ax = fabsf(x), t = ax * ax;
    float p = -0x1.268bc2p-11f;
    p = fmaf(t, p, 0x1.420828p-8f);
    p = fmaf(t, p, -0x1.b5937p-6f);
    p = fmaf(t, p, 0x1.ce077cp-4f);
    p = fmaf(t, p, -0x1.81266p-2f);
    p = fmaf(t, p, 0x1.06eba0p-3f);
    const float r1 = fmaf(ax, p, ax);
    float q = 0x1.1d3156p-16f;
    q = fmaf(ax, q, -0x1.8d129p-12f);
    q = fmaf(ax, q, 0x1.f9a6d2p-9f);
    q = fmaf(ax, q, -0x1.8c3164p-6f);
    q = fmaf(ax, q, 0x1.b4e9c8p-4f);
    q = fmaf(ax, q, 0x1.4515fap-1f);
    q = fmaf(ax, q, 0x1.078e50p-3f);
    q = fmaf(q, ax, ax);
    const float r2 = 1.f - __builtin_amdgcn_exp2f(q * -1.4426950408889634f);
    return copysignf(ax < 1.f ? r1 : r2, x);
}

template <int ACT>
__global__ __launch_bounds__(FT, 2) void ffn_f16x3_kernel(const FfnF16Args p) {
    __shared__ u32x4 s_ring[RING + (NPAR + NSTAT) / 4];
    float* const s_par = reinterpret_cast<float*>(s_ring + RING);
    float* const s_stat = s_par + NPAR;

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * TMF;
    const bool tok_wave = wu < 4;
    const int lw = wu & 3;

    // 16-byte piece index of this lane's hi / lo fragment inside a tile whose first row is a multiple of 16
    const int lane_hi = swz(i32, 2 * h), lane_lo = swz(i32, 2 * h + 1);

    const u32x4* A0 = reinterpret_cast<const u32x4*>(p.A0);
    const u32x4* A1 = reinterpret_cast<const u32x4*>(p.A1);
    const u32x4* W0 = reinterpret_cast<const u32x4*>(p.W0);
    const u32x4* W3 = reinterpret_cast<const u32x4*>(p.W3);

    // ---- phase 1: H = W0f . A^T ----------------------------------------------------------------------------------
    u32x4* const s_tok = s_ring;
    u32x4* const s_w1 = s_ring + TS1 * TOK_STAGE;
    auto issue_tok = [&](int kt) {  // 128 rows x 64 B = 8 pieces of 1 KB: waves 0-3, two each
        const int k0 = kt * 16;
        const u32x4* src = k0 < 256 ? A0 : A1;
        const int kk = k0 & 255;
        u32x4* ring = s_tok + (kt % TS1) * TOK_STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slot = (lw * 2 + j) * 64 + lane, row = slot >> 2, qp = slot & 3;
            const int q = qp ^ ((row >> 2) & 3);
            const int gm = min(m0 + row, p.T - 1);
            __builtin_amdgcn_global_load_lds(src + (((long long)gm * 256 + kk) >> 2) + q, ring + (lw * 2 + j) * 64, 16, 0, 0);
        }
    };
    auto issue_w1 = [&](int kt) {  // 512 rows x 64 B = 32 pieces: waves 4-7, eight each
        u32x4* ring = s_w1 + (kt % WS1) * W1_STAGE;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int slot = (lw * 8 + j) * 64 + lane, row = slot >> 2, qp = slot & 3;
            const int q = qp ^ ((row >> 2) & 3);
            __builtin_amdgcn_global_load_lds(W0 + ((long long)kt * HID + row) * 4 + q, ring + (lw * 8 + j) * 64, 16, 0, 0);
        }
    };

    f32x16 acc[2][4];  // [hidden tile of this wave][token tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NK1 = HID / 16;  // K of ffn.0 = 512
    if (tok_wave) {
#pragma unroll 1
        for (int s0 = 0; s0 < TS1 - 1; ++s0) issue_tok(s0);
    } else {
#pragma unroll 1
        for (int s0 = 0; s0 < WS1 - 1; ++s0) issue_w1(s0);
    }
    for (int i = t; i < NPAR; i += FT) {
        float v;
        if (i < HID) v = p.b0[i];
        else if (i < 2 * HID) v = ACT == ACT_LN_GELU ? p.gamma[i - HID] : 1.f;
        else if (i < 3 * HID) v = ACT == ACT_LN_GELU ? p.beta[i - 2 * HID] : 0.f;
        else v = p.b3[i - 3 * HID];
        s_par[i] = v;
    }
#pragma unroll 1
    for (int kt = 0; kt < NK1; ++kt) {
        if (tok_wave) wait_stages<2>(min(TS1 - 2, NK1 - 1 - kt));
        else wait_stages<8>(min(WS1 - 2, NK1 - 1 - kt));
        __builtin_amdgcn_s_barrier();
        if (tok_wave) {
            if (kt + TS1 - 1 < NK1) issue_tok(kt + TS1 - 1);
        } else {
            if (kt + WS1 - 1 < NK1) issue_w1(kt + WS1 - 1);
        }
        // all fragment addresses are (per-lane base) + (compile-time constant): rows 32 apart share the swizzle term
        const f16x8* st = reinterpret_cast<const f16x8*>(s_tok + (kt % TS1) * TOK_STAGE);
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w1 + (kt % WS1) * W1_STAGE) + wu * 64 * 4;
        f16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            ah[ft] = sw[lane_hi + ft * 32 * 4];
            al[ft] = sw[lane_lo + ft * 32 * 4];
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            bh[tt] = st[lane_hi + tt * 32 * 4];
            bl[tt] = st[lane_lo + tt * 32 * 4];
        }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[ft][tt] = mfma16(ah[ft], bh[tt], acc[ft][tt]);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[ft][tt] = mfma16(ah[ft], bl[tt], acc[ft][tt]);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[ft][tt] = mfma16(al[ft], bh[tt], acc[ft][tt]);
    }
    __builtin_amdgcn_s_barrier();  // every wave has read its last phase-1 fragments: the rings may be re-used

    // ---- phase 3 set-up: W3 stream starts now, under the LayerNorm / GELU arithmetic ---------------------------------
    u32x4* const s_h = s_ring;                      // H' ring
    u32x4* const s_w3 = s_ring + HR * TOK_STAGE;
    constexpr int NK3 = HID / 16;
    auto issue_w3 = [&](int kt) {  // 256 rows x 64 B = 16 pieces: waves 4-7, four each
        u32x4* ring = s_w3 + (kt % WS3) * W3_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int slot = (lw * 4 + j) * 64 + lane, row = slot >> 2, qp = slot & 3;
            const int q = qp ^ ((row >> 2) & 3);
            __builtin_amdgcn_global_load_lds(W3 + ((long long)kt * DOUT + row) * 4 + q, ring + (lw * 4 + j) * 64, 16, 0, 0);
        }
    };
    if (!tok_wave) {
#pragma unroll 1
        for (int s0 = 0; s0 < WS3 - 1; ++s0) issue_w3(s0);
    }

    // ---- phase 2: bias, LayerNorm, GELU (or bias, ReLU) on the accumulators -------------------------------------------
    {
        const float* b0 = s_par + wu * 64 + 4 * h;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b0 + ft * 32 + 8 * g);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    acc[ft][tt][4 * g] += bv.x; acc[ft][tt][4 * g + 1] += bv.y; acc[ft][tt][4 * g + 2] += bv.z; acc[ft][tt][4 * g + 3] += bv.w;
                }
            }
    }
    if (ACT == ACT_LN_GELU) {
        float mean[4], rstd[4];
        // pass 1: sum over the 512 features of each token
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[ft][tt][r];
            s += __shfl_xor(s, 32);
            if (h == 0) s_stat[wu * TMF + tt * 32 + i32] = s;
        }
        lds_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += s_stat[w * TMF + tt * 32 + i32];
            mean[tt] = s * (1.f / HID);
        }
        // pass 2: sum of squared deviations
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[ft][tt][r] - mean[tt];
                    s += d * d;
                }
            s += __shfl_xor(s, 32);
            if (h == 0) s_stat[8 * TMF + wu * TMF + tt * 32 + i32] = s;
        }
        lds_barrier();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += s_stat[8 * TMF + w * TMF + tt * 32 + i32];
            rstd[tt] = 1.f / sqrtf(s * (1.f / HID) + p.eps);
        }
        const float* gm = s_par + HID + wu * 64 + 4 * h;
        const float* bt = s_par + 2 * HID + wu * 64 + 4 * h;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(gm + ft * 32 + 8 * g);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bt + ft * 32 + 8 * g);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = (acc[ft][tt][4 * g + e] - mean[tt]) * rstd[tt] * gv[e] + bv[e];
                        acc[ft][tt][4 * g + e] = 0.5f * y * (1.f + erf_select(y * 0.70710678118654752440f));
                    }
            }
    } else {
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ft][tt][r] = fmaxf(acc[ft][tt][r], 0.f);
    }

    // quarter q = 2 ft + s of this wave's block -> ring slot: piece 2h / 2h+1 of a token row = hi / lo halves of registers 8s..8s+7
    auto write_quarter = [&](int q, int slot) {
        const int ft = q >> 1, s = q & 1;
        f16x8* dst = reinterpret_cast<f16x8*>(s_h + slot * TOK_STAGE);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const f32x8 v = {acc[ft][tt][8 * s], acc[ft][tt][8 * s + 1], acc[ft][tt][8 * s + 2], acc[ft][tt][8 * s + 3],
                             acc[ft][tt][8 * s + 4], acc[ft][tt][8 * s + 5], acc[ft][tt][8 * s + 6], acc[ft][tt][8 * s + 7]};
            f16x8 hi, lo;
            split8(v, hi, lo);
            dst[lane_hi + tt * 32 * 4] = hi;
            dst[lane_lo + tt * 32 * 4] = lo;
        }
    };
    write_quarter(0, wu);  // k-steps 0..7
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // visible to the other waves behind the first barrier of the loop

    // ---- phase 3: OUT = W3 . H'^T ---------------------------------------------------------------------------------------
    const int wf3 = wu >> 1, wt3 = wu & 1;
    f32x16 o[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][j][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NK3; ++kt) {  // fully unrolled: which quarter is written where is a compile-time fact
        if (!tok_wave) wait_stages<4>(min(WS3 - 2, NK3 - 1 - kt));
        __builtin_amdgcn_s_barrier();  // W3 stage kt landed; H' k-step kt complete (written >= 7 iterations ago); slot of k-step kt-1 is free
        if (!tok_wave && kt + WS3 - 1 < NK3) issue_w3(kt + WS3 - 1);
        if (kt >= 1 && kt + HR - 1 < NK3 && wu == ((kt + HR - 1) & 7)) write_quarter((kt + HR - 1) >> 3, (kt + HR - 1) % HR);
        const f16x8* sh = reinterpret_cast<const f16x8*>(s_h + (kt % HR) * TOK_STAGE) + wt3 * 64 * 4;
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w3 + (kt % WS3) * W3_STAGE) + wf3 * 64 * 4;
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            bh[tt] = sh[lane_hi + tt * 32 * 4];
            bl[tt] = sh[lane_lo + tt * 32 * 4];
        }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            const f16x8 ah = sw[lane_hi + ft * 32 * 4], al = sw[lane_lo + ft * 32 * 4];
            o[ft][0] = mfma16(ah, bh[0], o[ft][0]);
            o[ft][1] = mfma16(ah, bh[1], o[ft][1]);
            o[ft][0] = mfma16(ah, bl[0], o[ft][0]);
            o[ft][1] = mfma16(ah, bl[1], o[ft][1]);
            o[ft][0] = mfma16(al, bh[0], o[ft][0]);
            o[ft][1] = mfma16(al, bh[1], o[ft][1]);
        }
    }

    // ---- epilogue: + b3, + residual, fp32 row and SH8 copy --------------------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> inline-asm permlane reads
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        const int fb = wf3 * 64 + ft * 32 + 16 * h;  // first of this lane's 16 consecutive output features
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tok = m0 + wt3 * 64 + tt * 32 + i32;
            const bool tok_ok = tok < p.T;
            const long long tokc = tok_ok ? tok : 0;
            f32x16 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = o[ft][tt][e], b0 = o[ft][tt][8 + e], a1 = o[ft][tt][4 + e], b1 = o[ft][tt][12 + e];
                half_swap(a0, b0);
                half_swap(a1, b1);
                w[e] = a0; w[4 + e] = b0; w[8 + e] = a1; w[12 + e] = b1;
            }
            const float* bp = s_par + 3 * HID + fb;
            const float* rs = p.X + tokc * DOUT + fb;
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + j);
                const f32x4 rv = *reinterpret_cast<const f32x4*>(rs + j);
                w[j] = (w[j] + bv.x) + rv.x; w[j + 1] = (w[j + 1] + bv.y) + rv.y; w[j + 2] = (w[j + 2] + bv.z) + rv.z; w[j + 3] = (w[j + 3] + bv.w) + rv.w;
            }
            const int ql = i32 & 3;
            const int tq = m0 + wt3 * 64 + tt * 32 + (i32 & ~3);  // first token of this lane's quad
            {
                u32x4 q0 = __builtin_bit_cast(u32x4, f32x4{w[0], w[1], w[2], w[3]}), q1 = __builtin_bit_cast(u32x4, f32x4{w[4], w[5], w[6], w[7]});
                u32x4 q2 = __builtin_bit_cast(u32x4, f32x4{w[8], w[9], w[10], w[11]}), q3 = __builtin_bit_cast(u32x4, f32x4{w[12], w[13], w[14], w[15]});
                quad_transpose(q0, q1, q2, q3, ql);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
                    if (tq + sidx < p.T)
                        *reinterpret_cast<u32x4*>(p.X + (long long)(tq + sidx) * DOUT + fb + 4 * ql) = sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3;
            }
            {
                const f32x8 g0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
                const f32x8 g1 = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
                f16x8 hi0, lo0, hi1, lo1;
                split8(g0, hi0, lo0);
                split8(g1, hi1, lo1);
                u32x4 q0 = __builtin_bit_cast(u32x4, hi0), q1 = __builtin_bit_cast(u32x4, lo0);
                u32x4 q2 = __builtin_bit_cast(u32x4, hi1), q3 = __builtin_bit_cast(u32x4, lo1);
                quad_transpose(q0, q1, q2, q3, ql);
                u32x4* ob = reinterpret_cast<u32x4*>(p.XS);
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx)
                    if (tq + sidx < p.T)
                        ob[(((long long)(tq + sidx) * DOUT + fb) >> 2) + ql] = sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3;
            }
        }
    }
}

}  // namespace

void launch_ffn_f16x3(const FfnF16Args& a, hipStream_t s) {
    const dim3 grid((a.T + TMF - 1) / TMF);
    if (a.act == 0) hipLaunchKernelGGL((ffn_f16x3_kernel<ACT_LN_GELU>), grid, dim3(FT), 0, s, a);
    else hipLaunchKernelGGL((ffn_f16x3_kernel<ACT_RELU>), grid, dim3(FT), 0, s, a);
}
