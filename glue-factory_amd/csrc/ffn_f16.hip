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
//            accumulator registers); K = 512 in stages of 16.  Token rows (HBM) and weight rows (L2) arrive by LDS-DMA in
//            three-slot rings; every wave issues its share (1 + 4 pieces of 1 KB per stage) BETWEEN its MFMAs, three stages
//            ahead, and reads the fragments of stage k+1 into a second register set while the MFMAs of stage k run -- one
//            barrier per stage, and nothing between the barrier and the first MFMA but one DMA issue.
//   phase 2  bias, LayerNorm over the 512 features of a token (two passes: sum, then sum of squared deviations; the
//            8 waves' partial sums meet in LDS and are added in wave order), GELU(erf), split into fp16 halves -- all on
//            the accumulator registers.
//   phase 3  OUT[256 x 128] = W3 . H'^T : the hidden features are now the contraction index, and they are spread over the
//            waves.  They go through LDS one k-step (16 features x 128 tokens, 8 KB) at a time, in a ring of 8 slots: a
//            wave's 64 x 128 block is four k-steps ("quarters"); k-step j of the contraction is quarter j/8 of wave j%8, so
//            every wave writes its first quarter before the loop and one more quarter every eighth iteration, eight
//            k-steps ahead of its use, under the other waves' MFMAs -- and a quarter's 32 registers are dead once written.
//            Wave (wf, wt) accumulates output features [64 wf, +64) x tokens [64 wt, +64).  An accumulator register block
//            of a lane is used as it stands: registers 8s..8s+7 of lane (token j, half h) become the 16-byte fragment of
//            half h -- which fixes the order of the contraction index: position 8h + 4a + e of k-step j is hidden feature
//            64 (j%8) + 16 (j/8) + 8a + 4h + e; W3 is tiled in that order at upload (api.hip: kperm).  Same pipeline as
//            phase 1 (W3 stages by LDS-DMA, fragments of k-step k+1 read under the MFMAs of k-step k); the residual rows
//            are fetched under the last eight k-steps, when the hidden block's registers are free.
//   epilogue + b3, + residual, fp32 row and its SH8 copy; the lane's 64-byte pieces go through a wave-private LDS image and leave as
//            full 128-byte lines (round 4; the quad transposes in registers of gemm_f16.hip cost 1 175 VALU instructions per wave here).
#include "common.h"
#include "kernels.h"

#include <type_traits>

namespace {

constexpr int FT = 512;                  // threads: 8 waves
constexpr int TMF = 128;                 // tokens per workgroup
constexpr int HID = 512, DOUT = 256;     // hidden width, output width
constexpr int RS = 3;                    // phase-1 DMA ring slots (stage k+3 is issued into the slot stage k just left)
constexpr int RS3 = 4;                   // phase-3 W3 ring slots (two stages per barrier interval)
constexpr int HR = 8;                    // H' ring: k-steps of 128 tokens x 64 B

constexpr int TOK_STAGE = TMF * 4;       // u32x4 per stage
constexpr int W1_STAGE = HID * 4;
constexpr int W3_STAGE = DOUT * 4;
constexpr int RING1 = RS * (TOK_STAGE + W1_STAGE);  // u32x4
constexpr int RING3 = HR * TOK_STAGE + RS3 * W3_STAGE;
constexpr int RING = RING1 > RING3 ? RING1 : RING3;
constexpr int NPAR = 3 * HID + DOUT;     // b0 | gamma | beta | b3
constexpr int NSTAT = 2 * 8 * TMF;       // partial sums / partial squared deviations per wave and token

__device__ __forceinline__ int swz(int row, int q) { return row * 4 + (q ^ ((row >> 2) & 3)); }  // 64-byte rows

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// lgkmcnt(0) as the BUILTIN (0xC07F = vmcnt 63, expcnt 7, lgkmcnt 0 in the gfx9 encoding): hipcc's wait-count pass sees it and knows that
// every fragment read issued before it has landed; an asm statement is opaque to that pass, which then protects the first MFMA behind
// the barrier with a full lgkmcnt(0) of its own -- after the reads of the next k-step have been issued in front of that MFMA
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }
// LDS accesses of this wave are complete, then the workgroup barrier (the raw s_barrier does not wait for them; DMA stays in flight)
__device__ __forceinline__ void lds_barrier() {
    wait_lds();
    __builtin_amdgcn_s_barrier();
}
#define PIN() __builtin_amdgcn_sched_barrier(0)
// LDS-DMA as `buffer_load_dwordx4 ... offen lds` in inline asm (round 4): 1 KB per wave and request, lane l's 16 bytes land at m0 + 16 l.
// (a) its issue is cheaper than the global_load_lds builtin's (tests/micro/mfma_gap.hip, profiles/r04_ad_mfma_gap_vmem.txt: one request every
// fourth MFMA costs a lone wave 14 cycles instead of 28, every eighth 0 instead of 28: a 32-bit offset per lane instead of a 64-bit
// address); (b) rows beyond the end of the buffer read as zeros by the range check -- no clamped row index; (c) hipcc does not see it,
// and that is wanted: with the builtin in the instruction stream its wait-count pass has two kinds of LDS event pending and answers every
// use of a fragment with lgkmcnt(0).  The vmcnt waits of this kernel were counted by hand already.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(const i32x4 rsrc, unsigned lds_byte, int voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, long long bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);  // stride 0, no swizzle
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffll ? 0x7fffffffll : bytes));
    r[3] = 0x00020000;
    return r;
}
#ifndef MG_FFN_EXP  // timing ablations only (wrong results, finite values): bit 0/1 no DMA in the phase-3/1 loop, 2/5 no barriers, 3 no quarter writes,
                    // 6 (with the stamp instantiation) per-wave barrier waits of phase 3 instead of the phase stamps.  NOTE: without its barriers a
                    // workgroup's stamps are wave 0's only -- the other waves may still be running (how round 2 overestimated the barrier cost)
#define MG_FFN_EXP 0
#endif
constexpr int EXP = MG_FFN_EXP;

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

// STAMP: diagnostic instantiation only (debug tap "ffn.stamps"): wave 0 of every workgroup records s_memtime at the phase
// boundaries, as cycle counts relative to its start, into a buffer nothing else reads.  The production kernel holds no stamp.
template <int ACT, bool STAMP>
__global__ __launch_bounds__(FT, 2) void ffn_f16x3_kernel(const FfnF16Args p) {
    __shared__ u32x4 s_ring[RING + (NPAR + NSTAT) / 4];
    float* const s_par = reinterpret_cast<float*>(s_ring + RING);
    float* const s_stat = s_par + NPAR;

    const int t = threadIdx.x, lane = t & 63;
    const int wu = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i32 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * TMF;
    unsigned long long t_start = 0;
    auto stamp = [&](int slot) {
        if constexpr (STAMP) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (slot == 0) t_start = now;
            if (t == 0) p.stamps[blockIdx.x * 8 + slot] = (float)(now - t_start);
        }
    };
    stamp(0);

    // 16-byte piece index of this lane's hi / lo fragment inside a tile whose first row is a multiple of 16
    const int lane_hi = swz(i32, 2 * h), lane_lo = swz(i32, 2 * h + 1);

    auto lds_of = [](const u32x4* q) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)reinterpret_cast<const char*>(q); };
    const long long tok_bytes = (long long)min(TMF, p.T - m0) * 1024;  // this workgroup's rows of the (T, 256) SH8 operands; rows beyond read as zeros
    const i32x4 rs_a0 = make_rsrc(reinterpret_cast<const char*>(p.A0) + (long long)m0 * 1024, tok_bytes);
    const i32x4 rs_a1 = make_rsrc(reinterpret_cast<const char*>(p.A1) + (long long)m0 * 1024, tok_bytes);
    const i32x4 rs_w0 = make_rsrc(p.W0, (long long)(HID / 16) * W1_STAGE * 16);
    const i32x4 rs_w3 = make_rsrc(p.W3, (long long)(HID / 16) * W3_STAGE * 16);

    // per-feature parameters -> LDS by DMA as well (a plain load here would make hipcc drain the whole DMA prologue at its use)
    {
        u32x4* dst = s_ring + RING;
        const int piece = wu;  // 7 pieces of 1 KB: b0 (2), gamma (2), beta (2), b3 (1)
        const float* src = nullptr;
        if (piece < 2) src = p.b0 + piece * 256;
        else if (piece < 4) src = ACT == ACT_LN_GELU ? p.gamma + (piece - 2) * 256 : nullptr;
        else if (piece < 6) src = ACT == ACT_LN_GELU ? p.beta + (piece - 4) * 256 : nullptr;
        else if (piece == 6) src = p.b3;
        if (src) dma_piece(make_rsrc(src, 1024), __builtin_amdgcn_readfirstlane(lds_of(dst + piece * 64)), lane * 16, 0);
    }

    // ---- phase 1: H = W0f . A^T ----------------------------------------------------------------------------------
    u32x4* const s_tok = s_ring;
    u32x4* const s_w1 = s_ring + RS * TOK_STAGE;
    // DMA sources of this lane: token piece wu (rows 16 wu ..), weight pieces 4 wu + j; the LDS image is linear, the XOR swizzle
    // is applied to the source piece
    int tok_vo, w_vo;  // byte offsets of this lane's 16 bytes: token piece wu (row-in-tile x 1024 + swizzled quarter), weight pieces 4 wu + j (+ 1024 j)
    {
        const int slot = wu * 64 + lane, row = slot >> 2, q = (slot & 3) ^ ((row >> 2) & 3);
        tok_vo = row * 1024 + q * 16;
        const int sl = (wu * 4) * 64 + lane, r = sl >> 2;
        w_vo = (r * 4 + ((sl & 3) ^ ((r >> 2) & 3))) * 16;  // piece 4 wu + j is 16 rows further: same swizzle term, + 1024 j bytes
    }
    const unsigned lds_tok = lds_of(s_tok) + wu * 1024, lds_w1 = lds_of(s_w1) + wu * 4096;
    auto dma_tok = [&](int kt) {
        dma_piece(kt < 16 ? rs_a0 : rs_a1, __builtin_amdgcn_readfirstlane(lds_tok + (kt % RS) * (TOK_STAGE * 16)), tok_vo, (kt & 15) * 64);
    };
    auto dma_w1 = [&](int kt, int j) {
        dma_piece(rs_w0, __builtin_amdgcn_readfirstlane(lds_w1 + (kt % RS) * (W1_STAGE * 16) + j * 1024), w_vo, kt * (W1_STAGE * 16) + j * 1024);
    };

    f32x16 acc[2][4];  // [hidden tile of this wave][token tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NK1 = HID / 16;  // K of ffn.0 = 512
#pragma unroll
    for (int s0 = 0; s0 < RS; ++s0) {
        dma_tok(s0);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_w1(s0, j);
    }
    f16x8 ah[2][2], al[2][2], bh[2][4], bl[2][4];  // [register set][tile]
    auto read1 = [&](int set, int kt, int i) {  // fragment read i (0..11) of stage kt: lo halves last (they are used last)
        const f16x8* st = reinterpret_cast<const f16x8*>(s_tok + (kt % RS) * TOK_STAGE);
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w1 + (kt % RS) * W1_STAGE) + wu * 64 * 4;
        if (i < 2) ah[set][i] = sw[lane_hi + i * 32 * 4];
        else if (i < 6) bh[set][i - 2] = st[lane_hi + (i - 2) * 32 * 4];
        else if (i < 10) bl[set][i - 6] = st[lane_lo + (i - 6) * 32 * 4];
        else al[set][i - 10] = sw[lane_lo + (i - 10) * 32 * 4];
    };
    wait_vm<10>();  // parameters + stage 0 (issued first) have landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 12; ++i) read1(0, 0, i);

    // one stage: [my pieces of stage kt+1 landed] [my reads of stage kt done] barrier, then the MFMAs of stage kt with, between
    // them, the DMA of stage kt+3 (into the slot stage kt just left) and the fragment reads of stage kt+1 (other register set)
    auto stage1 = [&](auto set_c, int kt) {
        constexpr int S = decltype(set_c)::value;
        if (kt + 2 <= NK1 - 1) wait_vm<5>(); else wait_vm<0>();
        if (!(EXP & 32)) lds_barrier();
        const bool more_dma = kt + RS < NK1 && !(EXP & 2), more_rd = kt + 1 < NK1;
        // 24 MFMAs in product-major order (hi.hi, hi.lo, lo.hi; consecutive MFMAs hit different accumulators).  The head is
        // MFMA-only (the pipe starts the moment the barrier opens); from the third on, every MFMA is followed by ONE other
        // instruction -- the five DMA pieces of stage kt+3, then the twelve fragment reads of stage kt+1 -- which issue in its shadow.
        int slot_i = 0;
        auto filler = [&]() {
            const int f = slot_i++ - 2;
            if (f >= 0 && f < 5) { if (more_dma) { if (f == 0) dma_tok(kt + RS); else dma_w1(kt + RS, f - 1); } }
            else if (f >= 5 && f < 17) { if (more_rd) read1(1 - S, kt + 1, f - 5); }
            PIN();
        };
        PIN();
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { acc[ft][tt] = mfma16(ah[S][ft], bh[S][tt], acc[ft][tt]); filler(); }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { acc[ft][tt] = mfma16(ah[S][ft], bl[S][tt], acc[ft][tt]); filler(); }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) { acc[ft][tt] = mfma16(al[S][ft], bh[S][tt], acc[ft][tt]); filler(); }
    };
#pragma unroll 1
    for (int kt = 0; kt < NK1; kt += 2) {
        stage1(std::integral_constant<int, 0>{}, kt);
        stage1(std::integral_constant<int, 1>{}, kt + 1);
    }
    lds_barrier();  // every wave has read its last phase-1 fragments: the rings may be re-used
    stamp(1);

    // ---- phase 3 set-up: W3 stream starts now, under the LayerNorm / GELU arithmetic ---------------------------------
    u32x4* const s_h = s_ring;                      // H' ring
    u32x4* const s_w3 = s_ring + HR * TOK_STAGE;
    constexpr int NK3 = HID / 16;
    int w3_vo;  // pieces 2 wu + j of a 16-piece stage (256 rows x 64 B)
    {
        const int sl = (wu * 2) * 64 + lane, r = sl >> 2;
        w3_vo = (r * 4 + ((sl & 3) ^ ((r >> 2) & 3))) * 16;
    }
    const unsigned lds_w3 = lds_of(s_w3) + wu * 2048;
    auto dma_w3 = [&](int kt, int j) {
        dma_piece(rs_w3, __builtin_amdgcn_readfirstlane(lds_w3 + (kt % RS3) * (W3_STAGE * 16) + j * 1024), w3_vo, kt * (W3_STAGE * 16) + j * 1024);
    };
#pragma unroll
    for (int s0 = 0; s0 < 3; ++s0) {
        dma_w3(s0, 0);
        dma_w3(s0, 1);
    }

    // ---- phase 2: bias, LayerNorm, GELU (or bias, ReLU) on the accumulators -------------------------------------------
    {
        mfma_results_guard();
        const float* b0 = s_par + wu * 64 + 4 * h;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b0 + ft * 32 + 8 * g);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    acc[ft][tt][4 * g] = fma_winv_asm(acc[ft][tt][4 * g], p.winv0, bv.x); acc[ft][tt][4 * g + 1] = fma_winv_asm(acc[ft][tt][4 * g + 1], p.winv0, bv.y);
                    acc[ft][tt][4 * g + 2] = fma_winv_asm(acc[ft][tt][4 * g + 2], p.winv0, bv.z); acc[ft][tt][4 * g + 3] = fma_winv_asm(acc[ft][tt][4 * g + 3], p.winv0, bv.w);
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
    stamp(2);
    // Split into fp16 halves HERE, by all waves at once: in the loop below a quarter is written by one wave per iteration while the
    // other seven wait for it at the next barrier, so that write must be nothing but eight LDS stores.
    // Quarter q = 2 ft + s of this wave's block = registers 8s..8s+7 of acc[ft][*]: hi / lo become pieces 2h / 2h+1 of a token row.
    f16x8 Hh[4][4], Hl[4][4];  // [quarter][token tile]
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f32x8 v = {acc[ft][tt][8 * s], acc[ft][tt][8 * s + 1], acc[ft][tt][8 * s + 2], acc[ft][tt][8 * s + 3],
                                 acc[ft][tt][8 * s + 4], acc[ft][tt][8 * s + 5], acc[ft][tt][8 * s + 6], acc[ft][tt][8 * s + 7]};
                split8(v, Hh[2 * ft + s][tt], Hl[2 * ft + s][tt]);
                // materialise the halves now: without this hipcc sinks the GELU + split of the later quarters into the
                // (single-wave) write blocks of the loop below, where seven waves would wait for it
                asm volatile("" : "+v"(Hh[2 * ft + s][tt]), "+v"(Hl[2 * ft + s][tt]));
            }
    auto write_quarter = [&](int q, int slot) {
        f16x8* dst = reinterpret_cast<f16x8*>(s_h + slot * TOK_STAGE);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            dst[lane_hi + tt * 32 * 4] = Hh[q][tt];
            dst[lane_lo + tt * 32 * 4] = Hl[q][tt];
        }
    };
    write_quarter(0, wu);  // k-steps 0..7
    stamp(3);

    // ---- phase 3: OUT = W3 . H'^T ---------------------------------------------------------------------------------------
    const int wf3 = wu >> 1, wt3 = wu & 1;
    f32x16 o[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][j][r] = 0.f;
    f16x8 ch[2][2], cl[2][2], dh[2][2], dl[2][2];  // [register set][tile]: W3 (c) and H' (d) fragments
    auto read3 = [&](int set, int kt, int i) {  // fragment read i (0..7) of k-step kt, in the order the MFMAs need them
        const f16x8* sh = reinterpret_cast<const f16x8*>(s_h + (kt % HR) * TOK_STAGE) + wt3 * 64 * 4;
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_w3 + (kt % RS3) * W3_STAGE) + wf3 * 64 * 4;
        if (i == 0) ch[set][0] = sw[lane_hi];
        else if (i == 1) dh[set][0] = sh[lane_hi];
        else if (i == 2) dh[set][1] = sh[lane_hi + 32 * 4];
        else if (i == 3) ch[set][1] = sw[lane_hi + 32 * 4];
        else if (i == 4) dl[set][0] = sh[lane_lo];
        else if (i == 5) dl[set][1] = sh[lane_lo + 32 * 4];
        else if (i == 6) cl[set][0] = sw[lane_lo];
        else cl[set][1] = sw[lane_lo + 32 * 4];
    };
    auto write_piece = [&](int q, int slot, int i) {  // store i (0..7) of quarter q
        f16x8* dst = reinterpret_cast<f16x8*>(s_h + slot * TOK_STAGE);
        if (i & 1) dst[lane_lo + (i >> 1) * 32 * 4] = Hl[q][i >> 1];
        else dst[lane_hi + (i >> 1) * 32 * 4] = Hh[q][i >> 1];
    };
    wait_vm<4>();  // W3 stage 0 (stages 1, 2 stay in flight)
    lds_barrier(); // ... and everybody's first quarter
#pragma unroll
    for (int i = 0; i < 8; ++i) read3(0, 0, i);
    float resid[2][2][16];
    // Two k-steps per barrier interval (a = 2j from register set 0, b = 2j+1 from set 1): a 12-MFMA k-step is too short to pay for a
    // barrier of its own (measured: 29 % of this phase).  At the barrier of interval j every wave's pieces of stages 2j+1, 2j+2 have
    // landed and the reads of k-steps <= 2j are done, so: DMA stages 2j+3, 2j+4 into the slots of stages 2j-1, 2j; k-steps 2j+7 and
    // 2j+8 of the hidden block into the slots of k-steps 2j-1 and 2j, by their owner waves.  Fully unrolled: which quarter goes where is
    // a compile-time fact.  12 MFMAs per k-step, MFMA-first, four accumulators in rotation; every later MFMA carries one filler.
    auto kstep = [&](int S, int kt, int j, bool second) {
        const int rd_kt = kt + 1;                    // fragments to fetch under these MFMAs (into the other register set)
        int slot_i = 0;
        auto filler = [&]() {
            const int f = slot_i++ - 1;
            // both stages' DMA right behind the barrier (they must land by the next one), then the fragment reads
            if (!second && f >= 0 && f < 2) {
                if (2 * j + 3 + f < NK3 && !(EXP & 1)) { dma_w3(2 * j + 3 + f, 0); dma_w3(2 * j + 3 + f, 1); }
            }
            const int r = second ? f : f - 2;
            if (r >= 0 && r < 8 && rd_kt < NK3) read3(1 - S, rd_kt, r);
            PIN();
        };
        o[0][0] = mfma16(ch[S][0], dh[S][0], o[0][0]); filler();
        o[0][1] = mfma16(ch[S][0], dh[S][1], o[0][1]); filler();
        o[1][0] = mfma16(ch[S][1], dh[S][0], o[1][0]); filler();
        o[1][1] = mfma16(ch[S][1], dh[S][1], o[1][1]); filler();
        o[0][0] = mfma16(ch[S][0], dl[S][0], o[0][0]); filler();
        o[0][1] = mfma16(ch[S][0], dl[S][1], o[0][1]); filler();
        o[1][0] = mfma16(ch[S][1], dl[S][0], o[1][0]); filler();
        o[1][1] = mfma16(ch[S][1], dl[S][1], o[1][1]); filler();
        o[0][0] = mfma16(cl[S][0], dh[S][0], o[0][0]); filler();
        o[0][1] = mfma16(cl[S][0], dh[S][1], o[0][1]); filler();
        o[1][0] = mfma16(cl[S][1], dh[S][0], o[1][0]); filler();
        o[1][1] = mfma16(cl[S][1], dh[S][1], o[1][1]); filler();
    };
    // the hidden slabs of k-steps 2j+7, 2j+8 (pre-loop wrote 0..7), by their owner waves, AFTER the interval's MFMAs (round 4): behind a
    // branch hipcc cannot count LDS operations, and with the stores among the fillers every k-step began with a full lgkmcnt(0)
    auto publish = [&](int j) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int wr_kt = 2 * j + 7 + k;
            if (wr_kt >= HR && wr_kt < NK3 && !(EXP & 8) && wu == (wr_kt & 7)) {
#pragma unroll
                for (int r = 0; r < 8; ++r) write_piece(wr_kt >> 3, wr_kt % HR, r);
            }
        }
    };
    unsigned long long t_wait = 0, t_vm = 0;  // STAMP only: cycles wave 0 spends in the vmcnt wait + LDS wait + barrier of the 16 intervals / in the vmcnt wait alone
#pragma unroll
    for (int j = 0; j < NK3 / 2; ++j) {
        unsigned long long t_a = 0;
        if constexpr (STAMP) t_a = __builtin_amdgcn_s_memtime();
        wait_vm<0>();  // stages 2j+1, 2j+2 (and, once, the residual rows) -- everything this wave has in flight
        if constexpr (STAMP) t_vm += __builtin_amdgcn_s_memtime() - t_a;
        if (!(EXP & 4)) lds_barrier();
        if constexpr (STAMP) t_wait += __builtin_amdgcn_s_memtime() - t_a;
        PIN();
        if (j == NK3 / 2 - 3) {  // the hidden block's registers are free from here on: fetch the residual rows under the last MFMAs
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int tok = min(m0 + wt3 * 64 + tt * 32 + i32, p.T - 1);
                    const float* rs = p.X + (long long)tok * DOUT + wf3 * 64 + ft * 32 + 16 * h;
#pragma unroll
                    for (int jj = 0; jj < 16; jj += 4) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(rs + jj);
                        resid[ft][tt][jj] = rv.x; resid[ft][tt][jj + 1] = rv.y; resid[ft][tt][jj + 2] = rv.z; resid[ft][tt][jj + 3] = rv.w;
                    }
                }
            PIN();
        }
        kstep(0, 2 * j, j, false);
        kstep(1, 2 * j + 1, j, true);
        publish(j);
    }

    stamp(4);
    if constexpr (STAMP) {
        if (t == 0) { p.stamps[blockIdx.x * 8 + 6] = (float)t_wait; p.stamps[blockIdx.x * 8 + 7] = (float)t_vm; }
    }
    // ---- epilogue: + b3, + residual, fp32 row and SH8 copy --------------------------------------------------------------
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> inline-asm permlane reads
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        const int fb = wf3 * 64 + ft * 32 + 16 * h;  // first of this lane's 16 consecutive output features
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            f32x16 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = o[ft][tt][e], b0 = o[ft][tt][8 + e], a1 = o[ft][tt][4 + e], b1 = o[ft][tt][12 + e];
                half_swap(a0, b0);
                half_swap(a1, b1);
                w[e] = a0; w[4 + e] = b0; w[8 + e] = a1; w[12 + e] = b1;
            }
            const float* bp = s_par + 3 * HID + fb;
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + j);
                w[j] = fma_winv_asm(w[j], p.winv3, bv.x) + resid[ft][tt][j]; w[j + 1] = fma_winv_asm(w[j + 1], p.winv3, bv.y) + resid[ft][tt][j + 1];
                w[j + 2] = fma_winv_asm(w[j + 2], p.winv3, bv.z) + resid[ft][tt][j + 2]; w[j + 3] = fma_winv_asm(w[j + 3], p.winv3, bv.w) + resid[ft][tt][j + 3];
            }
            // The lane holds 64 contiguous bytes (4 pieces) of its token's row, twice (fp32 and SH8).  They go through a 4 KB LDS image
            // private to the wave -- [token 32][piece 8], piece XOR (token & 7) -- and come back so that eight consecutive lanes hold the
            // 128 contiguous bytes of one token: every store instruction writes eight full lines.  (Until round 4 a 4 x 4 quad transpose
            // in registers did this: ~100 VALU instructions per transpose, 1 175 of them per wave in an epilogue no MFMA hides; the image
            // costs 4 + 4 LDS instructions.  It lives in slabs 0-3 of the hidden ring, dead since interval 13.)
            u32x4* const stg = s_ring + wu * 256;
            const int tb = m0 + wt3 * 64 + tt * 32;                 // first token of the block
            const int wr_idx = i32 * 8, wr_sw = i32 & 7, rd_tok = lane >> 3, rd_p = lane & 7, rd_sw = rd_p ^ (rd_tok & 7);
            auto through_lds = [&](const u32x4 q0, const u32x4 q1, const u32x4 q2, const u32x4 q3, u32x4* out_rows /* 16-byte pieces, row = DOUT / 4 */) {
                stg[wr_idx + ((4 * h + 0) ^ wr_sw)] = q0;
                stg[wr_idx + ((4 * h + 1) ^ wr_sw)] = q1;
                stg[wr_idx + ((4 * h + 2) ^ wr_sw)] = q2;
                stg[wr_idx + ((4 * h + 3) ^ wr_sw)] = q3;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int tok = rd_tok + 8 * it;
                    const u32x4 v = stg[tok * 8 + rd_sw];
                    if (tb + tok < p.T) out_rows[(long long)(tb + tok) * (DOUT / 4) + (wf3 * 64 + ft * 32) / 4 + rd_p] = v;
                }
            };
            through_lds(__builtin_bit_cast(u32x4, f32x4{w[0], w[1], w[2], w[3]}), __builtin_bit_cast(u32x4, f32x4{w[4], w[5], w[6], w[7]}),
                        __builtin_bit_cast(u32x4, f32x4{w[8], w[9], w[10], w[11]}), __builtin_bit_cast(u32x4, f32x4{w[12], w[13], w[14], w[15]}),
                        reinterpret_cast<u32x4*>(p.X));
            {
                const f32x8 g0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
                const f32x8 g1 = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
                f16x8 hi0, lo0, hi1, lo1;
                split8(g0, hi0, lo0);
                split8(g1, hi1, lo1);
                through_lds(__builtin_bit_cast(u32x4, hi0), __builtin_bit_cast(u32x4, lo0), __builtin_bit_cast(u32x4, hi1), __builtin_bit_cast(u32x4, lo1),
                            reinterpret_cast<u32x4*>(p.XS));
            }
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(5);
        if (EXP & 64) {  // diagnostic variant: slot w = what wave w spent in the waits + barriers of phase 3 (replaces the phase stamps)
            __syncthreads();
            if (lane == 0) p.stamps[blockIdx.x * 8 + wu] = (float)t_wait;
        }
    }
}

}  // namespace

void launch_ffn_f16x3(const FfnF16Args& a, hipStream_t s) {
    const dim3 grid((a.T + TMF - 1) / TMF);
    if (a.stamps) hipLaunchKernelGGL((ffn_f16x3_kernel<ACT_LN_GELU, true>), grid, dim3(FT), 0, s, a);
    else if (a.act == 0) hipLaunchKernelGGL((ffn_f16x3_kernel<ACT_LN_GELU, false>), grid, dim3(FT), 0, s, a);
    else hipLaunchKernelGGL((ffn_f16x3_kernel<ACT_RELU, false>), grid, dim3(FT), 0, s, a);
}
