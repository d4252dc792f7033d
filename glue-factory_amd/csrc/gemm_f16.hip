// C[tokens, features] = A[tokens, K] * W[features, K]^T on the f16 MFMA with split-fp16 (SH8) operands.
//
// Serves the Linear layers of LightGlue and SuperPoint's 1x1 heads in MG_PREC_F16X3 mode.  Re-cut for
// the 16x faster pipe: a 256x256 tile (8 waves x 128 features x 64 tokens, K in stages of 32,
// 144 KB of LDS rings, one workgroup per CU) -- at this MFMA rate the 128x128 tile of gemm.hip
// is bound by L2->LDS traffic (42 B/clk/CU at full rate); 256x256 halves the bytes per MFMA:
//   * both operands arrive SH8, so a 16-deep stage row is 64 contiguous bytes = four 16-byte pieces,
//     moved to LDS by LDS-DMA (global_load_lds) with the piece index XOR (row>>2)&3 applied on the
//     source side -> no ds_write traffic, every fragment ds_read_b128 conflict-free;
//   * one barrier per stage and nothing between it and the first MFMA: the fragments of stage k+1 are read into a
//     second register set under the MFMAs of stage k, and every wave issues its four DMA pieces of stage k+3 (into the
//     slot stage k just left) between those MFMAs -- MFMA-first, one filler per MFMA (the structure of ffn_f16.hip, where
//     the in-kernel stamps showed the older "dedicated loader waves + reads behind the barrier" loop at 56 % MFMA occupancy);
//   * the weight tile is the A operand (rows = features) and the token tile the B operand (lanes), so a
//     lane owns one token: after one v_permlane32_swap per register pair it holds 16 consecutive features
//     -- rotary pairs are adjacent registers (no shuffle), the residual / fp32 store is 64 contiguous
//     bytes, and the SH8 copy for the next MFMA consumer is produced in the same epilogue.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

#include <type_traits>

namespace {

// timing experiments of DESIGN.md section 4 (phase ablation): build a second library with -DMG_GEMM_EXP=1 (no stores) / 2 (no K loop);
// results are wrong by design, the shipped library is built without it
#ifndef MG_GEMM_EXP
#define MG_GEMM_EXP 0
#endif
constexpr int GEMM_EXP = MG_GEMM_EXP;
constexpr int TN = 256, BK = 16;  // 2 (feature halves) x TOKW (token groups) waves, each 128 features x 64 tokens
constexpr int RS = 3;             // ring slots per operand: stage k+3 is issued into the slot stage k just left
constexpr int TSLOTS = RS, WSLOTS = RS;
// TOKW = 4: 256 x 256 tile, 8 waves, 99 KB of LDS, one workgroup per CU.  TOKW = 2: 256 features x 128 tokens, 4 waves, 75 KB,
// TWO workgroups per CU: with K = 256-512 the loop is short and load / MFMA / epilogue of one tile run back to back, so
// the second workgroup's loop fills the first one's epilogue (VALU + stores) and vice versa.
#define PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ int swz(int row, int q) { return row * 4 + (q ^ ((row >> 2) & 3)); }  // 64-byte rows

// Tile that straddles N (the 65-channel detector head): element-wise, fp32 output only (no rotary / SH8
// at a ragged edge).  Kept out of line so the hot epilogue stays small.
__device__ __attribute__((noinline)) void epilogue_ragged(const GemmF16Args& p, float* C, f32x16 w, int fb, long long tokc, bool tok_ok) {
    if (!tok_ok || !C) return;
    for (int j = 0; j < 16; ++j) {
        const int f = fb + j;
        if (f >= p.N) break;
        float x = w[j] * p.winv;
        if (p.bias) x += p.bias[f];
        if (p.relu) x = fmaxf(x, 0.f);
        if (p.post_scale) x = __fadd_rn(__fmul_rn(x, p.post_scale[f]), p.post_shift[f]);
        if (f < p.scale_cols) x *= p.scale;
        if (p.residual) x += p.residual[tokc * p.ldr + f];
        C[tokc * p.ldc + f] = x;
    }
}

template <bool W_ROWMAJOR, bool RAGGED, int TOKW, bool L2NORM = false>
__global__ __launch_bounds__(128 * TOKW, 2) void gemm_f16x3_kernel(const GemmF16Args p) {
    constexpr int TM = 64 * TOKW, GT = 128 * TOKW;
    constexpr int NWP = 16 / (2 * TOKW), NPW = 2 + NWP;  // weight pieces / all DMA pieces per wave and stage (token pieces: always 2)
    // ONE LDS array (a second __shared__ object next to LDS-DMA staging makes hipcc drain vmcnt before
    // every fragment read): [token ring: TSLOTS x 256 rows x 64 B][weight ring: WSLOTS x 256 rows x 64 B] = 144 KB
    __shared__ u32x4 s_ring[(TSLOTS * TM + WSLOTS * TN) * 4 + 3 * TN / 4];  // + bias | post_scale | post_shift of this feature tile
    u32x4* const s_tokr = s_ring;
    u32x4* const s_wr = s_ring + TSLOTS * TM * 4;
    float* const s_par = reinterpret_cast<float*>(s_ring + (TSLOTS * TM + WSLOTS * TN) * 4);

    // Two workgroups share a CU (TOKW = 2).  Started together they stay in lockstep -- both in the K loop, then both in the
    // epilogue: the phase ablation (MG_GEMM_DEBUG) shows loop, epilogue arithmetic and stores adding up to the launch time, nothing
    // overlaps.  The workgroups the dispatcher places second on each CU (ids [CUs, 2 CUs) of the first round) therefore sleep about
    // half a tile first; later workgroups inherit the offset from the slots they take over.
    if ((int)blockIdx.x >= p.stagger_lo && (int)blockIdx.x < p.stagger_hi)
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wf = wave / TOKW, wt = wave % TOKW;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private L2s), so the
    // feature tiles of one token tile are given ids 8 apart -- same XCD, adjacent in time.
    const int nft = (p.N + TN - 1) / TN, ntt = (p.M + TM - 1) / TM;
    const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int ti = (slot0 / nft) * 8 + xcd, fj = slot0 % nft;
    if (ti >= ntt) return;
    const int m0 = ti * TM, n0 = fj * TN;
    const long long z = blockIdx.z;
    const u32x4* A0 = reinterpret_cast<const u32x4*>(p.A0) + z * (p.strideA >> 2);
    const u32x4* A1 = p.A1 ? reinterpret_cast<const u32x4*>(p.A1) + z * (p.strideA >> 2) : nullptr;
    const u32x4* Wg = reinterpret_cast<const u32x4*>(p.W) + z * (p.strideW >> 2);

    f32x16 acc[4][2];  // [feature tile][token tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int i32 = lane & 31, h = lane >> 5;
    const int lane_hi = swz(i32, 2 * h), lane_lo = swz(i32, 2 * h + 1);  // fragment piece index inside a tile whose first row is a multiple of 16

    // A stage is one 16-deep k-step: 256 token rows x 64 B + 256 feature rows x 64 B, moved to LDS by LDS-DMA
    // (global_load_lds_dwordx4: no VGPR round trip, no ds_write; the DMA writes LDS linearly, so the XOR swizzle is applied
    // to the per-lane SOURCE piece; rows past the edge are clamped, their outputs are never stored).  32 pieces of 1 KB per
    // stage: every wave issues two token pieces (2 wu, 2 wu + 1) and NWP weight pieces.
    long long tok_off[2];
    long long w_off[NWP];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = (wu * 2 + j) * 64 + lane, row = slot >> 2, q = (slot & 3) ^ ((row >> 2) & 3);
        tok_off[j] = (long long)min(m0 + row, p.M - 1);  // row index; the row stride depends on the source (A0 / A1)
        tok_off[j] = tok_off[j] * 4 + q;                 // packed: row * 4 + piece (lda is applied per stage)
    }
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int slot = (wu * NWP + j) * 64 + lane, row = slot >> 2, q = (slot & 3) ^ ((row >> 2) & 3);
        if (W_ROWMAJOR) w_off[j] = (long long)min(n0 + row, p.N - 1) * 4 + q;
        else w_off[j] = (long long)(n0 + row) * 4 + q;
    }
    auto dma_tok = [&](int kt, int j) {
        const int k0 = kt * BK;
        const u32x4* Asrc;
        int lda, kk;
        if (k0 < p.ksplit) { Asrc = A0; lda = p.lda0; kk = k0; } else { Asrc = A1; lda = p.lda1; kk = k0 - p.ksplit; }
        const long long row = tok_off[j] >> 2;
        const int q = (int)(tok_off[j] & 3);
        __builtin_amdgcn_global_load_lds(Asrc + ((row * lda + kk) >> 2) + q, s_tokr + (kt % RS) * TM * 4 + (wu * 2 + j) * 64, 16, 0, 0);
    };
    auto dma_w = [&](int kt, int j) {
        const int k0 = kt * BK;
        const u32x4* src;
        if (W_ROWMAJOR) src = Wg + (((w_off[j] >> 2) * p.ldw + k0) >> 2) + (w_off[j] & 3);
        else src = Wg + (long long)(k0 >> 4) * p.Npad * 4 + w_off[j];
        __builtin_amdgcn_global_load_lds(src, s_wr + (kt % RS) * TN * 4 + (wu * NWP + j) * 64, 16, 0, 0);
    };
    const int npro = min(RS, nk);
#pragma unroll 1
    for (int s0 = 0; s0 < npro; ++s0) {
        dma_tok(s0, 0); dma_tok(s0, 1);
#pragma unroll
        for (int j = 0; j < NWP; ++j) dma_w(s0, j);
    }
    // per-feature epilogue parameters -> LDS (one global round trip per workgroup, under the DMA prologue)
    for (int i = t; i < TN; i += GT) {
        const int f = n0 + i;
        const bool ok = f < p.N;
        s_par[i] = (p.bias && ok) ? p.bias[f] : 0.f;
        s_par[TN + i] = (p.post_scale && ok) ? p.post_scale[f] : 1.f;
        s_par[2 * TN + i] = (p.post_shift && ok) ? p.post_shift[f] : 0.f;
    }
    f16x8 ah[2][4], al[2][4], bh[2][2], bl[2][2];  // [register set][tile]
    auto read_frag = [&](int set, int kt, int i) {  // fragment read i (0..11) of stage kt: hi halves first (they are used first)
        const f16x8* st = reinterpret_cast<const f16x8*>(s_tokr + (kt % RS) * TM * 4) + wt * 64 * 4;
        const f16x8* sw = reinterpret_cast<const f16x8*>(s_wr + (kt % RS) * TN * 4) + wf * 128 * 4;
        if (i < 4) ah[set][i] = sw[lane_hi + i * 32 * 4];
        else if (i < 6) bh[set][i - 4] = st[lane_hi + (i - 4) * 32 * 4];
        else if (i < 8) bl[set][i - 6] = st[lane_lo + (i - 6) * 32 * 4];
        else al[set][i - 8] = sw[lane_lo + (i - 8) * 32 * 4];
    };
    if (nk > 0 && !(GEMM_EXP & 2)) {
        // stage 0 landed (mine): all but the younger stages' pieces (NPW per stage)
        if (npro >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
        else if (npro == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 12; ++i) read_frag(0, 0, i);
    }
    // one stage: [my pieces of stage kt+1 landed] [my fragment reads of stage kt done] barrier, then the 24 MFMAs of stage kt
    // (product-major: hi.hi, hi.lo, lo.hi; consecutive MFMAs hit different accumulators) with, from the third on, one filler each:
    // the four DMA pieces of stage kt+3, then the twelve fragment reads of stage kt+1 into the other register set
    auto stage = [&](auto set_c, int kt) {
        constexpr int S = decltype(set_c)::value;
        if (kt + 2 <= nk - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more_dma = kt + RS < nk, more_rd = kt + 1 < nk;
        int slot_i = 0;
        auto filler = [&]() {
            const int f = slot_i++ - 2;
            if (f >= 0 && f < NPW) { if (more_dma) { if (f < 2) dma_tok(kt + RS, f); else dma_w(kt + RS, f - 2); } }
            else if (f >= NPW && f < NPW + 12) { if (more_rd) read_frag(1 - S, kt + 1, f - NPW); }
            PIN();
        };
        PIN();
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { acc[ft][tt] = mfma16(ah[S][ft], bh[S][tt], acc[ft][tt]); filler(); }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { acc[ft][tt] = mfma16(ah[S][ft], bl[S][tt], acc[ft][tt]); filler(); }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) { acc[ft][tt] = mfma16(al[S][ft], bh[S][tt], acc[ft][tt]); filler(); }
    };
    if (!(GEMM_EXP & 2)) {
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            stage(std::integral_constant<int, 0>{}, kt);
            if (kt + 1 < nk) stage(std::integral_constant<int, 1>{}, kt + 1);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // All tile-level decisions are made on wave-uniform values (block / wave indices), so the fast
    // path is straight-line code with 16-byte loads and stores; only a tile that straddles N (the
    // 65-channel detector head) takes the element-wise path.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> inline-asm permlane reads
    __builtin_amdgcn_s_barrier();                         // s_par visible (also when the k-loop was empty)
    float* C = p.C ? p.C + z * p.strideC : nullptr;
    const int wfu = wu / TOKW;
    // Fused L2 normalisation of the output rows (SuperPoint's descriptor head: convDb then F.normalize over the 256 channels,
    // superpoint.py:233-236 -- N == TN, the whole row lives in this workgroup).  Pass 1: bias in place, squared sums per token
    // (a lane and its partner at lane ^ 32 hold the two halves of a 32-feature tile of the SAME token; the other 128 features sit in
    // the wave with the other wfu and meet through LDS -- the rings are free after the barrier above).  The store loop below then
    // scales by 1 / max(||row||, 1e-12) instead of adding the bias.
    float rnorm[2] = {1.f, 1.f};
    constexpr bool l2 = L2NORM;  // its own instantiation (in the hot ones the extra live values cost qkv 10 %); launcher: N == 256, fp32 output,
    if (l2) {                    // no rotary / ReLU / affine / residual
        float ss[2] = {0.f, 0.f};
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float b = s_par[wfu * 128 + ft * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const float v = fma_winv(acc[ft][tt][r], p.winv, b);
                    acc[ft][tt][r] = v;
                    ss[tt] = __builtin_fmaf(v, v, ss[tt]);
                }
            }
        float* s_red = reinterpret_cast<float*>(s_ring);  // [feature half][token of the tile]
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            ss[tt] += __shfl_xor(ss[tt], 32);
            if (h == 0) s_red[wfu * TM + wt * 64 + tt * 32 + i32] = ss[tt];
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tl = wt * 64 + tt * 32 + i32;
            rnorm[tt] = 1.f / fmaxf(sqrtf(s_red[tl] + s_red[TM + tl]), 1e-12f);
        }
    }
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const int fbase = n0 + wfu * 128 + ft * 32;  // uniform
        const int fb = fbase + 16 * h;               // first of this lane's 16 consecutive features
        const bool full = fbase + 32 <= p.N;
        if (fbase >= p.N) continue;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tok = m0 + wt * 64 + tt * 32 + i32;
            const bool tok_ok = tok < p.M;
            const long long tokc = tok_ok ? tok : 0;
            f32x16 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = acc[ft][tt][e], b0 = acc[ft][tt][8 + e], a1 = acc[ft][tt][4 + e], b1 = acc[ft][tt][12 + e];
                half_swap(a0, b0);
                half_swap(a1, b1);
                w[e] = a0; w[4 + e] = b0; w[8 + e] = a1; w[12 + e] = b1;
            }
            if (full) {
                if (l2) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) w[j] *= rnorm[tt];
                } else {
                    const float* bp = s_par + (fb - n0);
#pragma unroll
                    for (int j = 0; j < 16; j += 4) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + j);  // acc x winv (power of two: exact) + bias, one rounding
                        w[j] = fma_winv(w[j], p.winv, bv.x); w[j + 1] = fma_winv(w[j + 1], p.winv, bv.y);
                        w[j + 2] = fma_winv(w[j + 2], p.winv, bv.z); w[j + 3] = fma_winv(w[j + 3], p.winv, bv.w);
                    }
                }
                if (fbase < p.rotary_cols) {  // lightglue.py:36-43, pairs (2i, 2i+1); 8 frequencies per lane
                    const float* er = p.enc + tokc * 64 + ((fb & 63) >> 1);
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(er), c1 = *reinterpret_cast<const f32x4*>(er + 4);
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(er + 32), s1 = *reinterpret_cast<const f32x4*>(er + 36);
                    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xe = w[2 * j], xo = w[2 * j + 1];
                        w[2 * j] = __fadd_rn(__fmul_rn(xe, cs[j]), __fmul_rn(-xo, sn[j]));
                        w[2 * j + 1] = __fadd_rn(__fmul_rn(xo, cs[j]), __fmul_rn(xe, sn[j]));
                    }
                }
                if (p.relu) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) w[j] = fmaxf(w[j], 0.f);
                }
                if (p.post_scale) {
#pragma unroll
                    for (int j = 0; j < 16; j += 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(s_par + TN + (fb - n0) + j);
                        const f32x4 c = *reinterpret_cast<const f32x4*>(s_par + 2 * TN + (fb - n0) + j);
                        w[j] = __fadd_rn(__fmul_rn(w[j], a.x), c.x); w[j + 1] = __fadd_rn(__fmul_rn(w[j + 1], a.y), c.y);
                        w[j + 2] = __fadd_rn(__fmul_rn(w[j + 2], a.z), c.z); w[j + 3] = __fadd_rn(__fmul_rn(w[j + 3], a.w), c.w);
                    }
                }
                if (fbase < p.scale_cols) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) w[j] *= p.scale;
                }
                if (p.residual && tok_ok) {
                    const float* rs = p.residual + tokc * p.ldr + fb;
#pragma unroll
                    for (int j = 0; j < 16; j += 4) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(rs + j);
                        w[j] += rv.x; w[j + 1] += rv.y; w[j + 2] += rv.z; w[j + 3] += rv.w;
                    }
                }
                // Stores: the lane's 64 contiguous bytes go out as four 16-byte pieces.  A quad transpose first
                // (common.h) makes store s write token s of every quad with 4 lanes x 16 B, and lanes L / L+32
                // complete the 128-byte line: full-line writes instead of 64 partial lines per instruction.
                const int ql = i32 & 3;
                const int tq = m0 + wt * 64 + tt * 32 + (i32 & ~3);  // first token of this lane's quad
                if (C && (p.ldc & 3) == 0) {
                    u32x4 q0 = __builtin_bit_cast(u32x4, f32x4{w[0], w[1], w[2], w[3]}), q1 = __builtin_bit_cast(u32x4, f32x4{w[4], w[5], w[6], w[7]});
                    u32x4 q2 = __builtin_bit_cast(u32x4, f32x4{w[8], w[9], w[10], w[11]}), q3 = __builtin_bit_cast(u32x4, f32x4{w[12], w[13], w[14], w[15]});
                    quad_transpose(q0, q1, q2, q3, ql);
                    if (!(GEMM_EXP & 1)) {
#pragma unroll
                        for (int sidx = 0; sidx < 4; ++sidx)
                            if (tq + sidx < p.M)
                                *reinterpret_cast<u32x4*>(C + (long long)(tq + sidx) * p.ldc + fb + 4 * ql) = sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3;
                    }
                } else if (C && tok_ok && !(GEMM_EXP & 1)) {
                    float* o = C + tokc * p.ldc + fb;
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[j] = w[j];
                }
                if (p.Cs) {
                    const f32x8 g0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
                    const f32x8 g1 = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
                    f16x8 hi0, lo0, hi1, lo1;
                    split8(g0, hi0, lo0);
                    split8(g1, hi1, lo1);
                    u32x4 q0 = __builtin_bit_cast(u32x4, hi0), q1 = __builtin_bit_cast(u32x4, lo0);
                    u32x4 q2 = __builtin_bit_cast(u32x4, hi1), q3 = __builtin_bit_cast(u32x4, lo1);
                    quad_transpose(q0, q1, q2, q3, ql);
                    u32x4* ob = reinterpret_cast<u32x4*>(p.Cs) + z * (p.strideCs >> 2);
                    if (!(GEMM_EXP & 1)) {
#pragma unroll
                        for (int sidx = 0; sidx < 4; ++sidx)
                            if (tq + sidx < p.M)
                                ob[(((long long)(tq + sidx) * p.ldcs + fb) >> 2) + ql] = sidx == 0 ? q0 : sidx == 1 ? q1 : sidx == 2 ? q2 : q3;
                    }
                }
            } else {
                if (RAGGED) epilogue_ragged(p, C, w, fb, tokc, tok_ok);  // only instantiated when N % 32 != 0 (keeps scratch out of the hot kernel)
            }
        }
    }
}

// fp32 rows -> SH8 rows (entry of the LightGlue residual stream; C % 8 == 0).  One thread per group of 8.
__global__ __launch_bounds__(256) void to_sh8_kernel(const float* __restrict__ in, void* __restrict__ out, long long groups, float scale) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(in + g * 8) * scale, b = *reinterpret_cast<const f32x4*>(in + g * 8 + 4) * scale;
    const f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 hi, lo;
    split8(v, hi, lo);
    f16x8* o = reinterpret_cast<f16x8*>(out) + g * 2;
    o[0] = hi;
    o[1] = lo;
}

// Entry of the LightGlue residual stream in f16x3 mode: the two descriptor sets -> the fp32 stream X and its SH8 copy, in ONE pass
// (round 4; was two device copies and a conversion: 268 MB moved instead of 201, three launches instead of one)
__global__ __launch_bounds__(256) void copy2_to_sh8_kernel(const float* __restrict__ in0, long long groups0, const float* __restrict__ in1, long long groups,
                                                           float* __restrict__ x, void* __restrict__ out) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const float* in = g < groups0 ? in0 + g * 8 : in1 + (g - groups0) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(in), b = *reinterpret_cast<const f32x4*>(in + 4);
    *reinterpret_cast<f32x4*>(x + g * 8) = a;
    *reinterpret_cast<f32x4*>(x + g * 8 + 4) = b;
    const f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 hi, lo;
    split8(v, hi, lo);
    f16x8* o = reinterpret_cast<f16x8*>(out) + g * 2;
    o[0] = hi;
    o[1] = lo;
}

// SH8 rows -> fp32 rows (test entries only)
__global__ __launch_bounds__(256) void from_sh8_kernel(const void* __restrict__ in, float* __restrict__ out, long long groups) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= groups) return;
    const f16x8* i = reinterpret_cast<const f16x8*>(in) + g * 2;
    const f16x8 hi = i[0], lo = i[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) out[g * 8 + j] = (float)hi[j] + (float)lo[j];
}

}  // namespace

void launch_gemm_f16x3(const GemmF16Args& a0, hipStream_t s) {
    if (!a0.generic_only && launch_gemm_ws_f16x3(a0, s)) return;  // the QKV projections (K = 256): weights stationary in registers
    GemmF16Args a = a0;
    // two 4-wave workgroups per CU (TOKW = 2; the 8-wave tile of round 1 measured 1-5 % slower)
    constexpr int tokw = 2;
    // second-placed workgroups start `stagger` sleeps late; sweep on one box: 0 / 1 / 2 / 3 / 5 -> qkv 1.081 / 1.058 / 1.033 / 1.025 / 1.090 ms per step
    constexpr int stagger = 3;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    a.stagger = tokw == 2 ? stagger : 0;
    a.stagger_lo = cus;
    a.stagger_hi = 2 * cus;
    const int TM = 64 * tokw;
    const int ntt = (a.M + TM - 1) / TM, nft = (a.N + TN - 1) / TN;
    dim3 grid(((ntt + 7) / 8) * 8 * nft, 1, a.batch);
    const bool ragged = (a.N & 31) != 0;
#define MG_GEMM_LAUNCH(RM, RG)                                                                                      \
    do {                                                                                                            \
        hipLaunchKernelGGL((gemm_f16x3_kernel<RM, RG, 2>), grid, dim3(256), 0, s, a);                                 \
    } while (0)
    if (a.l2norm) {
        hipLaunchKernelGGL((gemm_f16x3_kernel<false, false, 2, true>), grid, dim3(256), 0, s, a);
    } else if (a.w_rowmajor) {
        if (ragged) MG_GEMM_LAUNCH(true, true);
        else MG_GEMM_LAUNCH(true, false);
    } else {
        if (ragged) MG_GEMM_LAUNCH(false, true);
        else MG_GEMM_LAUNCH(false, false);
    }
#undef MG_GEMM_LAUNCH
}

void launch_to_sh8(const float* in, void* out, long long elems, float scale, hipStream_t s) {
    const long long groups = elems / 8;
    hipLaunchKernelGGL(to_sh8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, in, out, groups, scale);
}

void launch_copy2_to_sh8(const float* in0, long long elems0, const float* in1, long long elems1, float* x, void* out, hipStream_t s) {
    const long long groups0 = elems0 / 8, groups = groups0 + elems1 / 8;
    if (groups > 0) hipLaunchKernelGGL(copy2_to_sh8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, in0, groups0, in1, groups, x, out);
}

void launch_from_sh8(const void* in, float* out, long long elems, hipStream_t s) {
    const long long groups = elems / 8;
    hipLaunchKernelGGL(from_sh8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, in, out, groups);
}
