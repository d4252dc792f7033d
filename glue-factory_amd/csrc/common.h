// Shared device helpers for the gfx950 kernels (wave64, fp32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // 16-byte staging piece

#define MG_WAVE 64

// v_mfma_f32_32x32x2_f32: D(32x32) += A(32x2) * B(2x32), exact fp32 FMA chain.
// Lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
// D layout: col j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5) for register r in [0,16).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One 16-byte fragment per operand feeds four MFMAs: lanes 0-31 hold k = k0..k0+3 and lanes 32-63
// hold k = k0+4..k0+7 of the same row, so register e contracts over {k0+e, k0+4+e}.  A and B use
// the same pairing, and the sum over an 8-wide k group is complete after the four steps.
__device__ __forceinline__ void mma_k8(const f32x4 a, const f32x4 b, f32x16& c) {
    c = mfma32(a.x, b.x, c);
    c = mfma32(a.y, b.y, c);
    c = mfma32(a.z, b.z, c);
    c = mfma32(a.w, b.w, c);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Order-preserving map float -> uint32 (larger float => larger key), and back.
__device__ __host__ __forceinline__ uint32_t float_to_key(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float key_to_float(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}

// ---- split-fp16 ("f16x3") arithmetic ------------------------------------------------------------
// x = hi + lo with hi = fp16(x), lo = fp16(x - hi): 22 significant bits.  A product is evaluated as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the lo*lo term, 2^-22
// relative, is dropped): fp32-class accuracy at 1/3 of the dense f16 MFMA rate = 5.3x the fp32 MFMA rate.
//
// SH8 storage of a (rows, C) fp32-equivalent tensor: per row, channels in groups of 8, each group
// = 16 B of hi halves followed by 16 B of lo halves -> the same 4 B/element as fp32, and every MFMA
// fragment (8 consecutive k for one row) is one aligned 16-byte piece.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// c += a*b with a = ah+al, b = bh+bl
__device__ __forceinline__ void mma3(const f16x8 ah, const f16x8 al, const f16x8 bh, const f16x8 bl, f32x16& c) {
    c = mfma16(ah, bh, c);
    c = mfma16(ah, bl, c);
    c = mfma16(al, bh, c);
}
// two accumulators sharing the a operand, products interleaved (consecutive MFMAs independent)
__device__ __forceinline__ void mma3x2(const f16x8 ah, const f16x8 al, const f16x8 bh0, const f16x8 bl0, const f16x8 bh1, const f16x8 bl1, f32x16& c0,
                                       f32x16& c1) {
    c0 = mfma16(ah, bh0, c0);
    c1 = mfma16(ah, bh1, c1);
    c0 = mfma16(ah, bl0, c0);
    c1 = mfma16(ah, bl1, c1);
    c0 = mfma16(al, bh0, c0);
    c1 = mfma16(al, bh1, c1);
}
// acc * winv + bias with winv a wave-uniform power of two (the scaling of uploaded weights, api.hip: pow2_scale): the product is
// exact, so the fma rounds once, exactly like "accumulate unscaled, then add the bias".
__device__ __forceinline__ float fma_winv(float acc, float winv, float bias) { return __builtin_fmaf(acc, winv, bias); }
// The same as ONE scalar-operand v_fma_f32 written in asm, for the translation unit that keeps hipcc's packed-fp32 code generation
// (ffn_f16.hip): there a run of plain fmas becomes v_pk_fma_f32 with an op_sel broadcast of the scalar, which build.py / pk_scan.py
// keep out of barrier intervals that hold MFMAs.  hipcc's hazard recognizer does not look inside asm statements, and MFMA builtins
// are free to move across an asm statement they share no operand with: use it only where scheduling fences (sched_barrier after the
// last MFMA) and mfma_results_guard() -- both volatile, so they keep their order -- separate it from the MFMAs that wrote `acc`.
__device__ __forceinline__ float fma_winv_asm(float acc, float winv, float bias) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(acc), "s"(winv), "v"(bias));
    return r;
}
__device__ __forceinline__ void mfma_results_guard() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8(const f32x8 v, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        hi[j] = h;
        lo[j] = (_Float16)(v[j] - (float)h);
    }
}
__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
}
// 4x4 transpose of 16-byte pieces across the four lanes of a quad: afterwards piece s of quad-lane l holds what
// was piece l of quad-lane s.  Used by the epilogues: a lane owns 64 contiguous output bytes (4 pieces) of one
// token / pixel, and storing them directly makes every store instruction touch 64 partial lines (measured
// 3.0 TB/s, tests/micro/bw_pattern.hip); after the transpose instruction s writes row s of every quad with 4 lanes
// x 16 B, lanes L and L+32 complete the 128-byte line, and the stores run at the coalesced rate (6.5 TB/s).
__device__ __forceinline__ unsigned dpp_quad_xor1_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned dpp_quad_xor2_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
}
__device__ __forceinline__ void quad_transpose(u32x4& p0, u32x4& p1, u32x4& p2, u32x4& p3, int lane) {
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned r = dpp_quad_xor1_u(odd ? p0[e] : p1[e]);
        p0[e] = odd ? r : p0[e];
        p1[e] = odd ? p1[e] : r;
        r = dpp_quad_xor1_u(odd ? p2[e] : p3[e]);
        p2[e] = odd ? r : p2[e];
        p3[e] = odd ? p3[e] : r;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned r = dpp_quad_xor2_u(hi ? p0[e] : p2[e]);
        p0[e] = hi ? r : p0[e];
        p2[e] = hi ? p2[e] : r;
        r = dpp_quad_xor2_u(hi ? p1[e] : p3[e]);
        p1[e] = hi ? r : p1[e];
        p3[e] = hi ? p3[e] : r;
    }
}
// piece q of (p0..p3) for quad-lane q (all four lanes hold the same data, e.g. after a 2x2 max-pool)
__device__ __forceinline__ u32x4 quad_pick(const u32x4& p0, const u32x4& p1, const u32x4& p2, const u32x4& p3, int lane) {
    const u32x4 a = (lane & 1) ? p1 : p0, b = (lane & 1) ? p3 : p2;
    return (lane & 2) ? b : a;
}
// Exchange: lanes 32-63 of a <-> lanes 0-31 of b (v_permlane32_swap).  Inline asm on purpose: with
// ROCm 7.2's hipcc the second result of __builtin_amdgcn_permlane32_swap is lowered to the first one
// (tests/micro/swap_test.hip shows it).  The two v_nop cover the VALU-write -> permlane-read hazard.
__device__ __forceinline__ void half_swap(float& a, float& b) {
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
