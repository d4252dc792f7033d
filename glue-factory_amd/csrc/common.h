// Shared device helpers for the gfx950 kernels (wave64, fp32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
