// Operand / result layout of v_mfma_f32_16x16x32_f16 as conv1ab's conv1a step assumes it (not part of the product):
//   A[i][k]: lane i + 16 (k / 8), element k % 8;   B[k][j]: lane j + 16 (k / 8), element k % 8;   D[i][j]: lane j + 16 (i / 4), register i % 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D) {  // A (16,32) row-major, B (32,16) row-major, D (16,16)
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 32 + 8 * q + e]; b[e] = (_Float16)B[(8 * q + e) * 16 + i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r];
}

int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    unsigned s = 1;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 20) % 17 - 8); }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 20) % 13 - 6); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float t = 0; for (int k = 0; k < 32; ++k) t += hA[i * 32 + k] * hB[k * 16 + j]; ref[i * 16 + j] = t; }
    float *dA, *dB, *dD;
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    printf("v_mfma_f32_16x16x32_f16 layout: %d of 256 entries differ from A.B (%s)\n", bad, bad ? "ASSUMED LAYOUT WRONG" : "layout as assumed");
    return bad != 0;
}
