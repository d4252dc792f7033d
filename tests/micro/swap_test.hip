#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../glue-factory_amd/csrc/common.h"
__global__ void k(float* out) {
    int l = threadIdx.x;
    float a = (float)l, b = 100.f + l;
    half_swap(a, b);
    out[l] = a; out[64 + l] = b;
    // dpp test
    float v = (float)l;
    out[128 + l] = dpp_quad_xor1(v);
    out[192 + l] = dpp_quad_xor2(v);
}
int main() {
    float* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    printf("a: "); for (int i = 0; i < 64; i += 8) printf("[%d]=%g ", i, h[i]); printf("\nb: ");
    for (int i = 0; i < 64; i += 8) printf("[%d]=%g ", i, h[64 + i]);
    printf("\nxor1: %g %g %g %g  xor2: %g %g %g %g\n", h[128], h[129], h[130], h[131], h[192], h[193], h[194], h[195]);
    return 0;
}
