// Micro-check (not part of the product): is { hi = v_cvt_pk_f16_f32, lo = v_fma_mixlo/hi_f16(x - hi) } bit-identical to common.h's split8
// ( hi = (_Float16) x, lo = (_Float16)(x - (float) hi) ) on every input class -- normal values, values whose lo half is an fp16
// subnormal, negative values, values beyond the fp16 range?   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o split_check split_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* ref_hi, unsigned* ref_lo, unsigned* new_hi, unsigned* new_lo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    h16x2 rh = {h0, h1}, rl = {l0, l1};
    ref_hi[i] = __builtin_bit_cast(unsigned, rh);
    ref_lo[i] = __builtin_bit_cast(unsigned, rl);
    const f32x2 pr = {x0, x1};
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, h16x2));
    unsigned lo;
    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(x0));
    asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(x1));
    new_hi[i] = hi;
    new_lo[i] = lo;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const double e = -7.0 + 10.5 * (rand() / (double)RAND_MAX);  // 1e-7 .. 3e3
        h[i] = (float)((rand() & 1 ? 1 : -1) * pow(10.0, e) * (0.5 + rand() / (double)RAND_MAX));
    }
    float* dx; unsigned *a, *b, *c, *d;
    hipMalloc(&dx, n * 4); hipMalloc(&a, n * 2); hipMalloc(&b, n * 2); hipMalloc(&c, n * 2); hipMalloc(&d, n * 2);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, a, b, c, d, n);
    std::vector<unsigned> ra(n / 2), rb(n / 2), rc(n / 2), rd(n / 2);
    hipMemcpy(ra.data(), a, n * 2, hipMemcpyDeviceToHost); hipMemcpy(rb.data(), b, n * 2, hipMemcpyDeviceToHost);
    hipMemcpy(rc.data(), c, n * 2, hipMemcpyDeviceToHost); hipMemcpy(rd.data(), d, n * 2, hipMemcpyDeviceToHost);
    long bad_hi = 0, bad_lo = 0, bad_lo_sub = 0, shown = 0;
    for (int i = 0; i < n / 2; ++i) {
        if (ra[i] != rc[i]) ++bad_hi;
        if (rb[i] != rd[i]) {
            ++bad_lo;
            for (int s = 0; s < 2; ++s) {
                const unsigned r = (rb[i] >> (16 * s)) & 0xffff, w = (rd[i] >> (16 * s)) & 0xffff;
                if (r != w && (r & 0x7c00) == 0) ++bad_lo_sub;
                if (r != w && shown < 6) { printf("x = %.9g: lo ref 0x%04x new 0x%04x (hi 0x%04x)\n", h[2 * i + s], r, w, (ra[i] >> (16 * s)) & 0xffff); ++shown; }
            }
        }
    }
    printf("%d pairs: hi words differing %ld, lo words differing %ld (lo halves that are fp16 subnormals in the reference: %ld)\n", n / 2, bad_hi, bad_lo, bad_lo_sub);
    return hipDeviceSynchronize() != hipSuccess;
}
