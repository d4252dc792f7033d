// Micro-benchmark (not part of the product): issue cost of the VALU instructions the attention soft-max is made of, measured instead
// of taken from a price list.  One workgroup per CU, 256 or 512 threads = 1 or 2 waves per SIMD; every wave
// runs a loop of 64 INDEPENDENT instances of one instruction (8 registers x 8 rounds per iteration, dependency distance 8), so the
// figure is the issue interval, not the latency.  Reported: shader cycles per instruction and SIMD (s_memtime runs at 100 MHz, the
// shader clock is taken from a v_add_f32 loop in the same launch configuration, whose cost is 4 cycles per wave64 instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(512) void loop(float* out, int iters) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = 0.001f * (threadIdx.x + 1) + 0.37f * i;
    float c = 0.999f;
    double d[8];
    for (int i = 0; i < 8; ++i) d[i] = 1e300 * (threadIdx.x + 1 + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int round = 0; round < 8; ++round) {
#define ONE(i)                                                                                                                            \
    if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                                                              \
    else if (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));                                                                      \
    else if (OP == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));                                               \
    else if (OP == 3) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(c));            \
    else if (OP == 4) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));                                                    \
    else if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));                                                     \
    else if (OP == 6) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(c));                                                 \
    else if (OP == 7) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));                   \
    else if (OP == 8) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));                                                                  \
    else if (OP == 9) asm volatile("v_log_f32 %0, %0" : "+v"(r[i]));                                                                      \
    else if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));                                                                     \
    else if (OP == 11) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(d[i]));
            REP8(ONE)
#undef ONE
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i] + (float)d[i];
    if (s == 12345.f) out[threadIdx.x] = s + c;
}

template <int OP>
static double run(int threads, int iters) {
    float* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(loop<OP>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(loop<OP>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1e-3 / (64.0 * iters * (threads / 256));  // seconds per instruction and SIMD (threads / 256 waves per SIMD, 64 instructions per iteration each)
}

int main() {
    const char* names[] = {"v_add_f32", "v_exp_f32", "v_cvt_pkrtz_f16_f32", "v_fma_mixlo_f16", "v_max3_f32", "v_fma_f32", "v_permlane32_swap_b32",
                           "v_mov_b32_dpp (quad_perm)", "v_cvt_f16_f32", "v_log_f32", "v_rcp_f32", "v_lshrrev_b64"};
    for (int threads : {256, 512}) {
        const int iters = 20000;
        double t[12];
        t[0] = run<0>(threads, iters); t[1] = run<1>(threads, iters); t[2] = run<2>(threads, iters); t[3] = run<3>(threads, iters);
        t[4] = run<4>(threads, iters); t[5] = run<5>(threads, iters); t[6] = run<6>(threads, iters); t[7] = run<7>(threads, iters);
        t[8] = run<8>(threads, iters); t[9] = run<9>(threads, iters); t[10] = run<10>(threads, iters); t[11] = run<11>(threads, iters);
        const double ghz = 4.0 / t[0] / 1e9;  // v_add_f32 issues every 4 cycles per wave64
        printf("%d waves per SIMD (shader clock from the v_add_f32 loop: %.2f GHz)\n", threads / 256, ghz);
        for (int i = 0; i < 12; ++i) printf("   %-28s %6.2f cycles per instruction\n", names[i], t[i] / t[0] * 4.0);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
