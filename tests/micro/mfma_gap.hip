// Micro-benchmark (not part of the product): what fits into the shadow of a v_mfma_f32_32x32x16_f16 when ONE wave per SIMD issues both
// the MFMAs and the vector work (the regime of attention_pw_f16.hip).  One workgroup of 256 threads per CU; every wave loops over
// 8 x { MFMA (two alternating accumulators); NF fillers } and the cost is read in SHADER cycles (s_memtime) per MFMA.  Varied: the
// accumulator file of the MFMA (arch VGPRs, as the S chains of the attention kernel, or AGPRs, as its O chains), the filler kind and
// the number of fillers per gap.  Fillers rotate over 8 registers (dependency distance 8) except in the "dependent" kinds, which chain
// every filler on the previous one -- the latency a single wave cannot hide.
//     hipcc --offload-arch=gfx950 -O3 -o mfma_gap mfma_gap.hip && ./mfma_gap
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum { F_ADD = 0, F_EXP = 1, F_MIX = 2, F_ADD_DEP = 3, F_EXP_DEP = 4, F_SOFTMAX = 5, F_LDS = 6, F_CVT = 7, F_ACCREAD = 8, F_WAITCNT = 9, F_SNOP = 10, F_SALU = 11, F_CNDMASK = 12, F_DPP = 13, F_FMA = 14, F_DMA = 15, F_GLOAD = 16, F_DMA_BUF = 17 };

template <int ACC_AGPR, int KIND, int NF>
__global__ __launch_bounds__(256) void gap_kernel(unsigned long long* cycles, float* sink, int iters, const float* src = nullptr) {
    __shared__ float lds[4096 + 4 * 4 * 256];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 0.001f;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * ((threadIdx.x * 7 + j * 13) % 97)); b[j] = (_Float16)(0.02f * ((threadIdx.x * 11 + j * 5) % 89)); }
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = 0.001f * (threadIdx.x + 1) + 0.37f * i;
    float c = 0.999f;
    float spare[8];
    for (int i = 0; i < 8; ++i) spare[i] = 0.5f + i;
    int sreg = 0;
    const float* lp = lds + (threadIdx.x & 63) * 4;
    float4 ld = {0, 0, 0, 0};
    float4 gl[4] = {};
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t rsrc;
    {
        const unsigned long long a = (unsigned long long)src;
        rsrc[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
        rsrc[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffff);
        rsrc[2] = 0x7fffffffu;
        rsrc[3] = 0x00020000u;
    }
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + 4096 + (threadIdx.x >> 6) * 1024));
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (ACC_AGPR == 3) {  // accumulators AND the A operand in the accumulator file (weights of gemm_ws_f16.hip, Q of attention_pw_f16.hip)
                if (g & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "a"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "a"(a), "v"(b));
            } else if (ACC_AGPR == 2) {  // VGPR accumulators, A operand in the accumulator file
                if (g & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "a"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "a"(a), "v"(b));
            } else if (ACC_AGPR) {
                if (g & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            } else {
                if (g & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            }
            if ((KIND == F_DMA || KIND == F_GLOAD || KIND == F_DMA_BUF) && NF > 0 && g % NF == 0) {  // VMEM kinds: ONE request every NF-th gap (L2-resident source)
                const float* gp = src + ((blockIdx.x & 15) * 8 + g) * 256 + (threadIdx.x & 63) * 4;
                if (KIND == F_DMA) {
                    __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(gp), (__attribute__((address_space(3))) void*)(lds + 4096 + (threadIdx.x >> 6) * 1024 + (g & 3) * 256), 16, 0, 0);
                } else if (KIND == F_GLOAD) {
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl[g & 3]) : "v"(gp) : "memory");
                } else {
                    asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"((unsigned)(((blockIdx.x & 15) * 8 + g) * 1024 + (threadIdx.x & 63) * 16)), "s"(rsrc), "s"(m0v + (g & 3) * 1024) : "memory");
                }
            }
#pragma unroll
            for (int f = 0; f < ((KIND == F_DMA || KIND == F_GLOAD || KIND == F_DMA_BUF) ? 0 : NF); ++f) {
                const int i = (g * NF + f) & 7;
                if (KIND == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                else if (KIND == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                else if (KIND == F_MIX) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(c));
                else if (KIND == F_CVT) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                else if (KIND == F_ADD_DEP) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(c));
                else if (KIND == F_EXP_DEP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[0]));
                else if (KIND == F_SOFTMAX) {  // the attention mix: of every 5 fillers one exp, two adds, one cvt, one mix (independent)
                    const int k = f % 5;
                    if (k == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                    else if (k == 1 || k == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                    else if (k == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
                    else asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(c));
                } else if (KIND == F_ACCREAD) {  // a VGPR <- AGPR copy of a register no MFMA of the loop touches (what hipcc emits when the vector pipe reads an accumulator)
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r[i]) : "a"(spare[i]));
                } else if (KIND == F_WAITCNT) {
                    asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                } else if (KIND == F_SNOP) {
                    asm volatile("s_nop 0" ::: "memory");
                } else if (KIND == F_SALU) {
                    asm volatile("s_add_i32 %0, %0, 1" : "+s"(sreg));
                } else if (KIND == F_CNDMASK) {
                    asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c) : );
                } else if (KIND == F_DPP) {
                    asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
                } else if (KIND == F_FMA) {
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
                } else if (KIND == F_LDS) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)lp) : "memory");
                }
            }
        }
        if (KIND == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (KIND == F_DMA || KIND == F_GLOAD || KIND == F_DMA_BUF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = ld.x + ld.y + (float)sreg + gl[0].x + gl[1].y + gl[2].z + gl[3].w + lds[4096 + threadIdx.x];
    for (int i = 0; i < 8; ++i) s += r[i];
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACC_AGPR, int KIND, int NF>
static double run(int iters) {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 256 * 4 * 8);
    hipMalloc(&sink, 4096);
    static float* src = nullptr;
    if (!src) { hipMalloc(&src, 16 * 8 * 1024 + 4096); hipMemset(src, 0, 16 * 8 * 1024 + 4096); }
    hipLaunchKernelGGL((gap_kernel<ACC_AGPR, KIND, NF>), dim3(256), dim3(256), 0, 0, d, sink, iters, src);
    hipLaunchKernelGGL((gap_kernel<ACC_AGPR, KIND, NF>), dim3(256), dim3(256), 0, 0, d, sink, iters, src);
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    hipFree(sink);
    std::sort(h.begin(), h.end());
    return (double)h[512] / (8.0 * iters);  // median over the 1024 waves: shader cycles per MFMA gap
}

template <int KIND>
static void row(const char* name) {
    const int it = 4000;
    printf("%-34s VGPR acc:", name);
    printf(" %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f |", run<0, KIND, 0>(it), run<0, KIND, 2>(it), run<0, KIND, 3>(it), run<0, KIND, 4>(it), run<0, KIND, 5>(it),
           run<0, KIND, 6>(it), run<0, KIND, 8>(it));
    printf(" AGPR acc: %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f %5.1f\n", run<1, KIND, 0>(it), run<1, KIND, 2>(it), run<1, KIND, 3>(it), run<1, KIND, 4>(it),
           run<1, KIND, 5>(it), run<1, KIND, 6>(it), run<1, KIND, 8>(it));
}

int main() {
    printf("A operand in the accumulator file, no fillers / 4 v_add: VGPR acc %.1f / %.1f, AGPR acc %.1f / %.1f cycles per MFMA\n", run<2, F_ADD, 0>(4000), run<2, F_ADD, 4>(4000),
           run<3, F_ADD, 0>(4000), run<3, F_ADD, 4>(4000));
    printf("shader cycles per MFMA gap (v_mfma_f32_32x32x16_f16, one wave per SIMD); fillers per gap: 0 2 3 4 5 6 8\n");
    row<F_ADD>("v_add_f32 (independent)");
    row<F_EXP>("v_exp_f32 (independent)");
    row<F_MIX>("v_fma_mixlo_f16 (independent)");
    row<F_CVT>("v_cvt_pkrtz_f16_f32 (independent)");
    row<F_SOFTMAX>("soft-max mix exp/add/add/cvt/mix");
    row<F_ADD_DEP>("v_add_f32 (one dependent chain)");
    row<F_EXP_DEP>("v_exp_f32 (one dependent chain)");
    row<F_LDS>("ds_read_b128");
    row<F_FMA>("v_fma_f32 (independent)");
    row<F_CNDMASK>("v_cndmask_b32 (vcc)");
    row<F_DPP>("v_mov_b32_dpp quad_perm");
    row<F_ACCREAD>("v_accvgpr_read_b32");
    row<F_WAITCNT>("s_waitcnt lgkmcnt(7) (nothing pending)");
    row<F_SNOP>("s_nop 0");
    row<F_SALU>("s_add_i32");
    printf("VMEM requests of 1 KB per wave (16 B per lane, L2-resident source), one every N-th gap; cycles per MFMA gap for N = none 1 2 4 8\n");
    printf("%-34s AGPR acc: %5.1f %5.1f %5.1f %5.1f %5.1f\n", "global_load_lds_dwordx4 (LDS-DMA)", run<1, F_DMA, 0>(2000), run<1, F_DMA, 1>(2000), run<1, F_DMA, 2>(2000), run<1, F_DMA, 4>(2000), run<1, F_DMA, 8>(2000));
    printf("%-34s AGPR acc: %5.1f %5.1f %5.1f %5.1f %5.1f\n", "buffer_load_dwordx4 ... lds", run<1, F_DMA_BUF, 0>(2000), run<1, F_DMA_BUF, 1>(2000), run<1, F_DMA_BUF, 2>(2000), run<1, F_DMA_BUF, 4>(2000), run<1, F_DMA_BUF, 8>(2000));
    printf("%-34s AGPR acc: %5.1f %5.1f %5.1f %5.1f %5.1f\n", "global_load_dwordx4 (registers)", run<1, F_GLOAD, 0>(2000), run<1, F_GLOAD, 1>(2000), run<1, F_GLOAD, 2>(2000), run<1, F_GLOAD, 4>(2000), run<1, F_GLOAD, 8>(2000));
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
