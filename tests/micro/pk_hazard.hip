// Minimal reproducer for "hazard 1" of DESIGN.md section 4 (not part of the product): does a packed-fp32 FMA whose operand is
// broadcast through op_sel / op_sel_hi return wrong halves while MFMAs execute on the same SIMD (gfx950)?
//
// Every lane evaluates r = {g, g} * {w0, w1} + {c0, c1} with ONE v_pk_fma_f32 in four operand forms and checks both halves against
// scalar v_fma_f32 results computed in the same iteration:
//   form 0  real pair   v_pk_fma_f32 r, gg, w, c                       (gg = {g, g} materialised in two registers)
//   form 1  op_sel_hi   v_pk_fma_f32 r, gp, w, c op_sel_hi:[0,1,1]      (gp = {g, junk}: high lane reads the LOW half of src0)
//   form 2  op_sel      v_pk_fma_f32 r, pg, w, c op_sel:[1,0,0]         (pg = {junk, g}: low lane reads the HIGH half of src0)
//   form 3  src1 form   v_pk_fma_f32 r, w, gp, c op_sel_hi:[1,0,1]      (the form hipcc emitted in conv1ab: broadcast on src1)
// in three surroundings: (a) no MFMA anywhere in the kernel, (b) the wave's own independent MFMAs issued right before and after each
// packed instruction, (c) this wave runs no MFMA but its SIMD partner wave does (a workgroup's waves go to the SIMDs cyclically, so waves w
// and w + 4 of a 512-thread workgroup share a SIMD: waves 4-7 run the MFMAs, waves 0-3 the packed FMAs).  Counts of wrong LOW and wrong HIGH halves per
// (surrounding, form) are printed; all registers of the packed instruction are disjoint from the MFMA operands and accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rnd(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return (float)((int)(s >> 8) & 0xffff) * (1.0f / 32768.0f) - 1.0f;
}

#define PK(form, R, G, W, C)                                                                                         \
    if (form == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(R) : "v"(G), "v"(W), "v"(C));                 \
    else if (form == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(R) : "v"(G), "v"(W), "v"(C)); \
    else if (form == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(R) : "v"(G), "v"(W), "v"(C));   \
    else asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel_hi:[1,0,1]" : "=v"(R) : "v"(G), "v"(W), "v"(C));

// surround: 0 = no MFMA, 1 = own MFMAs around every packed instruction, 2 = waves 4-7 run only MFMAs, their SIMD partners 0-3 only the packed FMAs
template <int SURROUND, int FORM>
__global__ __launch_bounds__(512) void pk_probe(unsigned* bad, float* sink, int iters) {
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 977u * FORM + 131u * SURROUND;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)rnd(s); b[j] = (_Float16)rnd(s); }
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = SURROUND == 1 || (SURROUND == 2 && (wave & 4));
    const bool do_pk = SURROUND != 2 || !(wave & 4);
    unsigned bad_lo = 0, bad_hi = 0;
    for (int i = 0; i < iters; ++i) {
        const float g = rnd(s), w0 = rnd(s), w1 = rnd(s), c0 = rnd(s), c1 = rnd(s), junk = rnd(s) * 1024.0f;
        f32x2 G = FORM == 0 ? f32x2{g, g} : FORM == 2 ? f32x2{junk, g} : f32x2{g, junk};
        f32x2 W = {w0, w1}, C = {c0, c1}, R = {0, 0};
        float e0, e1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(g), "v"(w0), "v"(c0));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(g), "v"(w1), "v"(c1));
        if (do_mfma) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        }
        if (do_pk) { PK(FORM, R, G, W, C) }
        if (do_mfma) {
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc3, 0, 0, 0);
        }
        if (do_pk) {
            bad_lo += (R.x != e0);
            bad_hi += (R.y != e1);
        }
    }
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (t == 12345.678f) sink[threadIdx.x] = t;
    if (bad_lo) atomicAdd(&bad[0], bad_lo);
    if (bad_hi) atomicAdd(&bad[1], bad_hi);
}

template <int S, int F>
static void run(unsigned* bad, float* sink, int iters) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL((pk_probe<S, F>), dim3(1024), dim3(512), 0, 0, bad, sink, iters);
    unsigned h[2];
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    static const char* sn[] = {"no MFMA in the kernel      ", "own MFMAs around each pk   ", "MFMAs on SIMD partner waves"};
    static const char* fn[] = {"real pair          ", "src0 op_sel_hi:[0,1,1]", "src0 op_sel:[1,0,0]   ", "src1 op_sel_hi:[1,0,1]"};
    const double n = (S == 2 ? 0.5 : 1.0) * 1024.0 * 512.0 * iters;
    printf("%s | %s | wrong low halves %10u  wrong high halves %10u  of %.3g packed FMAs\n", sn[S], fn[F], h[0], h[1], n);
}

int main() {
    unsigned* bad; float* sink;
    hipMalloc(&bad, 8); hipMalloc(&sink, 4096);
    const int iters = 20000;
    run<0, 0>(bad, sink, iters); run<0, 1>(bad, sink, iters); run<0, 2>(bad, sink, iters); run<0, 3>(bad, sink, iters);
    run<1, 0>(bad, sink, iters); run<1, 1>(bad, sink, iters); run<1, 2>(bad, sink, iters); run<1, 3>(bad, sink, iters);
    run<2, 0>(bad, sink, iters); run<2, 1>(bad, sink, iters); run<2, 2>(bad, sink, iters); run<2, 3>(bad, sink, iters);
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
