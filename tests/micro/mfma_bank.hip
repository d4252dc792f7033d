// Micro-benchmark (not part of the product): does the VGPR bank alignment of an MFMA's operands change its issue rate?
// One wave per SIMD, 4 accumulators in rotation, hard-coded registers.  "same": A, B and C all start at a register index = 2 (mod 4) --
// what hipcc allocated in ffn_f16.hip (v[98:101], v[70:73], v[50:65]); the other rows move A, B or C to index = 0 (mod 4).
//     hipcc --offload-arch=gfx950 -O3 -o mfma_bank mfma_bank.hip && ./mfma_bank
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CLOB "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33", \
             "v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65", \
             "v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v96","v97","v98","v99","v100","v101","v102","v103","v104"

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void bank_kernel(unsigned long long* cycles, int iters) {
    unsigned long long t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // A 2, B 2, C 2 (the allocation found in ffn_f16.hip)
            asm volatile("v_mfma_f32_32x32x16_f16 v[50:65], v[98:101], v[70:73], v[50:65]\n\tv_mfma_f32_32x32x16_f16 v[34:49], v[98:101], v[66:69], v[34:49]\n\tv_mfma_f32_32x32x16_f16 v[18:33], v[82:85], v[70:73], v[18:33]\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[82:85], v[66:69], v[2:17]\n\tv_mfma_f32_32x32x16_f16 v[50:65], v[98:101], v[78:81], v[50:65]\n\tv_mfma_f32_32x32x16_f16 v[34:49], v[98:101], v[74:77], v[34:49]\n\tv_mfma_f32_32x32x16_f16 v[18:33], v[82:85], v[78:81], v[18:33]\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[82:85], v[74:77], v[2:17]\n\t" ::: CLOB);
        }
        else if (MODE == 1) {  // A 0, B 0, C 2
            asm volatile("v_mfma_f32_32x32x16_f16 v[50:65], v[96:99], v[68:71], v[50:65]\n\tv_mfma_f32_32x32x16_f16 v[34:49], v[96:99], v[72:75], v[34:49]\n\tv_mfma_f32_32x32x16_f16 v[18:33], v[100:103], v[68:71], v[18:33]\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[100:103], v[72:75], v[2:17]\n\tv_mfma_f32_32x32x16_f16 v[50:65], v[96:99], v[76:79], v[50:65]\n\tv_mfma_f32_32x32x16_f16 v[34:49], v[96:99], v[80:83], v[34:49]\n\tv_mfma_f32_32x32x16_f16 v[18:33], v[100:103], v[76:79], v[18:33]\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[100:103], v[80:83], v[2:17]\n\t" ::: CLOB);
        }
        else if (MODE == 2) {  // A 0, B 2, C 2
            asm volatile("v_mfma_f32_32x32x16_f16 v[50:65], v[96:99], v[70:73], v[50:65]\n\tv_mfma_f32_32x32x16_f16 v[34:49], v[96:99], v[66:69], v[34:49]\n\tv_mfma_f32_32x32x16_f16 v[18:33], v[100:103], v[70:73], v[18:33]\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[100:103], v[66:69], v[2:17]\n\tv_mfma_f32_32x32x16_f16 v[50:65], v[96:99], v[78:81], v[50:65]\n\tv_mfma_f32_32x32x16_f16 v[34:49], v[96:99], v[74:77], v[34:49]\n\tv_mfma_f32_32x32x16_f16 v[18:33], v[100:103], v[78:81], v[18:33]\n\tv_mfma_f32_32x32x16_f16 v[2:17], v[100:103], v[74:77], v[2:17]\n\t" ::: CLOB);
        }
        else if (MODE == 3) {  // A 2, B 2, C 0
            asm volatile("v_mfma_f32_32x32x16_f16 v[52:67], v[98:101], v[70:73], v[52:67]\n\tv_mfma_f32_32x32x16_f16 v[36:51], v[98:101], v[74:77], v[36:51]\n\tv_mfma_f32_32x32x16_f16 v[20:35], v[82:85], v[70:73], v[20:35]\n\tv_mfma_f32_32x32x16_f16 v[4:19], v[82:85], v[74:77], v[4:19]\n\tv_mfma_f32_32x32x16_f16 v[52:67], v[98:101], v[78:81], v[52:67]\n\tv_mfma_f32_32x32x16_f16 v[36:51], v[98:101], v[86:89], v[36:51]\n\tv_mfma_f32_32x32x16_f16 v[20:35], v[82:85], v[78:81], v[20:35]\n\tv_mfma_f32_32x32x16_f16 v[4:19], v[82:85], v[86:89], v[4:19]\n\t" ::: CLOB);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int WAVES>
static double run(int iters) {
    unsigned long long* d;
    hipMalloc(&d, 256 * WAVES * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bank_kernel<MODE, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    std::vector<unsigned long long> h(256 * WAVES);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / (8.0 * iters * (WAVES / 4));
}

int main() {
    const int it = 20000;
    printf("cycles per MFMA of a SIMD (v_mfma_f32_32x32x16_f16, VGPR accumulators), by the alignment (register index mod 4) of A / B / C:\n");
    printf("  (tuples must start at an even register: the alignments are 0 or 2)\n");
    printf("  one wave per SIMD:  2/2/2 %5.2f   0/0/2 %5.2f   0/2/2 %5.2f   2/2/0 %5.2f\n", run<0, 4>(it), run<1, 4>(it), run<2, 4>(it), run<3, 4>(it));
    printf("  two waves per SIMD: 2/2/2 %5.2f   0/0/2 %5.2f   0/2/2 %5.2f   2/2/0 %5.2f\n", run<0, 8>(it), run<1, 8>(it), run<2, 8>(it), run<3, 8>(it));
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
