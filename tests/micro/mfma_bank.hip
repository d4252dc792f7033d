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

template <int MODE, int WAVES, int DATA>
__global__ __launch_bounds__(WAVES * 64) void bank_kernel(unsigned long long* cycles, int iters, const unsigned* rnd) {
    unsigned long long t0, t1;
    {
        const unsigned* q = rnd + (DATA ? (threadIdx.x & 63) * 64 : 0);  // DATA = 0: every lane and register the same value (zero)
        asm volatile("v_mov_b32 v66, %0" ::"v"(DATA ? q[0] : 0u) : "v66");
        asm volatile("v_mov_b32 v67, %0" ::"v"(DATA ? q[1] : 0u) : "v67");
        asm volatile("v_mov_b32 v68, %0" ::"v"(DATA ? q[2] : 0u) : "v68");
        asm volatile("v_mov_b32 v69, %0" ::"v"(DATA ? q[3] : 0u) : "v69");
        asm volatile("v_mov_b32 v70, %0" ::"v"(DATA ? q[4] : 0u) : "v70");
        asm volatile("v_mov_b32 v71, %0" ::"v"(DATA ? q[5] : 0u) : "v71");
        asm volatile("v_mov_b32 v72, %0" ::"v"(DATA ? q[6] : 0u) : "v72");
        asm volatile("v_mov_b32 v73, %0" ::"v"(DATA ? q[7] : 0u) : "v73");
        asm volatile("v_mov_b32 v74, %0" ::"v"(DATA ? q[8] : 0u) : "v74");
        asm volatile("v_mov_b32 v75, %0" ::"v"(DATA ? q[9] : 0u) : "v75");
        asm volatile("v_mov_b32 v76, %0" ::"v"(DATA ? q[10] : 0u) : "v76");
        asm volatile("v_mov_b32 v77, %0" ::"v"(DATA ? q[11] : 0u) : "v77");
        asm volatile("v_mov_b32 v78, %0" ::"v"(DATA ? q[12] : 0u) : "v78");
        asm volatile("v_mov_b32 v79, %0" ::"v"(DATA ? q[13] : 0u) : "v79");
        asm volatile("v_mov_b32 v80, %0" ::"v"(DATA ? q[14] : 0u) : "v80");
        asm volatile("v_mov_b32 v81, %0" ::"v"(DATA ? q[15] : 0u) : "v81");
        asm volatile("v_mov_b32 v82, %0" ::"v"(DATA ? q[16] : 0u) : "v82");
        asm volatile("v_mov_b32 v83, %0" ::"v"(DATA ? q[17] : 0u) : "v83");
        asm volatile("v_mov_b32 v84, %0" ::"v"(DATA ? q[18] : 0u) : "v84");
        asm volatile("v_mov_b32 v85, %0" ::"v"(DATA ? q[19] : 0u) : "v85");
        asm volatile("v_mov_b32 v86, %0" ::"v"(DATA ? q[20] : 0u) : "v86");
        asm volatile("v_mov_b32 v87, %0" ::"v"(DATA ? q[21] : 0u) : "v87");
        asm volatile("v_mov_b32 v88, %0" ::"v"(DATA ? q[22] : 0u) : "v88");
        asm volatile("v_mov_b32 v89, %0" ::"v"(DATA ? q[23] : 0u) : "v89");
        asm volatile("v_mov_b32 v90, %0" ::"v"(DATA ? q[24] : 0u) : "v90");
        asm volatile("v_mov_b32 v91, %0" ::"v"(DATA ? q[25] : 0u) : "v91");
        asm volatile("v_mov_b32 v92, %0" ::"v"(DATA ? q[26] : 0u) : "v92");
        asm volatile("v_mov_b32 v93, %0" ::"v"(DATA ? q[27] : 0u) : "v93");
        asm volatile("v_mov_b32 v94, %0" ::"v"(DATA ? q[28] : 0u) : "v94");
        asm volatile("v_mov_b32 v95, %0" ::"v"(DATA ? q[29] : 0u) : "v95");
        asm volatile("v_mov_b32 v96, %0" ::"v"(DATA ? q[30] : 0u) : "v96");
        asm volatile("v_mov_b32 v97, %0" ::"v"(DATA ? q[31] : 0u) : "v97");
        asm volatile("v_mov_b32 v98, %0" ::"v"(DATA ? q[32] : 0u) : "v98");
        asm volatile("v_mov_b32 v99, %0" ::"v"(DATA ? q[33] : 0u) : "v99");
        asm volatile("v_mov_b32 v100, %0" ::"v"(DATA ? q[34] : 0u) : "v100");
        asm volatile("v_mov_b32 v101, %0" ::"v"(DATA ? q[35] : 0u) : "v101");
        asm volatile("v_mov_b32 v102, %0" ::"v"(DATA ? q[36] : 0u) : "v102");
        asm volatile("v_mov_b32 v103, %0" ::"v"(DATA ? q[37] : 0u) : "v103");
        asm volatile("v_mov_b32 v2, 0" ::: "v2");
        asm volatile("v_mov_b32 v3, 0" ::: "v3");
        asm volatile("v_mov_b32 v4, 0" ::: "v4");
        asm volatile("v_mov_b32 v5, 0" ::: "v5");
        asm volatile("v_mov_b32 v6, 0" ::: "v6");
        asm volatile("v_mov_b32 v7, 0" ::: "v7");
        asm volatile("v_mov_b32 v8, 0" ::: "v8");
        asm volatile("v_mov_b32 v9, 0" ::: "v9");
        asm volatile("v_mov_b32 v10, 0" ::: "v10");
        asm volatile("v_mov_b32 v11, 0" ::: "v11");
        asm volatile("v_mov_b32 v12, 0" ::: "v12");
        asm volatile("v_mov_b32 v13, 0" ::: "v13");
        asm volatile("v_mov_b32 v14, 0" ::: "v14");
        asm volatile("v_mov_b32 v15, 0" ::: "v15");
        asm volatile("v_mov_b32 v16, 0" ::: "v16");
        asm volatile("v_mov_b32 v17, 0" ::: "v17");
        asm volatile("v_mov_b32 v18, 0" ::: "v18");
        asm volatile("v_mov_b32 v19, 0" ::: "v19");
        asm volatile("v_mov_b32 v20, 0" ::: "v20");
        asm volatile("v_mov_b32 v21, 0" ::: "v21");
        asm volatile("v_mov_b32 v22, 0" ::: "v22");
        asm volatile("v_mov_b32 v23, 0" ::: "v23");
        asm volatile("v_mov_b32 v24, 0" ::: "v24");
        asm volatile("v_mov_b32 v25, 0" ::: "v25");
        asm volatile("v_mov_b32 v26, 0" ::: "v26");
        asm volatile("v_mov_b32 v27, 0" ::: "v27");
        asm volatile("v_mov_b32 v28, 0" ::: "v28");
        asm volatile("v_mov_b32 v29, 0" ::: "v29");
        asm volatile("v_mov_b32 v30, 0" ::: "v30");
        asm volatile("v_mov_b32 v31, 0" ::: "v31");
        asm volatile("v_mov_b32 v32, 0" ::: "v32");
        asm volatile("v_mov_b32 v33, 0" ::: "v33");
        asm volatile("v_mov_b32 v34, 0" ::: "v34");
        asm volatile("v_mov_b32 v35, 0" ::: "v35");
        asm volatile("v_mov_b32 v36, 0" ::: "v36");
        asm volatile("v_mov_b32 v37, 0" ::: "v37");
        asm volatile("v_mov_b32 v38, 0" ::: "v38");
        asm volatile("v_mov_b32 v39, 0" ::: "v39");
        asm volatile("v_mov_b32 v40, 0" ::: "v40");
        asm volatile("v_mov_b32 v41, 0" ::: "v41");
        asm volatile("v_mov_b32 v42, 0" ::: "v42");
        asm volatile("v_mov_b32 v43, 0" ::: "v43");
        asm volatile("v_mov_b32 v44, 0" ::: "v44");
        asm volatile("v_mov_b32 v45, 0" ::: "v45");
        asm volatile("v_mov_b32 v46, 0" ::: "v46");
        asm volatile("v_mov_b32 v47, 0" ::: "v47");
        asm volatile("v_mov_b32 v48, 0" ::: "v48");
        asm volatile("v_mov_b32 v49, 0" ::: "v49");
        asm volatile("v_mov_b32 v50, 0" ::: "v50");
        asm volatile("v_mov_b32 v51, 0" ::: "v51");
        asm volatile("v_mov_b32 v52, 0" ::: "v52");
        asm volatile("v_mov_b32 v53, 0" ::: "v53");
        asm volatile("v_mov_b32 v54, 0" ::: "v54");
        asm volatile("v_mov_b32 v55, 0" ::: "v55");
        asm volatile("v_mov_b32 v56, 0" ::: "v56");
        asm volatile("v_mov_b32 v57, 0" ::: "v57");
        asm volatile("v_mov_b32 v58, 0" ::: "v58");
        asm volatile("v_mov_b32 v59, 0" ::: "v59");
        asm volatile("v_mov_b32 v60, 0" ::: "v60");
        asm volatile("v_mov_b32 v61, 0" ::: "v61");
        asm volatile("v_mov_b32 v62, 0" ::: "v62");
        asm volatile("v_mov_b32 v63, 0" ::: "v63");
        asm volatile("v_mov_b32 v64, 0" ::: "v64");
        asm volatile("v_mov_b32 v65, 0" ::: "v65");
    }
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

template <int MODE, int WAVES, int DATA = 0>
static double run(int iters) {
    static unsigned* rnd = nullptr;
    if (!rnd) {
        std::vector<unsigned> hr(64 * 64);
        unsigned x = 12345u;
        for (auto& v : hr) {  // two fp16 values per word: random sign and mantissa, exponent 13..16 (|x| in [0.25, 4))
            x = x * 1664525u + 1013904223u; const unsigned a = (x >> 9) & 0x83ffu | ((13u + ((x >> 3) & 3u)) << 10);
            x = x * 1664525u + 1013904223u; const unsigned b = (x >> 9) & 0x83ffu | ((13u + ((x >> 3) & 3u)) << 10);
            v = a | (b << 16);
        }
        hipMalloc(&rnd, hr.size() * 4);
        hipMemcpy(rnd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
    }
    unsigned long long* d;
    hipMalloc(&d, 256 * WAVES * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bank_kernel<MODE, WAVES, DATA>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters, rnd);
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
    printf("operand DATA (alignment 2/2/2): all-zero registers %5.2f, random fp16 values in [0.25, 4) %5.2f cycles per MFMA (one wave per SIMD); two waves per SIMD %5.2f / %5.2f; 10 x longer %5.2f / %5.2f\n",
           run<0, 4, 0>(it), run<0, 4, 1>(it), run<0, 8, 0>(it), run<0, 8, 1>(it), run<0, 8, 0>(10 * it), run<0, 8, 1>(10 * it));
    hipError_t e = hipDeviceSynchronize();
    printf("status %s\n", hipGetErrorString(e));
    return e != hipSuccess;
}
