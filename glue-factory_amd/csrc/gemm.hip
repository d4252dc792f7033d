// C[M,N] = A[M,K] * W[N,K]^T with a fused epilogue, on the fp32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Serves every Linear of LightGlue (lightglue.py:133-142,144-157,186-215,273-281) and the 1x1
// convolutions of SuperPoint's heads (superpoint.py:190,196) on NHWC activations.
//
// Tiling: 128x128 block, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).
// K advances in stages of 32; both operands are staged in LDS as [row][32 + 4 pad] floats (144-byte
// rows = 9 x 16 B, so ds_read_b128 by 16 consecutive rows is conflict-free) and double-buffered with a
// register prefetch, one barrier per stage.  Each ds_read_b128 feeds four MFMAs (common.h: mma_k8).
//
// PRECISE (GemmArgs::precise; LightGlue's final projection and similarity, lightglue.py:273-281, both arithmetic modes): the fp32
// MFMA chain restarts from zero at every 32-deep K stage and the stage sums are added up in float64; bias and scale are applied in
// float64 and the result is rounded to fp32 once.  An entry of log_assignment is 2 sim - lse_row - lse_col: with |sim| ~ 1e2 an
// error of three fp32 ulps of sim is already north_star's whole 1e-4, and one 256-term fp32 chain (let alone 22-bit split-fp16
// operands) does not keep that -- measured against the float64 oracle the f16x3 similarity put log_assignment 3.5x further from
// the exact value than the fp32 CPU path, the single-chain fp32 MFMA 1.2-2x (gpurun_out/r03_a_tests.log).  Eight 32-term chains +
// float64 sums leave ~0.3 ulp of sim.  Cost: the two GEMMs are 26 of the step's 5900 GFLOP.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

template <bool W_ROWMAJOR, bool PRECISE>
__global__ __launch_bounds__(256, PRECISE ? 1 : 2) void gemm_kernel(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDT];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const long long z = blockIdx.z;
    const float* A0 = p.A0 + z * p.strideA;
    const float* A1 = p.A1 ? p.A1 + z * p.strideA : nullptr;
    const float* Wg = p.W + z * p.strideW;
    float* C = p.C + z * p.strideC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    double dacc[PRECISE ? 2 : 1][PRECISE ? 2 : 1][PRECISE ? 16 : 1];
    if (PRECISE) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[PRECISE ? i : 0][PRECISE ? j : 0][PRECISE ? r : 0] = 0.0;
    }
    f32x4 ra[4], rw[4];
    const int nk = p.K / BK;

    auto load_stage = [&](int kt) {
        const int k0 = kt * BK;
        const float* Asrc;
        int lda, kk;
        if (k0 < p.ksplit) { Asrc = A0; lda = p.lda0; kk = k0; } else { Asrc = A1; lda = p.lda1; kk = k0 - p.ksplit; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i, row = idx >> 3, c4 = idx & 7;
            const int gm = m0 + row;
            ra[i] = (gm < p.M) ? *reinterpret_cast<const f32x4*>(Asrc + (long long)gm * lda + kk + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (W_ROWMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, row = idx >> 3, c4 = idx & 7;
                const int gn = n0 + row;
                rw[i] = (gn < p.N) ? *reinterpret_cast<const f32x4*>(Wg + (long long)gn * p.ldw + k0 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, kg = idx >> 8, rem = idx & 255, n = rem >> 1, hf = rem & 1;
                rw[i] = *reinterpret_cast<const f32x4*>(Wg + ((long long)((k0 >> 3) + kg) * p.Npad + n0 + n) * 8 + hf * 4);
            }
        }
    };
    auto store_stage = [&](int buf) {
        float* As = smem[buf];
        float* Ws = As + BM * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = t + 256 * i, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<f32x4*>(As + row * LDT + c4 * 4) = ra[i];
        }
        if (W_ROWMAJOR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, row = idx >> 3, c4 = idx & 7;
                *reinterpret_cast<f32x4*>(Ws + row * LDT + c4 * 4) = rw[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, kg = idx >> 8, rem = idx & 255, n = rem >> 1, hf = rem & 1;
                *reinterpret_cast<f32x4*>(Ws + n * LDT + kg * 8 + hf * 4) = rw[i];
            }
        }
    };

    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int ar = (wm * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int br = (wn * 64 + (lane & 31)) * LDT + (lane >> 5) * 4;

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_stage(kt + 1);
        const float* As = smem[buf];
        const float* Ws = As + BM * LDT;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + ar + kg * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + ar + 32 * LDT + kg * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Ws + br + kg * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Ws + br + 32 * LDT + kg * 8);
            mma_k8(a0, b0, acc[0][0]);
            mma_k8(a0, b1, acc[0][1]);
            mma_k8(a1, b0, acc[1][0]);
            mma_k8(a1, b1, acc[1][1]);
        }
        if (PRECISE) {  // close this stage's fp32 chains: float64 running sums, fresh accumulators
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        dacc[PRECISE ? i : 0][PRECISE ? j : 0][PRECISE ? r : 0] += (double)acc[i][j][r];
                        acc[i][j][r] = 0.f;
                    }
        }
        if (kt + 1 < nk) store_stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
            const bool col_ok = col < p.N;
            const float bias = (p.bias && col_ok) ? p.bias[col] : 0.f;
            const float ps = (p.post_scale && col_ok) ? p.post_scale[col] : 1.f;
            const float pt = (p.post_shift && col_ok) ? p.post_shift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + mi * 32 + acc_row(r, lane);
                const bool row_ok = row < p.M;
                float v;
                if (PRECISE) {  // bias and scale in float64, one rounding
                    double d = dacc[PRECISE ? mi : 0][PRECISE ? ni : 0][PRECISE ? r : 0] + (double)bias;
                    if (col < p.scale_cols) d *= (double)p.scale;
                    v = (float)d;
                    if (row_ok && col_ok) C[(long long)row * p.ldc + col] = v;
                    continue;
                }
                v = acc[mi][ni][r] + bias;
                if (p.rotary_cols > 0) {  // lightglue.py:36-43 on adjacent feature pairs (2i,2i+1)
                    const float partner = __shfl_xor(v, 1);
                    if (col < p.rotary_cols) {
                        const long long er = (long long)(row_ok ? row : 0) * 64 + ((col & 63) >> 1);
                        const float c = p.enc[er], s = p.enc[er + 32];
                        const float rot = (col & 1) ? partner : -partner;
                        v = __fadd_rn(__fmul_rn(v, c), __fmul_rn(rot, s));
                    }
                }
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.post_scale) v = __fadd_rn(__fmul_rn(v, ps), pt);
                if (col < p.scale_cols) v *= p.scale;
                if (row_ok && col_ok) {
                    if (p.residual) v += p.residual[(long long)row * p.ldr + col];
                    C[(long long)row * p.ldc + col] = v;
                }
            }
        }
    }
}

}  // namespace

void launch_gemm(const GemmArgs& a, hipStream_t s) {
    dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.batch);
    if (a.precise) {  // plain C = scale (A W^T + bias): no rotary / ReLU / affine / residual in this form
        if (a.w_rowmajor) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<false, true>), grid, dim3(256), 0, s, a);
    } else if (a.w_rowmajor)
        hipLaunchKernelGGL((gemm_kernel<true, false>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_kernel<false, false>), grid, dim3(256), 0, s, a);
}
