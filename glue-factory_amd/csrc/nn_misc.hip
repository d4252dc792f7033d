// Nearest-neighbour matcher (gluefactory/models/matchers/nearest_neighbor_matcher.py:15-72).
// The similarity matrix comes from the batched GEMM; these byte-bound kernels sweep it once by rows and once by
// columns (top-2 for the ratio test, log-sum-exp for the double log-softmax), apply the thresholds and the mutual
// check, and optionally write the (M+1)x(N+1) log-assignment.  Ties: lowest index (torch.topk's order among equal
// values is unspecified).
#include "common.h"
#include "kernels.h"

namespace {

struct Top2 {
    float v1, v2;
    int i1;
};
__device__ __forceinline__ void top2_push(Top2& t, float v, int i) {
    if (v > t.v1 || (v == t.v1 && i < t.i1)) { t.v2 = t.v1; t.v1 = v; t.i1 = i; }
    else t.v2 = fmaxf(t.v2, v);
}
__device__ __forceinline__ void top2_merge(Top2& a, float v1, float v2, int i1) {
    if (v1 > a.v1 || (v1 == a.v1 && i1 < a.i1)) { a.v2 = fmaxf(a.v1, v2); a.v1 = v1; a.i1 = i1; }
    else a.v2 = fmaxf(a.v2, v1);
}
// nearest_neighbor_matcher.py:17-25 on the two best similarities
__device__ __forceinline__ int nn_decide(const NnArgs& p, const Top2& t) {
    const float d0 = __fmul_rn(2.f, __fsub_rn(1.f, t.v1)), d1 = __fmul_rn(2.f, __fsub_rn(1.f, t.v2));
    bool ok = true;
    if (p.use_ratio) ok = ok && (d0 <= __fmul_rn(p.ratio2, d1));
    if (p.use_dist) ok = ok && (d0 <= p.dist2);
    return (ok && t.i1 != 0x7fffffff) ? t.i1 : -1;  // the sentinel survives a row of NaNs: no candidate, never an index
}

// one wave per row (b, i)
__global__ __launch_bounds__(256) void nn_rows_kernel(const NnArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * p.M) return;
    const float* s = p.sim + row * p.N;
    const float NEG = -__builtin_inff();
    Top2 t{NEG, NEG, 0x7fffffff};
    float m = NEG, l = 0.f;
    for (int j = lane; j < p.N; j += 64) {
        const float x = s[j];
        top2_push(t, x, j);
        if (x > m) { l = l * expf(m - x) + 1.f; m = x; } else l += expf(x - m);
    }
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) {
        const float ov1 = __shfl_xor(t.v1, of), ov2 = __shfl_xor(t.v2, of), om = __shfl_xor(m, of), ol = __shfl_xor(l, of);
        const int oi = __shfl_xor(t.i1, of);
        top2_merge(t, ov1, ov2, oi);
        const float nm = fmaxf(m, om);
        l = (l > 0.f ? l * expf(m - nm) : 0.f) + (ol > 0.f ? ol * expf(om - nm) : 0.f);
        m = nm;
    }
    if (lane == 0) {
        p.row_lse[row] = logf(l) + m;
        p.cand0[row] = nn_decide(p, t);
    }
}

// 64 columns per block, 4 waves take interleaved rows
__global__ __launch_bounds__(256) void nn_cols_kernel(const NnArgs p) {
    __shared__ float s_v1[4][64], s_v2[4][64], s_m[4][64], s_l[4][64];
    __shared__ int s_i1[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* s = p.sim + (long long)b * p.M * p.N;
    const float NEG = -__builtin_inff();
    Top2 t{NEG, NEG, 0x7fffffff};
    float m = NEG, l = 0.f;
    if (j < p.N)
        for (int i = g; i < p.M; i += 4) {
            const float x = s[(long long)i * p.N + j];
            top2_push(t, x, i);
            if (x > m) { l = l * expf(m - x) + 1.f; m = x; } else l += expf(x - m);
        }
    s_v1[g][cl] = t.v1; s_v2[g][cl] = t.v2; s_i1[g][cl] = t.i1; s_m[g][cl] = m; s_l[g][cl] = l;
    __syncthreads();
    if (g == 0 && j < p.N) {
        for (int k = 1; k < 4; ++k) {
            top2_merge(t, s_v1[k][cl], s_v2[k][cl], s_i1[k][cl]);
            const float om = s_m[k][cl], ol = s_l[k][cl], nm = fmaxf(m, om);
            l = (l > 0.f ? l * expf(m - nm) : 0.f) + (ol > 0.f ? ol * expf(om - nm) : 0.f);
            m = nm;
        }
        p.col_lse[(long long)b * p.N + j] = logf(l) + m;
        p.cand1[(long long)b * p.N + j] = nn_decide(p, t);
    }
}

// mutual check (:29-36) and outputs
__global__ __launch_bounds__(256) void nn_final_kernel(const NnArgs p) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int* c0 = p.cand0 + (long long)b * p.M;
    const int* c1 = p.cand1 + (long long)b * p.N;
    if (i < p.M) {
        int mt = c0[i];
        if (p.mutual && mt > -1 && c1[mt] != i) mt = -1;
        p.m0[(long long)b * p.M + i] = mt;
        p.ms0[(long long)b * p.M + i] = mt > -1 ? 1.f : 0.f;
    }
    if (i < p.N) {
        int mt = c1[i];
        if (p.mutual && mt > -1 && c0[mt] != i) mt = -1;
        p.m1[(long long)b * p.N + i] = mt;
        p.ms1[(long long)b * p.N + i] = mt > -1 ? 1.f : 0.f;
    }
}

// log_assignment[:, :M, :N] = log_softmax(sim, -1) + log_softmax(sim, -2); zero last row / column (:60-61)
__global__ __launch_bounds__(256) void nn_la_kernel(const NnArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * (p.M + 1)) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    float* o = p.la + row * (p.N + 1);
    if (i == p.M) {
        for (int j = lane; j <= p.N; j += 64) o[j] = 0.f;
        return;
    }
    const float* s = p.sim + ((long long)b * p.M + i) * p.N;
    const float rl = p.row_lse[(long long)b * p.M + i];
    const float* cl = p.col_lse + (long long)b * p.N;
    for (int j = lane; j < p.N; j += 64) o[j] = __fadd_rn(__fsub_rn(s[j], rl), __fsub_rn(s[j], cl[j]));
    if (lane == 0) o[p.N] = 0.f;
}

}  // namespace

void launch_nn_match(const NnArgs& a, hipStream_t s) {
    const long long rows = (long long)a.B * a.M;
    hipLaunchKernelGGL(nn_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(nn_cols_kernel, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, a);
    const int L = a.M > a.N ? a.M : a.N;
    hipLaunchKernelGGL(nn_final_kernel, dim3((L + 255) / 256, a.B), dim3(256), 0, s, a);
    if (a.la) {
        const long long rows1 = (long long)a.B * (a.M + 1);
        hipLaunchKernelGGL(nn_la_kernel, dim3((unsigned)((rows1 + 3) / 4)), dim3(256), 0, s, a);
    }
}
