// SuperGlue-specific byte-bound kernels (gluefactory_nonfree/superglue.py):
// keypoint-encoder input assembly and the log-domain Sinkhorn optimal transport (:173-201).
// The attentional GNN itself runs on the shared GEMM / attention kernels.
#include "common.h"
#include "kernels.h"

namespace {

// superglue.py:74-86 + :98-103 -- rows of [x_n, y_n, score, 0 x 29]: (kp - size/2) / (0.7 * max(size)).
__global__ __launch_bounds__(256) void sg_kenc_input_kernel(const float* __restrict__ k0, const float* __restrict__ k1,
                                                           const float* __restrict__ s0, const float* __restrict__ s1,
                                                           const float* __restrict__ size0, const float* __restrict__ size1, int B, int M, int N,
                                                           float* __restrict__ out) {
    const long long T0 = (long long)B * M, T = T0 + (long long)B * N;
    const long long tok = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tok >= T) return;
    const int side = tok >= T0;
    const long long r = side ? tok - T0 : tok;
    const int L = side ? N : M;
    const int b = (int)(r / L);
    const float* kp = (side ? k1 : k0) + r * 2;
    const float* sz = (side ? size1 : size0) + 2 * b;
    const float sc = (side ? s1 : s0)[r];
    const float w = sz[0], h = sz[1];
    const float scale = __fmul_rn(fmaxf(w, h), 0.7f);
    float* o = out + tok * 32;
    o[0] = __fdiv_rn(__fsub_rn(kp[0], w / 2.f), scale);
    o[1] = __fdiv_rn(__fsub_rn(kp[1], h / 2.f), scale);
    o[2] = sc;
#pragma unroll
    for (int j = 3; j < 32; ++j) o[j] = 0.f;
}

// Couplings Z (M+1)x(N+1): Z[i<M][j<N] = cost, everything in the last row / column = alpha.
struct SinkArgs {
    const float* cost;
    float alpha;
    int B, M, N;
    float* u;  // (B, M+1)
    float* v;  // (B, N+1)
    float norm, log_mu_bin, log_nu_bin;  // -log(M+N), log(N)+norm, log(M)+norm
};

// u[i] = log_mu[i] - logsumexp_j(Z[i,j] + v[j]).  One wave per row (rows 0..M of every pair).
__global__ __launch_bounds__(256) void sinkhorn_rows_kernel(const SinkArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * (p.M + 1)) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    const float* v = p.v + (long long)b * (p.N + 1);
    const float* c = p.cost + ((long long)b * p.M + i) * p.N;
    float m = -__builtin_inff();
    for (int j = lane; j <= p.N; j += 64) {
        const float z = (i < p.M && j < p.N) ? c[j] : p.alpha;
        m = fmaxf(m, z + v[j]);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j <= p.N; j += 64) {
        const float z = (i < p.M && j < p.N) ? c[j] : p.alpha;
        s += expf(z + v[j] - m);
    }
    s = wave_sum(s);
    if (lane == 0) p.u[row] = (i < p.M ? p.norm : p.log_mu_bin) - (logf(s) + m);
}

// v[j] = log_nu[j] - logsumexp_i(Z[i,j] + u[i]).  64 columns per block, 4 row groups, online max/sum.
__global__ __launch_bounds__(256) void sinkhorn_cols_kernel(const SinkArgs p) {
    __shared__ float s_m[4][64], s_l[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* u = p.u + (long long)b * (p.M + 1);
    const float* c = p.cost + (long long)b * p.M * p.N;
    float m = -__builtin_inff(), l = 0.f;
    if (j <= p.N)
        for (int i = g; i <= p.M; i += 4) {
            const float z = (i < p.M && j < p.N) ? c[(long long)i * p.N + j] : p.alpha;
            const float x = z + u[i];
            if (x > m) { l = l * expf(m - x) + 1.f; m = x; } else { l += expf(x - m); }
        }
    s_m[g][cl] = m;
    s_l[g][cl] = l;
    __syncthreads();
    if (g == 0 && j <= p.N) {
        const float mm = fmaxf(fmaxf(s_m[0][cl], s_m[1][cl]), fmaxf(s_m[2][cl], s_m[3][cl]));
        float ll = 0.f;
        for (int k = 0; k < 4; ++k)
            if (s_l[k][cl] > 0.f) ll += s_l[k][cl] * expf(s_m[k][cl] - mm);
        p.v[(long long)b * (p.N + 1) + j] = (j < p.N ? p.norm : p.log_nu_bin) - (logf(ll) + mm);
    }
}

// log_assignment = Z + u + v - norm (optional dump) and the row arg-max over the inner block.
__global__ __launch_bounds__(256) void sinkhorn_final_rows(const SinkArgs p, float* __restrict__ la, float* __restrict__ row_best, int* __restrict__ row_arg) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * (p.M + 1)) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    const float* v = p.v + (long long)b * (p.N + 1);
    const float* c = p.cost + ((long long)b * p.M + i) * p.N;
    const float ui = p.u[row];
    float* o = la ? la + row * (p.N + 1) : nullptr;
    float best = -__builtin_inff();
    int arg = 0x7fffffff;
    for (int j = lane; j <= p.N; j += 64) {
        const float z = (i < p.M && j < p.N) ? c[j] : p.alpha;
        const float x = __fsub_rn(__fadd_rn(__fadd_rn(z, ui), v[j]), p.norm);
        if (o) o[j] = x;
        if (i < p.M && j < p.N && x > best) { best = x; arg = j; }
    }
    if (i >= p.M) return;
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) {
        const float ob = __shfl_xor(best, of);
        const int oa = __shfl_xor(arg, of);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        row_best[(long long)b * p.M + i] = best;
        row_arg[(long long)b * p.M + i] = arg;
    }
}

__global__ __launch_bounds__(256) void sinkhorn_final_cols(const SinkArgs p, float* __restrict__ col_best, int* __restrict__ col_arg) {
    __shared__ float s_b[4][64];
    __shared__ int s_a[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* u = p.u + (long long)b * (p.M + 1);
    const float* c = p.cost + (long long)b * p.M * p.N;
    float best = -__builtin_inff();
    int arg = 0x7fffffff;
    if (j < p.N) {
        const float vj = p.v[(long long)b * (p.N + 1) + j];
        for (int i = g; i < p.M; i += 4) {
            const float x = __fsub_rn(__fadd_rn(__fadd_rn(c[(long long)i * p.N + j], u[i]), vj), p.norm);
            if (x > best) { best = x; arg = i; }
        }
    }
    s_b[g][cl] = best;
    s_a[g][cl] = arg;
    __syncthreads();
    if (g == 0 && j < p.N) {
        for (int k = 1; k < 4; ++k)
            if (s_b[k][cl] > best || (s_b[k][cl] == best && s_a[k][cl] < arg)) { best = s_b[k][cl]; arg = s_a[k][cl]; }
        col_best[(long long)b * p.N + j] = best;
        col_arg[(long long)b * p.N + j] = arg;
    }
}

}  // namespace

void launch_sg_kenc_input(const float* k0, const float* k1, const float* s0, const float* s1, const float* size0, const float* size1, int B,
                          int M, int N, float* out, hipStream_t s) {
    const long long T = (long long)B * (M + N);
    hipLaunchKernelGGL(sg_kenc_input_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, k0, k1, s0, s1, size0, size1, B, M, N, out);
}

void launch_sinkhorn(const SinkhornArgs& a, hipStream_t s) {
    SinkArgs p;
    p.cost = a.cost; p.alpha = a.alpha; p.B = a.B; p.M = a.M; p.N = a.N; p.u = a.u; p.v = a.v;
    p.norm = -logf((float)a.M + (float)a.N);
    p.log_mu_bin = logf((float)a.N) + p.norm;
    p.log_nu_bin = logf((float)a.M) + p.norm;
    (void)hipMemsetAsync(a.u, 0, sizeof(float) * a.B * (a.M + 1), s);
    (void)hipMemsetAsync(a.v, 0, sizeof(float) * a.B * (a.N + 1), s);
    const long long rows = (long long)a.B * (a.M + 1);
    const dim3 rgrid((unsigned)((rows + 3) / 4)), cgrid((a.N + 1 + 63) / 64, a.B);
    for (int it = 0; it < a.iters; ++it) {
        hipLaunchKernelGGL(sinkhorn_rows_kernel, rgrid, dim3(256), 0, s, p);
        hipLaunchKernelGGL(sinkhorn_cols_kernel, cgrid, dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(sinkhorn_final_rows, rgrid, dim3(256), 0, s, p, a.log_assignment, a.row_best, a.row_arg);
    hipLaunchKernelGGL(sinkhorn_final_cols, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, p, a.col_best, a.col_arg);
}
