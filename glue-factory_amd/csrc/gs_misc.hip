// GlueStick-specific byte-bound kernels (gluefactory/models/matchers/gluestick.py).
// The attentional GNN runs on the shared GEMM / attention kernels exactly like SuperGlue's; these kernels add the
// line side: endpoint-encoder input rows, the per-endpoint message gather and the per-junction mean of the line
// layer (:581-612, :656-689), the double log-softmax with a dustbin (:758-770) and the line score gather (:331-352).
#include "common.h"
#include "kernels.h"

namespace {

// gluestick.py:498-517 -- rows of [x, y, dx, dy, score, 0 x 27] per line endpoint, endpoints normalised as
// normalize_keypoints does (:474-485); (dx,dy) = other endpoint - this endpoint.  Rows: image-0 endpoints, then image 1.
__global__ __launch_bounds__(256) void gs_lenc_input_kernel(const float* __restrict__ l0, const float* __restrict__ l1, const float* __restrict__ s0,
                                                           const float* __restrict__ s1, const float* __restrict__ size0,
                                                           const float* __restrict__ size1, int B, int n0, int n1, float* __restrict__ out) {
    const long long E0 = (long long)B * 2 * n0, E = E0 + (long long)B * 2 * n1;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int side = e >= E0;
    const long long r = side ? e - E0 : e;
    const int n = side ? n1 : n0;
    const int b = (int)(r / (2 * n));
    const long long line = r >> 1;
    const int end = (int)(r & 1);
    const float* L = (side ? l1 : l0) + line * 4;  // (x0, y0, x1, y1)
    const float* sz = (side ? size1 : size0) + 2 * b;
    const float w = sz[0], h = sz[1];
    const float scale = __fmul_rn(fmaxf(w, h), 0.7f);
    const float x0 = __fdiv_rn(__fsub_rn(L[0], w / 2.f), scale), y0 = __fdiv_rn(__fsub_rn(L[1], h / 2.f), scale);
    const float x1 = __fdiv_rn(__fsub_rn(L[2], w / 2.f), scale), y1 = __fdiv_rn(__fsub_rn(L[3], h / 2.f), scale);
    const float dx = __fsub_rn(x1, x0), dy = __fsub_rn(y1, y0);
    float* o = out + e * 32;
    o[0] = end ? x1 : x0;
    o[1] = end ? y1 : y0;
    o[2] = end ? -dx : dx;
    o[3] = end ? -dy : dy;
    // scores.repeat(1, 2) TILES the per-line scores over the 2n endpoints (gluestick.py:515): endpoint e gets score[e mod n]
    const int el = (int)(r - (long long)b * 2 * n);
    o[4] = (side ? s1 : s0)[(long long)b * n + (el % n)];
#pragma unroll
    for (int j = 5; j < 32; ++j) o[j] = 0.f;
}

// message row of an endpoint = [desc of its junction | desc of the line's other junction | endpoint encoding]
// (3 x 256 values; 4-byte elements in either storage, so the copy is layout-agnostic: fp32 rows or SH8 rows)
__global__ __launch_bounds__(256) void gs_gather_msg_kernel(const GsLineArgs p) {
    const int lane = threadIdx.x & 63;
    const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long E0 = (long long)p.B * 2 * p.n0, E = E0 + (long long)p.B * 2 * p.n1;
    if (e >= E) return;
    const int side = e >= E0;
    const long long r = side ? e - E0 : e;
    const int n = side ? p.n1 : p.n0, K = side ? p.N : p.M;
    const int b = (int)(r / (2 * n));
    const long long* idx = (side ? p.idx1 : p.idx0);
    const long long tok0 = side ? (long long)p.B * p.M + (long long)b * p.N : (long long)b * p.M;
    const int own = (int)idx[r], other = (int)idx[r ^ 1];
    (void)K;
    const u32x4* X = reinterpret_cast<const u32x4*>(p.x);
    const u32x4* En = reinterpret_cast<const u32x4*>(p.enc);
    u32x4* o = reinterpret_cast<u32x4*>(p.msg) + e * 192;
    o[lane] = X[(tok0 + own) * 64 + lane];
    o[64 + lane] = X[(tok0 + other) * 64 + lane];
    o[128 + lane] = En[e * 64 + lane];
}

// desc[j] += mean over the endpoints whose junction is j (scatter_reduce "mean", include_self=False; untouched
// junctions keep their value).  One wave per keypoint: the endpoint list of its image is scanned 64 at a time and
// matching endpoints are added in ascending order (deterministic, the order of the sequential CPU reduction).
__global__ __launch_bounds__(256) void gs_scatter_mean_kernel(const GsLineArgs p) {
    const int lane = threadIdx.x & 63;
    const long long tok = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long T0 = (long long)p.B * p.M, T = T0 + (long long)p.B * p.N;
    if (tok >= T) return;
    const int side = tok >= T0;
    const long long r = side ? tok - T0 : tok;
    const int K = side ? p.N : p.M, n = side ? p.n1 : p.n0;
    const int b = (int)(r / K), j = (int)(r - (long long)b * K);
    const long long* idx = (side ? p.idx1 : p.idx0) + (long long)b * 2 * n;
    const long long e0 = side ? (long long)p.B * 2 * p.n0 + (long long)b * 2 * n : (long long)b * 2 * n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int base = 0; base < 2 * n; base += 64) {
        const int e = base + lane;
        const bool hit = e < 2 * n && (int)idx[e] == j;
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1;
            const f32x4 u = *reinterpret_cast<const f32x4*>(p.upd + (e0 + base + k) * 256 + lane * 4);
            acc += u;
            ++cnt;
        }
    }
    if (cnt == 0) return;
    const float c = (float)cnt;
    float* x = p.x32 + tok * 256 + lane * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(x);
    v[0] += acc[0] / c; v[1] += acc[1] / c; v[2] += acc[2] / c; v[3] += acc[3] / c;
    *reinterpret_cast<f32x4*>(x) = v;
    if (p.xs) {  // refresh the SH8 copy of this row: lanes 2g, 2g+1 hold the 8 channels of group g
        const f32x4 w = {__shfl_xor(v[0], 1), __shfl_xor(v[1], 1), __shfl_xor(v[2], 1), __shfl_xor(v[3], 1)};
        const f32x8 g8 = (lane & 1) ? f32x8{w[0], w[1], w[2], w[3], v[0], v[1], v[2], v[3]} : f32x8{v[0], v[1], v[2], v[3], w[0], w[1], w[2], w[3]};
        f16x8 hi, lo;
        split8(g8, hi, lo);
        f16x8* o = reinterpret_cast<f16x8*>(p.xs) + tok * 64 + (lane >> 1) * 2;
        o[lane & 1] = (lane & 1) ? lo : hi;
    }
}

// ---- double log-softmax with a dustbin (gluestick.py:758-770) -------------------------------------------------------
// lse over [row | bin] and [column | bin]; S[i,j] = ((s - lr_i) + (s - lc_j)) / 2, S[i,N] = bin - lr_i, S[M,j] = bin - lc_j.
__global__ __launch_bounds__(256) void gs_ds_row_lse(const GsSoftmaxArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * p.M) return;
    const float* s = p.scores + row * p.N;
    float m = (lane == 0) ? p.bin : -__builtin_inff();
    for (int j = lane; j < p.N; j += 64) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float l = (lane == 0) ? expf(p.bin - m) : 0.f;
    for (int j = lane; j < p.N; j += 64) l += expf(s[j] - m);
    l = wave_sum(l);
    if (lane == 0) p.row_lse[row] = logf(l) + m;
}
__global__ __launch_bounds__(256) void gs_ds_col_lse(const GsSoftmaxArgs p) {
    __shared__ float s_m[4][64], s_l[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* s = p.scores + (long long)b * p.M * p.N;
    float m = -__builtin_inff(), l = 0.f;
    if (j < p.N)
        for (int i = g; i < p.M; i += 4) {
            const float x = s[(long long)i * p.N + j];
            if (x > m) { l = l * expf(m - x) + 1.f; m = x; } else l += expf(x - m);
        }
    s_m[g][cl] = m; s_l[g][cl] = l;
    __syncthreads();
    if (g == 0 && j < p.N) {
        float mm = p.bin;
        for (int k = 0; k < 4; ++k) mm = fmaxf(mm, s_m[k][cl]);
        float ll = expf(p.bin - mm);
        for (int k = 0; k < 4; ++k)
            if (s_l[k][cl] > 0.f) ll += s_l[k][cl] * expf(s_m[k][cl] - mm);
        p.col_lse[(long long)b * p.N + j] = logf(ll) + mm;
    }
}
__global__ __launch_bounds__(256) void gs_ds_rows(const GsSoftmaxArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * (p.M + 1)) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    float* o = p.out ? p.out + row * (p.N + 1) : nullptr;
    const float* lc = p.col_lse + (long long)b * p.N;
    if (i == p.M) {
        if (o) {
            for (int j = lane; j < p.N; j += 64) o[j] = __fsub_rn(p.bin, lc[j]);
            if (lane == 0) o[p.N] = 0.f;
        }
        return;
    }
    const float* s = p.scores + ((long long)b * p.M + i) * p.N;
    const float lr = p.row_lse[(long long)b * p.M + i];
    float best = -__builtin_inff();
    int arg = 0x7fffffff;
    for (int j = lane; j < p.N; j += 64) {
        const float x = __fdiv_rn(__fadd_rn(__fsub_rn(s[j], lr), __fsub_rn(s[j], lc[j])), 2.f);
        if (o) o[j] = x;
        if (x > best) { best = x; arg = j; }
    }
    if (o && lane == 0) o[p.N] = __fsub_rn(p.bin, lr);
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) {
        const float ob = __shfl_xor(best, of);
        const int oa = __shfl_xor(arg, of);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        p.row_best[(long long)b * p.M + i] = best;
        p.row_arg[(long long)b * p.M + i] = arg < p.N ? arg : 0;  // (never an index outside the row: a row of NaNs leaves the sentinel, and the mutual check below indexes with this value)
    }
}
__global__ __launch_bounds__(256) void gs_ds_cols(const GsSoftmaxArgs p) {
    __shared__ float s_b[4][64];
    __shared__ int s_a[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* s = p.scores + (long long)b * p.M * p.N;
    const float* lr = p.row_lse + (long long)b * p.M;
    float best = -__builtin_inff();
    int arg = 0x7fffffff;
    if (j < p.N) {
        const float lc = p.col_lse[(long long)b * p.N + j];
        for (int i = g; i < p.M; i += 4) {
            const float v = s[(long long)i * p.N + j];
            const float x = __fdiv_rn(__fadd_rn(__fsub_rn(v, lr[i]), __fsub_rn(v, lc)), 2.f);
            if (x > best) { best = x; arg = i; }
        }
    }
    s_b[g][cl] = best; s_a[g][cl] = arg;
    __syncthreads();
    if (g == 0 && j < p.N) {
        for (int k = 1; k < 4; ++k)
            if (s_b[k][cl] > best || (s_b[k][cl] == best && s_a[k][cl] < arg)) { best = s_b[k][cl]; arg = s_a[k][cl]; }
        p.col_best[(long long)b * p.N + j] = best;
        p.col_arg[(long long)b * p.N + j] = arg < p.M ? arg : 0;
    }
}

// gluestick.py:339-358: raw[b,i,j] = 0.5 * max(J[a0,c0] + J[a1,c1], J[a0,c1] + J[a1,c0]) on the endpoint similarity J
__global__ __launch_bounds__(256) void gs_line_scores_kernel(const float* __restrict__ J, const long long* __restrict__ idx0,
                                                            const long long* __restrict__ idx1, int B, int n0, int n1, float* __restrict__ raw) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * n0 * n1) return;
    const int j = (int)(t % n1), i = (int)((t / n1) % n0), b = (int)(t / ((long long)n0 * n1));
    const float* Jb = J + (long long)b * (2 * n0) * (2 * n1);
    const int a0 = (int)idx0[((long long)b * n0 + i) * 2], a1 = (int)idx0[((long long)b * n0 + i) * 2 + 1];
    const int c0 = (int)idx1[((long long)b * n1 + j) * 2], c1 = (int)idx1[((long long)b * n1 + j) * 2 + 1];
    const long long ld = 2 * n1;
    const float s00 = Jb[a0 * ld + c0], s11 = Jb[a1 * ld + c1], s01 = Jb[a0 * ld + c1], s10 = Jb[a1 * ld + c0];
    raw[t] = __fmul_rn(0.5f, fmaxf(__fadd_rn(s00, s11), __fadd_rn(s01, s10)));
}

}  // namespace

void launch_gs_lenc_input(const float* l0, const float* l1, const float* s0, const float* s1, const float* size0, const float* size1, int B, int n0,
                          int n1, float* out, hipStream_t s) {
    const long long E = (long long)B * 2 * (n0 + n1);
    hipLaunchKernelGGL(gs_lenc_input_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, l0, l1, s0, s1, size0, size1, B, n0, n1, out);
}
void launch_gs_gather_msg(const GsLineArgs& a, hipStream_t s) {
    const long long E = (long long)a.B * 2 * (a.n0 + a.n1);
    hipLaunchKernelGGL(gs_gather_msg_kernel, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, s, a);
}
void launch_gs_scatter_mean(const GsLineArgs& a, hipStream_t s) {
    const long long T = (long long)a.B * (a.M + a.N);
    hipLaunchKernelGGL(gs_scatter_mean_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, a);
}
void launch_gs_double_softmax(const GsSoftmaxArgs& a, hipStream_t s) {
    const long long rows = (long long)a.B * a.M, rows1 = (long long)a.B * (a.M + 1);
    hipLaunchKernelGGL(gs_ds_row_lse, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gs_ds_col_lse, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gs_ds_rows, dim3((unsigned)((rows1 + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gs_ds_cols, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, a);
}
void launch_gs_line_scores(const float* J, const long long* idx0, const long long* idx1, int B, int n0, int n1, float* raw, hipStream_t s) {
    const long long t = (long long)B * n0 * n1;
    hipLaunchKernelGGL(gs_line_scores_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, J, idx0, idx1, B, n0, n1, raw);
}
