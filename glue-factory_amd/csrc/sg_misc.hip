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
    float* v;  // (B, vstride)
    int vstride;  // roundup4(N+1): 16-byte aligned v vectors
    float norm, log_mu_bin, log_nu_bin;  // -log(M+N), log(N)+norm, log(M)+norm
};

// u[i] = log_mu[i] - logsumexp_j(Z[i,j] + v[j]).  One wave per row (rows 0..M of every pair).  A row of up to
// 2048 couplings stays in registers (8 x 16-byte loads per lane, all in flight together), so the matrix is read
// once per half-iteration; longer or unaligned rows take the two-pass loop (second pass served by the cache).
__global__ __launch_bounds__(256) void sinkhorn_rows_kernel(const SinkArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * (p.M + 1)) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    const float* v = p.v + (long long)b * p.vstride;
    const float* c = p.cost + ((long long)b * p.M + i) * p.N;
    float m, s;
    if (i < p.M && (p.N & 3) == 0 && p.N <= 2048) {
        f32x4 x[8];
        const int nq = p.N >> 2;  // float4 groups in the row
        m = (lane == 0) ? p.alpha + v[p.N] : -__builtin_inff();  // dustbin column
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = lane + 64 * k;
            if (q < nq) {
                x[k] = *reinterpret_cast<const f32x4*>(c + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) x[k][e] += v[4 * q + e];
                m = fmaxf(m, fmaxf(fmaxf(x[k][0], x[k][1]), fmaxf(x[k][2], x[k][3])));
            }
        }
        m = wave_max(m);
        s = (lane == 0) ? expf(p.alpha + v[p.N] - m) : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (lane + 64 * k < nq) s += (expf(x[k][0] - m) + expf(x[k][1] - m)) + (expf(x[k][2] - m) + expf(x[k][3] - m));
    } else {
        m = -__builtin_inff();
        for (int j = lane; j <= p.N; j += 64) {
            const float z = (i < p.M && j < p.N) ? c[j] : p.alpha;
            m = fmaxf(m, z + v[j]);
        }
        m = wave_max(m);
        s = 0.f;
        for (int j = lane; j <= p.N; j += 64) {
            const float z = (i < p.M && j < p.N) ? c[j] : p.alpha;
            s += expf(z + v[j] - m);
        }
    }
    s = wave_sum(s);
    if (lane == 0) p.u[row] = (i < p.M ? p.norm : p.log_mu_bin) - (logf(s) + m);
}

// v[j] = log_nu[j] - logsumexp_i(Z[i,j] + u[i]).  64 columns per block (a wave reads 256 contiguous bytes of a
// row), 8 waves take interleaved rows in batches of 8 independent loads with one running (max, sum) pair per
// lane: 9 exponentials per 8 couplings, 16 KB in flight per workgroup.
constexpr int COLS_WAVES = 8, COLS_BATCH = 8;
__global__ __launch_bounds__(64 * COLS_WAVES) void sinkhorn_cols_kernel(const SinkArgs p) {
    __shared__ float s_m[COLS_WAVES][64], s_l[COLS_WAVES][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* u = p.u + (long long)b * (p.M + 1);
    const float* c = p.cost + (long long)b * p.M * p.N;
    float m = -__builtin_inff(), l = 0.f;
    if (j <= p.N) {
        const bool inner = j < p.N;
        const float* cj = c + (inner ? j : 0);
        // rows i = g + COLS_WAVES * t; the last row (i == M) is the dustbin row
        for (int i0 = g; i0 <= p.M; i0 += COLS_WAVES * COLS_BATCH) {
            float x[COLS_BATCH];
#pragma unroll
            for (int k = 0; k < COLS_BATCH; ++k) {
                const int i = i0 + COLS_WAVES * k;
                x[k] = -__builtin_inff();
                if (i <= p.M) x[k] = ((inner && i < p.M) ? cj[(long long)i * p.N] : p.alpha) + u[i];
            }
            float bm = x[0];
#pragma unroll
            for (int k = 1; k < COLS_BATCH; ++k) bm = fmaxf(bm, x[k]);
            const float nm = fmaxf(m, bm);  // finite: every batch holds at least row i0
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < COLS_BATCH; ++k) acc += expf(x[k] - nm);  // exp(-inf) = 0 for the padding
            l = l * expf(m - nm) + acc;
            m = nm;
        }
    }
    s_m[g][cl] = m;
    s_l[g][cl] = l;
    __syncthreads();
    if (g == 0 && j <= p.N) {
        float mm = s_m[0][cl];
#pragma unroll
        for (int k = 1; k < COLS_WAVES; ++k) mm = fmaxf(mm, s_m[k][cl]);
        float ll = 0.f;
#pragma unroll
        for (int k = 0; k < COLS_WAVES; ++k)
            if (s_l[k][cl] > 0.f) ll += s_l[k][cl] * expf(s_m[k][cl] - mm);
        p.v[(long long)b * p.vstride + j] = (j < p.N ? p.norm : p.log_nu_bin) - (logf(ll) + mm);
    }
}

// ---- fused iteration (N % 4 == 0, N <= 2048; v stride padded to a multiple of 4) -----------------------------
// One launch per Sinkhorn iteration reads the couplings ONCE: a workgroup owns 16 consecutive rows (8 waves x 2
// rows, a row = 8 x 16 bytes per lane held in registers, all 16 loads of a lane in flight together).  Row
// half-step: u_i from the registers.  Column half-step: with its own fresh u_i every wave reduces its RPW rows to
// one (max, sum) pair per column, the 4 waves combine through LDS and the workgroup writes ONE partial per column;
// sinkhorn_combine_kernel folds the ceil((M+1)/16) partials of a column into v_j.  Two workgroups share a CU
// (66 KB LDS), so one's load phase runs under the other's exponentials.  Round 3: two rows per wave instead of four and eight waves
// instead of four (128 VGPRs, 4 waves per SIMD): 16 waves per CU keep twice as many independent load phases in flight, with the
// same 16 rows -> one partial per column per workgroup (measured 11.4 -> 9.1 ms per 50 iterations at 32 x 2049^2).
constexpr int FW = 8, RPW = 2, FR = RPW * FW;  // waves per workgroup, rows per wave (round 3: 2, was 4), rows per workgroup
struct SinkFusedArgs {
    SinkArgs a;
    int nblk;      // row blocks per pair
    float* pm;     // (B, nblk, vstride) partial column maxima
    float* pl;     // (B, nblk, vstride) partial column sums
};

// exp(x) for x <= 0 on the hardware exp2 (1 ulp); the product rounds x * log2(e) to 6e-8 relative, i.e. < 2e-6
// relative on every term that is not already negligible (x > -30)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// Exponentials are spent ONCE per coupling: the row half-step leaves e_ij = exp(z_ij + v_j - m_i) in the registers
// that held z_ij; with a_i = exp(u_i + m_i) (one value per row) the column sums are
//   sum_i exp(z_ij + u_i) = exp(-v_j) * sum_i e_ij * a_i,
// i.e. plain FMAs.  e and a are <= 1, so nothing overflows; a column whose total falls below 1e-20 (its potential
// moved by more than ~46 in one iteration: possible in the first iterations on extreme costs) is recomputed exactly,
// in the log domain, by sinkhorn_combine_kernel.
__global__ __launch_bounds__(64 * FW) __attribute__((amdgpu_waves_per_eu(4, 4))) void sinkhorn_fused_kernel(const SinkFusedArgs q) {
    const SinkArgs& p = q.a;
    __shared__ float s_l[FW][2052];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int b = blockIdx.y, blk = blockIdx.x;
    const int nq = p.N >> 2;
    const float* v = p.v + (long long)b * p.vstride;
    const float* c = p.cost + (long long)b * p.M * p.N;
    float* u = p.u + (long long)b * (p.M + 1);
    const float vbin = v[p.N];

    f32x4 z[RPW][8];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int i = blk * FR + w + FW * k;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int qd = lane + 64 * g;
            z[k][g] = f32x4{p.alpha, p.alpha, p.alpha, p.alpha};
            if (i < p.M && qd < nq) z[k][g] = *reinterpret_cast<const f32x4*>(c + (long long)i * p.N + 4 * qd);
        }
    }
    f32x4 vv[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int qd = lane + 64 * g;
        vv[g] = qd < nq ? *reinterpret_cast<const f32x4*>(v + 4 * qd) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float ebin[RPW], ai[RPW];  // dustbin-column exponential and a_i of this wave's rows
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int i = blk * FR + w + FW * k;
        const bool live = i <= p.M;
        // row half-step: z <- z + v, row max, z <- exp(z - m), row sum
        float m = p.alpha + vbin;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (lane + 64 * g < nq) {
                z[k][g] += vv[g];
                m = fmaxf(m, fmaxf(fmaxf(z[k][g][0], z[k][g][1]), fmaxf(z[k][g][2], z[k][g][3])));
            }
        m = wave_max(m);
        ebin[k] = fexp(p.alpha + vbin - m);
        float sum = (lane == 0) ? ebin[k] : 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (lane + 64 * g < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e) z[k][g][e] = fexp(z[k][g][e] - m);
                sum += (z[k][g][0] + z[k][g][1]) + (z[k][g][2] + z[k][g][3]);
            }
        sum = wave_sum(sum);
        const float ui = (i < p.M ? p.norm : p.log_mu_bin) - (logf(sum) + m);
        if (lane == 0 && live) u[i] = ui;
        ai[k] = live ? fexp(ui + m) : 0.f;  // = mu_i / sum <= 1; 0 switches a row past the end off
    }
    // column half-step: S_j = sum over this wave's rows of e_ij * a_i
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int qd = lane + 64 * g;
        if (qd >= nq) continue;
        f32x4 cl = z[0][g] * ai[0];
#pragma unroll
        for (int k = 1; k < RPW; ++k) cl += z[k][g] * ai[k];
        *reinterpret_cast<f32x4*>(&s_l[w][4 * qd]) = cl;
    }
    if (lane == 0) {
        float eb = ebin[0] * ai[0];
#pragma unroll
        for (int k = 1; k < RPW; ++k) eb += ebin[k] * ai[k];
        s_l[w][p.N] = eb;
    }
    __syncthreads();
    float* pl = q.pl + ((long long)b * q.nblk + blk) * p.vstride;
    for (int j = t; j <= p.N; j += 64 * FW) {
        float ll = s_l[0][j];
#pragma unroll
        for (int k = 1; k < FW; ++k) ll += s_l[k][j];
        pl[j] = ll;
    }
}

// v_j = log_nu_j - log(sum_i exp(z_ij + u_i)) with sum_i exp(z_ij + u_i) = exp(-v_j) * (sum of the row-block partials).
// 64 columns per workgroup, 4 waves take interleaved blocks in batches of 8 independent loads.  A column whose total
// underflowed is recomputed exactly from the couplings (log-sum-exp over its M+1 rows with the fresh u).
__global__ __launch_bounds__(256) void sinkhorn_combine_kernel(const SinkFusedArgs q) {
    const SinkArgs& p = q.a;
    __shared__ float s_l[4][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, j = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    float l = 0.f;
    if (j <= p.N) {
        const float* pl = q.pl + (long long)b * q.nblk * p.vstride + j;
        for (int k0 = g; k0 < q.nblk; k0 += 32) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + 4 * e;
                x[e] = k < q.nblk ? pl[(long long)k * p.vstride] : 0.f;
            }
            l += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        }
    }
    s_l[g][cl] = l;
    __syncthreads();
    if (g == 0 && j <= p.N) {
        const float ll = (s_l[0][cl] + s_l[1][cl]) + (s_l[2][cl] + s_l[3][cl]);
        float* vj = p.v + (long long)b * p.vstride + j;
        const float log_nu = j < p.N ? p.norm : p.log_nu_bin;
        if (ll > 1e-20f) {
            *vj = log_nu - (logf(ll) - *vj);
        } else {  // rare: exact log-domain column (the other lanes of the wave idle meanwhile)
            const float* u = p.u + (long long)b * (p.M + 1);
            const float* c = p.cost + (long long)b * p.M * p.N;
            float m = -__builtin_inff(), acc = 0.f;
            for (int i = 0; i <= p.M; ++i) {
                const float x = ((i < p.M && j < p.N) ? c[(long long)i * p.N + j] : p.alpha) + u[i];
                if (x > m) { acc = acc * expf(m - x) + 1.f; m = x; } else acc += expf(x - m);
            }
            *vj = log_nu - (logf(acc) + m);
        }
    }
}

// log_assignment = Z + u + v - norm (optional dump) and the row arg-max over the inner block.
__global__ __launch_bounds__(256) void sinkhorn_final_rows(const SinkArgs p, float* __restrict__ la, float* __restrict__ row_best, int* __restrict__ row_arg) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * (p.M + 1)) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    const float* v = p.v + (long long)b * p.vstride;
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
        row_arg[(long long)b * p.M + i] = arg < p.N ? arg : 0;  // (never an index outside the row: a row of NaNs leaves the sentinel, and the mutual check below indexes with this value)
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
        const float vj = p.v[(long long)b * p.vstride + j];
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
        col_arg[(long long)b * p.N + j] = arg < p.M ? arg : 0;
    }
}

}  // namespace

void launch_sg_kenc_input(const float* k0, const float* k1, const float* s0, const float* s1, const float* size0, const float* size1, int B,
                          int M, int N, float* out, hipStream_t s) {
    const long long T = (long long)B * (M + N);
    hipLaunchKernelGGL(sg_kenc_input_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, k0, k1, s0, s1, size0, size1, B, M, N, out);
}

size_t sinkhorn_work_floats(int B, int M, int N) {
    const size_t vs = (size_t)(N + 4) / 4 * 4, nblk = (size_t)(M + 1 + FR - 1) / FR;
    return (size_t)B * (M + 1) + (size_t)B * vs + 2 * (size_t)B * nblk * vs + 16;
}

void launch_sinkhorn(const SinkhornArgs& a, hipStream_t s) {
    SinkArgs p;
    p.cost = a.cost; p.alpha = a.alpha; p.B = a.B; p.M = a.M; p.N = a.N;
    p.vstride = (a.N + 4) / 4 * 4;
    p.u = a.work;
    p.v = a.work + (size_t)a.B * (a.M + 1);
    p.norm = -logf((float)a.M + (float)a.N);
    p.log_mu_bin = logf((float)a.N) + p.norm;
    p.log_nu_bin = logf((float)a.M) + p.norm;
    launch_zero_words(p.v, (size_t)a.B * p.vstride, s);  // (a kernel, not a memset node: kernels.h)
    const long long rows = (long long)a.B * (a.M + 1);
    const dim3 rgrid((unsigned)((rows + 3) / 4)), cgrid((a.N + 1 + 63) / 64, a.B);
    const bool fused = (a.N & 3) == 0 && a.N <= 2048 && (reinterpret_cast<uintptr_t>(a.cost) & 15) == 0;
    if (fused) {
        SinkFusedArgs q;
        q.a = p;
        q.nblk = (a.M + 1 + FR - 1) / FR;
        q.pm = p.v + (size_t)a.B * p.vstride;
        q.pl = q.pm + (size_t)a.B * q.nblk * p.vstride;
        for (int it = 0; it < a.iters; ++it) {
            hipLaunchKernelGGL(sinkhorn_fused_kernel, dim3(q.nblk, a.B), dim3(64 * FW), 0, s, q);
            hipLaunchKernelGGL(sinkhorn_combine_kernel, dim3((a.N + 1 + 63) / 64, a.B), dim3(256), 0, s, q);
        }
    } else {
        for (int it = 0; it < a.iters; ++it) {
            hipLaunchKernelGGL(sinkhorn_rows_kernel, rgrid, dim3(256), 0, s, p);
            hipLaunchKernelGGL(sinkhorn_cols_kernel, cgrid, dim3(64 * COLS_WAVES), 0, s, p);
        }
    }
    if (a.iters == 0) launch_zero_words(p.u, (size_t)a.B * (a.M + 1), s);
    hipLaunchKernelGGL(sinkhorn_final_rows, rgrid, dim3(256), 0, s, p, a.log_assignment, a.row_best, a.row_arg);
    hipLaunchKernelGGL(sinkhorn_final_cols, dim3((a.N + 63) / 64, a.B), dim3(256), 0, s, p, a.col_best, a.col_arg);
}
