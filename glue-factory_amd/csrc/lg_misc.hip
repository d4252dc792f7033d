// LightGlue byte-bound kernels: positional encoding, LayerNorm+GELU, matchability, dual-softmax
// assignment and mutual-arg-max filtering.  Coalesced row/column sweeps, wave shuffles; no MFMA.
#include "common.h"
#include "kernels.h"

namespace {

// lightglue.py:22-33 + :46-59.  enc[token] = cos(Wr*kn)[32] | sin(Wr*kn)[32]; the reference's
// repeat_interleave(2) is an indexing matter handled by the rotary epilogue (frequency = d>>1).
// POSENC_CHUNKS blocks per sequence (image b of side s), each encoding a contiguous slice of its tokens (one block per sequence, until
// round 4, ran 64 blocks on 256 CUs: 73 us for 16 MB of output); size == NULL falls back to the keypoint extent, which every block of
// the sequence recomputes from the whole sequence (2 L floats).
constexpr int POSENC_CHUNKS = 8;
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ k0, const float* __restrict__ k1,
                                                    const float* __restrict__ size0, const float* __restrict__ size1,
                                                    const float* __restrict__ Wr, int B, int M, int N, float* __restrict__ enc) {
    __shared__ float s_red[4][4];
    __shared__ float s_size[2];
    const int seq = blockIdx.x / POSENC_CHUNKS, chunk = blockIdx.x - seq * POSENC_CHUNKS, side = seq >= B, b = seq - side * B;
    const int L = side ? N : M;
    const float* kp = (side ? k1 : k0) + (long long)b * L * 2;
    const float* size = side ? size1 : size0;
    float* e = enc + (side ? ((long long)B * M + (long long)b * N) : (long long)b * M) * 64;
    if (size) {
        if (threadIdx.x < 2) s_size[threadIdx.x] = size[2 * b + threadIdx.x];
    } else {
        float mnx = __builtin_inff(), mny = __builtin_inff(), mxx = -__builtin_inff(), mxy = -__builtin_inff();
        for (int i = threadIdx.x; i < L; i += 256) {
            const float x = kp[2 * i], y = kp[2 * i + 1];
            mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
        }
        mnx = -wave_max(-mnx); mny = -wave_max(-mny); mxx = wave_max(mxx); mxy = wave_max(mxy);
        if ((threadIdx.x & 63) == 0) {
            const int w = threadIdx.x >> 6;
            s_red[w][0] = mnx; s_red[w][1] = mny; s_red[w][2] = mxx; s_red[w][3] = mxy;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) {
                s_red[0][0] = fminf(s_red[0][0], s_red[w][0]); s_red[0][1] = fminf(s_red[0][1], s_red[w][1]);
                s_red[0][2] = fmaxf(s_red[0][2], s_red[w][2]); s_red[0][3] = fmaxf(s_red[0][3], s_red[w][3]);
            }
            s_size[0] = __fsub_rn(__fadd_rn(1.f, s_red[0][2]), s_red[0][0]);
            s_size[1] = __fsub_rn(__fadd_rn(1.f, s_red[0][3]), s_red[0][1]);
        }
    }
    __syncthreads();
    const float sw = s_size[0], sh = s_size[1];
    const float shx = sw / 2.f, shy = sh / 2.f, sc = fmaxf(sw, sh) / 2.f;
    const int per = (L + POSENC_CHUNKS - 1) / POSENC_CHUNKS, t_lo = chunk * per, t_hi = min(L, t_lo + per);
    for (int i = t_lo * 32 + threadIdx.x; i < t_hi * 32; i += 256) {
        const int tok = i >> 5, f = i & 31;
        const float x = __fdiv_rn(__fsub_rn(kp[2 * tok], shx), sc);
        const float y = __fdiv_rn(__fsub_rn(kp[2 * tok + 1], shy), sc);
        const float pr = __fadd_rn(__fmul_rn(x, Wr[2 * f]), __fmul_rn(y, Wr[2 * f + 1]));
        e[(long long)tok * 64 + f] = cosf(pr);
        e[(long long)tok * 64 + 32 + f] = sinf(pr);
    }
}

// lightglue.py:139-140 -- LayerNorm(512, eps) then exact (erf) GELU, in place.  One wave per row, 8
// consecutive features per lane; with `sh8` the result is written as split-fp16 over the same bytes
// (a lane's 32 bytes of fp32 become its own hi/lo pieces, so the in-place rewrite is race-free).
__global__ __launch_bounds__(256) void ln_gelu_kernel(float* __restrict__ x, long long rows, int dim, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int sh8) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = x + row * dim + lane * 8;
    f32x4 v[2];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(p + i * 4);
        sum += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mean = wave_sum(sum) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            sq += d * d;
        }
    const float var = wave_sum(sq) / (float)dim;
    const float rstd = 1.f / sqrtf(var + eps);
    f32x8 o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 8 + i * 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + lane * 8 + i * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float y = (v[i][j] - mean) * rstd * g[j] + bb[j];
            o[i * 4 + j] = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
        }
    }
    if (sh8) {
        f16x8 hi, lo;
        split8(o, hi, lo);
        f16x8* q = reinterpret_cast<f16x8*>(p);
        q[0] = hi;
        q[1] = lo;
    } else {
        *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
}

// lightglue.py:270,279-280 -- matchability logit z = <x, w> + b per token.  One wave per row of 256.
__global__ __launch_bounds__(256) void matchability_kernel(const float* __restrict__ x, long long rows, const float* __restrict__ w, float bias,
                                                          float* __restrict__ z) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * 256 + lane * 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(w + lane * 4);
    const float d = wave_sum(a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w);
    if (lane == 0) z[row] = d + bias;
}

__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

// Row statistics of sim: max_j and log sum_j exp(sim - max).  One wave per row.
__global__ __launch_bounds__(256) void assign_row_stats(const AssignArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * p.M) return;
    const float* s = p.sim + row * p.N;
    float m = -__builtin_inff();
    for (int j = lane; j < p.N; j += 64) m = fmaxf(m, s[j]);
    m = wave_max(m);
    float acc = 0.f;
    for (int j = lane; j < p.N; j += 64) acc += expf(s[j] - m);
    acc = wave_sum(acc);
    if (lane == 0) { p.row_mx[row] = m; p.row_ls[row] = logf(acc); }
}

// log sigmoid(z) and log sigmoid(-z) of every matchability logit, once (the assembly kernels below would otherwise
// evaluate them M*N times)
__global__ __launch_bounds__(256) void assign_prep(const float* __restrict__ z, long long n, float* __restrict__ ls, float* __restrict__ lsn) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ls[i] = log_sigmoid(z[i]);
    lsn[i] = log_sigmoid(-z[i]);
}

constexpr int ACW = 8, ACB = 4;  // column sweeps: waves per block (interleaved rows), independent loads per batch

// Column statistics: 64 columns per block, online max/sum per lane then an LDS combine.
__global__ __launch_bounds__(64 * ACW) void assign_col_stats(const AssignArgs p) {
    __shared__ float s_m[ACW][64], s_l[ACW][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* s = p.sim + (long long)b * p.M * p.N;
    float m = -__builtin_inff(), l = 0.f;
    if (c < p.N)
        for (int i0 = g; i0 < p.M; i0 += ACW * ACB) {
            float v[ACB];
#pragma unroll
            for (int k = 0; k < ACB; ++k) {
                const int i = i0 + ACW * k;
                v[k] = i < p.M ? s[(long long)i * p.N + c] : -__builtin_inff();
            }
            float bm = v[0];
#pragma unroll
            for (int k = 1; k < ACB; ++k) bm = fmaxf(bm, v[k]);
            const float nm = fmaxf(m, bm);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < ACB; ++k) acc += expf(v[k] - nm);
            l = l * expf(m - nm) + acc;
            m = nm;
        }
    s_m[g][cl] = m;
    s_l[g][cl] = l;
    __syncthreads();
    if (g == 0 && c < p.N) {
        float mm = s_m[0][cl];
#pragma unroll
        for (int k = 1; k < ACW; ++k) mm = fmaxf(mm, s_m[k][cl]);
        float ll = 0.f;
#pragma unroll
        for (int k = 0; k < ACW; ++k)
            if (s_l[k][cl] > 0.f) ll += s_l[k][cl] * expf(s_m[k][cl] - mm);
        p.col_mx[(long long)b * p.N + c] = mm;
        p.col_ls[(long long)b * p.N + c] = logf(ll);
    }
}

__device__ __forceinline__ float assign_value(float sim, float rmx, float rls, float cmx, float cls, float cert) {
    // (log_softmax rows + log_softmax cols) + certainties, in the reference's association (lightglue.py:255-259)
    const float s0 = __fsub_rn(__fsub_rn(sim, rmx), rls);
    const float s1 = __fsub_rn(__fsub_rn(sim, cmx), cls);
    return __fadd_rn(__fadd_rn(s0, s1), cert);
}

// Assemble S (optional write of the (M+1)x(N+1) matrix) and the row arg-max.  One wave per row.
__global__ __launch_bounds__(256) void assign_rows(const AssignArgs p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long rows = (long long)p.B * (p.M + 1);
    if (row >= rows) return;
    const int b = (int)(row / (p.M + 1)), i = (int)(row - (long long)b * (p.M + 1));
    float* la = p.log_assignment ? p.log_assignment + row * (p.N + 1) : nullptr;
    if (i == p.M) {  // last row: unmatched bin of image 1
        if (la) {
            for (int j = lane; j < p.N; j += 64) la[j] = p.lsn1[(long long)b * p.N + j];
            if (lane == 0) la[p.N] = 0.f;
        }
        return;
    }
    const long long r = (long long)b * p.M + i;
    const float* s = p.sim + r * p.N;
    const float rmx = p.row_mx[r], rls = p.row_ls[r];
    const float ls0 = p.ls0[r];
    float best = -__builtin_inff();
    int arg = 0x7fffffff;
    for (int j = lane; j < p.N; j += 64) {
        const long long c = (long long)b * p.N + j;
        const float cert = __fadd_rn(ls0, p.ls1[c]);
        const float v = assign_value(s[j], rmx, rls, p.col_mx[c], p.col_ls[c], cert);
        if (la) la[j] = v;
        if (v > best) { best = v; arg = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oa = __shfl_xor(arg, o);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) {
        p.row_best[r] = best;
        p.row_arg[r] = arg < p.N ? arg : 0;  // (never an index outside the row: a row of NaNs leaves the sentinel, and the mutual check below indexes with this value)
        if (la) la[p.N] = p.lsn0[r];
    }
}

// Column arg-max of S: 64 columns per block, interleaved rows, first maximum wins.
__global__ __launch_bounds__(64 * ACW) void assign_cols(const AssignArgs p) {
    __shared__ float s_b[ACW][64];
    __shared__ int s_a[ACW][64];
    const int b = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, g = threadIdx.x >> 6;
    const float* s = p.sim + (long long)b * p.M * p.N;
    float best = -__builtin_inff();
    int arg = 0x7fffffff;
    if (c < p.N) {
        const long long cc = (long long)b * p.N + c;
        const float cmx = p.col_mx[cc], cls = p.col_ls[cc], ls1 = p.ls1[cc];
        for (int i0 = g; i0 < p.M; i0 += ACW * ACB) {
            float v[ACB];
#pragma unroll
            for (int k = 0; k < ACB; ++k) {
                const int i = i0 + ACW * k;
                v[k] = i < p.M ? s[(long long)i * p.N + c] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < ACB; ++k) {
                const int i = i0 + ACW * k;
                if (i < p.M) {
                    const long long r = (long long)b * p.M + i;
                    const float x = assign_value(v[k], p.row_mx[r], p.row_ls[r], cmx, cls, __fadd_rn(p.ls0[r], ls1));
                    if (x > best) { best = x; arg = i; }
                }
            }
        }
    }
    s_b[g][cl] = best;
    s_a[g][cl] = arg;
    __syncthreads();
    if (g == 0 && c < p.N) {
        for (int k = 1; k < ACW; ++k)
            if (s_b[k][cl] > best || (s_b[k][cl] == best && s_a[k][cl] < arg)) { best = s_b[k][cl]; arg = s_a[k][cl]; }
        p.col_best[(long long)b * p.N + c] = best;
        p.col_arg[(long long)b * p.N + c] = arg < p.M ? arg : 0;
    }
}

// lightglue.py:287-303 -- mutual check, exp, threshold.
__global__ __launch_bounds__(256) void assign_filter(const AssignArgs p) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int* ra = p.row_arg + (long long)b * p.M;
    const int* ca = p.col_arg + (long long)b * p.N;
    const float* rb = p.row_best + (long long)b * p.M;
    if (i < p.M) {
        const int j = ra[i];
        const bool mutual = ca[j] == i;
        const float sc = mutual ? expf(rb[i]) : 0.f;
        p.ms0[(long long)b * p.M + i] = sc;
        p.m0[(long long)b * p.M + i] = (mutual && sc > p.threshold) ? j : -1;
    }
    if (i < p.N) {
        const int k = ca[i];          // row index matched to column i
        const bool mutual1 = ra[k] == i;
        const bool mutual0 = ca[ra[k]] == k;
        const float sc0 = mutual0 ? expf(rb[k]) : 0.f;
        const float sc = mutual1 ? sc0 : 0.f;
        p.ms1[(long long)b * p.N + i] = sc;
        p.m1[(long long)b * p.N + i] = (mutual1 && mutual0 && sc0 > p.threshold) ? k : -1;
    }
}

// ---- adaptive depth / width (batch 1) ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lg_adaptive_init_kernel(int* ind, int* prune0, int* prune1, int M, int N) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < M) { ind[i] = i; prune0[i] = 1; }
    else if (i < M + N) { ind[i] = i - M; prune1[i - M] = 1; }
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

__global__ __launch_bounds__(256) void lg_count_below_kernel(const float* __restrict__ z, int T, float thr, int* count) {
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int n = 0;
    for (int i = threadIdx.x; i < T; i += 256) n += sigmoidf_(z[i]) < thr ? 1 : 0;
    atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) *count = s_n;
}

// get_pruning_mask (lightglue.py:541-549): keep = matchability > 1 - width_confidence, OR the token is not yet confident.
// One workgroup per image: ordered compaction of the original-index list, source map for the row gather, prune counter of survivors.
__global__ __launch_bounds__(256) void lg_prune_kernel(const LgPruneArgs p) {
    __shared__ int s_wsum[4];
    __shared__ int s_base;
    const int img = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = img ? p.N : p.M, off = img ? p.M : 0;
    int* prune = img ? p.prune1 : p.prune0;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 256) {
        const int k = k0 + t;
        int keep = 0;
        if (k < n) {
            keep = sigmoidf_(p.z[off + k]) > p.thr_keep;
            if (p.zt) keep |= sigmoidf_(p.zt[off + k]) <= p.thr_tok;
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wave] = __popcll(m);
        __syncthreads();
        int o = s_base;
        for (int w = 0; w < wave; ++w) o += s_wsum[w];
        if (keep) {
            const int d = o + before, orig = p.ind[off + k];
            p.src[off + d] = k;          // per-image source row; destinations are re-based once both counts are known
            p.ind_out[off + d] = orig;   // staged at the OLD offset of the image, moved by the gather pass
            prune[orig] += 1;
        }
        __syncthreads();
        if (t == 0) s_base += s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
        __syncthreads();
    }
    if (t == 0) p.counts[img] = s_base;
}

// rows follow: new row r (image-0 rows [0, Mn), image-1 rows [Mn, Mn+Nn)) <- old row off_old + src; one wave per row
__global__ __launch_bounds__(256) void lg_prune_gather_kernel(const LgPruneArgs p, int Mn, int Nn, int* ind_final) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= Mn + Nn) return;
    const int img = r >= Mn, rr = img ? r - Mn : r, off_old = img ? p.M : 0;
    const int srow = off_old + p.src[off_old + rr];
    *reinterpret_cast<f32x4*>(p.x_out + (long long)r * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(p.x + (long long)srow * 256 + lane * 4);
    if (p.xs) reinterpret_cast<u32x4*>(p.xs_out)[(long long)r * 64 + lane] = reinterpret_cast<const u32x4*>(p.xs)[(long long)srow * 64 + lane];
    p.enc_out[(long long)r * 64 + lane] = p.enc[(long long)srow * 64 + lane];
    if (lane == 0) ind_final[r] = p.ind_out[off_old + rr];
}

// lightglue.py:500-508: m0_[ind0] = where(m0 == -1, -1, ind1[m0]); scores likewise; everything else -1 / 0
__global__ __launch_bounds__(256) void lg_scatter_matches_kernel(const long long* m0, const long long* m1, const float* s0, const float* s1, const int* ind,
                                                                int Mc, int Nc, int M, int N, long long* o0, long long* o1, float* os0, float* os1, int phase) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (phase == 0) {
        if (i < M) { o0[i] = -1; os0[i] = 0.f; }
        else if (i < M + N) { o1[i - M] = -1; os1[i - M] = 0.f; }
    } else {
        if (i < Mc) {
            const long long j = m0[i];
            o0[ind[i]] = j < 0 ? -1 : (long long)ind[Mc + j];
            os0[ind[i]] = s0[i];
        } else if (i < Mc + Nc) {
            const int k = i - Mc;
            const long long j = m1[k];
            o1[ind[Mc + k]] = j < 0 ? -1 : (long long)ind[j];
            os1[ind[Mc + k]] = s1[k];
        }
    }
}

}  // namespace

void launch_lg_adaptive_init(int* ind, int* prune0, int* prune1, int M, int N, hipStream_t s) {
    hipLaunchKernelGGL(lg_adaptive_init_kernel, dim3((M + N + 255) / 256), dim3(256), 0, s, ind, prune0, prune1, M, N);
}

void launch_lg_count_below(const float* z, int T, float thr, int* count, hipStream_t s) {
    hipLaunchKernelGGL(lg_count_below_kernel, dim3(1), dim3(256), 0, s, z, T, thr, count);
}

void launch_lg_prune(const LgPruneArgs& a, hipStream_t s) { hipLaunchKernelGGL(lg_prune_kernel, dim3(2), dim3(256), 0, s, a); }

void launch_lg_prune_gather(const LgPruneArgs& a, int M_new, int N_new, hipStream_t s) {
    // the compacted index list is staged in ind_out at the old image offsets; its final copy (new offsets) goes to the tail of `src`'s buffer
    // and is copied back, so that ind_out ends up [image-0 survivors | image-1 survivors]
    int* tmp = a.src + a.M + a.N;
    hipLaunchKernelGGL(lg_prune_gather_kernel, dim3((M_new + N_new + 3) / 4), dim3(256), 0, s, a, M_new, N_new, tmp);
    (void)hipMemcpyAsync(a.ind_out, tmp, sizeof(int) * (size_t)(M_new + N_new), hipMemcpyDeviceToDevice, s);
}

void launch_lg_scatter_matches(const long long* m0, const long long* m1, const float* s0, const float* s1, const int* ind, int Mc, int Nc, int M, int N,
                               long long* out_m0, long long* out_m1, float* out_s0, float* out_s1, hipStream_t s) {
    hipLaunchKernelGGL(lg_scatter_matches_kernel, dim3((M + N + 255) / 256), dim3(256), 0, s, m0, m1, s0, s1, ind, Mc, Nc, M, N, out_m0, out_m1, out_s0, out_s1, 0);
    hipLaunchKernelGGL(lg_scatter_matches_kernel, dim3((Mc + Nc + 255) / 256), dim3(256), 0, s, m0, m1, s0, s1, ind, Mc, Nc, M, N, out_m0, out_m1, out_s0, out_s1, 1);
}

void launch_posenc(const float* kpts0, const float* kpts1, const float* size0, const float* size1, const float* Wr, int B, int M,
                   int N, float* enc, hipStream_t s) {
    hipLaunchKernelGGL(posenc_kernel, dim3(2 * B * POSENC_CHUNKS), dim3(256), 0, s, kpts0, kpts1, size0, size1, Wr, B, M, N, enc);
}

void launch_ln_gelu(float* x, long long rows, int dim, const float* gamma, const float* beta, float eps, int sh8, hipStream_t s) {
    hipLaunchKernelGGL(ln_gelu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, dim, gamma, beta, eps, sh8);
}

void launch_matchability(const float* x, long long rows, const float* w, float bias, float* z, hipStream_t s) {
    hipLaunchKernelGGL(matchability_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, w, bias, z);
}

void launch_assignment(const AssignArgs& a, hipStream_t s) {
    const long long rows = (long long)a.B * a.M, T = (long long)a.B * (a.M + a.N);
    hipLaunchKernelGGL(assign_prep, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, a.z0, T, a.ls0, a.lsn0);  // z0 | z1 are contiguous
    hipLaunchKernelGGL(assign_row_stats, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(assign_col_stats, dim3((a.N + 63) / 64, a.B), dim3(64 * ACW), 0, s, a);
    const long long rows1 = (long long)a.B * (a.M + 1);
    hipLaunchKernelGGL(assign_rows, dim3((unsigned)((rows1 + 3) / 4)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(assign_cols, dim3((a.N + 63) / 64, a.B), dim3(64 * ACW), 0, s, a);
    const int L = a.M > a.N ? a.M : a.N;
    hipLaunchKernelGGL(assign_filter, dim3((L + 255) / 256, a.B), dim3(256), 0, s, a);
}

void launch_assign_filter(const AssignArgs& a, hipStream_t s) {
    const int L = a.M > a.N ? a.M : a.N;
    hipLaunchKernelGGL(assign_filter, dim3((L + 255) / 256, a.B), dim3(256), 0, s, a);
}
