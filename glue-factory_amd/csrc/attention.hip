// Flash-style multi-head attention over keypoint tokens on the fp32-input MFMA.
//
// Self attention (lightglue.py:105-115, SDPA fp32) and both directions of the cross attention
// (lightglue.py:200-210) run through this one kernel: softmax(Q K^T) V per (sequence, head), never
// materialising the (K x K) score matrix.  Scaling is folded into Q/K by the producing GEMM epilogue.
//
// A block = 4 waves x 32 queries.  Per 32-key tile a wave computes S^T = K_tile Q^T (keys on MFMA rows,
// queries on lanes), so the soft-max statistics of a query live in one lane (plus its lane+32 twin),
// and the exponentiated accumulator registers are *already* the B operand of O^T += V^T P^T -- no LDS
// round trip or shuffle for P.  K tiles sit in LDS as [key][64+4], V tiles transposed as [d][32+4]
// (both 16-byte-odd row strides -> conflict-free ds_read_b128), double-buffered with register prefetch.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT = 32;        // keys per tile
constexpr int KS = 68;        // K tile row stride (floats)
constexpr int VS = 36;        // V^T tile row stride (floats)

__global__ __launch_bounds__(256, 2) void attention_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) float s_k[2][KT * KS];
    __shared__ __attribute__((aligned(16))) float s_v[2][64 * VS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int head = blockIdx.y;
    const int seq = blockIdx.z;
    const int side = seq >= p.B ? 1 : 0;
    const int b = seq - side * p.B;
    const int Lq = side ? p.N : p.M;
    const int kside = p.cross ? (1 - side) : side;
    const int Lkv = kside ? p.N : p.M;
    const long long qbase = side ? ((long long)p.B * p.M + (long long)b * p.N) : (long long)b * p.M;
    const long long kbase = kside ? ((long long)p.B * p.M + (long long)b * p.N) : (long long)b * p.M;
    const int q0 = blockIdx.x * 128;
    if (q0 >= Lq) return;

    // Q fragments: lane = (query l&31, k-half l>>5), 8 groups of 8 along d
    const int qi = min(q0 + wave * 32 + (lane & 31), Lq - 1);
    const float* qrow = p.q + (qbase + qi) * p.ld + p.q_col + head * 64 + (lane >> 5) * 4;
    f32x4 qf[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) qf[g] = *reinterpret_cast<const f32x4*>(qrow + g * 8);

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -__builtin_inff(), l_run = 0.f;

    const int ntiles = (Lkv + KT - 1) / KT;
    f32x4 rk[2], rv[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 256 * i, key = idx >> 4, c4 = idx & 15;
            const long long row = kbase + min(kt * KT + key, Lkv - 1);
            rk[i] = *reinterpret_cast<const f32x4*>(p.k + row * p.ld + p.k_col + head * 64 + c4 * 4);
            rv[i] = *reinterpret_cast<const f32x4*>(p.v + row * p.ld + p.v_col + head * 64 + c4 * 4);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 256 * i, key = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4*>(&s_k[buf][key * KS + c4 * 4]) = rk[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) s_v[buf][(c4 * 4 + j) * VS + key] = rv[i][j];
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frag = (lane & 31) * KS + (lane >> 5) * 4;
    const int vfrag = (lane & 31) * VS + (lane >> 5) * 4;

    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) load_tile(kt + 1);
        const float* sk = s_k[buf];
        const float* sv = s_v[buf];

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sk + frag + g * 8);
            mma_k8(a, qf[g], s);
        }
        // mask the tail keys, then online soft-max for this lane's query
        const int kleft = Lkv - kt * KT;
        float mt = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (acc_row(r, lane) >= kleft) s[r] = -__builtin_inff();
            mt = fmaxf(mt, s[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
            ps += s[r];
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        // O^T += V^T P^T : registers 4g..4g+3 of s are keys 8g+4h+{0..3} == the k order of a b128 fragment
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 pb = {s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(sv + vfrag + g * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(sv + vfrag + 32 * VS + g * 8);
            mma_k8(v0, pb, o[0]);
            mma_k8(v1, pb, o[1]);
        }
        if (kt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    const int q = q0 + wave * 32 + (lane & 31);
    if (q < Lq) {
        float* orow = p.out + (qbase + q) * p.ldo + head * 64 + (lane >> 5) * 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(orow + dt * 32 + g * 8) = v;
            }
    }
}

}  // namespace

void launch_attention(const AttnArgs& a, hipStream_t s) {
    const int L = a.M > a.N ? a.M : a.N;
    dim3 grid((L + 127) / 128, a.heads, 2 * a.B);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, s, a);
}
