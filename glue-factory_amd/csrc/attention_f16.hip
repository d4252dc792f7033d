// Flash-style attention on the f16 MFMA with split-fp16 (SH8) operands -- MG_PREC_F16X3 twin of
// attention.hip (same math: lightglue.py:105-115 self, :200-210 cross).
//
// Per 32-key tile a wave (32 queries) issues 12 MFMAs for S^T = K Q^T (4 k-steps x {hi.hi, hi.lo, lo.hi})
// and 12 for O^T += V^T P^T.  The probabilities are split into hi/lo halves in registers and are used
// directly as the B operand (accumulator-as-operand: register 8s+4a+e of a lane is key 16s+8a+4h+e, so
// the V^T image keeps the keys of a 16-block in the order [h][a][e]).  K tiles are staged as
// [key][256 B] with the 16-byte piece index XOR (key&15); V tiles are transposed on the way into LDS
// ([d][hi 64 B | lo 64 B], piece index XOR (d>>1)&7): every ds_read_b128 is conflict-free.  The output
// leaves as SH8 (16 consecutive d per lane after one permlane32 swap per register pair).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT = 32;

__global__ __launch_bounds__(256, 2) void attention_f16x3_kernel(const AttnF16Args p) {
    __shared__ u32x4 s_k[2][KT * 16];
    __shared__ u32x4 s_v[2][64 * 8];

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

    const int i32 = lane & 31, h = lane >> 5;
    const int ld4 = p.ld >> 2;  // 16-byte pieces per row
    const u32x4* Q = reinterpret_cast<const u32x4*>(p.q);
    const u32x4* Kp = reinterpret_cast<const u32x4*>(p.k);
    const u32x4* Vp = reinterpret_cast<const u32x4*>(p.v);

    // Q fragments (B operand): lane = (query, k-half); k-step s covers d = 16s..16s+15 -> group 2s+h
    const int qi = min(q0 + wave * 32 + i32, Lq - 1);
    const u32x4* qrow = Q + (qbase + qi) * ld4 + ((p.q_col + head * 64) >> 2);
    f16x8 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qh[s] = __builtin_bit_cast(f16x8, qrow[(2 * s + h) * 2]);
        ql[s] = __builtin_bit_cast(f16x8, qrow[(2 * s + h) * 2 + 1]);
    }

    f32x16 o[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
    float m_run = -__builtin_inff(), l_run = 0.f;

    const int ntiles = (Lkv + KT - 1) / KT;
    u32x4 rk[2], rv[2];
    // K loader: piece (key = idx>>4, c = idx&15).  V loader: key = t&31 across lanes (bank spread of the
    // transposing 2-byte writes), piece c = (t>>5) + 8i.
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 256 * i, key = idx >> 4, c = idx & 15;
            const long long row = kbase + min(kt * KT + key, Lkv - 1);
            rk[i] = Kp[row * ld4 + ((p.k_col + head * 64) >> 2) + c];
            const int vkey = t & 31, vc = (t >> 5) + 8 * i;
            const long long vrow = kbase + min(kt * KT + vkey, Lkv - 1);
            rv[i] = Vp[vrow * ld4 + ((p.v_col + head * 64) >> 2) + vc];
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = t + 256 * i, key = idx >> 4, c = idx & 15;
            s_k[buf][key * 16 + (c ^ (key & 15))] = rk[i];
            const int vkey = t & 31, vc = (t >> 5) + 8 * i, g = vc >> 1, part = vc & 1;
            // key = 16s + 8a + 4hh + e  ->  position 16s + 8hh + 4a + e inside the tile row
            const int pos = (vkey & 16) | ((vkey & 4) << 1) | ((vkey & 8) >> 1) | (vkey & 3);
            _Float16* vt = reinterpret_cast<_Float16*>(s_v[buf]);
            const f16x8 hv = __builtin_bit_cast(f16x8, rv[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = 8 * g + j;
                vt[(d * 8 + ((part * 4 + (pos >> 3)) ^ ((d >> 1) & 7))) * 8 + (pos & 7)] = hv[j];
            }
        }
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ntiles) load_tile(kt + 1);
        const f16x8* sk = reinterpret_cast<const f16x8*>(s_k[buf]);
        const f16x8* sv = reinterpret_cast<const f16x8*>(s_v[buf]);

        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int c = (2 * st + h) * 2;
            const f16x8 kh = sk[i32 * 16 + (c ^ (i32 & 15))];
            const f16x8 kl = sk[i32 * 16 + ((c + 1) ^ (i32 & 15))];
            mma3(kh, kl, qh[st], ql[st], s);
        }
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
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const f32x8 pv = {s[8 * s2], s[8 * s2 + 1], s[8 * s2 + 2], s[8 * s2 + 3], s[8 * s2 + 4], s[8 * s2 + 5], s[8 * s2 + 6], s[8 * s2 + 7]};
            f16x8 ph, pl;
            split8(pv, ph, pl);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = dt * 32 + i32, f = (d >> 1) & 7;
                const f16x8 vh = sv[d * 8 + ((2 * s2 + h) ^ f)];
                const f16x8 vl = sv[d * 8 + ((4 + 2 * s2 + h) ^ f)];
                mma3(vh, vl, ph, pl, o[dt]);
            }
        }
        if (kt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float inv = 1.f / (l_run + __shfl_xor(l_run, 32));
    const int q = q0 + wave * 32 + i32;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        f32x16 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a0 = o[dt][e] * inv, b0 = o[dt][8 + e] * inv, a1 = o[dt][4 + e] * inv, b1 = o[dt][12 + e] * inv;
            half_swap(a0, b0);
            half_swap(a1, b1);
            w[e] = a0; w[4 + e] = b0; w[8 + e] = a1; w[12 + e] = b1;
        }
        if (q < Lq) {
            const f32x8 g0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
            const f32x8 g1 = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
            f16x8 hi0, lo0, hi1, lo1;
            split8(g0, hi0, lo0);
            split8(g1, hi1, lo1);
            // this lane's 16 consecutive d: dt*32 + 16h ..
            f16x8* orow = reinterpret_cast<f16x8*>(p.out) + (qbase + q) * (p.ldo >> 2) + ((head * 64 + dt * 32 + 16 * h) >> 2);
            orow[0] = hi0; orow[1] = lo0; orow[2] = hi1; orow[3] = lo1;
        }
    }
}

}  // namespace

void launch_attention_f16x3(const AttnF16Args& a, hipStream_t s) {
    const int L = a.M > a.N ? a.M : a.N;
    dim3 grid((L + 127) / 128, a.heads, 2 * a.B);
    hipLaunchKernelGGL(attention_f16x3_kernel, grid, dim3(256), 0, s, a);
}
