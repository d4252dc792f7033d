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
//
// VALU diet (the kernel is VALU-bound next to a 16x faster matrix pipe): logits arrive pre-multiplied
// by log2(e) (folded into the projection epilogue) so the soft-max is one v_exp_f32 per element; the
// running maximum is only raised when some query of the wave outgrows it by more than 2^8 (deferred
// rescale -- P stays < 2^8.5, far inside fp16 range for its hi half, sums are fp32); tail masking runs
// only on a partial tile; P is split with mask/sub/cvt_pkrtz (3 VALU per element).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int KT = 32, AT = 512, QW = 64;  // keys per tile; 8 waves x 64 queries (two MFMA tiles) per workgroup
constexpr float RESCALE_THR = 8.f;  // in log2 units

// The probabilities are kept as P' = 2^P_SHIFT * exp2(s - m_run): with the deferred rescale s - m_run <= RESCALE_THR, so
// P' <= 2^15 stays inside fp16 range, and the fp16-subnormal zone of the halves (< 2^-14) starts at P < 2^-21 instead of
// 2^-14: what a 2048-key row can lose to it is < 2048 * 2^-24-7 = 1e-6 of the row sum (it was 1.2e-4, the whole tolerance,
// for an adversarial row before the shift).  l_run and the accumulators carry the same factor; it cancels in O / l.
constexpr float P_SHIFT = 7.f;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// x (>= 0, fp16 range) -> packed (hi, lo) halves of a pair: hi = x rounded toward zero to fp16 (v_cvt_pkrtz: truncation to 11
// significant bits), lo = fp16(x - hi) rounded to nearest -> 22 bits, unbiased.  Round 3: 1.5 VALU instructions per element instead
// of 3 (mask, subtract, two packed conversions): the mixed-precision FMA reads the fp16 hi half directly, evaluates x - hi in fp32
// (exact) and writes the rounded fp16 result into one half of the destination -- of the 281 VALU instructions a full key tile executes
// in this VALU-issue-bound loop (branch-aware count of the ISA, profiles/r03_attention_isa.txt), 96 were the split; now 239.  Written as asm: left to itself hipcc converts hi back to fp32 first (no gain), and the
// translation unit is the only one built with the mix instructions enabled (build.py; DESIGN.md section 4, hazard 2, is about
// COMPILER-chosen v_fma_mixlo on a product -- the epilogue below pins its products in fp32 registers for that reason).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo)
        : "v"(hi), "v"(x0), "v"(x1));
}

__global__ __launch_bounds__(AT, 2) void attention_f16x3_kernel(const AttnF16Args p) {
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
    const int q0 = blockIdx.x * (8 * QW);
    if (q0 >= Lq) return;

    const int i32 = lane & 31, h = lane >> 5;
    const int ld4 = p.ld >> 2;
    const u32x4* Q = reinterpret_cast<const u32x4*>(p.q);
    const u32x4* Kp = reinterpret_cast<const u32x4*>(p.k);
    const u32x4* Vp = reinterpret_cast<const u32x4*>(p.v);

    f16x8 qh[2][4], ql[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = min(q0 + wave * QW + qt * 32 + i32, Lq - 1);
        const u32x4* qrow = Q + (qbase + qi) * ld4 + ((p.q_col + head * 64) >> 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[qt][s] = __builtin_bit_cast(f16x8, qrow[(2 * s + h) * 2]);
            ql[qt][s] = __builtin_bit_cast(f16x8, qrow[(2 * s + h) * 2 + 1]);
        }
    }

    f32x16 o[2][2];  // [query tile][d tile]
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][0][r] = 0.f; o[0][1][r] = 0.f; o[1][0][r] = 0.f; o[1][1][r] = 0.f; }
    float m_run[2] = {-__builtin_inff(), -__builtin_inff()}, l_run[2] = {0.f, 0.f};

    const int ntiles = (Lkv + KT - 1) / KT;
    u32x4 rk, rv;
    // K: piece (key = t>>4, c = t&15).  V: key = t&31 across lanes (bank spread of the transposing
    // 2-byte writes), piece c = t>>5.
    auto load_tile = [&](int kt) {
        const long long row = kbase + min(kt * KT + (t >> 4), Lkv - 1);
        rk = Kp[row * ld4 + ((p.k_col + head * 64) >> 2) + (t & 15)];
        const long long vrow = kbase + min(kt * KT + (t & 31), Lkv - 1);
        rv = Vp[vrow * ld4 + ((p.v_col + head * 64) >> 2) + (t >> 5)];
    };
    auto store_tile = [&](int buf) {
        const int key = t >> 4, c = t & 15;
        s_k[buf][key * 16 + (c ^ (key & 15))] = rk;
        const int vkey = t & 31, vc = t >> 5, g = vc >> 1, part = vc & 1;
        // key = 16s + 8a + 4hh + e  ->  position 16s + 8hh + 4a + e inside the tile row
        const int pos = (vkey & 16) | ((vkey & 4) << 1) | ((vkey & 8) >> 1) | (vkey & 3);
        _Float16* vt = reinterpret_cast<_Float16*>(s_v[buf]);
        const f16x8 hv = __builtin_bit_cast(f16x8, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = 8 * g + j;
            vt[(d * 8 + ((part * 4 + (pos >> 3)) ^ ((d >> 1) & 7))) * 8 + (pos & 7)] = hv[j];
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

        f32x16 s[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int c = (2 * st + h) * 2;
            const f16x8 kh = sk[i32 * 16 + (c ^ (i32 & 15))];
            const f16x8 kl = sk[i32 * 16 + ((c + 1) ^ (i32 & 15))];
            mma3x2(kh, kl, qh[0][st], ql[0][st], qh[1][st], ql[1][st], s[0], s[1]);
        }
        const int kleft = Lkv - kt * KT;
        f16x8 ph[2][2], pl[2][2];  // [query tile][k-step]
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (kleft < KT) {  // partial last tile only
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (acc_row(r, lane) >= kleft) s[qt][r] = -__builtin_inff();
            }
            float mt = s[qt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[qt][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            if (__any(mt > m_run[qt] + RESCALE_THR)) {  // wave-uniform, rare after the first tiles
                const float m_new = fmaxf(m_run[qt], mt);
                const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                l_run[qt] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[qt][0][r] *= alpha; o[qt][1][r] *= alpha; }
                m_run[qt] = m_new;
            }
            float ps = 0.f;
            const float m_off = m_run[qt] - P_SHIFT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[qt][r] = __builtin_amdgcn_exp2f(s[qt][r] - m_off);
                ps += s[qt][r];
            }
            l_run[qt] += ps;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 phu, plu;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hi, lo;
                    split_pair(s[qt][8 * s2 + 2 * j], s[qt][8 * s2 + 2 * j + 1], hi, lo);
                    phu[j] = hi;
                    plu[j] = lo;
                }
                ph[qt][s2] = __builtin_bit_cast(f16x8, phu);
                pl[qt][s2] = __builtin_bit_cast(f16x8, plu);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const int d = dt * 32 + i32, f = (d >> 1) & 7;
                const f16x8 vh = sv[d * 8 + ((2 * s2 + h) ^ f)];
                const f16x8 vl = sv[d * 8 + ((4 + 2 * s2 + h) ^ f)];
                mma3x2(vh, vl, ph[0][s2], pl[0][s2], ph[1][s2], pl[1][s2], o[0][dt], o[1][dt]);
            }
        }
        if (kt + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float inv = 1.f / (l_run[qt] + __shfl_xor(l_run[qt], 32));
        const int q = q0 + wave * QW + qt * 32 + i32;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            f32x16 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a0 = o[qt][dt][e] * inv, b0 = o[qt][dt][8 + e] * inv, a1 = o[qt][dt][4 + e] * inv, b1 = o[qt][dt][12 + e] * inv;
                half_swap(a0, b0);  // (asm: the products above are fp32 registers here, never fused into the conversions of split8)
                half_swap(a1, b1);
                w[e] = a0; w[4 + e] = b0; w[8 + e] = a1; w[12 + e] = b1;
            }
            const f32x8 g0 = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
            const f32x8 g1 = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
            f16x8 hi0, lo0, hi1, lo1;
            split8(g0, hi0, lo0);
            split8(g1, hi1, lo1);
            // quad transpose (common.h): store s writes query s of every quad with 4 lanes x 16 bytes -> full lines
            u32x4 p0 = __builtin_bit_cast(u32x4, hi0), p1 = __builtin_bit_cast(u32x4, lo0);
            u32x4 p2 = __builtin_bit_cast(u32x4, hi1), p3 = __builtin_bit_cast(u32x4, lo1);
            const int ql = i32 & 3;
            quad_transpose(p0, p1, p2, p3, ql);
            const int qq = q - ql;  // first query of the quad
            u32x4* ob = reinterpret_cast<u32x4*>(p.out) + ((head * 64 + dt * 32 + 16 * h) >> 2) + ql;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
                if (qq + sidx < Lq) ob[(qbase + qq + sidx) * (p.ldo >> 2)] = sidx == 0 ? p0 : sidx == 1 ? p1 : sidx == 2 ? p2 : p3;
        }
    }
}

}  // namespace

void launch_attention_f16x3(const AttnF16Args& a, hipStream_t s) {
    if (a.variant != 1) return launch_attention_f16x3_pw(a, s);
    const int L = a.M > a.N ? a.M : a.N;
    dim3 grid((L + 8 * QW - 1) / (8 * QW), a.heads, 2 * a.B);
    hipLaunchKernelGGL(attention_f16x3_kernel, grid, dim3(AT), 0, s, a);
}
