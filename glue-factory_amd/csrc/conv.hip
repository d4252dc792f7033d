// SuperPoint encoder convolutions on NHWC activations (superpoint.py:209-219,224,233).
//
// conv3x3 is an implicit GEMM on the fp32-input MFMA: M = output pixels of a spatial tile, N = 64
// output channels, K = 9 taps x Cin.  Cin advances in chunks of 8; per chunk the (TH+2)x(TW+2) halo
// tile of the input ([pixel][8] floats) and the [tap][64][8] weight slice are staged in LDS
// (double-buffered, register prefetch, one barrier per chunk) and each tap is a shifted window of the
// same LDS tile.  One ds_read_b128 per operand feeds four MFMAs (common.h: mma_k8).
//
// A wave owns 4 rows x 16 cols of pixels = two 32-row MFMA tiles.  MFMA row m of a tile maps to
// pixel (dy,dx) = ((m>>1)&1, m&1) of 2x2 block m>>2, so the four accumulator registers 4g..4g+3 of a
// lane are exactly one 2x2 pooling window: ReLU + (optional affine) + 2x2 max-pool happen in
// registers and the full-resolution map of a pooled layer never reaches HBM.
#include "common.h"
#include "kernels.h"

namespace {

template <int WY, int WX>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const ConvArgs p) {
    constexpr int TH = 4 * WY, TW = 16 * WX, PW = TW + 2, NPIX = (TH + 2) * PW;
    constexpr int NIN = (2 * NPIX + 255) / 256;  // float4 loads per thread for the input chunk
    constexpr int NWT = (9 * 64 * 2 + 255) / 256;  // float4 loads per thread for the weight chunk
    __shared__ __attribute__((aligned(16))) float s_in[2][NPIX * 8];
    __shared__ __attribute__((aligned(16))) float s_w[2][9 * 64 * 8];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tiles_x = (p.W + TW - 1) / TW;
    const int y0 = (blockIdx.x / tiles_x) * TH, x0 = (blockIdx.x % tiles_x) * TW;
    const int n0 = blockIdx.y * 64;
    const int b = blockIdx.z;
    const float* in = p.in + (long long)b * p.H * p.W * p.Cin;

    // per-thread staging slots (independent of the chunk)
    int in_goff[NIN], in_loff[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
        const int idx = t + 256 * i;
        const int pix = idx >> 1, hf = idx & 1;
        const int py = pix / PW, px = pix - py * PW;
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        const bool ok = (idx < 2 * NPIX) && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        in_goff[i] = ok ? ((gy * p.W + gx) * p.Cin + hf * 4) : -1;
        in_loff[i] = (idx < 2 * NPIX) ? (pix * 8 + hf * 4) : -1;
    }
    int w_goff[NWT], w_loff[NWT];
#pragma unroll
    for (int i = 0; i < NWT; ++i) {
        const int idx = t + 256 * i;
        const int tap = idx >> 7, rem = idx & 127, n = rem >> 1, hf = rem & 1;
        const bool ok = idx < 9 * 64 * 2;
        w_goff[i] = ok ? ((tap * p.CoutPad + n0 + n) * 8 + hf * 4) : -1;
        w_loff[i] = ok ? ((tap * 64 + n) * 8 + hf * 4) : -1;
    }

    f32x4 rin[NIN], rwt[NWT];
    const int nchunks = p.Cin >> 3;
    const long long wchunk = (long long)9 * p.CoutPad * 8;

    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            rin[i] = (in_goff[i] >= 0) ? *reinterpret_cast<const f32x4*>(in + in_goff[i] + c * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NWT; ++i)
            rwt[i] = (w_goff[i] >= 0) ? *reinterpret_cast<const f32x4*>(p.Wt + c * wchunk + w_goff[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NIN; ++i)
            if (in_loff[i] >= 0) *reinterpret_cast<f32x4*>(&s_in[buf][in_loff[i]]) = rin[i];
#pragma unroll
        for (int i = 0; i < NWT; ++i)
            if (w_loff[i] >= 0) *reinterpret_cast<f32x4*>(&s_w[buf][w_loff[i]]) = rwt[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wy0 = (wave / WX) * 4, wx0 = (wave % WX) * 16;
    const int i32 = lane & 31, hf4 = (lane >> 5) * 4;
    const int abase = ((wy0 + ((i32 >> 1) & 1)) * PW + wx0 + 2 * (i32 >> 2) + (i32 & 1)) * 8 + hf4;
    const int bbase = i32 * 8 + hf4;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) load_chunk(c + 1);
        const float* si = s_in[buf];
        const float* sw = s_w[buf];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int toff = (ky * PW + kx) * 8;
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(si + abase + toff);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(si + abase + toff + 2 * PW * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(sw + bbase + (ky * 3 + kx) * 64 * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(sw + bbase + (ky * 3 + kx) * 64 * 8 + 32 * 8);
                mma_k8(a0, b0, acc[0][0]);
                mma_k8(a0, b1, acc[0][1]);
                mma_k8(a1, b0, acc[1][0]);
                mma_k8(a1, b1, acc[1][1]);
            }
        }
        if (c + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, ReLU, optional BatchNorm affine, optional 2x2 max-pool, NHWC store --------
    const int Ho = p.pool ? p.H / 2 : p.H, Wo = p.pool ? p.W / 2 : p.W;
    float* out = p.out + (long long)b * Ho * Wo * p.Cout;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = n0 + nt * 32 + i32;
        const bool col_ok = col < p.Cout;
        const float bias = col_ok ? p.bias[col] : 0.f;
        const float ps = (p.post_scale && col_ok) ? p.post_scale[col] : 1.f;
        const float pt = (p.post_shift && col_ok) ? p.post_shift[col] : 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float x = acc[tt][nt][r] + bias;
                if (p.relu) x = fmaxf(x, 0.f);
                if (p.post_scale) x = __fadd_rn(__fmul_rn(x, ps), pt);
                v[r] = x;
            }
            if (p.pool) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float m = fmaxf(fmaxf(v[4 * g], v[4 * g + 1]), fmaxf(v[4 * g + 2], v[4 * g + 3]));
                    const int py = (y0 + wy0 + 2 * tt) >> 1;
                    const int px = ((x0 + wx0) >> 1) + 2 * g + (lane >> 5);
                    if (col_ok && py < Ho && px < Wo) out[((long long)py * Wo + px) * p.Cout + col] = m;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = acc_row(r, lane);
                    const int oy = y0 + wy0 + 2 * tt + ((m >> 1) & 1);
                    const int ox = x0 + wx0 + 2 * (m >> 2) + (m & 1);
                    if (col_ok && oy < Ho && ox < Wo) out[((long long)oy * Wo + ox) * p.Cout + col] = v[r];
                }
            }
        }
    }
}

// conv1a: 1 (or RGB->gray, superpoint.py:204-206) -> 64 channels, 3x3, ReLU.  16 lanes per pixel,
// 4 output channels each, so a wave stores 1 KiB contiguous NHWC.
__global__ __launch_bounds__(256) void conv1a_kernel(const Conv1aArgs p) {
    __shared__ float s_w[64 * 9 + 64 * 3];
    for (int i = threadIdx.x; i < 64 * 9; i += 256) s_w[i] = p.w[i];
    for (int i = threadIdx.x; i < 64; i += 256) {
        s_w[576 + i] = p.bias[i];
        s_w[640 + i] = p.post_scale ? p.post_scale[i] : 1.f;
        s_w[704 + i] = p.post_shift ? p.post_shift[i] : 0.f;
    }
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pix = gid >> 4;
    const int cg = (int)(gid & 15);
    const long long HW = (long long)p.H * p.W;
    if (pix >= (long long)p.B * HW) return;
    const int b = (int)(pix / HW);
    const int rem = (int)(pix - (long long)b * HW);
    const int y = rem / p.W, x = rem - y * p.W;
    const float* img = p.image + (long long)b * p.C * HW;
    float g[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            float v = 0.f;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
                const long long o = (long long)yy * p.W + xx;
                if (p.C == 3)
                    v = __fadd_rn(__fadd_rn(__fmul_rn(img[o], 0.299f), __fmul_rn(img[HW + o], 0.587f)), __fmul_rn(img[2 * HW + o], 0.114f));
                else
                    v = img[o];
            }
            g[ky * 3 + kx] = v;
        }
    f32x4 o4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cg * 4 + j;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) a = fmaf(s_w[c * 9 + k], g[k], a);
        a += s_w[576 + c];
        a = fmaxf(a, 0.f);
        if (p.post_scale) a = __fadd_rn(__fmul_rn(a, s_w[640 + c]), s_w[704 + c]);
        o4[j] = a;
    }
    *reinterpret_cast<f32x4*>(p.out + pix * 64 + cg * 4) = o4;
}

}  // namespace

void launch_conv3x3(const ConvArgs& a, hipStream_t s) {
    // pick the wave arrangement that wastes the fewest tile pixels for this map
    auto waste = [&](int th, int tw) {
        long long ty = (a.H + th - 1) / th, tx = (a.W + tw - 1) / tw;
        return ty * th * tx * tw;
    };
    const long long w22 = waste(8, 32), w41 = waste(16, 16), w14 = waste(4, 64);
    const int nslices = (a.Cout + 63) / 64;
    if (w22 <= w41 && w22 <= w14) {
        dim3 grid(((a.H + 7) / 8) * ((a.W + 31) / 32), nslices, a.B);
        hipLaunchKernelGGL((conv3x3_kernel<2, 2>), grid, dim3(256), 0, s, a);
    } else if (w41 <= w14) {
        dim3 grid(((a.H + 15) / 16) * ((a.W + 15) / 16), nslices, a.B);
        hipLaunchKernelGGL((conv3x3_kernel<4, 1>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid(((a.H + 3) / 4) * ((a.W + 63) / 64), nslices, a.B);
        hipLaunchKernelGGL((conv3x3_kernel<1, 4>), grid, dim3(256), 0, s, a);
    }
}

void launch_conv1a(const Conv1aArgs& a, hipStream_t s) {
    const long long threads = (long long)a.B * a.H * a.W * 16;
    hipLaunchKernelGGL(conv1a_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
}
