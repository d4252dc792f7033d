// SuperPoint post-processing kernels: detector soft-max + depth-to-space, L2 normalisation,
// NMS + border + threshold + top-k keypoint selection, bilinear descriptor sampling.
// All HBM-bound / LDS-bound byte work: coalesced loads, LDS tiles, wave shuffles -- no MFMA.
#include "common.h"
#include "kernels.h"

#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------
// superpoint.py:226-230 -- softmax over the 65 logits of a cell, drop the dustbin, and scatter the
// 64 probabilities to the 8x8 pixels of the cell.  One wave per cell, lane c = logit c.
__global__ __launch_bounds__(256) void softmax_d2s_kernel(const float* __restrict__ logits, int ld, int B, int hc, int wc,
                                                         float* __restrict__ heat) {
    const int lane = threadIdx.x & 63;
    const long long cell = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long ncell = (long long)B * hc * wc;
    if (cell >= ncell) return;
    const float* lg = logits + cell * ld;
    const float v = lg[lane];
    const float dust = lg[64];
    const float m = fmaxf(wave_max(v), dust);
    const float e = expf(v - m);
    const float sum = wave_sum(e) + expf(dust - m);
    const float pr = e / sum;
    const int b = (int)(cell / (hc * wc));
    const int rem = (int)(cell - (long long)b * hc * wc);
    const int cy = rem / wc, cx = rem - cy * wc;
    const int Ws = wc * 8;
    heat[((long long)b * hc * 8 + cy * 8 + (lane >> 3)) * Ws + cx * 8 + (lane & 7)] = pr;
}

// F.normalize(p=2, eps=1e-12) over rows of `dim` floats (dim == 256: one wave per row, 4 floats per lane).
__global__ __launch_bounds__(256) void l2norm_rows_kernel(float* __restrict__ x, long long rows, int dim) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = x + row * dim;
    float ss = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    ss = wave_sum(ss);
    const float d = fmaxf(sqrtf(ss), 1e-12f);
    for (int c = lane * 4; c < dim; c += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
        v.x /= d; v.y /= d; v.z /= d; v.w /= d;
        *reinterpret_cast<f32x4*>(p + c) = v;
    }
}

// (B,HW,C) <-> (B,C,HW) through a 32x33 LDS tile.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
    // in: (B,R,Cc) -> out: (B,Cc,R)
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (long long)b * R * Cc;
    float* dst = out + (long long)b * R * Cc;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < Cc) ? src[(long long)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cc) dst[(long long)c * R + r] = tile[tx][i];
    }
}

// ---------------------------------------------------------------------------------------------
// simple_nms (superpoint.py:60-80) + border removal (:243-254) + threshold (:257), one 32x32 output
// tile per block with a 5r halo (the dependency radius of the three keep/suppress rounds).
// max_pool2d pads with -inf, so everything outside the image is -inf / "not kept".
constexpr int NMS_TILE = 32;
constexpr int NMS_THREADS = 512;   // 4 blocks/CU (40 KB of LDS each) x 8 waves: the kernel is a chain of barriers with short tasks between them -- four independent chains per CU instead of two (round 4; 1024 threads: 2 blocks/CU)
constexpr int COUNT_STRIDE = 32;  // ints between per-image candidate counters (one 128-B line each)

// Separable (2R+1)^2 max-pool over a TxT LDS tile (row stride TP).  Each thread produces 4 adjacent
// outputs from a 4+2R register window, so a pass costs (4+2R)/4 LDS reads per element instead of
// 2R+1.  Indices are clamped into the tile: a duplicated edge value cannot change a maximum.
// `load(offset)` supplies the pooled quantity on the fly and `store(offset, pooled)` consumes the result, so the
// element-wise steps of simple_nms ride on the pooling passes instead of being LDS round trips of their own.
// M (round 4): the pass produces outputs only on [M, T - M)^2.  The k-th of the five pooling passes of simple_nms is right only
// where its window and those of its k - 1 predecessors stayed inside the tile, i.e. on [kR, T - kR)^2, and that is also all the next
// pass reads: computing the whole 62 x 62 tile five times (3.75 x the 32 x 32 interior) spent half of the kernel on values nobody
// uses.  With M = kR the passes cover 56, 50, 44, 38, 32 of the 62 columns: 52 % of the work.
template <int R, int T, int TP, int M, class Load, class Store>
__device__ __forceinline__ void pool_pass(Load load, float* tmp, Store store) {
    constexpr int W = T - 2 * M, G = (W + 3) / 4;  // outputs per row / column, groups of 4 along the pooled axis
    constexpr int Y0 = (M - R > 0) ? M - R : 0, Y1 = (T - M + R < T) ? T - M + R : T;  // rows the column sub-pass reads
    for (int i = threadIdx.x; i < (Y1 - Y0) * G; i += NMS_THREADS) {
        const int yy = i / G, y = Y0 + yy, x0 = M + (i - yy * G) * 4;
        float w[4 + 2 * R];
#pragma unroll
        for (int k = 0; k < 4 + 2 * R; ++k) w[k] = load(y * TP + min(max(x0 + k - R, 0), T - 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m = w[j];
#pragma unroll
            for (int k = 1; k <= 2 * R; ++k) m = fmaxf(m, w[j + k]);
            if (x0 + j < T - M) tmp[y * TP + x0 + j] = m;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W * G; i += NMS_THREADS) {
        const int g = i / W, x = M + (i - g * W), y0 = M + g * 4;
        float w[4 + 2 * R];
#pragma unroll
        for (int k = 0; k < 4 + 2 * R; ++k) w[k] = tmp[min(max(y0 + k - R, Y0), Y1 - 1) * TP + x];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float m = w[j];
#pragma unroll
            for (int k = 1; k <= 2 * R; ++k) m = fmaxf(m, w[j + k]);
            if (y0 + j < T - M) store((y0 + j) * TP + x, m);
        }
    }
    __syncthreads();
}

template <int R>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const DetectArgs p) {
    constexpr int halo = 5 * R, T = NMS_TILE + 2 * halo, TP = (T + 3) / 4 * 4 + 1, n = T * T;
    __shared__ float s_s[T * TP];
    __shared__ float s_t[(T * TP > 2048) ? T * TP : 2048];  // also the 1024-entry u64 candidate staging
    __shared__ unsigned char s_keep[T * TP];
    __shared__ unsigned char s_supp[T * TP];

    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (private L2s), so all tiles of one image get ids that are
    // equal mod 8: the image's score map (1.2 MB at VGA) is fetched into ONE L2 and the 5R-pixel halos of neighbouring tiles are L2
    // hits instead of being re-fetched by eight different XCDs (the (32 + 10R)^2 / 32^2 = 3.75x read amplification of r = 3).
    const int ntx = (p.Ws + NMS_TILE - 1) / NMS_TILE, nty = (p.Hs + NMS_TILE - 1) / NMS_TILE, per_img = ntx * nty;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / per_img) * 8 + xcd, tile = slot % per_img;
    if (b >= p.B) return;
    const int tile_y = tile / ntx, tile_x = tile - tile_y * ntx;
    const int y0 = tile_y * NMS_TILE - halo, x0 = tile_x * NMS_TILE - halo;
    const float* heat = p.heat + (long long)b * p.Hs * p.Ws;
    const float NEG = -__builtin_inff();

    for (int i = threadIdx.x; i < n; i += NMS_THREADS) {
        const int y = i / T, x = i - y * T;
        const int gy = y0 + y, gx = x0 + x;
        s_s[y * TP + x] = (gy >= 0 && gy < p.Hs && gx >= 0 && gx < p.Ws) ? heat[(long long)gy * p.Ws + gx] : NEG;
    }
    __syncthreads();
    if (R > 0) {
        // max_mask = scores == max_pool(scores)                                     (superpoint.py:71)
        pool_pass<R, T, TP, R>([&](int o) { return s_s[o]; }, s_t, [&](int o, float m) { s_keep[o] = (s_s[o] != NEG) && (s_s[o] == m); });
        auto round = [&](auto m_c) {
            constexpr int MS = decltype(m_c)::value;  // margin of the suppression pass of this round; the keep pass comes one radius further in
            // supp_mask = max_pool(max_mask.float()) > 0                             (:73)
            pool_pass<R, T, TP, MS>([&](int o) { return s_keep[o] ? 1.f : 0.f; }, s_t, [&](int o, float m) { s_supp[o] = m > 0.f; });
            // supp_scores = where(supp_mask, 0, scores); new_max = supp_scores == max_pool(supp_scores);
            // max_mask |= new_max & ~supp_mask                                        (:74-76)
            pool_pass<R, T, TP, MS + R>([&](int o) { return s_s[o] == NEG ? NEG : (s_supp[o] ? 0.f : s_s[o]); }, s_t,
                                        [&](int o, float m) { if (s_s[o] != NEG && !s_supp[o] && s_s[o] == m) s_keep[o] = 1; });
        };
        round(std::integral_constant<int, 2 * R>{});
        round(std::integral_constant<int, 4 * R>{});
    } else {
        for (int i = threadIdx.x; i < T * TP; i += NMS_THREADS) s_keep[i] = 1;  // radius 0: every pixel is its own maximum
        __syncthreads();
    }

    // interior 32x32: border mask, optional dump, candidate compaction.  Candidates are first packed
    // in LDS so the block makes ONE global atomic (per-image counters sit on separate 128-B lines).
    __shared__ int s_cnt, s_base;
    unsigned long long* s_cand = reinterpret_cast<unsigned long long*>(s_t);  // >= 1024 entries
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int hlim = p.Hs, wlim = p.Ws;
    if (p.image_size) {
        wlim = (int)p.image_size[2 * b];
        hlim = (int)p.image_size[2 * b + 1];
    }
    for (int i = threadIdx.x; i < NMS_TILE * NMS_TILE; i += NMS_THREADS) {
        const int ty = i >> 5, tx = i & 31;
        const int gy = tile_y * NMS_TILE + ty, gx = tile_x * NMS_TILE + tx;
        if (gy >= p.Hs || gx >= p.Ws) continue;
        const int li = (ty + halo) * TP + tx + halo;
        float v = s_keep[li] ? s_s[li] : 0.f;
        if (p.border > 0) {
            if (gy < p.border || gx < p.border || gy >= hlim - p.border || gx >= wlim - p.border) v = -1.f;
        }
        const unsigned idx = (unsigned)(gy * p.Ws + gx);
        if (p.nms_out) p.nms_out[(long long)b * p.Hs * p.Ws + idx] = v;
        if (v > p.threshold) {
            const int pos = atomicAdd(&s_cnt, 1);
            s_cand[pos] = ((unsigned long long)float_to_key(v) << 32) | (0xFFFFFFFFu - idx);
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(&p.counts[b * COUNT_STRIDE], cnt);
    __syncthreads();
    // The per-image list holds Hs*Ws entries, which a correct run can never exceed (one candidate per pixel at most).  The append is
    // clamped to that capacity all the same: a counter that was NOT zero when the launch started (the unordered memset node of a graph
    // replay, round 3) then costs wrong candidates, never a write outside the list.
    const int cap = p.Hs * p.Ws, base = s_base;
    unsigned long long* dst = p.cand + (long long)b * cap;
    for (int i = threadIdx.x; i < cnt; i += NMS_THREADS)
        if (base >= 0 && base + i < cap) dst[base + i] = s_cand[i];
}

// ---------------------------------------------------------------------------------------------
// top_k_keypoints (superpoint.py:83-87, :286-302): per image, the K candidates with the largest
// (score, then lowest index) key, sorted descending; if there are at most K candidates they are
// returned in row-major order.  Radix-select (8 x 8-bit passes over the 64-bit keys, per-image
// block) finds the K-th key, the survivors are compacted into LDS and bitonic-sorted there.
constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAXK = 4096;

__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const DetectArgs p) {
    __shared__ unsigned long long s_keys[SEL_MAXK];
    __shared__ int s_hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ int s_want, s_cnt;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = min(max(p.counts[b * COUNT_STRIDE], 0), p.Hs * p.Ws);  // never read past the list (see the clamped append in nms_kernel)
    const int K = p.K;
    const unsigned long long* cand = p.cand + (long long)b * p.Hs * p.Ws;
    int P = 1;
    while (P < K) P <<= 1;
    const int nsel = n < K ? n : K;

    for (int i = tid; i < P; i += SEL_THREADS) s_keys[i] = 0ull;
    __syncthreads();

    if (n <= K) {
        // row-major order == ascending index == descending (0xFFFFFFFF - idx): sort on the low word
        for (int i = tid; i < n; i += SEL_THREADS) {
            const unsigned long long k = cand[i];
            s_keys[i] = (k << 32) | (k >> 32);
        }
    } else {
        if (tid == 0) { s_prefix = 0ull; s_want = K; s_cnt = 0; }
        __syncthreads();
        for (int pass = 0; pass < 8; ++pass) {
            const int shift = 56 - 8 * pass;
            for (int i = tid; i < 256; i += SEL_THREADS) s_hist[i] = 0;
            __syncthreads();
            const unsigned long long prefix = s_prefix;
            for (int i0 = 0; i0 < n; i0 += SEL_THREADS) {
                const int i = i0 + tid;
                bool act = false;
                int bin = 0;
                if (i < n) {
                    const unsigned long long k = cand[i];
                    act = (pass == 0) || ((k >> (shift + 8)) == (prefix >> (shift + 8)));
                    bin = (int)((k >> shift) & 255);
                }
                // wave-aggregate when every active lane hits the same bin (the high bytes of scores
                // are nearly constant), otherwise plain LDS atomics
                const unsigned long long am = __ballot(act);
                if (am) {
                    const int lead = __ffsll((long long)am) - 1;
                    const int lbin = __shfl(bin, lead);
                    const unsigned long long same = __ballot(act && bin == lbin);
                    if (same == am) {
                        if (lane == lead) atomicAdd(&s_hist[lbin], __popcll(am));
                    } else if (act) {
                        atomicAdd(&s_hist[bin], 1);
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                int want = s_want, cum = 0, d = 255;
                for (; d > 0; --d) {
                    if (cum + s_hist[d] >= want) break;
                    cum += s_hist[d];
                }
                s_prefix = prefix | ((unsigned long long)d << shift);
                s_want = want - cum;
            }
            __syncthreads();
        }
        const unsigned long long kth = s_prefix;
        for (int i = tid; i < n; i += SEL_THREADS) {
            const unsigned long long k = cand[i];
            if (k >= kth) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < SEL_MAXK) s_keys[pos] = k;
            }
        }
    }
    __syncthreads();

    // bitonic sort, descending, P elements
    if (P <= SEL_THREADS) {
        // One key per thread, in a register (round 4).  Of the log2(P) (log2(P) + 1) / 2 compare-exchange stages (55 at P = 1024) only those
        // with a partner distance >= 64 cross a wave: they go through LDS, alternating between two halves of s_keys so that one barrier
        // per stage suffices (10 at P = 1024); the others are two 32-bit wave shuffles.  Was: every stage an LDS round trip + barrier.
        // Element i keeps the larger key of the pair iff it is the lower index of a descending pair or the upper one of an ascending pair.
        unsigned long long v = s_keys[tid < P ? tid : 0];
        int buf = 0;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1) {
            const bool desc = ((tid & k) == 0);
            for (int j = k >> 1; j > 0; j >>= 1) {
                unsigned long long pv;
                if (j >= 64) {
                    buf ^= 1;
                    s_keys[buf * (SEL_MAXK / 2) + tid] = v;
                    __syncthreads();
                    pv = s_keys[buf * (SEL_MAXK / 2) + (tid ^ j)];
                } else {
                    const unsigned lo = __shfl_xor((unsigned)v, j), hi = __shfl_xor((unsigned)(v >> 32), j);
                    pv = ((unsigned long long)hi << 32) | lo;
                }
                const bool lower = (tid & j) == 0;
                const bool want_max = lower == desc;
                v = want_max ? (v > pv ? v : pv) : (v < pv ? v : pv);
            }
        }
        __syncthreads();  // the last readers of either half are done
        if (tid < P) s_keys[tid] = v;
        __syncthreads();
    } else {
        for (int k = 2; k <= P; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += SEL_THREADS) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long a = s_keys[i], c = s_keys[ixj];
                        const bool desc = ((i & k) == 0);
                        if (desc ? (a < c) : (a > c)) { s_keys[i] = c; s_keys[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
    }

    for (int i = tid; i < K; i += SEL_THREADS) {
        float kx = 0.f, ky = 0.f, sc = 0.f;
        if (i < nsel) {
            unsigned long long k = s_keys[i];
            if (n <= K) k = (k << 32) | (k >> 32);
            const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
            sc = key_to_float((unsigned)(k >> 32));
            const int y = idx / p.Ws, x = idx - y * p.Ws;
            kx = (float)x + 0.5f;
            ky = (float)y + 0.5f;
        }
        p.keypoints[((long long)b * K + i) * 2] = kx;
        p.keypoints[((long long)b * K + i) * 2 + 1] = ky;
        p.scores[(long long)b * K + i] = sc;
    }
    if (tid == 0) p.num_valid[b] = nsel;
}

// ---------------------------------------------------------------------------------------------
// Descriptor sampling (superpoint.py:117-149, superpoint_open.py:20-31): grid_sample(bilinear, zero
// padding) of the normalised dense map at each keypoint, then L2-normalise.  One wave per keypoint,
// lane = 4 channels, so each corner is one coalesced 1 KiB read of the NHWC map.
__global__ __launch_bounds__(256) void sample_kernel(const SampleArgs p, float divx, float divy) {
    const int lane = threadIdx.x & 63;
    const long long kp = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kp >= (long long)p.B * p.K) return;
    const int b = (int)(kp / p.K);
    const float sh = p.mode == 3 ? 0.f : 0.5f;  // MG_SAMPLE_CORNER takes raw coordinates
    const float xi = p.keypoints[kp * 2] - sh, yi = p.keypoints[kp * 2 + 1] - sh;
    float gx, gy, ix, iy;
    if (p.mode == 0) {  // legacy: shifted grid, align_corners=True
        gx = __fadd_rn(__fsub_rn(xi, 4.f), 0.5f);
        gy = __fadd_rn(__fsub_rn(yi, 4.f), 0.5f);
    } else if (p.mode == 1 || p.mode == 3) {
        gx = xi; gy = yi;
    } else {
        gx = __fadd_rn(xi, 0.5f); gy = __fadd_rn(yi, 0.5f);
    }
    gx = __fsub_rn(__fmul_rn(__fdiv_rn(gx, divx), 2.f), 1.f);
    gy = __fsub_rn(__fmul_rn(__fdiv_rn(gy, divy), 2.f), 1.f);
    if (p.mode == 0) {
        ix = __fmul_rn(__fadd_rn(gx, 1.f), (float)(p.wc - 1) / 2.f);
        iy = __fmul_rn(__fadd_rn(gy, 1.f), (float)(p.hc - 1) / 2.f);
    } else {
        ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)p.wc / 2.f), 0.5f);
        iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)p.hc / 2.f), 0.5f);
    }
    const float xw = floorf(ix), yn = floorf(iy);
    const float w = __fsub_rn(ix, xw), e = __fsub_rn(1.f, w), nn = __fsub_rn(iy, yn), s = __fsub_rn(1.f, nn);
    const float wnw = __fmul_rn(s, e), wne = __fmul_rn(s, w), wsw = __fmul_rn(nn, e), wse = __fmul_rn(nn, w);
    const int x0 = (int)xw, y0 = (int)yn;
    const float* dense = p.dense + (long long)b * p.hc * p.wc * 256;
    auto corner = [&](int yy, int xx) -> f32x4 {
        if (yy < 0 || yy >= p.hc || xx < 0 || xx >= p.wc) return f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v = *reinterpret_cast<const f32x4*>(dense + ((long long)yy * p.wc + xx) * 256 + lane * 4);
        if (p.normalize_cells) {  // F.normalize of the cell, same arithmetic as l2norm_rows_kernel (the dense map is not needed)
            const float cs = wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
            const float cd = fmaxf(sqrtf(cs), 1e-12f);
            v.x /= cd; v.y /= cd; v.z /= cd; v.w /= cd;
        }
        return v;
    };
    const f32x4 vnw = corner(y0, x0), vne = corner(y0, x0 + 1), vsw = corner(y0 + 1, x0), vse = corner(y0 + 1, x0 + 1);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(vnw[j], wnw), __fmul_rn(vne[j], wne)), __fmul_rn(vsw[j], wsw)), __fmul_rn(vse[j], wse));
    float ss = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
    ss = wave_sum(ss);
    const float d = fmaxf(sqrtf(ss), 1e-12f);
    o.x /= d; o.y /= d; o.z /= d; o.w /= d;
    *reinterpret_cast<f32x4*>(p.desc + kp * 256 + lane * 4) = o;
}

}  // namespace

void launch_softmax_d2s(const float* logits, int ld, int B, int hc, int wc, float* heat, hipStream_t s) {
    const long long cells = (long long)B * hc * wc;
    hipLaunchKernelGGL(softmax_d2s_kernel, dim3((unsigned)((cells + 3) / 4)), dim3(256), 0, s, logits, ld, B, hc, wc, heat);
}

void launch_l2norm_rows(float* x, long long rows, int dim, hipStream_t s) {
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, dim);
}

void launch_nhwc_to_nchw(const float* in, float* out, int B, int HW, int C, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, in, out, HW, C);
}
void launch_nchw_to_nhwc(const float* in, float* out, int B, int HW, int C, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (HW + 31) / 32, B), dim3(256), 0, s, in, out, C, HW);
}

// The per-image candidate counters are cleared by a KERNEL, not by hipMemsetAsync: recorded into a hipGraph (stream capture), the memset
// node of ROCm 7.2 was not ordered before the NMS kernel that follows it -- a replay on new data mixed candidates of the previous image
// into the list (results differed from the eager ones, one run ended in a memory access fault), while the same stream executed eagerly
// is fine.  profiles/r03_g_graph_replay_diag.txt (memset) vs r03_g_diag_sp_zero_kernel.txt (kernel).
__global__ void zero_ints_kernel(int* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}
void launch_zero_words(void* p, size_t words, hipStream_t s) {
    if (!words) return;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, 2048);
    hipLaunchKernelGGL(zero_ints_kernel, dim3(blocks), dim3(256), 0, s, static_cast<int*>(p), words);
}
void launch_detect(const DetectArgs& a, hipStream_t s) {
    launch_zero_words(a.counts, (size_t)a.B * COUNT_STRIDE, s);
    const int per_img = ((a.Ws + NMS_TILE - 1) / NMS_TILE) * ((a.Hs + NMS_TILE - 1) / NMS_TILE);
    dim3 grid((unsigned)(8 * ((a.B + 7) / 8) * per_img));  // id -> (XCD = id % 8, image group, tile), see nms_kernel
    switch (a.radius) {
        case 0: hipLaunchKernelGGL(nms_kernel<0>, grid, dim3(NMS_THREADS), 0, s, a); break;
        case 1: hipLaunchKernelGGL(nms_kernel<1>, grid, dim3(NMS_THREADS), 0, s, a); break;
        case 2: hipLaunchKernelGGL(nms_kernel<2>, grid, dim3(NMS_THREADS), 0, s, a); break;
        case 3: hipLaunchKernelGGL(nms_kernel<3>, grid, dim3(NMS_THREADS), 0, s, a); break;
        default: hipLaunchKernelGGL(nms_kernel<4>, grid, dim3(NMS_THREADS), 0, s, a); break;
    }
    hipLaunchKernelGGL(select_kernel, dim3(a.B), dim3(SEL_THREADS), 0, s, a);
}

void launch_sample(const SampleArgs& a, hipStream_t s) {
    float divx, divy;
    if (a.mode == 0) {  // python double arithmetic then fp32, as torch.tensor([...]) does
        divx = (float)((double)a.wc * 8 - 4.0 - 0.5);
        divy = (float)((double)a.hc * 8 - 4.0 - 0.5);
    } else {
        divx = (float)a.wc * 8.f;
        divy = (float)a.hc * 8.f;
    }
    const long long n = (long long)a.B * a.K;
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, a, divx, divy);
}
