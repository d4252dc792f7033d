// Wireframe side of the GlueStick path (gluefactory/models/lines/wireframe.py:163-306), the part that runs on the GPU:
// keypoints too close to a line endpoint are dropped (:184-216) and the survivors compacted in their original order.
// (Junction descriptors are sampled by sample_kernel in its MG_SAMPLE_CORNER mode, sp_post.hip; LSD and the DBSCAN
// clustering of endpoints stay on the CPU, as in the reference.)
#include "common.h"
#include "kernels.h"

namespace {

constexpr int WFT = 256;

// One workgroup per image: flag = any endpoint closer than `radius` (torch.norm(kp - e) < r, :185-193), exclusive scan of the
// kept flags in LDS, scatter of keypoints / scores, and the source row of every output row for the descriptor pass.
__global__ __launch_bounds__(WFT) void wf_filter_kernel(const WfFilterArgs p) {
    __shared__ int s_wsum[WFT / 64];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* kp = p.keypoints + (long long)b * p.K * 2;
    const float* ep = p.endpoints + (long long)b * p.E * 2;
    const int nv = p.num_valid ? min(p.num_valid[b], p.K) : p.K;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < p.K; k0 += WFT) {
        const int k = k0 + t;
        int keep = 0;
        float x = 0.f, y = 0.f;
        if (k < nv) {
            x = kp[2 * k];
            y = kp[2 * k + 1];
            keep = 1;
            for (int e = 0; e < p.E; ++e) {
                const float dx = __fsub_rn(x, ep[2 * e]), dy = __fsub_rn(y, ep[2 * e + 1]);
                if (sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))) < p.radius) { keep = 0; break; }
            }
        }
        // order-preserving compaction: wave ballot prefix + per-wave totals
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wsum[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wsum[w];
        if (keep) {
            const int d = off + before;
            p.out_keypoints[((long long)b * p.K + d) * 2] = x;
            p.out_keypoints[((long long)b * p.K + d) * 2 + 1] = y;
            p.out_scores[(long long)b * p.K + d] = p.scores[(long long)b * p.K + k];
            p.src_row[(long long)b * p.K + d] = k;
        }
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int w = 0; w < WFT / 64; ++w) tot += s_wsum[w];
            s_base += tot;
        }
        __syncthreads();
    }
    const int n = s_base;
    if (t == 0) p.out_count[b] = n;
    for (int d = n + t; d < p.K; d += WFT) {  // rows past the survivors are padding
        p.out_keypoints[((long long)b * p.K + d) * 2] = 0.f;
        p.out_keypoints[((long long)b * p.K + d) * 2 + 1] = 0.f;
        p.out_scores[(long long)b * p.K + d] = 0.f;
        p.src_row[(long long)b * p.K + d] = -1;
    }
}

// descriptor rows follow their keypoints: one wave per output row (256 floats = 64 lanes x 16 B)
__global__ __launch_bounds__(256) void wf_gather_rows_kernel(const float* __restrict__ in, const int* __restrict__ src_row, float* __restrict__ out, int B, int K) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * K) return;
    const int b = (int)(row / K), s = src_row[row];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (s >= 0) v = *reinterpret_cast<const f32x4*>(in + ((long long)b * K + s) * 256 + lane * 4);
    *reinterpret_cast<f32x4*>(out + row * 256 + lane * 4) = v;
}

}  // namespace

void launch_wf_filter(const WfFilterArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(wf_filter_kernel, dim3(a.B), dim3(WFT), 0, s, a);
    const long long rows = (long long)a.B * a.K;
    hipLaunchKernelGGL(wf_gather_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, a.descriptors, a.src_row, a.out_descriptors, a.B, a.K);
}
