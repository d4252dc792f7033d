// Host-side launchers of the HIP kernels (internal; the public surface is include/matchglue.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- gemm.hip : C[M,N] = A[M,K] * W[N,K]^T (+ epilogue), fp32-input MFMA --------------------------
struct GemmArgs {
    const float* A0 = nullptr;  // rows of K0 = ksplit floats (row stride lda0)
    int lda0 = 0;
    const float* A1 = nullptr;  // optional second source for k >= ksplit (concat-free FFN)
    int lda1 = 0;
    int ksplit = 0;             // == K when A1 is unused
    const float* W = nullptr;   // tiled [K/8][Npad][8] (w_rowmajor = 0) or row-major [N][ldw]
    int Npad = 0;
    int w_rowmajor = 0;
    int ldw = 0;
    const float* bias = nullptr;
    float* C = nullptr;
    int ldc = 0;
    int M = 0, N = 0, K = 0;
    // epilogue: v = acc + bias; rotary (cols < rotary_cols); relu; v*post_scale+post_shift;
    //           v *= scale (cols < scale_cols); v += residual
    const float* enc = nullptr;  // (M,64): cos[32] | sin[32] per row
    int rotary_cols = 0;
    int relu = 0;
    const float* post_scale = nullptr;
    const float* post_shift = nullptr;
    float scale = 1.f;
    int scale_cols = 0;
    const float* residual = nullptr;
    int ldr = 0;
    // batching (blockIdx.z): element strides
    int batch = 1;
    long long strideA = 0, strideW = 0, strideC = 0;
    int precise = 0;  // fp32 chains of one 32-deep stage, stage sums / bias / scale in float64 (bias + scale epilogue only)
};
void launch_gemm(const GemmArgs& a, hipStream_t s);

// ---- conv.hip ------------------------------------------------------------------------------------
struct ConvArgs {
    const float* in = nullptr;   // NHWC (B,H,W,Cin)
    const float* Wt = nullptr;   // tiled [Cin/8][9][CoutPad][8]
    int CoutPad = 0;
    const float* bias = nullptr;
    const float* post_scale = nullptr;  // optional per-channel affine after ReLU (BatchNorm of SP-open)
    const float* post_shift = nullptr;
    float* out = nullptr;        // NHWC (B,Ho,Wo,Cout), Ho = pool ? H/2 : H
    int B = 0, H = 0, W = 0, Cin = 0, Cout = 0;
    int pool = 0;
    int relu = 1;
};
void launch_conv3x3(const ConvArgs& a, hipStream_t s);

struct Conv1aArgs {
    const float* image = nullptr;  // NCHW (B,C,H,W), C in {1,3}
    int B = 0, C = 0, H = 0, W = 0;
    const float* w = nullptr;      // [64][9]
    const float* bias = nullptr;
    const float* post_scale = nullptr;
    const float* post_shift = nullptr;
    float* out = nullptr;          // NHWC (B,H,W,64)
};
void launch_conv1a(const Conv1aArgs& a, hipStream_t s);

// ---- conv_f16.hip : same convolutions on the f16 MFMA with split-fp16 (SH8) operands -------------
struct ConvF16Args {
    const void* in = nullptr;    // SH8 NHWC (B,H,W,Cin)
    const void* Wt = nullptr;    // SH8 tiled [Cin/16][9][CoutPad][64 B], values scaled by 1/winv (a power of two, api.hip: pow2_scale)
    float winv = 1.f;            // accumulators x winv (+ bias) in the epilogue
    int CoutPad = 0;
    const float* bias = nullptr;
    const float* post_scale = nullptr;
    const float* post_shift = nullptr;
    void* out = nullptr;         // SH8 NHWC (or fp32 NHWC when out_f32)
    int B = 0, H = 0, W = 0, Cin = 0, Cout = 0;
    int pool = 0;
    int relu = 1;
    int out_f32 = 0;
    int staged = 0;              // debug requests only: 1 = the register-staged kernels
};
void launch_conv3x3_f16x3(const ConvF16Args& a, hipStream_t s);
void launch_conv1a_sh8(const Conv1aArgs& a, hipStream_t s);
// conv1a + conv1b + pool fused (input = the image itself; a.in is unused)
void launch_conv1ab_f16x3(const ConvF16Args& a, const Conv1aArgs& c1, hipStream_t s);

// ---- gemm_f16.hip : the same GEMM on the f16 MFMA, SH8 operands --------------------------------
struct GemmF16Args {
    const void* A0 = nullptr;   // SH8 rows (tokens); lda in elements
    int lda0 = 0;
    const void* A1 = nullptr;
    int lda1 = 0;
    int ksplit = 0;
    const void* W = nullptr;    // SH8 tiled [K/16][Npad][64 B] (w_rowmajor = 0, values x 1/winv) or SH8 rows [N][ldw]
    float winv = 1.f;           // accumulators x winv (+ bias) in the epilogue: undoes the power-of-two scaling of uploaded weights
    int l2norm = 0;             // N == 256 only: rows of C are L2-normalised (x / max(||x||, 1e-12)) after the bias -- SuperPoint's convDb
    int Npad = 0;
    int w_rowmajor = 0;
    int ldw = 0;
    const float* bias = nullptr;
    float* C = nullptr;         // optional fp32 output (tokens, ldc)
    int ldc = 0;
    void* Cs = nullptr;         // optional SH8 output (tokens, ldcs)
    int ldcs = 0;
    int M = 0, N = 0, K = 0;    // tokens, features, reduction
    const float* enc = nullptr;
    int rotary_cols = 0;
    int relu = 0;
    const float* post_scale = nullptr;
    const float* post_shift = nullptr;
    float scale = 1.f;
    int scale_cols = 0;
    const float* residual = nullptr;
    int ldr = 0;
    int batch = 1;
    long long strideA = 0, strideW = 0, strideC = 0, strideCs = 0;  // elements
    int stagger = 0, stagger_lo = 0, stagger_hi = 0;  // workgroups [lo, hi) sleep `stagger` x 8128 cycles before they start (set by the launcher)
    int generic_only = 0;       // 1: never the weight-stationary kernel (debug request "lg.gemm_generic": A/B and bit-identity tests)
};
void launch_gemm_f16x3(const GemmF16Args& a, hipStream_t s);   // tries launch_gemm_ws_f16x3 first
bool launch_gemm_ws_f16x3(const GemmF16Args& a, hipStream_t s); // gemm_ws_f16.hip: K = 256, SH8 output, bias / rotary / scale only; false = not taken
void launch_to_sh8(const float* in, void* out, long long elems, float scale, hipStream_t s);  // SH8(in * scale)
void launch_from_sh8(const void* in, float* out, long long elems, hipStream_t s);          // hi + lo back to fp32
void launch_copy2_to_sh8(const float* in0, long long elems0, const float* in1, long long elems1, float* x, void* out, hipStream_t s);  // x = [in0 | in1], out = SH8(x)

// ---- ffn_f16.hip : x += W3 . act(W0f . [x | ctx] + b0) + b3 in one kernel (f16x3 mode) ------------------------------
struct FfnF16Args {
    const void* A0 = nullptr;    // SH8 (T,256): the residual stream (k 0..255 of ffn.0)
    const void* A1 = nullptr;    // SH8 (T,256): attention context (k 256..511; the out-projection is folded into W0)
    const void* W0 = nullptr;    // SH8 tiled [512/16][512][64 B], values x 1/winv0
    float winv0 = 1.f, winv3 = 1.f;  // accumulators x winv (+ bias): the power-of-two scaling of the uploaded weights
    const float* b0 = nullptr;   // (512)
    const float* gamma = nullptr;  // LayerNorm affine (512), act == 0 only
    const float* beta = nullptr;
    const void* W3 = nullptr;    // SH8 tiled [512/16][256][64 B], contraction index in accumulator order (api.hip: kperm)
    const float* b3 = nullptr;   // (256)
    float* X = nullptr;          // fp32 residual stream (T,256), updated in place
    void* XS = nullptr;          // its SH8 copy, rewritten
    int T = 0;
    float eps = 1e-5f;
    int act = 0;                 // 0: LayerNorm + GELU(erf) (LightGlue), 1: ReLU (SuperGlue / GlueStick, BatchNorm folded)
    float* stamps = nullptr;     // diagnostic build only: 8 floats per workgroup (cycles since its start at the phase boundaries)
};
void launch_ffn_f16x3(const FfnF16Args& a, hipStream_t s);

// ---- heads_f16.hip : convPb + soft-max + depth-to-space in one kernel (f16x3 mode) ---------------------------------------
struct HeadsF16Args {
    const void* in = nullptr;    // SH8 (cells, 256): convPa output
    const void* W = nullptr;     // SH8 tiled [256/16][Npad][64 B], rows 0..64 real, values x 1/winv
    float winv = 1.f;
    int Npad = 0;
    const float* bias = nullptr;        // (>= 65)
    const float* post_scale = nullptr;  // optional BatchNorm affine of the open variant (>= 65)
    const float* post_shift = nullptr;
    float* heat = nullptr;       // (B, 8 hc, 8 wc)
    int B = 0, hc = 0, wc = 0;
};
void launch_heads_f16x3(const HeadsF16Args& a, hipStream_t s);

// ---- sp_post.hip ---------------------------------------------------------------------------------
void launch_softmax_d2s(const float* logits, int ld, int B, int hc, int wc, float* heat, hipStream_t s);
void launch_l2norm_rows(float* x, long long rows, int dim, hipStream_t s);
void launch_nhwc_to_nchw(const float* in, float* out, int B, int HW, int C, hipStream_t s);
void launch_nchw_to_nhwc(const float* in, float* out, int B, int HW, int C, hipStream_t s);
struct DetectArgs {
    const float* heat = nullptr;  // (B,Hs,Ws)
    int B = 0, Hs = 0, Ws = 0;
    int radius = 0, border = 0, K = 0;
    float threshold = 0.f;
    const float* image_size = nullptr;  // (B,2) (w,h) or null
    unsigned long long* cand = nullptr; // (B, Hs*Ws) candidate keys
    int* counts = nullptr;              // (B)
    float* nms_out = nullptr;           // optional
    float* keypoints = nullptr;         // (B,K,2)
    float* scores = nullptr;            // (B,K)
    int* num_valid = nullptr;           // (B)
};
void launch_detect(const DetectArgs& a, hipStream_t s);
// Zero `words` 32-bit words with a KERNEL (never hipMemsetAsync: as a memset node of a captured hipGraph it was not ordered before the
// kernel that follows it on ROCm 7.2 -- sp_post.hip).
void launch_zero_words(void* p, size_t words, hipStream_t s);
struct SampleArgs {
    const float* keypoints = nullptr;  // (B,K,2) (x,y)+0.5
    const float* dense = nullptr;      // NHWC (B,hc,wc,256), normalised
    int B = 0, K = 0, hc = 0, wc = 0;
    int mode = 0;
    int normalize_cells = 0;           // 1: `dense` is NOT normalised yet; L2-normalise the four sampled cells on the fly
    // mode 3 (MG_SAMPLE_CORNER): `keypoints` are raw pixel coordinates (no +0.5 to undo), grid = xy / (8 wc, 8 hc), align_corners=False
    // = sample_descriptors_corner_conv, gluefactory/models/lines/wireframe.py:8-20 (junction descriptors)
    float* desc = nullptr;             // (B,K,256)
};
void launch_sample(const SampleArgs& a, hipStream_t s);

// ---- attention.hip -------------------------------------------------------------------------------
struct AttnArgs {
    const float* q = nullptr;  // rows of `ld` floats; head h occupies cols [q_col + 64h, +64)
    const float* k = nullptr;
    const float* v = nullptr;
    int ld = 0;
    int q_col = 0, k_col = 0, v_col = 0;
    float* out = nullptr;      // (tokens,256): head h -> cols [64h, +64)
    int ldo = 0;
    int B = 0, M = 0, N = 0;   // pairs, tokens in image 0 / image 1 (rows: image-0 block, then image-1)
    int heads = 4;
    int cross = 0;             // 0: each sequence attends to itself; 1: to the other image of its pair
};
void launch_attention(const AttnArgs& a, hipStream_t s);

// ---- attention_f16.hip : SH8 operands, f16 MFMA --------------------------------------------------
struct AttnF16Args {
    const void* q = nullptr;  // SH8 rows of `ld` elements; head h occupies cols [q_col + 64h, +64)
    const void* k = nullptr;
    const void* v = nullptr;
    int ld = 0;
    int q_col = 0, k_col = 0, v_col = 0;
    void* out = nullptr;      // SH8 (tokens, ldo)
    int ldo = 0;
    int B = 0, M = 0, N = 0;
    int heads = 4;
    int cross = 0;
    int variant = 0;          // 0: attention_pw_f16.hip; 1: the round-1 kernel (8 waves, two per SIMD; debug request "lg.attn8");
                              // 16 + N: timing ablation N of the pw kernel (debug request "lg.attn.exp<N>", wrong results on purpose)
};
void launch_attention_f16x3(const AttnF16Args& a, hipStream_t s);       // dispatches on a.variant
void launch_attention_f16x3_pw(const AttnF16Args& a, hipStream_t s);    // 4 waves, one per SIMD, software-pipelined (attention_pw_f16.hip)

// ---- lg_misc.hip ---------------------------------------------------------------------------------
void launch_posenc(const float* kpts0, const float* kpts1, const float* size0, const float* size1, const float* Wr,
                   int B, int M, int N, float* enc, hipStream_t s);
void launch_ln_gelu(float* x, long long rows, int dim, const float* gamma, const float* beta, float eps, int sh8, hipStream_t s);
void launch_matchability(const float* x, long long rows, const float* w, float bias, float* z, hipStream_t s);
struct AssignArgs {
    const float* sim = nullptr;  // (B,M,N)
    const float* z0 = nullptr;   // (B*M) matchability logits of image 0
    const float* z1 = nullptr;   // (B*N), directly behind z0
    int B = 0, M = 0, N = 0;
    float threshold = 0.f;
    // workspace
    float* ls0 = nullptr;        // (B*M + B*N) log sigmoid(z), image 0 then image 1 (ls1 = ls0 + B*M)
    float* ls1 = nullptr;
    float* lsn0 = nullptr;       // (B*M + B*N) log sigmoid(-z)
    float* lsn1 = nullptr;
    float* row_mx = nullptr;     // (B,M) max_j sim
    float* row_ls = nullptr;     // (B,M) log sum_j exp(sim - max)
    float* col_mx = nullptr;     // (B,N)
    float* col_ls = nullptr;     // (B,N)
    float* row_best = nullptr;   // (B,M) max_j S[i,j]
    int* row_arg = nullptr;      // (B,M)
    float* col_best = nullptr;   // (B,N)
    int* col_arg = nullptr;      // (B,N)
    // outputs
    float* log_assignment = nullptr;  // optional (B,M+1,N+1)
    long long* m0 = nullptr;
    long long* m1 = nullptr;
    float* ms0 = nullptr;
    float* ms1 = nullptr;
};
void launch_assignment(const AssignArgs& a, hipStream_t s);

// adaptive depth / width (lightglue.py:62-88,473-508,536-560), batch 1: token rows = image-0 rows (M) then image-1 rows (N)
void launch_lg_adaptive_init(int* ind, int* prune0, int* prune1, int M, int N, hipStream_t s);   // ind = arange per image, prune = 1
void launch_lg_count_below(const float* z, int T, float thr, int* count, hipStream_t s);         // #{ sigmoid(z) < thr }
struct LgPruneArgs {
    const float* z = nullptr;    // (M+N) matchability logits of the current tokens
    const float* zt = nullptr;   // (M+N) token-confidence logits, or null (no early stopping)
    float thr_tok = 0.f;         // confidence_threshold(layer)
    float thr_keep = 0.f;        // 1 - width_confidence
    int M = 0, N = 0;            // current token counts
    const int* ind = nullptr;    // (M+N) original index of every current token
    int* ind_out = nullptr;      // compacted
    int* prune0 = nullptr;       // (M_orig) / (N_orig): += 1 for every survivor
    int* prune1 = nullptr;
    int* src = nullptr;          // (M+N) workspace: source row of every output row
    int* counts = nullptr;       // (2) survivors per image
    const float* x = nullptr;    // token rows to compact: fp32 (T,256), SH8 (T,256) (may be null), encodings (T,64)
    const void* xs = nullptr;
    const float* enc = nullptr;
    float* x_out = nullptr;
    void* xs_out = nullptr;
    float* enc_out = nullptr;
};
void launch_lg_prune(const LgPruneArgs& a, hipStream_t s);                                        // keep flags, ordered compaction of ind, src map, counts
void launch_lg_prune_gather(const LgPruneArgs& a, int M_new, int N_new, hipStream_t s);          // rows follow
void launch_lg_scatter_matches(const long long* m0, const long long* m1, const float* s0, const float* s1, const int* ind, int Mc, int Nc, int M, int N,
                               long long* out_m0, long long* out_m1, float* out_s0, float* out_s1, hipStream_t s);

// ---- sg_misc.hip : SuperGlue keypoint-encoder input + log-domain Sinkhorn ---------------------------
void launch_sg_kenc_input(const float* k0, const float* k1, const float* s0, const float* s1, const float* size0, const float* size1, int B,
                          int M, int N, float* out, hipStream_t s);
struct SinkhornArgs {
    const float* cost = nullptr;  // (B,M,N)
    float alpha = 1.f;            // bin score
    int B = 0, M = 0, N = 0, iters = 50;
    float* work = nullptr;        // sinkhorn_work_floats(B,M,N) floats: u, v and the column partials
    float* log_assignment = nullptr;  // optional (B,M+1,N+1)
    float* row_best = nullptr;    // (B,M)
    int* row_arg = nullptr;
    float* col_best = nullptr;    // (B,N)
    int* col_arg = nullptr;
};
size_t sinkhorn_work_floats(int B, int M, int N);
void launch_sinkhorn(const SinkhornArgs& a, hipStream_t s);
void launch_assign_filter(const AssignArgs& a, hipStream_t s);

// ---- nn_misc.hip : nearest-neighbour matcher on a (B,M,N) similarity matrix ---------------------
struct NnArgs {
    const float* sim = nullptr;
    int B = 0, M = 0, N = 0;
    float ratio2 = 0.f, dist2 = 0.f;  // squared thresholds (fp32 of the double square, as torch casts the scalar)
    int use_ratio = 0, use_dist = 0, mutual = 1;
    float* row_lse = nullptr;  // (B,M) workspace
    float* col_lse = nullptr;  // (B,N)
    int* cand0 = nullptr;      // (B,M)
    int* cand1 = nullptr;      // (B,N)
    long long* m0 = nullptr;
    long long* m1 = nullptr;
    float* ms0 = nullptr;
    float* ms1 = nullptr;
    float* la = nullptr;       // optional (B,M+1,N+1)
};
void launch_nn_match(const NnArgs& a, hipStream_t s);

// ---- gs_misc.hip : GlueStick line side -----------------------------------------------------------
struct GsLineArgs {
    int B = 0, M = 0, N = 0, n0 = 0, n1 = 0;
    const long long* idx0 = nullptr;  // (B, n0, 2) junction index of every line endpoint (into the keypoints of image 0)
    const long long* idx1 = nullptr;
    const void* x = nullptr;          // token rows as the GEMMs read them: fp32 (f32 mode) or SH8 (f16x3), 256 channels
    const void* enc = nullptr;        // endpoint encodings, same storage, rows = image-0 endpoints then image-1 endpoints
    void* msg = nullptr;              // out: (E, 768) message rows, same storage
    const float* upd = nullptr;       // (E, 256) fp32 endpoint updates (scatter input)
    float* x32 = nullptr;             // fp32 residual stream, updated in place
    void* xs = nullptr;               // SH8 copy, refreshed for touched rows (nullptr in f32 mode)
};
void launch_gs_lenc_input(const float* l0, const float* l1, const float* s0, const float* s1, const float* size0, const float* size1, int B, int n0,
                          int n1, float* out, hipStream_t s);
void launch_gs_gather_msg(const GsLineArgs& a, hipStream_t s);
void launch_gs_scatter_mean(const GsLineArgs& a, hipStream_t s);
struct GsSoftmaxArgs {
    const float* scores = nullptr;  // (B,M,N)
    float bin = 1.f;
    int B = 0, M = 0, N = 0;
    float* row_lse = nullptr;       // (B,M) workspace
    float* col_lse = nullptr;       // (B,N)
    float* out = nullptr;           // optional (B,M+1,N+1)
    float* row_best = nullptr;      // (B,M)
    int* row_arg = nullptr;
    float* col_best = nullptr;      // (B,N)
    int* col_arg = nullptr;
};
void launch_gs_double_softmax(const GsSoftmaxArgs& a, hipStream_t s);
void launch_gs_line_scores(const float* J, const long long* idx0, const long long* idx1, int B, int n0, int n1, float* raw, hipStream_t s);

// ---- wf_misc.hip : wireframe side of GlueStick (keypoint / line-endpoint distance filter) --------
struct WfFilterArgs {
    const float* keypoints = nullptr;    // (B,K,2)
    const float* scores = nullptr;       // (B,K)
    const float* descriptors = nullptr;  // (B,K,256)
    const int* num_valid = nullptr;      // optional (B): rows beyond it are padding already
    const float* endpoints = nullptr;    // (B,E,2)
    int B = 0, K = 0, E = 0;
    float radius = 0.f;
    float* out_keypoints = nullptr;      // (B,K,2) survivors first, original order, zero padding
    float* out_scores = nullptr;         // (B,K)
    float* out_descriptors = nullptr;    // (B,K,256)
    int* out_count = nullptr;            // (B)
    int* src_row = nullptr;              // (B,K) workspace: source row of every output row (-1: padding)
};
void launch_wf_filter(const WfFilterArgs& a, hipStream_t s);
