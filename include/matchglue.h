/*
 * matchglue.h -- C ABI of the MI355X-native SuperPoint + LightGlue forward path.
 *
 * This is the drop-in boundary for the one hot path of glue-factory's evaluation pipeline:
 *   extractor  gluefactory_nonfree/superpoint.py:202-350  (SuperPoint._forward, MagicLeap layout)
 *              gluefactory/models/extractors/superpoint_open.py:123-217 (open variant)
 *   matcher    gluefactory/models/matchers/lightglue.py:399-525 (LightGlue.forward)
 * The reference reaches that path through Python `nn.Module.__call__` on `BaseModel` subclasses
 * (gluefactory/models/base_model.py:103-114); it has no FFI of its own, so the entry points below
 * are what a binding for that path would call (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - plain C: opaque handle, raw pointers, sizes; no C++ exceptions cross the boundary.
 *   - every function returns 0 on success or a negative mg_status; mg_last_error() gives text.
 *   - all tensor pointers are DEVICE pointers owned by the caller unless marked "host".
 *   - launches are asynchronous on the `stream` argument (a hipStream_t passed as void*).
 *   - one handle per device; a handle is not re-entrant.
 *   - workspace and weights are owned by the handle (hipMalloc at create / first use of a shape).
 */
#ifndef MATCHGLUE_H
#define MATCHGLUE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_context* mg_handle;

enum mg_status {
    MG_OK = 0,
    MG_ERR_ARG = -1,      /* bad argument / unsupported shape */
    MG_ERR_HIP = -2,      /* HIP runtime error */
    MG_ERR_WEIGHTS = -3,  /* missing or mis-shaped weight tensor */
    MG_ERR_STATE = -4     /* call order (e.g. forward before weights are complete) */
};

enum mg_sp_variant { MG_SP_MAGICLEAP = 0, MG_SP_OPEN = 1 };
/* descriptor sampling conventions: superpoint.py:117-133 (legacy), :138-149 (fixed),
 * superpoint_open.py:20-31 (open) */
enum mg_sampling { MG_SAMPLE_LEGACY = 0, MG_SAMPLE_FIXED = 1, MG_SAMPLE_OPEN = 2,
                   MG_SAMPLE_CORNER = 3 /* raw pixel coordinates: sample_descriptors_corner_conv, models/lines/wireframe.py:8-20 */ };

/* Extractor configuration == the conf keys of superpoint.py:153-169 that the sparse path reads. */
typedef struct mg_sp_conf {
    int32_t variant;            /* mg_sp_variant */
    int32_t nms_radius;         /* 0..4 */
    int32_t remove_borders;     /* pixels, 0 = off */
    int32_t max_num_keypoints;  /* 1..4096 */
    int32_t sampling;           /* mg_sampling */
    float detection_threshold;
} mg_sp_conf;

/* Matcher configuration == lightglue.py:307-327 (depth/width adaptivity is dead in the reference). */
typedef struct mg_lg_conf {
    int32_t n_layers;   /* 9 */
    int32_t num_heads;  /* 4 (head dim 64, descriptor dim 256) */
    float filter_threshold;
} mg_lg_conf;

/* -- lifetime ------------------------------------------------------------------------------- */
int mg_create(int device, mg_handle* out);
int mg_destroy(mg_handle h);
const char* mg_last_error(void);
/* ABI version of this header (bumped on any signature change). */
#define MG_ABI_VERSION 11 /* bumped whenever an entry point or a conf struct changes */
int mg_abi_version(void);   /* == MG_ABI_VERSION of the header the library was built from */
/* Counts, in this process, every event that makes a RECORDED launch sequence stale: a re-allocation of handle-owned workspace
 * (workspaces grow on demand when a larger shape arrives, and a grown buffer is a NEW device address), the release of a model's
 * device weights before they are re-tiled (mg_weights_ready after new mg_set_weight calls), a change of mg_set_precision, a new
 * mg_debug_request (requests select kernel variants).  A caller that recorded the launches of a forward pass -- a hipGraph captured
 * from the stream passed to mg_*_forward, gluefactory_amd/graphed.py -- must re-record once this value has changed: the recorded
 * kernel arguments point into old buffers or name other kernels.  Nothing in the reference corresponds to it (eager PyTorch). */
unsigned long long mg_workspace_generation(void);

/* -- arithmetic ------------------------------------------------------------------------------
 * MG_PREC_F32   : fp32-input MFMA everywhere (v_mfma_f32_32x32x2_f32), bit-for-bit an fp32 FMA chain.
 * MG_PREC_F16X3 : matrix work on the f16 MFMA with split-fp16 operands (x = hi + lo, three products,
 *                 fp32 accumulation): ~2^-21 relative operand error, 5.3x the fp32 MFMA rate.        */
enum mg_precision { MG_PREC_F32 = 0, MG_PREC_F16X3 = 1 };
int mg_set_precision(mg_handle h, int mode);

/* -- weights ---------------------------------------------------------------------------------
 * Upload one tensor of a reference state_dict, addressed by its reference key name
 * ("conv1a.weight", "backbone.0.0.bn.running_var", "transformers.3.self_attn.Wqkv.weight", ...).
 * `data` is a HOST pointer to contiguous fp32 in the reference's own layout (OIHW / [out][in]);
 * the library re-tiles it for the kernels.  `which` selects the model: 0 = SuperPoint, 1 = LightGlue, 2 = SuperGlue, 3 = GlueStick.
 * mg_weights_ready() re-tiles and uploads; it returns 0 once every tensor the model needs has
 * arrived.  `variant_or_layers` is the mg_sp_variant for which == 0 and n_layers for which == 1. */
int mg_set_weight(mg_handle h, int which, const char* key, const float* data, const int64_t* shape, int ndim);
int mg_weights_ready(mg_handle h, int which, int variant_or_layers);

/* -- SuperPoint -------------------------------------------------------------------------------
 * image        (B,C,H,W) fp32 in [0,1], C in {1,3}, NCHW as the reference takes it
 * image_size   (B,2) fp32 (w,h) or NULL  -- superpoint.py:247-251 (MagicLeap variant only)
 * keypoints    (B,K,2) fp32 (x,y)+0.5     K = conf.max_num_keypoints
 * scores       (B,K)   fp32
 * descriptors  (B,K,256) fp32, L2-normalised
 * num_valid    (B) int32: min(#candidates, K); rows >= num_valid are zero-filled
 * heatmap      optional (B,8*(H/8),8*(W/8)) fp32 dense scores before NMS   (NULL to skip)
 * dense_desc   optional (B,256,H/8,W/8) fp32 NCHW normalised descriptors    (NULL to skip)
 */
int mg_sp_forward(mg_handle h, const mg_sp_conf* conf, const float* image, int B, int C, int H, int W,
                  const float* image_size, float* keypoints, float* scores, float* descriptors,
                  int32_t* num_valid, float* heatmap, float* dense_desc, void* stream);

/* Stage entry points (same kernels, used by the stage-wise parity tests and by callers that
 * already hold a dense map):
 *   mg_sp_detect : heatmap (B,Hs,Ws) -> NMS + border + threshold + top-k  (superpoint.py:241-302)
 *                  nms_out optional (B,Hs,Ws): the NMS'd, border-masked map the reference thresholds
 *   mg_sp_sample : integer keypoints + dense NCHW descriptors -> (B,K,256)  (superpoint.py:323-339)
 */
int mg_sp_detect(mg_handle h, const mg_sp_conf* conf, const float* heatmap, int B, int Hs, int Ws,
                 const float* image_size, float* keypoints, float* scores, int32_t* num_valid,
                 float* nms_out, void* stream);
int mg_sp_sample(mg_handle h, const mg_sp_conf* conf, const float* keypoints, const float* dense_nchw,
                 int B, int K, int Hc, int Wc, float* descriptors, void* stream);

/* -- LightGlue --------------------------------------------------------------------------------
 * kpts0 (B,M,2), kpts1 (B,N,2) fp32 pixel coords; desc0 (B,M,256), desc1 (B,N,256)
 * size0/size1  (B,2) fp32 (w,h) or NULL (then the keypoint extent is used, lightglue.py:25-26)
 * matches0 (B,M) int64, matches1 (B,N) int64 (-1 = unmatched); mscores0/1 fp32
 * log_assignment optional (B,M+1,N+1) fp32; ref_desc0/1 optional (B,M,256)/(B,N,256) = last layer
 */
int mg_lg_forward(mg_handle h, const mg_lg_conf* conf, const float* kpts0, const float* kpts1,
                  const float* desc0, const float* desc1, const float* size0, const float* size1,
                  int B, int M, int N, int64_t* matches0, int64_t* matches1, float* mscores0,
                  float* mscores1, float* log_assignment, float* ref_desc0, float* ref_desc1, void* stream);

/* Adaptive depth / width (lightglue.py:62-88,468-508,536-560), batch 1 as the reference asserts (:476,:481).
 * After every layer but the last:
 *   depth_confidence > 0: token confidences sigmoid(token_confidence.i(x)); stop when 1 - #{conf < t_i} / (M + N) > depth_confidence,
 *     t_i = clip(0.8 + 0.1 exp(-4 i / n_layers), 0, 1) (:536-539, :551-560).  The fork reads an undefined `self.confidence_thresholds`
 *     there (AttributeError at the first call), so this half follows its evident intent = upstream LightGlue; parity unpinned.
 *   width_confidence > 0: points with sigmoid(log_assignment.i.matchability(x)) <= 1 - width_confidence are dropped from both the
 *     descriptors and the encodings -- unless early stopping is on and the point is not yet confident (get_pruning_mask, :541-549).
 *     This half runs in the reference and is pinned by tests/golden/lightglue_adaptive.npz.
 * The assignment uses the head of the last layer that ran (:498-499).  matches / scores come back in the ORIGINAL indices (pruned
 * points: -1 / 0, :500-508); prune0/1 (M)/(N) int32 = 1 + pruning rounds survived (:463-466,:487,:495).
 * log_assignment (optional, room for (M+1)(N+1) floats) is (M'+1, N'+1) and ref_desc0/1 (optional, (M,256)/(N,256)) hold M' / N'
 * rows for the surviving points; out_dims (HOST, 3 ints) receives M', N' and the number of layers run.  Synchronises the stream
 * once per layer (the decisions are made on the host, as `if self.check_if_stop(...)` / `torch.where` do in the reference).
 * The two confidences are DOUBLES, as the reference holds them (Python floats): it compares fp32 tensors with the Python scalars
 * `1 - width_confidence` / `depth_confidence`, i.e. with fp32(1.0 - width) formed from the unrounded double -- rounding width to
 * fp32 first moves that threshold by ~1e-8 (ten fp32 ulps at 0.01). */
int mg_lg_forward_adaptive(mg_handle h, const mg_lg_conf* conf, double depth_confidence, double width_confidence,
                           const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                           const float* size0, const float* size1, int M, int N, int64_t* matches0, int64_t* matches1,
                           float* mscores0, float* mscores1, float* log_assignment, float* ref_desc0, float* ref_desc1,
                           int32_t* prune0, int32_t* prune1, int32_t* out_dims, void* stream);

/* -- SuperGlue -------------------------------------------------------------
 * gluefactory_nonfree/superglue.py:253-307: keypoint encoder (MLP on (x,y,score)), 18-layer attentional
 * GNN (self/cross alternating), final projection, log-domain Sinkhorn with a dustbin, mutual filter.
 * Weights: mg_set_weight(h, 2, key, ...) with the reference key names ("kenc.encoder.0.weight",
 * "gnn.layers.3.attn.proj.1.bias", "bin_score", ...), then mg_weights_ready(h, 2, n_layers).
 * scores0/1 (B,M)/(B,N) keypoint scores; size0/1 (B,2) (w,h) REQUIRED (superglue.py:268-273).
 * log_assignment optional (B,M+1,N+1); sinkhorn_cost optional (B,M,N). */
typedef struct mg_sg_conf {
    int32_t n_layers;            /* 18 */
    int32_t sinkhorn_iterations; /* 50 */
    float filter_threshold;
} mg_sg_conf;
int mg_sg_forward(mg_handle h, const mg_sg_conf* conf, const float* kpts0, const float* kpts1,
                  const float* desc0, const float* desc1, const float* scores0, const float* scores1,
                  const float* size0, const float* size1, int B, int M, int N, int64_t* matches0,
                  int64_t* matches1, float* mscores0, float* mscores1, float* log_assignment,
                  float* sinkhorn_cost, void* stream);

/* -- GlueStick ------------------------------------------------------------------
 * gluefactory/models/matchers/gluestick.py:136-312: SuperGlue-style GNN over keypoints whose leading entries are the
 * line junctions, a line layer after every self layer (per-endpoint MLP, mean per junction), double log-softmax
 * assignment with a dustbin for points and for lines.  Released shape only: input_dim = descriptor_dim = 256,
 * keypoint_encoder [32,64,128,256], num_line_iterations 1, line_attention off, no intermediate supervision.
 * Weights: mg_set_weight(h, 3, key, ...) with the reference key names, then mg_weights_ready(h, 3, n_layers).
 * lines0/1 (B,n,2,2) endpoint pixel coordinates; junc_idx0/1 (B,n,2) int64 keypoint index of every endpoint, all
 * < 2*n (the reference slices the junctions off the front of the keypoint list, :262-263); line_scores0/1 (B,n).
 * With n0 == 0 or n1 == 0 the line arguments may be NULL and no line output is written.
 * log_assignment (B,M+1,N+1), line_log_assignment (B,n0+1,n1+1), raw_line_scores (B,n0,n1) are optional. */
typedef struct mg_gs_conf {
    int32_t n_layers;            /* 18 */
    float filter_threshold;
} mg_gs_conf;
int mg_gs_forward(mg_handle h, const mg_gs_conf* conf, const float* kpts0, const float* kpts1, const float* desc0,
                  const float* desc1, const float* scores0, const float* scores1, const float* size0, const float* size1,
                  const float* lines0, const float* lines1, const int64_t* junc_idx0, const int64_t* junc_idx1,
                  const float* line_scores0, const float* line_scores1, int B, int M, int N, int n0, int n1,
                  int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* log_assignment,
                  int64_t* line_matches0, int64_t* line_matches1, float* line_mscores0, float* line_mscores1,
                  float* line_log_assignment, float* raw_line_scores, void* stream);

/* -- wireframe side of GlueStick -------------------------------------------------
 * gluefactory/models/lines/wireframe.py:184-216: keypoints closer than `radius` (torch.norm < r) to any line endpoint of
 * their image are dropped; the survivors are compacted to the front in their original order, rows past out_count[b] are
 * zero.  keypoints (B,K,2), scores (B,K), descriptors (B,K,256), endpoints (B,E,2) = lines.reshape(B,-1,2); num_valid
 * optional (B): rows beyond it are padding already.  Outputs are separate buffers of the same shapes.
 * Junction descriptors (wireframe.py:8-20,118) are mg_sp_sample with sampling = MG_SAMPLE_CORNER on the junction
 * coordinates; line detection (LSD) and the DBSCAN merge of endpoints stay on the CPU, as in the reference. */
int mg_wf_filter(mg_handle h, const float* keypoints, const float* scores, const float* descriptors,
                 const int32_t* num_valid, const float* endpoints, int B, int K, int E, float radius,
                 float* out_keypoints, float* out_scores, float* out_descriptors, int32_t* out_count, void* stream);

/* -- nearest-neighbour matcher ------------------------------------------------
 * gluefactory/models/matchers/nearest_neighbor_matcher.py:51-72: sim = desc0 . desc1^T, best (and second best for
 * the ratio test) per row and per column, thresholds on d = 2(1 - sim), optional mutual check. No weights.
 * desc0 (B,M,D), desc1 (B,N,D), any D (rows are zero-padded to a multiple of 16 internally). A threshold <= 0 is "off"
 * (the reference's None).
 * similarity optional (B,M,N); log_assignment optional (B,M+1,N+1) (double log-softmax, zero last row/column). */
typedef struct mg_nn_conf {
    float ratio_thresh;
    float distance_thresh;
    int32_t mutual_check;
} mg_nn_conf;
int mg_nn_forward(mg_handle h, const mg_nn_conf* conf, const float* desc0, const float* desc1, int B, int M, int N, int D,
                  int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* similarity,
                  float* log_assignment, void* stream);

/* -- introspection ----------------------------------------------------------------------------
 * Copy a named internal buffer of the LAST forward into `dst` (device, fp32), for stage-wise
 * parity tests: "lg.enc" (tokens,64), "lg.self<i>" / "lg.layer<i>" (tokens,256) descriptors after
 * the self / cross block of layer i (both images, image-0 rows first), "lg.qkv0" (tokens,768) the raw
 * rows of the first QKV projection (SH8 words in f16x3 mode).  Returns element count in *count.  Only
 * taps requested beforehand with mg_debug_request() are kept.  A request may also name a KERNEL VARIANT
 * kept for A/B timing and kernel-equivalence tests (no buffer behind it): "sp.conv_staged",
 * "sp.conv1ab_staged", "sp.conv_dma32", "lg.unfused_ffn", "sp.logits", "lg.attn8" (the 8-wave attention
 * kernel instead of attention_pw_f16.hip), "lg.gemm_generic" (the tiled GEMM instead of gemm_ws_f16.hip
 * for the QKV projections), "lg.attn.exp<N>" (timing ablations of the attention kernel: WRONG results on
 * purpose, for mg_test_attention only).                                                            */
int mg_debug_request(mg_handle h, const char* tap);
int mg_debug_read(mg_handle h, const char* tap, float* dst, size_t capacity, size_t* count, void* stream);

/* Timing of the dominant kernels of the last forward, measured with HIP events on `stream`
 * when enabled: mg_profile_enable(h,1); after a forward + stream sync, mg_profile_read() fills
 * names/ms for up to `cap` entries and returns the number available.                          */
int mg_profile_enable(mg_handle h, int on);
/* Restrict the timing to the one scope called `name` (NULL or "" = all scopes again).  An event pair costs about
 * 10 us of stream time per launch, so a throughput measurement times only the kernel its roofline is quoted on.  */
int mg_profile_filter(mg_handle h, const char* name);
int mg_profile_read(mg_handle h, char (*names)[48], float* ms, int cap);

/* Stand-alone self-test entries of the two matrix cores, in the arithmetic mode set by mg_set_precision()
 * (f32: fp32-input MFMA kernels; f16x3: the split-fp16 kernels the benchmark runs), for unit tests over wide
 * operand ranges against fp64.
 * mg_test_linear:    C[M,N] = A[M,K] * W[N,K]^T + bias; A, C device fp32, W / bias host fp32; K % 32 == 0.
 * mg_test_attention: out = softmax(q k^T) v per head (4 heads x 64, the caller has applied any logit scale to q):
 *                    q, k, v, out device fp32 (tokens, 256), rows = image-0 tokens (B*M) then image-1 tokens (B*N);
 *                    cross = 0: every sequence attends to itself, 1: to the other image of its pair
 *                    (lightglue.py:105-115, :200-210).                                                         */
int mg_test_linear(mg_handle h, const float* A, const float* W_host, const float* bias_host, int M, int N,
                   int K, float* C, void* stream);
int mg_test_attention(mg_handle h, const float* q, const float* k, const float* v, int B, int M, int N, int cross,
                      float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MATCHGLUE_H */
