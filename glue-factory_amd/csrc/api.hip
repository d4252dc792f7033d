// C ABI of libmatchglue (include/matchglue.h): handle, weight upload / re-tiling, workspace, and
// the launch sequences of the SuperPoint and LightGlue forward passes.
#include "../../include/matchglue.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(MG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

// mg_workspace_generation(): bumped by everything that makes the arguments or the kernel selection of a recorded launch sequence stale --
// a workspace re-allocation, the release of a model's device weights before a re-tiling (load_state_dict), a precision change, a new
// debug request (they select kernel variants).  A hipGraph captured at generation g may only be replayed while the counter still reads g.
static std::atomic<unsigned long long> g_ws_generation{0};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        g_ws_generation.fetch_add(1, std::memory_order_relaxed);
        if (p) {
            if (hipDeviceSynchronize() != hipSuccess) return -1;
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        size_t alloc = need + (need >> 3) + 256;
        if (hipMalloc(&p, alloc) != hipSuccess) {
            p = nullptr;
            return -1;
        }
        bytes = alloc;
        return 0;
    }
    float* f() const { return static_cast<float*>(p); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

struct ConvLayer {
    int Cin = 0, Cout = 0, CoutPad = 0;
    float *W = nullptr, *bias = nullptr, *ps = nullptr, *pt = nullptr;
    float* Wh = nullptr;  // SH8 copy [Cin/16][9][CoutPad][64 B] for the f16x3 kernels, values x 1/winv (pow2_scale below)
    float winv = 1.f;
};
struct LinLayer {
    int N = 0, K = 0, Npad = 0;
    float *W = nullptr, *bias = nullptr, *ps = nullptr, *pt = nullptr;
    float* Wh = nullptr;  // SH8 copy [K/16][Npad][64 B] for the f16x3 GEMM, values x 1/winv (a power of two, pow2_scale below)
    float winv = 1.f;     // what the f16x3 kernels multiply their accumulators by before the bias
    float* Whp = nullptr;  // the same with the contraction index of every 32-block in accumulator order (ffn_f16.hip phase 3)
};
struct LgBlock {
    LinLayer proj;  // self: Wqkv (q|k|v rows), cross: [to_qk ; to_v]
    LinLayer out;   // out_proj / to_out
    LinLayer ffn0, ffn3;
    LinLayer ffn0f;  // ffn.0 with the out-projection folded in (consumes cat[x, ctx])
    float *ln_g = nullptr, *ln_b = nullptr;
};

struct ProfEntry {
    std::string name;
    hipEvent_t a, b;
};

}  // namespace

struct mg_context {
    int device = 0;
    int precision = MG_PREC_F16X3;
    std::map<std::string, HostTensor> raw[4];
    std::vector<void*> owned[5];  // device allocations of prepared weights, per model (`which`; 4 = mg_test_* scratch)
    int cur_which = 0;            // the model whose weights are being prepared (upload() books into owned[cur_which])

    // SuperPoint
    int sp_variant = -1;
    float *c1a_w = nullptr, *c1a_b = nullptr, *c1a_ps = nullptr, *c1a_pt = nullptr;
    ConvLayer conv[9];  // conv1b,2a,2b,3a,3b,4a,4b,Pa,Da
    LinLayer convPb, convDb;
    // LightGlue
    int lg_layers = -1;
    std::vector<LgBlock> lg_self, lg_cross;
    LinLayer final_proj;
    float* match_w = nullptr;
    float match_b = 0.f;
    // adaptive depth / width: the assignment head and the token-confidence head of EVERY layer (prepared on first adaptive use)
    int lg_heads = -1;
    std::vector<LinLayer> lg_final;
    std::vector<float*> lg_match_w, lg_tok_w;
    std::vector<float> lg_match_b, lg_tok_b;
    float* Wr = nullptr;
    // SuperGlue
    int sg_layers = -1;
    LinLayer sg_kenc[5];
    struct SgLayer { LinLayer proj, mlp0f, mlp3; };
    std::vector<SgLayer> sg_gnn;
    LinLayer sg_final;
    float sg_bin = 1.f;
    // GlueStick
    int gs_layers = -1;
    LinLayer gs_kenc[5], gs_lenc[5];
    struct GsLine { LinLayer mlp0, mlp3; };
    std::vector<SgLayer> gs_gnn;
    std::vector<GsLine> gs_line;
    LinLayer gs_final, gs_final_line;
    float gs_bin = 1.f, gs_line_bin = 1.f;

    // workspace
    std::map<std::string, DevBuf> ws;
    std::map<std::string, DevBuf> taps;
    std::map<std::string, size_t> tap_count;

    bool profile = false;
    std::string prof_filter;  // when not empty only the scope of that name, or of that dotted family, is timed (mg_profile_filter)
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> event_pool;
    size_t event_used = 0;

    float* buf(const std::string& name, size_t floats) {
        DevBuf& b = ws[name];
        if (b.ensure(floats * sizeof(float)) != 0) return nullptr;
        return b.f();
    }
    hipEvent_t next_event() {
        if (event_used == event_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            event_pool.push_back(e);
        }
        return event_pool[event_used++];
    }
};

namespace {

struct ProfScope {
    mg_context* h;
    hipStream_t s;
    size_t idx = 0;
    bool on;
    // the filter names one scope ("sp.conv1ab") or a dotted family ("lg.attn" = "lg.attn.self" + "lg.attn.cross": one GPU function)
    static bool selected(const std::string& f, const char* name) {
        return f.empty() || f == name || (std::strncmp(name, f.c_str(), f.size()) == 0 && name[f.size()] == '.');
    }
    ProfScope(mg_context* h_, hipStream_t s_, const char* name) : h(h_), s(s_), on(h_->profile && selected(h_->prof_filter, name)) {
        if (!on) return;
        ProfEntry e;
        e.name = name;
        e.a = h->next_event();
        e.b = h->next_event();
        (void)hipEventRecord(e.a, s);
        idx = h->prof.size();
        h->prof.push_back(e);
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(h->prof[idx].b, s);
    }
    ProfScope(mg_context* h_, hipStream_t s_, const std::string& name) : ProfScope(h_, s_, name.c_str()) {}
};

int upload(mg_context* h, const std::vector<float>& v, float** out) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, v.size() * sizeof(float)));
    HIPCHK(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    h->owned[h->cur_which].push_back(p);
    *out = static_cast<float*>(p);
    return 0;
}

// Re-tiling after a new load_state_dict: the previous device copies of that model are released first (after a device
// sync: launches of the old weights may still be in flight on some stream).
int begin_prepare(mg_context* h, int which) {
    h->cur_which = which;
    g_ws_generation.fetch_add(1, std::memory_order_relaxed);  // recorded launches point at the weight buffers released below
    if (h->owned[which].empty()) return 0;
    HIPCHK(hipDeviceSynchronize());
    for (void* q : h->owned[which]) (void)hipFree(q);
    h->owned[which].clear();
    return 0;
}

// which attention kernel the f16x3 mode launches: 0 = attention_pw_f16.hip (default), 1 = the round-1 8-wave kernel (debug request
// "lg.attn8"), 16 + N = timing ablation N of the default kernel (debug request "lg.attn.exp<N>": wrong results on purpose, bench A/B only)
int attn_variant(mg_context* h) {
    if (h->taps.count("lg.attn8")) return 1;
    for (auto& kv : h->taps)
        if (kv.first.rfind("lg.attn.exp", 0) == 0) return 16 + atoi(kv.first.c_str() + 11);
    return 0;
}

const HostTensor* find(mg_context* h, int which, const std::string& key) {
    auto it = h->raw[which].find(key);
    return it == h->raw[which].end() ? nullptr : &it->second;
}

int need(mg_context* h, int which, const std::string& key, std::initializer_list<int64_t> shape, const HostTensor** out) {
    const HostTensor* t = find(h, which, key);
    if (!t) return fail(MG_ERR_WEIGHTS, "missing weight tensor '" + key + "'");
    std::vector<int64_t> want(shape);
    int64_t n = 1;
    for (auto s : want) n *= s;
    if (t->numel() != n) return fail(MG_ERR_WEIGHTS, "weight tensor '" + key + "' has the wrong number of elements");
    *out = t;
    return 0;
}

// OIHW 3x3 -> [Cin/8][9][CoutPad][8]
// Power-of-two pre-scaling of a weight tensor for the split-fp16 copy.  hi = fp16(w) carries 11 bits and lo = fp16(w - hi) the next
// 11 -- but only while lo is a NORMAL fp16 number (>= 2^-14): for |w| < 2^-3 the lo half is subnormal (spacing 2^-24) and the pair holds
// w to an ABSOLUTE 2^-25 only.  Trained / He-initialised weights sit at 1e-2..1e-1: a weight of 0.01 was held to 3e-6 relative, 17 bits
// instead of 22.  Measured with the float64 oracle (round 3): that alone put the f16x3 log_assignment 3.5x further from the exact value
// than the fp32 CPU path (ffn.3's weights, |w| < 0.013, accounted for 90 % of it).  Scaling the tensor by 2^s so that max |w| lands in
// [2^14, 2^15) is exact, keeps hi inside fp16 range and moves every lo half that matters into the normal range; the kernels multiply
// their fp32 accumulators by winv = 2^-s (exact) in the fused-multiply-add that adds the bias.
float pow2_scale(float maxabs) {
    if (!(maxabs > 0.f) || !std::isfinite(maxabs)) return 1.f;
    int e = 0;
    std::frexp(maxabs, &e);            // maxabs = m 2^e, m in [0.5, 1)
    return std::ldexp(1.f, 15 - e);    // maxabs * scale in [2^14, 2^15)
}

int prep_conv3(mg_context* h, int which, const std::string& wkey, const std::string& bkey, int Cin, int Cout, ConvLayer* L) {
    const HostTensor *w, *b;
    if (int rc = need(h, which, wkey, {Cout, Cin, 3, 3}, &w)) return rc;
    if (int rc = need(h, which, bkey, {Cout}, &b)) return rc;
    L->Cin = Cin;
    L->Cout = Cout;
    L->CoutPad = (Cout + 63) / 64 * 64;
    std::vector<float> t((size_t)(Cin / 8) * 9 * L->CoutPad * 8, 0.f);
    for (int n = 0; n < Cout; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int tap = 0; tap < 9; ++tap)
                t[(((size_t)(c >> 3) * 9 + tap) * L->CoutPad + n) * 8 + (c & 7)] = w->data[((size_t)n * Cin + c) * 9 + tap];
    if (int rc = upload(h, t, &L->W)) return rc;
    // SH8 split-fp16 copy: [Cin/16][9][CoutPad][ (8 hi | 8 lo) x 2 groups ]
    std::vector<float> sh((size_t)(Cin / 16) * 9 * L->CoutPad * 16, 0.f);
    _Float16* hp = reinterpret_cast<_Float16*>(sh.data());
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)Cout * Cin * 9; ++i) wmax = std::max(wmax, std::fabs(w->data[i]));
    const float wsc = pow2_scale(wmax);
    L->winv = 1.f / wsc;
    for (int n = 0; n < Cout; ++n)
        for (int c = 0; c < Cin; ++c)
            for (int tap = 0; tap < 9; ++tap) {
                const float v = w->data[((size_t)n * Cin + c) * 9 + tap] * wsc;
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const size_t row = ((size_t)(c >> 4) * 9 + tap) * L->CoutPad + n;  // 64-byte row = 32 halves
                const int grp = (c >> 3) & 1, j = c & 7;
                hp[row * 32 + grp * 16 + j] = hi;
                hp[row * 32 + grp * 16 + 8 + j] = lo;
            }
    if (int rc = upload(h, sh, &L->Wh)) return rc;
    return upload(h, b->data, &L->bias);
}

// [N][K] -> [K/8][Npad][8]; `perm` optionally maps new row -> old row
int prep_linear_rows(mg_context* h, const std::vector<const float*>& rows, const std::vector<float>& bias, int K, LinLayer* L, bool kperm = false) {
    const int N = (int)rows.size();
    L->N = N;
    L->K = K;
    L->Npad = (N + 255) / 256 * 256;
    std::vector<float> t((size_t)(K / 8) * L->Npad * 8, 0.f);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) t[((size_t)(k >> 3) * L->Npad + n) * 8 + (k & 7)] = rows[n][k];
    if (int rc = upload(h, t, &L->W)) return rc;
    if (K % 16 == 0) {
        std::vector<float> sh((size_t)(K / 16) * L->Npad * 16, 0.f);
        _Float16* hp = reinterpret_cast<_Float16*>(sh.data());
        float wmax = 0.f;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) wmax = std::max(wmax, std::fabs(rows[n][k]));
        const float wsc = pow2_scale(wmax);
        L->winv = 1.f / wsc;
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                const float v = rows[n][k] * wsc;
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)(v - (float)hi);
                const size_t row = (size_t)(k >> 4) * L->Npad + n;
                const int grp = (k >> 3) & 1, j = k & 7;
                hp[row * 32 + grp * 16 + j] = hi;
                hp[row * 32 + grp * 16 + 8 + j] = lo;
            }
        if (int rc = upload(h, sh, &L->Wh)) return rc;
        if (kperm && K % 512 == 0) {
            // ffn_f16.hip phase 3: k-step j of the contraction is quarter j/8 of wave j%8, and position 8h + 4a + e inside it is hidden
            // feature 64 (j%8) + 16 (j/8) + 8a + 4h + e -- the order in which the lanes' accumulator registers present the hidden row
            std::fill(sh.begin(), sh.end(), 0.f);
            for (int n = 0; n < N; ++n)
                for (int kp = 0; kp < K; ++kp) {
                    const int blk = kp / 512, kj = (kp % 512) >> 4, pp = kp & 15;
                    const int ks = blk * 512 + 64 * (kj & 7) + 16 * (kj >> 3) + ((pp & 8) >> 1) + ((pp & 4) << 1) + (pp & 3);
                    const float v = rows[n][ks] * wsc;
                    const _Float16 hi = (_Float16)v;
                    const _Float16 lo = (_Float16)(v - (float)hi);
                    const size_t row = (size_t)(kp >> 4) * L->Npad + n;
                    const int grp = (kp >> 3) & 1, j = kp & 7;
                    hp[row * 32 + grp * 16 + j] = hi;
                    hp[row * 32 + grp * 16 + 8 + j] = lo;
                }
            if (int rc = upload(h, sh, &L->Whp)) return rc;
        }
    }
    if (!bias.empty()) {
        std::vector<float> bp(L->Npad, 0.f);  // padded: a launch may round N up to a multiple of 32 (zero weights, zero bias)
        std::copy(bias.begin(), bias.end(), bp.begin());
        return upload(h, bp, &L->bias);
    }
    return 0;
}

int prep_linear(mg_context* h, int which, const std::string& prefix, int N, int K, LinLayer* L, bool kperm = false) {
    const HostTensor *w, *b;
    if (int rc = need(h, which, prefix + ".weight", {N, K}, &w)) return rc;
    if (int rc = need(h, which, prefix + ".bias", {N}, &b)) return rc;
    std::vector<const float*> rows(N);
    for (int n = 0; n < N; ++n) rows[n] = w->data.data() + (size_t)n * K;
    return prep_linear_rows(h, rows, b->data, K, L, kperm);
}

// BatchNorm (eval) -> per-channel affine after the activation
int prep_bn(mg_context* h, const std::string& prefix, int C, float** ps, float** pt) {
    const HostTensor *g, *b, *m, *v;
    if (int rc = need(h, 0, prefix + ".weight", {C}, &g)) return rc;
    if (int rc = need(h, 0, prefix + ".bias", {C}, &b)) return rc;
    if (int rc = need(h, 0, prefix + ".running_mean", {C}, &m)) return rc;
    if (int rc = need(h, 0, prefix + ".running_var", {C}, &v)) return rc;
    std::vector<float> s(C), t(C);
    for (int c = 0; c < C; ++c) {
        const float invstd = 1.0f / std::sqrt(v->data[c] + 1e-3f);
        const float alpha = invstd * g->data[c];
        s[c] = alpha;
        t[c] = b->data[c] - m->data[c] * alpha;
    }
    if (int rc = upload(h, s, ps)) return rc;
    return upload(h, t, pt);
}

int prepare_sp(mg_context* h, int variant) {
    if (h->sp_variant == variant) return 0;
    h->sp_variant = -1;
    if (int rc = begin_prepare(h, 0)) return rc;
    const int cin[9] = {64, 64, 64, 64, 128, 128, 128, 128, 128};
    const int cout[9] = {64, 64, 64, 128, 128, 128, 128, 256, 256};
    if (variant == MG_SP_MAGICLEAP) {
        const char* names[9] = {"conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa"};
        const HostTensor *w, *b;
        if (int rc = need(h, 0, "conv1a.weight", {64, 1, 3, 3}, &w)) return rc;
        if (int rc = need(h, 0, "conv1a.bias", {64}, &b)) return rc;
        if (int rc = upload(h, w->data, &h->c1a_w)) return rc;
        if (int rc = upload(h, b->data, &h->c1a_b)) return rc;
        h->c1a_ps = h->c1a_pt = nullptr;
        for (int i = 0; i < 9; ++i) {
            h->conv[i] = ConvLayer();
            if (int rc = prep_conv3(h, 0, std::string(names[i]) + ".weight", std::string(names[i]) + ".bias", cin[i], cout[i], &h->conv[i])) return rc;
        }
        h->convPb = LinLayer();
        h->convDb = LinLayer();
        if (int rc = prep_linear(h, 0, "convPb", 65, 256, &h->convPb)) return rc;
        if (int rc = prep_linear(h, 0, "convDb", 256, 256, &h->convDb)) return rc;
    } else if (variant == MG_SP_OPEN) {
        const char* names[9] = {"backbone.0.1", "backbone.1.0", "backbone.1.1", "backbone.2.0", "backbone.2.1",
                                "backbone.3.0", "backbone.3.1", "detector.0", "descriptor.0"};
        const HostTensor *w, *b;
        if (int rc = need(h, 0, "backbone.0.0.conv.weight", {64, 1, 3, 3}, &w)) return rc;
        if (int rc = need(h, 0, "backbone.0.0.conv.bias", {64}, &b)) return rc;
        if (int rc = upload(h, w->data, &h->c1a_w)) return rc;
        if (int rc = upload(h, b->data, &h->c1a_b)) return rc;
        if (int rc = prep_bn(h, "backbone.0.0.bn", 64, &h->c1a_ps, &h->c1a_pt)) return rc;
        for (int i = 0; i < 9; ++i) {
            h->conv[i] = ConvLayer();
            const std::string n = names[i];
            if (int rc = prep_conv3(h, 0, n + ".conv.weight", n + ".conv.bias", cin[i], cout[i], &h->conv[i])) return rc;
            if (int rc = prep_bn(h, n + ".bn", cout[i], &h->conv[i].ps, &h->conv[i].pt)) return rc;
        }
        h->convPb = LinLayer();
        h->convDb = LinLayer();
        if (int rc = prep_linear(h, 0, "detector.1.conv", 65, 256, &h->convPb)) return rc;
        if (int rc = prep_bn(h, "detector.1.bn", 65, &h->convPb.ps, &h->convPb.pt)) return rc;
        if (int rc = prep_linear(h, 0, "descriptor.1.conv", 256, 256, &h->convDb)) return rc;
        if (int rc = prep_bn(h, "descriptor.1.bn", 256, &h->convDb.ps, &h->convDb.pt)) return rc;
    } else {
        return fail(MG_ERR_ARG, "unknown SuperPoint variant");
    }
    h->sp_variant = variant;
    return 0;
}

// ffn.0 consumes cat[x, message] with message = out(ctx) (lightglue.py:156-157, :213-214); the message
// is used nowhere else, so the out-projection is folded into the second half of ffn.0 at upload:
//   ffn0(cat[x, ctx Wo^T + bo]) = x W0a^T + ctx (W0b Wo)^T + (b0 + W0b bo)      (products in fp64)
// which removes one GEMM + one (tokens,256) round trip per block.  `ffn0f` is that folded layer.
int prep_ffn0_folded(mg_context* h, const std::string& p, const std::string& outname, LinLayer* L) {
    const HostTensor *w0, *b0, *wo, *bo;
    if (int rc = need(h, 1, p + ".ffn.0.weight", {512, 512}, &w0)) return rc;
    if (int rc = need(h, 1, p + ".ffn.0.bias", {512}, &b0)) return rc;
    if (int rc = need(h, 1, p + "." + outname + ".weight", {256, 256}, &wo)) return rc;
    if (int rc = need(h, 1, p + "." + outname + ".bias", {256}, &bo)) return rc;
    std::vector<float> W((size_t)512 * 512), bias(512);
    for (int n = 0; n < 512; ++n) {
        const float* w0n = w0->data.data() + (size_t)n * 512;
        for (int k = 0; k < 256; ++k) W[(size_t)n * 512 + k] = w0n[k];
        for (int k = 0; k < 256; ++k) {
            double acc = 0.0;
            for (int j = 0; j < 256; ++j) acc += (double)w0n[256 + j] * (double)wo->data[(size_t)j * 256 + k];
            W[(size_t)n * 512 + 256 + k] = (float)acc;
        }
        double bb = b0->data[n];
        for (int j = 0; j < 256; ++j) bb += (double)w0n[256 + j] * (double)bo->data[j];
        bias[n] = (float)bb;
    }
    std::vector<const float*> rows(512);
    for (int n = 0; n < 512; ++n) rows[n] = W.data() + (size_t)n * 512;
    return prep_linear_rows(h, rows, bias, 512, L);
}

int prep_ffn(mg_context* h, const std::string& p, LgBlock* blk) {
    if (int rc = prep_linear(h, 1, p + ".ffn.0", 512, 512, &blk->ffn0)) return rc;
    if (int rc = prep_linear(h, 1, p + ".ffn.3", 256, 512, &blk->ffn3, true)) return rc;
    const HostTensor *g, *b;
    if (int rc = need(h, 1, p + ".ffn.1.weight", {512}, &g)) return rc;
    if (int rc = need(h, 1, p + ".ffn.1.bias", {512}, &b)) return rc;
    if (int rc = upload(h, g->data, &blk->ln_g)) return rc;
    return upload(h, b->data, &blk->ln_b);
}

int prepare_lg(mg_context* h, int n_layers) {
    if (h->lg_layers == n_layers) return 0;
    h->lg_layers = -1;
    h->lg_heads = -1;
    if (int rc = begin_prepare(h, 1)) return rc;
    h->lg_self.assign(n_layers, LgBlock());
    h->lg_cross.assign(n_layers, LgBlock());
    for (int i = 0; i < n_layers; ++i) {
        const std::string ps = "transformers." + std::to_string(i) + ".self_attn";
        const std::string pc = "transformers." + std::to_string(i) + ".cross_attn";
        // Wqkv: reference feature f = head*192 + dim*3 + {q,k,v} (lightglue.py:151-152); re-ordered here to
        // [q | k | v] x [head][dim] so rotary pairs are adjacent columns and attention reads 64-wide head slices.
        const HostTensor *w, *b;
        if (int rc = need(h, 1, ps + ".Wqkv.weight", {768, 256}, &w)) return rc;
        if (int rc = need(h, 1, ps + ".Wqkv.bias", {768}, &b)) return rc;
        std::vector<const float*> rows(768);
        std::vector<float> bias(768);
        for (int which = 0; which < 3; ++which)
            for (int hd = 0; hd < 4; ++hd)
                for (int d = 0; d < 64; ++d) {
                    const int nn = which * 256 + hd * 64 + d, old = hd * 192 + d * 3 + which;
                    rows[nn] = w->data.data() + (size_t)old * 256;
                    bias[nn] = b->data[old];
                }
        if (int rc = prep_linear_rows(h, rows, bias, 256, &h->lg_self[i].proj)) return rc;
        if (int rc = prep_linear(h, 1, ps + ".out_proj", 256, 256, &h->lg_self[i].out)) return rc;
        if (int rc = prep_ffn(h, ps, &h->lg_self[i])) return rc;
        if (int rc = prep_ffn0_folded(h, ps, "out_proj", &h->lg_self[i].ffn0f)) return rc;
        // cross: one GEMM for [to_qk ; to_v]
        const HostTensor *wq, *bq, *wv, *bv;
        if (int rc = need(h, 1, pc + ".to_qk.weight", {256, 256}, &wq)) return rc;
        if (int rc = need(h, 1, pc + ".to_qk.bias", {256}, &bq)) return rc;
        if (int rc = need(h, 1, pc + ".to_v.weight", {256, 256}, &wv)) return rc;
        if (int rc = need(h, 1, pc + ".to_v.bias", {256}, &bv)) return rc;
        std::vector<const float*> rows2(512);
        std::vector<float> bias2(512);
        for (int n = 0; n < 256; ++n) {
            rows2[n] = wq->data.data() + (size_t)n * 256;
            bias2[n] = bq->data[n];
            rows2[256 + n] = wv->data.data() + (size_t)n * 256;
            bias2[256 + n] = bv->data[n];
        }
        if (int rc = prep_linear_rows(h, rows2, bias2, 256, &h->lg_cross[i].proj)) return rc;
        if (int rc = prep_linear(h, 1, pc + ".to_out", 256, 256, &h->lg_cross[i].out)) return rc;
        if (int rc = prep_ffn(h, pc, &h->lg_cross[i])) return rc;
        if (int rc = prep_ffn0_folded(h, pc, "to_out", &h->lg_cross[i].ffn0f)) return rc;
    }
    const std::string pa = "log_assignment." + std::to_string(n_layers - 1);
    if (int rc = prep_linear(h, 1, pa + ".final_proj", 256, 256, &h->final_proj)) return rc;
    const HostTensor *mw, *mb, *wr;
    if (int rc = need(h, 1, pa + ".matchability.weight", {1, 256}, &mw)) return rc;
    if (int rc = need(h, 1, pa + ".matchability.bias", {1}, &mb)) return rc;
    if (int rc = upload(h, mw->data, &h->match_w)) return rc;
    h->match_b = mb->data[0];
    if (int rc = need(h, 1, "posenc.Wr.weight", {32, 2}, &wr)) return rc;
    if (int rc = upload(h, wr->data, &h->Wr)) return rc;
    h->lg_layers = n_layers;
    return 0;
}

// heads used only by the adaptive path: log_assignment.{i} for every layer (the loop may stop at any of them, lightglue.py:495-496)
// and token_confidence.{i} for i < n_layers - 1 (:352-354)
int prepare_lg_heads(mg_context* h, int n_layers, bool need_tokens) {
    if (int rc = prepare_lg(h, n_layers)) return rc;
    if (h->lg_heads == n_layers + (need_tokens ? 1000 : 0)) return 0;
    h->cur_which = 1;
    h->lg_final.assign(n_layers, LinLayer());
    h->lg_match_w.assign(n_layers, nullptr);
    h->lg_match_b.assign(n_layers, 0.f);
    h->lg_tok_w.assign(n_layers, nullptr);
    h->lg_tok_b.assign(n_layers, 0.f);
    for (int i = 0; i < n_layers; ++i) {
        const std::string pa = "log_assignment." + std::to_string(i);
        if (int rc = prep_linear(h, 1, pa + ".final_proj", 256, 256, &h->lg_final[i])) return rc;
        const HostTensor *mw, *mb;
        if (int rc = need(h, 1, pa + ".matchability.weight", {1, 256}, &mw)) return rc;
        if (int rc = need(h, 1, pa + ".matchability.bias", {1}, &mb)) return rc;
        if (int rc = upload(h, mw->data, &h->lg_match_w[i])) return rc;
        h->lg_match_b[i] = mb->data[0];
        if (need_tokens && i < n_layers - 1) {
            const std::string pt = "token_confidence." + std::to_string(i) + ".token.0";
            const HostTensor *tw, *tb;
            if (int rc = need(h, 1, pt + ".weight", {1, 256}, &tw)) return rc;
            if (int rc = need(h, 1, pt + ".bias", {1}, &tb)) return rc;
            if (int rc = upload(h, tw->data, &h->lg_tok_w[i])) return rc;
            h->lg_tok_b[i] = tb->data[0];
        }
    }
    h->lg_heads = n_layers + (need_tokens ? 1000 : 0);
    return 0;
}

// ---- SuperGlue weights (gluefactory_nonfree/superglue.py:63-72,118-141,208-250) --------------------
// Conv1d(k=1) == Linear.  BatchNorm1d (eval) after a conv is folded into that conv in fp64; the
// attention heads use channel = dim*4 + head (:123-128), re-ordered here to [head][dim]; `merge` is
// folded into mlp.0 exactly like LightGlue's out-projection (the message only feeds the MLP, :139-141).
struct BnFold { std::vector<double> s, t; };

int sg_bn(mg_context* h, int which, const std::string& prefix, int C, BnFold* out) {
    const HostTensor *g, *b, *m, *v;
    if (int rc = need(h, which, prefix + ".weight", {C}, &g)) return rc;
    if (int rc = need(h, which, prefix + ".bias", {C}, &b)) return rc;
    if (int rc = need(h, which, prefix + ".running_mean", {C}, &m)) return rc;
    if (int rc = need(h, which, prefix + ".running_var", {C}, &v)) return rc;
    out->s.resize(C);
    out->t.resize(C);
    for (int c = 0; c < C; ++c) {
        const double s = (double)g->data[c] / std::sqrt((double)v->data[c] + 1e-5);
        out->s[c] = s;
        out->t[c] = (double)b->data[c] - (double)m->data[c] * s;
    }
    return 0;
}

int sg_linear(mg_context* h, const std::vector<std::vector<double>>& W, const std::vector<double>& bias, int Kpad, LinLayer* L, bool kperm = false) {
    const int N = (int)W.size();
    std::vector<float> flat((size_t)N * Kpad, 0.f), bf(N);
    std::vector<const float*> rows(N);
    for (int n = 0; n < N; ++n) {
        for (size_t k = 0; k < W[n].size(); ++k) flat[(size_t)n * Kpad + k] = (float)W[n][k];
        rows[n] = flat.data() + (size_t)n * Kpad;
        bf[n] = (float)bias[n];
    }
    return prep_linear_rows(h, rows, bf, Kpad, L, kperm);
}

// Conv1d(k=1) `name` (N x K) with an optional BatchNorm1d `bn_name` folded in; K padded up to Kpad
int sg_prep_conv(mg_context* h, int which, const std::string& name, int N, int K, const std::string& bn_name, int Kpad, LinLayer* out, bool kperm = false) {
    const HostTensor *w, *b;
    if (int rc = need(h, which, name + ".weight", {N, K, 1}, &w)) return rc;
    if (int rc = need(h, which, name + ".bias", {N}, &b)) return rc;
    BnFold bn;
    const bool has_bn = !bn_name.empty();
    if (has_bn)
        if (int rc = sg_bn(h, which, bn_name, N, &bn)) return rc;
    std::vector<std::vector<double>> W(N, std::vector<double>(K));
    std::vector<double> bias(N);
    for (int n = 0; n < N; ++n) {
        const double sc = has_bn ? bn.s[n] : 1.0, t = has_bn ? bn.t[n] : 0.0;
        for (int k = 0; k < K; ++k) W[n][k] = (double)w->data[(size_t)n * K + k] * sc;
        bias[n] = (double)b->data[n] * sc + t;
    }
    *out = LinLayer();
    return sg_linear(h, W, bias, Kpad, out, kperm);
}

// MLP([cin, 32, 64, 128, 256, 256]) of the keypoint / endpoint encoders (superglue.py:89-97, gluestick.py:488-517)
int sg_prep_encoder(mg_context* h, int which, const std::string& prefix, int cin, LinLayer* out5) {
    const int chans[6] = {cin, 32, 64, 128, 256, 256};
    for (int i = 0; i < 5; ++i) {
        const std::string cn = prefix + "." + std::to_string(3 * i);
        const std::string bn = i < 4 ? prefix + "." + std::to_string(3 * i + 1) : std::string();
        if (int rc = sg_prep_conv(h, which, cn, chans[i + 1], chans[i], bn, chans[i] < 32 ? 32 : chans[i], &out5[i])) return rc;
    }
    return 0;
}

// one AttentionalPropagation block under prefix `p` (superglue.py:118-141, gluestick.py:520-560)
int sg_prep_layer(mg_context* h, int which, const std::string& p, mg_context::SgLayer* L) {
    // [q ; k ; v] with rows re-ordered to [head][dim]
    std::vector<std::vector<double>> W(768, std::vector<double>(256));
    std::vector<double> bias(768);
    for (int part = 0; part < 3; ++part) {
        const HostTensor *w, *b;
        if (int rc = need(h, which, p + ".attn.proj." + std::to_string(part) + ".weight", {256, 256, 1}, &w)) return rc;
        if (int rc = need(h, which, p + ".attn.proj." + std::to_string(part) + ".bias", {256}, &b)) return rc;
        for (int hd = 0; hd < 4; ++hd)
            for (int d = 0; d < 64; ++d) {
                const int nn = part * 256 + hd * 64 + d, old = d * 4 + hd;
                for (int k = 0; k < 256; ++k) W[nn][k] = w->data[(size_t)old * 256 + k];
                bias[nn] = b->data[old];
            }
    }
    if (int rc = sg_linear(h, W, bias, 256, &L->proj)) return rc;
    // mlp.0 with merge folded in (context columns re-ordered to [head][dim]) and BatchNorm folded on top
    const HostTensor *w0, *b0, *wm, *bm;
    if (int rc = need(h, which, p + ".mlp.0.weight", {512, 512, 1}, &w0)) return rc;
    if (int rc = need(h, which, p + ".mlp.0.bias", {512}, &b0)) return rc;
    if (int rc = need(h, which, p + ".attn.merge.weight", {256, 256, 1}, &wm)) return rc;
    if (int rc = need(h, which, p + ".attn.merge.bias", {256}, &bm)) return rc;
    BnFold bn;
    if (int rc = sg_bn(h, which, p + ".mlp.1", 512, &bn)) return rc;
    std::vector<std::vector<double>> W0(512, std::vector<double>(512));
    std::vector<double> bias0(512);
    for (int n = 0; n < 512; ++n) {
        const float* wn = w0->data.data() + (size_t)n * 512;
        for (int k = 0; k < 256; ++k) W0[n][k] = (double)wn[k] * bn.s[n];
        for (int hd = 0; hd < 4; ++hd)
            for (int d = 0; d < 64; ++d) {
                double acc = 0.0;
                for (int j = 0; j < 256; ++j) acc += (double)wn[256 + j] * (double)wm->data[(size_t)j * 256 + d * 4 + hd];
                W0[n][256 + hd * 64 + d] = acc * bn.s[n];
            }
        double bb = b0->data[n];
        for (int j = 0; j < 256; ++j) bb += (double)wn[256 + j] * (double)bm->data[j];
        bias0[n] = bb * bn.s[n] + bn.t[n];
    }
    if (int rc = sg_linear(h, W0, bias0, 512, &L->mlp0f)) return rc;
    return sg_prep_conv(h, which, p + ".mlp.3", 256, 512, "", 512, &L->mlp3, true);
}

int prepare_sg(mg_context* h, int n_layers) {
    if (h->sg_layers == n_layers) return 0;
    h->sg_layers = -1;
    if (int rc = begin_prepare(h, 2)) return rc;
    if (int rc = sg_prep_encoder(h, 2, "kenc.encoder", 3, h->sg_kenc)) return rc;
    h->sg_gnn.assign(n_layers, mg_context::SgLayer());
    for (int l = 0; l < n_layers; ++l)
        if (int rc = sg_prep_layer(h, 2, "gnn.layers." + std::to_string(l), &h->sg_gnn[l])) return rc;
    if (int rc = sg_prep_conv(h, 2, "final_proj", 256, 256, "", 256, &h->sg_final)) return rc;
    const HostTensor* bs;
    if (int rc = need(h, 2, "bin_score", {1}, &bs)) return rc;
    h->sg_bin = bs->data[0];
    h->sg_layers = n_layers;
    return 0;
}

// GlueStick (gluefactory/models/matchers/gluestick.py:62-100): SuperGlue's blocks under `gnn.layers.{i}.update`, an
// endpoint encoder, one line layer per self layer (MLP([768, 512, 256]) with BatchNorm), two final projections.
int prepare_gs(mg_context* h, int n_layers) {
    if (h->gs_layers == n_layers) return 0;
    h->gs_layers = -1;
    if (int rc = begin_prepare(h, 3)) return rc;
    if (int rc = sg_prep_encoder(h, 3, "kenc.encoder", 3, h->gs_kenc)) return rc;
    if (int rc = sg_prep_encoder(h, 3, "lenc.encoder", 5, h->gs_lenc)) return rc;
    h->gs_gnn.assign(n_layers, mg_context::SgLayer());
    for (int l = 0; l < n_layers; ++l)
        if (int rc = sg_prep_layer(h, 3, "gnn.layers." + std::to_string(l) + ".update", &h->gs_gnn[l])) return rc;
    h->gs_line.assign((n_layers + 1) / 2, mg_context::GsLine());
    for (int l = 0; l < (n_layers + 1) / 2; ++l) {
        const std::string p = "gnn.line_layers." + std::to_string(l);
        if (int rc = sg_prep_conv(h, 3, p + ".mlp.0", 512, 768, p + ".mlp.1", 768, &h->gs_line[l].mlp0)) return rc;
        if (int rc = sg_prep_conv(h, 3, p + ".mlp.3", 256, 512, "", 512, &h->gs_line[l].mlp3)) return rc;
    }
    if (int rc = sg_prep_conv(h, 3, "final_proj", 256, 256, "", 256, &h->gs_final)) return rc;
    if (int rc = sg_prep_conv(h, 3, "final_line_proj", 256, 256, "", 256, &h->gs_final_line)) return rc;
    const HostTensor *bs, *lbs;
    if (int rc = need(h, 3, "bin_score", {1}, &bs)) return rc;
    if (int rc = need(h, 3, "line_bin_score", {1}, &lbs)) return rc;
    h->gs_bin = bs->data[0];
    h->gs_line_bin = lbs->data[0];
    h->gs_layers = n_layers;
    return 0;
}

GemmArgs lin_args(const LinLayer& L, const float* A, int lda, int M, float* C, int ldc) {
    GemmArgs g;
    g.A0 = A;
    g.lda0 = lda;
    g.ksplit = L.K;
    g.W = L.W;
    g.Npad = L.Npad;
    g.bias = L.bias;
    g.post_scale = L.ps;
    g.post_shift = L.pt;
    g.C = C;
    g.ldc = ldc;
    g.M = M;
    g.N = L.N;
    g.K = L.K;
    return g;
}

GemmF16Args lin_args_f16(const LinLayer& L, const void* A, int lda, int M) {
    GemmF16Args g;
    g.A0 = A;
    g.lda0 = lda;
    g.ksplit = L.K;
    g.W = L.Wh;
    g.winv = L.winv;
    g.Npad = L.Npad;
    g.bias = L.bias;
    g.post_scale = L.ps;
    g.post_shift = L.pt;
    g.M = M;
    g.N = L.N;
    g.K = L.K;
    return g;
}

int tap_copy(mg_context* h, const std::string& name, const float* src, size_t count, hipStream_t s) {
    auto it = h->taps.find(name);
    if (it == h->taps.end()) return 0;
    if (it->second.ensure(count * sizeof(float)) != 0) return fail(MG_ERR_HIP, "tap allocation failed");
    HIPCHK(hipMemcpyAsync(it->second.p, src, count * sizeof(float), hipMemcpyDeviceToDevice, s));
    h->tap_count[name] = count;
    return 0;
}

int run_detect(mg_context* h, const mg_sp_conf* conf, const float* heat, int B, int Hs, int Ws, const float* image_size,
               float* keypoints, float* scores, int32_t* num_valid, float* nms_out, hipStream_t s) {
    if (conf->nms_radius < 0 || conf->nms_radius > 4) return fail(MG_ERR_ARG, "nms_radius must be in [0,4]");
    if (conf->max_num_keypoints < 1 || conf->max_num_keypoints > 4096) return fail(MG_ERR_ARG, "max_num_keypoints must be in [1,4096]");
    DetectArgs d;
    d.heat = heat;
    d.B = B;
    d.Hs = Hs;
    d.Ws = Ws;
    d.radius = conf->nms_radius;
    d.border = conf->remove_borders;
    d.K = conf->max_num_keypoints;
    d.threshold = conf->detection_threshold;
    d.image_size = (conf->variant == MG_SP_MAGICLEAP) ? image_size : nullptr;
    d.cand = reinterpret_cast<unsigned long long*>(h->buf("sp.cand", (size_t)B * Hs * Ws * 2));
    d.counts = reinterpret_cast<int*>(h->buf("sp.counts", (size_t)B * 32));
    if (!d.cand || !d.counts) return fail(MG_ERR_HIP, "workspace allocation failed");
    d.nms_out = nms_out;
    d.keypoints = keypoints;
    d.scores = scores;
    d.num_valid = num_valid;
    ProfScope ps(h, s, "sp.detect");
    launch_detect(d, s);
    return 0;
}

}  // namespace

// ================================================================================================
extern "C" {

int mg_abi_version(void) { return MG_ABI_VERSION; }
unsigned long long mg_workspace_generation(void) { return g_ws_generation.load(std::memory_order_relaxed); }

const char* mg_last_error(void) { return g_err.c_str(); }

int mg_create(int device, mg_handle* out) {
    if (!out) return fail(MG_ERR_ARG, "out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(MG_ERR_ARG, "no such HIP device");
    HIPCHK(hipSetDevice(device));
    mg_context* h = new mg_context();
    h->device = device;
    *out = h;
    return 0;
}

int mg_destroy(mg_handle h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (auto& v : h->owned)
        for (void* p : v) (void)hipFree(p);
    for (auto& kv : h->ws) kv.second.release();
    for (auto& kv : h->taps) kv.second.release();
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

int mg_set_precision(mg_handle h, int mode) {
    if (!h || (mode != MG_PREC_F32 && mode != MG_PREC_F16X3)) return fail(MG_ERR_ARG, "bad precision mode");
    if (h->precision != mode) g_ws_generation.fetch_add(1, std::memory_order_relaxed);  // other kernels, other buffers
    h->precision = mode;
    return 0;
}

int mg_set_weight(mg_handle h, int which, const char* key, const float* data, const int64_t* shape, int ndim) {
    if (!h || !key || !data || which < 0 || which > 3 || ndim < 0 || (ndim > 0 && !shape)) return fail(MG_ERR_ARG, "bad argument");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data, data + t.numel());
    h->raw[which][key] = std::move(t);
    if (which == 0) h->sp_variant = -1; else if (which == 1) h->lg_layers = -1; else if (which == 2) h->sg_layers = -1; else h->gs_layers = -1;  // re-tile on next use
    return 0;
}

int mg_weights_ready(mg_handle h, int which, int variant_or_layers) {
    if (!h) return fail(MG_ERR_ARG, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    if (which == 0) return prepare_sp(h, variant_or_layers);
    if (which == 2) return prepare_sg(h, variant_or_layers > 0 ? variant_or_layers : 18);
    if (which == 3) return prepare_gs(h, variant_or_layers > 0 ? variant_or_layers : 18);
    return prepare_lg(h, variant_or_layers > 0 ? variant_or_layers : 9);
}

int mg_debug_request(mg_handle h, const char* tap) {
    if (!h || !tap) return fail(MG_ERR_ARG, "bad argument");
    if (h->taps.find(tap) == h->taps.end()) g_ws_generation.fetch_add(1, std::memory_order_relaxed);  // requests select kernel variants / add taps
    h->taps[tap];
    return 0;
}

int mg_debug_read(mg_handle h, const char* tap, float* dst, size_t capacity, size_t* count, void* stream) {
    if (!h || !tap || !dst) return fail(MG_ERR_ARG, "bad argument");
    auto it = h->tap_count.find(tap);
    if (it == h->tap_count.end()) return fail(MG_ERR_STATE, std::string("tap '") + tap + "' was not captured");
    if (it->second > capacity) return fail(MG_ERR_ARG, "destination too small");
    HIPCHK(hipMemcpyAsync(dst, h->taps[tap].p, it->second * sizeof(float), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    if (count) *count = it->second;
    return 0;
}

int mg_profile_enable(mg_handle h, int on) {
    if (!h) return fail(MG_ERR_ARG, "handle is NULL");
    h->profile = on != 0;
    h->prof.clear();
    h->event_used = 0;
    return 0;
}

int mg_profile_filter(mg_handle h, const char* name) {
    if (!h) return fail(MG_ERR_ARG, "handle is NULL");
    h->prof_filter = name ? name : "";
    return 0;
}

int mg_profile_read(mg_handle h, char (*names)[48], float* ms, int cap) {
    if (!h) return fail(MG_ERR_ARG, "handle is NULL");
    int n = 0;
    for (auto& e : h->prof) {
        if (n >= cap) break;
        HIPCHK(hipEventSynchronize(e.b));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, e.a, e.b));
        std::snprintf(names[n], 48, "%s", e.name.c_str());
        ms[n] = t;
        ++n;
    }
    const int total = (int)h->prof.size();
    h->prof.clear();
    h->event_used = 0;
    return total < cap ? total : n;
}

// ------------------------------------------------------------------------------------------------
int mg_sp_detect(mg_handle h, const mg_sp_conf* conf, const float* heatmap, int B, int Hs, int Ws, const float* image_size,
                 float* keypoints, float* scores, int32_t* num_valid, float* nms_out, void* stream) {
    if (!h || !conf || !heatmap || !keypoints || !scores || !num_valid || B < 1 || Hs < 1 || Ws < 1) return fail(MG_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    if (int rc = run_detect(h, conf, heatmap, B, Hs, Ws, image_size, keypoints, scores, num_valid, nms_out, static_cast<hipStream_t>(stream))) return rc;
    HIPCHK(hipGetLastError());
    return 0;
}

int mg_sp_sample(mg_handle h, const mg_sp_conf* conf, const float* keypoints, const float* dense_nchw, int B, int K, int Hc, int Wc,
                 float* descriptors, void* stream) {
    if (!h || !conf || !keypoints || !dense_nchw || !descriptors || B < 1 || K < 1) return fail(MG_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* nhwc = h->buf("sp.sample_nhwc", (size_t)B * Hc * Wc * 256);
    if (!nhwc) return fail(MG_ERR_HIP, "workspace allocation failed");
    launch_nchw_to_nhwc(dense_nchw, nhwc, B, Hc * Wc, 256, s);
    SampleArgs a;
    a.keypoints = keypoints;
    a.dense = nhwc;
    a.B = B;
    a.K = K;
    a.hc = Hc;
    a.wc = Wc;
    a.mode = conf->sampling;
    a.desc = descriptors;
    launch_sample(a, s);
    HIPCHK(hipGetLastError());
    return 0;
}

int mg_sp_forward(mg_handle h, const mg_sp_conf* conf, const float* image, int B, int C, int H, int W, const float* image_size,
                  float* keypoints, float* scores, float* descriptors, int32_t* num_valid, float* heatmap, float* dense_desc,
                  void* stream) {
    if (!h || !conf || !image || !keypoints || !scores || !descriptors || !num_valid) return fail(MG_ERR_ARG, "NULL argument");
    if (B < 1 || (C != 1 && C != 3) || H < 16 || W < 16) return fail(MG_ERR_ARG, "unsupported image shape");
    HIPCHK(hipSetDevice(h->device));
    if (int rc = prepare_sp(h, conf->variant)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);

    const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, hc = H4 / 2, wc = W4 / 2;
    const size_t n1 = (size_t)B * H * W, n2 = (size_t)B * H2 * W2, n4 = (size_t)B * H4 * W4, n8 = (size_t)B * hc * wc;
    const bool want_a1 = h->precision != MG_PREC_F16X3 || h->taps.count("sp.conv1a");
    float* a1 = h->buf("sp.a1", want_a1 ? n1 * 64 : 64);
    float* p1 = h->buf("sp.p1", n2 * 64);
    float* a2 = h->buf("sp.a2", n2 * 64);
    float* p2 = h->buf("sp.p2", n4 * 64);
    float* a3 = h->buf("sp.a3", n4 * 128);
    float* p3 = h->buf("sp.p3", n8 * 128);
    float* a4 = h->buf("sp.a4", n8 * 128);
    float* f4 = h->buf("sp.f4", n8 * 128);
    float* cP = h->buf("sp.cP", n8 * 256);
    float* cD = h->buf("sp.cD", n8 * 256);
    // detector logits: 65 channels; the f16x3 GEMM writes rows of 96 (N rounded up to its 32-feature blocks, zero
    // weights / bias in the padding) so that it stays on the straight-line epilogue instead of the ragged-N path
    const int ldl = (h->precision == MG_PREC_F16X3 && !h->convPb.ps) ? 96 : 65;
    float* lg = h->buf("sp.logits", n8 * 96);
    float* dd = h->buf("sp.dense", n8 * 256);
    float* heat = h->buf("sp.heat", n8 * 64);
    if (!a1 || !p1 || !a2 || !p2 || !a3 || !p3 || !a4 || !f4 || !cP || !cD || !lg || !dd || !heat) return fail(MG_ERR_HIP, "workspace allocation failed");

    const bool f16 = h->precision == MG_PREC_F16X3;
    // f16x3: conv1a is recomputed inside conv1b's tile loop (the 64-channel full-resolution map never
    // reaches HBM) unless a test asked to see it.
    const bool fuse1 = f16 && !h->taps.count("sp.conv1a");
    Conv1aArgs c1;
    c1.image = image; c1.B = B; c1.C = C; c1.H = H; c1.W = W;
    c1.w = h->c1a_w; c1.bias = h->c1a_b; c1.post_scale = h->c1a_ps; c1.post_shift = h->c1a_pt; c1.out = a1;
    if (!fuse1) {
        ProfScope ps(h, s, "sp.conv1a");
        if (f16) launch_conv1a_sh8(c1, s); else launch_conv1a(c1, s);
    }
    if (int rc = tap_copy(h, "sp.conv1a", a1, n1 * 64, s)) return rc;
    struct Step { const char* name; int layer; const float* in; float* out; int Hh, Ww, pool; };
    const Step steps[9] = {
        {"sp.conv1b", 0, a1, p1, H, W, 1},   {"sp.conv2a", 1, p1, a2, H2, W2, 0}, {"sp.conv2b", 2, a2, p2, H2, W2, 1},
        {"sp.conv3a", 3, p2, a3, H4, W4, 0}, {"sp.conv3b", 4, a3, p3, H4, W4, 1}, {"sp.conv4a", 5, p3, a4, hc, wc, 0},
        {"sp.conv4b", 6, a4, f4, hc, wc, 0}, {"sp.convPa", 7, f4, cP, hc, wc, 0}, {"sp.convDa", 8, f4, cD, hc, wc, 0}};
    for (const Step& st : steps) {
        ProfScope ps(h, s, (fuse1 && st.layer == 0) ? "sp.conv1ab" : st.name);
        const ConvLayer& L = h->conv[st.layer];
        if (fuse1 && st.layer == 0) {
            ConvF16Args a;
            a.Wt = L.Wh; a.winv = L.winv; a.CoutPad = L.CoutPad; a.bias = L.bias; a.post_scale = L.ps; a.post_shift = L.pt;
            a.out = st.out; a.B = B; a.H = st.Hh; a.W = st.Ww; a.Cin = L.Cin; a.Cout = L.Cout; a.pool = 1; a.relu = 1;
            a.staged = (h->taps.count("sp.conv_staged") || h->taps.count("sp.conv1ab_staged")) ? 1 : 0;
            launch_conv1ab_f16x3(a, c1, s);
        } else if (f16) {
            ConvF16Args a;
            a.in = st.in; a.Wt = L.Wh; a.winv = L.winv; a.CoutPad = L.CoutPad; a.bias = L.bias; a.post_scale = L.ps; a.post_shift = L.pt;
            a.out = st.out; a.B = B; a.H = st.Hh; a.W = st.Ww; a.Cin = L.Cin; a.Cout = L.Cout; a.pool = st.pool; a.relu = 1;
            a.out_f32 = 0;
            a.staged = h->taps.count("sp.conv_staged") ? 1 : h->taps.count("sp.conv_dma32") ? 2 : 0;
            launch_conv3x3_f16x3(a, s);
        } else {
            ConvArgs a;
            a.in = st.in; a.Wt = L.W; a.CoutPad = L.CoutPad; a.bias = L.bias; a.post_scale = L.ps; a.post_shift = L.pt;
            a.out = st.out; a.B = B; a.H = st.Hh; a.W = st.Ww; a.Cin = L.Cin; a.Cout = L.Cout; a.pool = st.pool; a.relu = 1;
            launch_conv3x3(a, s);
        }
        if (!h->taps.empty()) {
            const int Ho = st.pool ? st.Hh / 2 : st.Hh, Wo = st.pool ? st.Ww / 2 : st.Ww;
            if (int rc = tap_copy(h, std::string(st.name), st.out, (size_t)B * Ho * Wo * L.Cout, s)) return rc;
        }
        continue;
    }
    if (f16 && !h->taps.count("sp.logits")) {  // (a request for the logits tap selects the three-launch form)
        ProfScope ps(h, s, "sp.heads");  // convPb + soft-max + depth-to-space (heads_f16.hip)
        HeadsF16Args a;
        a.in = cP; a.W = h->convPb.Wh; a.winv = h->convPb.winv; a.Npad = h->convPb.Npad; a.bias = h->convPb.bias; a.post_scale = h->convPb.ps; a.post_shift = h->convPb.pt;
        a.heat = heat; a.B = B; a.hc = hc; a.wc = wc;
        launch_heads_f16x3(a, s);
    } else {
        {
            ProfScope ps(h, s, "sp.convPb");
            if (f16) {
                GemmF16Args g = lin_args_f16(h->convPb, cP, 256, (int)n8);
                g.C = lg; g.ldc = ldl; g.N = ldl;
                launch_gemm_f16x3(g, s);
            } else {
                GemmArgs g = lin_args(h->convPb, cP, 256, (int)n8, lg, 65);
                launch_gemm(g, s);
            }
        }
        {
            ProfScope ps(h, s, "sp.softmax_d2s");
            launch_softmax_d2s(lg, ldl, B, hc, wc, heat, s);
        }
    }
    {
        ProfScope ps(h, s, "sp.convDb");
        if (f16) {
            GemmF16Args g = lin_args_f16(h->convDb, cD, 256, (int)n8);
            g.C = dd; g.ldc = 256;
            g.l2norm = h->convDb.ps ? 0 : 1;  // descriptor head + F.normalize in one epilogue (the open variant's BatchNorm affine comes first: separate pass)
            launch_gemm_f16x3(g, s);
        } else {
            GemmArgs g = lin_args(h->convDb, cD, 256, (int)n8, dd, 256);
            launch_gemm(g, s);
        }
    }
    // the dense map is L2-normalised in place only when somebody wants it; otherwise the sampler normalises the four
    // cells it reads (same arithmetic) and the 2 x 315 MB sweep over the map disappears
    const bool want_dense = dense_desc || h->taps.count("sp.dense");
    const bool dense_normalised = f16 && !h->convDb.ps;  // fused into convDb's epilogue (gemm_f16.hip: l2norm)
    if (want_dense && !dense_normalised) {
        ProfScope ps(h, s, "sp.l2norm");
        launch_l2norm_rows(dd, (long long)n8, 256, s);
    }
    if (int rc = run_detect(h, conf, heat, B, hc * 8, wc * 8, image_size, keypoints, scores, num_valid, nullptr, s)) return rc;
    {
        ProfScope ps(h, s, "sp.sample");
        SampleArgs a;
        a.keypoints = keypoints; a.dense = dd; a.B = B; a.K = conf->max_num_keypoints; a.hc = hc; a.wc = wc;
        a.mode = conf->sampling; a.desc = descriptors; a.normalize_cells = (want_dense || dense_normalised) ? 0 : 1;
        launch_sample(a, s);
    }
    if (heatmap) HIPCHK(hipMemcpyAsync(heatmap, heat, n8 * 64 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (dense_desc) launch_nhwc_to_nchw(dd, dense_desc, B, hc * wc, 256, s);
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The LightGlue forward; `ad` != nullptr selects the adaptive depth / width variant (batch 1).
struct LgAdaptive {
    double depth = -1.0, width = -1.0;  // Python floats in the reference: kept unrounded up to the point of comparison
    int32_t* prune0 = nullptr;  // (M) / (N) int32, device
    int32_t* prune1 = nullptr;
    int32_t* out_dims = nullptr;  // host: M', N', layers run
};

static int lg_run(mg_handle h, const mg_lg_conf* conf, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                  const float* size0, const float* size1, int B, int M, int N, int64_t* matches0, int64_t* matches1, float* mscores0,
                  float* mscores1, float* log_assignment, float* ref_desc0, float* ref_desc1, void* stream, const LgAdaptive* ad) {
    if (!h || !conf || !kpts0 || !kpts1 || !desc0 || !desc1 || !matches0 || !matches1 || !mscores0 || !mscores1) return fail(MG_ERR_ARG, "NULL argument");
    if (B < 1 || M < 1 || N < 1) return fail(MG_ERR_ARG, "empty input");
    if (conf->num_heads != 4) return fail(MG_ERR_ARG, "num_heads must be 4 (descriptor_dim 256, head_dim 64)");
    if (conf->n_layers < 1) return fail(MG_ERR_ARG, "n_layers must be >= 1");
    const bool do_stop = ad && ad->depth > 0.0, do_prune = ad && ad->width > 0.0;
    if (ad && B != 1) return fail(MG_ERR_ARG, "adaptive depth / width needs batch 1 (lightglue.py:476,:481 assert it)");
    HIPCHK(hipSetDevice(h->device));
    if (ad) {
        if (int rc = prepare_lg_heads(h, conf->n_layers, do_stop)) return rc;
    } else if (int rc = prepare_lg(h, conf->n_layers)) {
        return rc;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);

    const size_t T0_in = (size_t)B * M, T1_in = (size_t)B * N, Tin = T0_in + T1_in;
    float* X = h->buf("lg.x", Tin * 256);
    float* QKV = h->buf("lg.qkv", Tin * 768);
    float* CTX = h->buf("lg.ctx", Tin * 256);
    float* MSG = h->buf("lg.msg", Tin * 256);
    float* HID = h->buf("lg.hid", Tin * 512);
    float* ENC = h->buf("lg.enc", Tin * 64);
    float* SIM = h->buf("lg.sim", (size_t)B * M * N);
    float* Z = h->buf("lg.z", Tin);
    float* ST = h->buf("lg.stats", 6 * Tin + 16);
    if (!X || !QKV || !CTX || !MSG || !HID || !ENC || !SIM || !Z || !ST) return fail(MG_ERR_HIP, "workspace allocation failed");

    const bool f16 = h->precision == MG_PREC_F16X3;
    float* XS = nullptr;
    if (f16) {  // descriptors -> the residual stream and its SH8 copy in one pass
        XS = h->buf("lg.xs", Tin * 256);
        if (!XS) return fail(MG_ERR_HIP, "workspace allocation failed");
        launch_copy2_to_sh8(desc0, (long long)T0_in * 256, desc1, (long long)T1_in * 256, X, XS, s);
    } else {
        HIPCHK(hipMemcpyAsync(X, desc0, T0_in * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(X + T0_in * 256, desc1, T1_in * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    {
        ProfScope ps(h, s, "lg.posenc");
        launch_posenc(kpts0, kpts1, size0, size1, h->Wr, B, M, N, ENC, s);
    }
    if (int rc = tap_copy(h, "lg.enc", ENC, Tin * 64, s)) return rc;
    // current token counts: only the adaptive variant changes them (point pruning)
    int Mc = M, Nc = N;
    size_t T0 = T0_in, T = Tin;

    // the three-launch form stays reachable for A/B timing (environment) and for the equivalence test (debug tap request)
    const bool unfused_ffn = h->taps.count("lg.unfused_ffn") != 0;
    auto ffn = [&](const LgBlock& blk) {
        if (f16 && !unfused_ffn) {
            ProfScope ps(h, s, "lg.ffn");
            FfnF16Args a;
            a.A0 = XS; a.A1 = CTX; a.W0 = blk.ffn0f.Wh; a.winv0 = blk.ffn0f.winv; a.b0 = blk.ffn0f.bias; a.gamma = blk.ln_g; a.beta = blk.ln_b;
            a.W3 = blk.ffn3.Whp; a.winv3 = blk.ffn3.winv; a.b3 = blk.ffn3.bias; a.X = X; a.XS = XS; a.T = (int)T; a.eps = 1e-5f; a.act = 0;
            auto st = h->taps.find("ffn.stamps");  // diagnostic: per-workgroup phase stamps of the last launch
            if (st != h->taps.end()) {
                const size_t n = (T + 127) / 128 * 8;
                if (st->second.ensure(n * sizeof(float)) == 0) { a.stamps = st->second.f(); h->tap_count["ffn.stamps"] = n; }
            }
            launch_ffn_f16x3(a, s);
            return;
        }
        {
            ProfScope ps(h, s, "lg.gemm.ffn0");
            if (f16) {
                GemmF16Args g = lin_args_f16(blk.ffn0f, XS, 256, (int)T);
                g.A1 = CTX; g.lda1 = 256; g.ksplit = 256;
                g.C = HID; g.ldc = 512;
                launch_gemm_f16x3(g, s);
            } else {
                GemmArgs g = lin_args(blk.ffn0, X, 256, (int)T, HID, 512);
                g.A1 = MSG; g.lda1 = 256; g.ksplit = 256;
                launch_gemm(g, s);
            }
        }
        {
            ProfScope ps(h, s, "lg.ln_gelu");
            launch_ln_gelu(HID, (long long)T, 512, blk.ln_g, blk.ln_b, 1e-5f, f16 ? 1 : 0, s);
        }
        {
            ProfScope ps(h, s, "lg.gemm.ffn3");
            if (f16) {
                GemmF16Args g = lin_args_f16(blk.ffn3, HID, 512, (int)T);
                g.residual = X; g.ldr = 256;
                g.C = X; g.ldc = 256; g.Cs = XS; g.ldcs = 256;
                launch_gemm_f16x3(g, s);
            } else {
                GemmArgs g = lin_args(blk.ffn3, HID, 512, (int)T, X, 256);
                g.residual = X; g.ldr = 256;
                launch_gemm(g, s);
            }
        }
    };
    auto proj = [&](const LgBlock& blk, const char* name, int ncols, float scale, bool rotary) {
        ProfScope ps(h, s, name);
        if (f16) {
            GemmF16Args g = lin_args_f16(blk.proj, XS, 256, (int)T);
            g.generic_only = h->taps.count("lg.gemm_generic") ? 1 : 0;  // A/B and bit-identity tests of gemm_ws_f16.hip
            g.Cs = QKV; g.ldcs = ncols;
            if (rotary) { g.enc = ENC; g.rotary_cols = 512; }
            g.scale = scale; g.scale_cols = 256;
            launch_gemm_f16x3(g, s);
        } else {
            GemmArgs g = lin_args(blk.proj, X, 256, (int)T, QKV, ncols);
            if (rotary) { g.enc = ENC; g.rotary_cols = 512; }
            g.scale = scale; g.scale_cols = 256;
            launch_gemm(g, s);
        }
    };
    auto attend = [&](const char* name, int ld, int kcol, int vcol, int cross) {
        ProfScope ps(h, s, name);
        if (f16) {
            AttnF16Args a;
            a.q = QKV; a.k = QKV; a.v = QKV; a.ld = ld; a.q_col = 0; a.k_col = kcol; a.v_col = vcol;
            a.out = CTX; a.ldo = 256; a.B = B; a.M = Mc; a.N = Nc; a.heads = 4; a.cross = cross; a.variant = attn_variant(h);
            launch_attention_f16x3(a, s);
        } else {
            AttnArgs a;
            a.q = QKV; a.k = QKV; a.v = QKV; a.ld = ld; a.q_col = 0; a.k_col = kcol; a.v_col = vcol;
            a.out = CTX; a.ldo = 256; a.B = B; a.M = Mc; a.N = Nc; a.heads = 4; a.cross = cross;
            launch_attention(a, s);
        }
    };
    auto outproj = [&](const LgBlock& blk) {
        if (f16) return;  // folded into ffn.0 (prep_ffn0_folded)
        ProfScope ps(h, s, "lg.gemm.out");
        {
            GemmArgs g = lin_args(blk.out, CTX, 256, (int)T, MSG, 256);
            launch_gemm(g, s);
        }
    };

    // adaptive state (batch 1): original index of every surviving token, pruning rounds survived, alternate buffers for the compaction
    int32_t *IND = nullptr, *IND2 = nullptr, *CNT = nullptr;
    float *X2 = nullptr, *XS2 = nullptr, *ENC2 = nullptr, *ZT = nullptr;
    if (ad) {
        IND = reinterpret_cast<int32_t*>(h->buf("lg.ind", Tin));
        IND2 = reinterpret_cast<int32_t*>(h->buf("lg.ind2", Tin));
        CNT = reinterpret_cast<int32_t*>(h->buf("lg.cnt", 8));
        X2 = h->buf("lg.x2", Tin * 256);
        XS2 = h->buf("lg.xs2", Tin * 256);
        ENC2 = h->buf("lg.enc2", Tin * 64);
        ZT = h->buf("lg.zt", Tin);
        if (!IND || !IND2 || !CNT || !X2 || !XS2 || !ENC2 || !ZT) return fail(MG_ERR_HIP, "workspace allocation failed");
        launch_lg_adaptive_init(IND, ad->prune0, ad->prune1, M, N, s);
    }
    int last = conf->n_layers - 1;  // the layer whose assignment head is used
    for (int i = 0; i < conf->n_layers; ++i) {
        const LgBlock& sb = h->lg_self[i];
        // f16x3: log2(e) is folded into the logit scale so the attention kernel uses bare v_exp_f32
        proj(sb, "lg.gemm.qkv", 768, f16 ? 0.125f * 1.4426950408889634f : 0.125f, true);
        if (i == 0) { if (int rc = tap_copy(h, "lg.qkv0", QKV, T * 768, s)) return rc; }  // raw rows of the first projection (SH8 words in f16x3 mode): kernel-equivalence tests
        attend("lg.attn.self", 768, 256, 512, 0);
        outproj(sb);
        ffn(sb);
        if (int rc = tap_copy(h, "lg.self" + std::to_string(i), X, T * 256, s)) return rc;

        const LgBlock& cb = h->lg_cross[i];
        // (dim_head^-0.5)^0.5 on both sides, lightglue.py:200 (x sqrt(log2 e) each in f16x3 mode)
        proj(cb, "lg.gemm.qkv_cross", 512, f16 ? 0.35355339059327373f * 1.2011224087864498f : 0.35355339059327373f, false);
        attend("lg.attn.cross", 512, 0, 256, 1);
        outproj(cb);
        ffn(cb);
        if (int rc = tap_copy(h, "lg.layer" + std::to_string(i), X, T * 256, s)) return rc;

        if (!ad || i == conf->n_layers - 1) continue;  // no early stopping or adaptive width at the last layer (lightglue.py:469-472)
        const float thr = (float)std::min(1.0, std::max(0.0, 0.8 + 0.1 * std::exp(-4.0 * i / conf->n_layers)));  // confidence_threshold(i), :536-539
        if (do_stop) {  // :474-479 with :551-560; the pruned points count as confident (num_points stays m + n)
            launch_matchability(X, (long long)T, h->lg_tok_w[i], h->lg_tok_b[i], ZT, s);
            launch_lg_count_below(ZT, (int)T, thr, CNT + 2, s);
            int32_t below = 0;
            HIPCHK(hipMemcpyAsync(&below, CNT + 2, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            const float ratio = 1.0f - (float)below / (float)(M + N);
            if (ratio > (float)ad->depth) { last = i; break; }  // an fp32 tensor compared with the Python scalar: in fp32 (:559)
        }
        if (do_prune) {  // :480-497 with get_pruning_mask :541-549
            launch_matchability(X, (long long)T, h->lg_match_w[i], h->lg_match_b[i], Z, s);
            LgPruneArgs pa;
            pa.z = Z; pa.zt = do_stop ? ZT : nullptr; pa.thr_tok = thr; pa.thr_keep = (float)(1.0 - ad->width);  // fp32(1 - width_confidence) from the double, lightglue.py:545
            pa.M = Mc; pa.N = Nc; pa.ind = IND; pa.ind_out = IND2; pa.prune0 = ad->prune0; pa.prune1 = ad->prune1; pa.src = reinterpret_cast<int32_t*>(ST);
            pa.counts = CNT;
            pa.x = X; pa.xs = f16 ? XS : nullptr; pa.enc = ENC; pa.x_out = X2; pa.xs_out = f16 ? XS2 : nullptr; pa.enc_out = ENC2;  // no SH8 copy in f32 mode
            launch_lg_prune(pa, s);
            int32_t cnt[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(cnt, CNT, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            launch_lg_prune_gather(pa, cnt[0], cnt[1], s);
            std::swap(X, X2); std::swap(ENC, ENC2); std::swap(IND, IND2);
            if (f16) std::swap(XS, XS2);
            Mc = cnt[0]; Nc = cnt[1];
            T0 = (size_t)Mc; T = (size_t)Mc + Nc;
            if (Mc < 1 || Nc < 1) return fail(MG_ERR_STATE, "point pruning removed every keypoint of an image");
        }
    }

    // assignment head (lightglue.py:265-303) of the last layer that ran
    const LinLayer& fproj = ad ? h->lg_final[last] : h->final_proj;
    const float* mw = ad ? h->lg_match_w[last] : h->match_w;
    const float mbias = ad ? h->lg_match_b[last] : h->match_b;
    const size_t T1 = T - T0;
    float* MD = QKV;  // reuse: (T,256) projected descriptors
    {
        ProfScope ps(h, s, "lg.gemm.final_proj");
        // both arithmetic modes: the projection and the similarity below run on the fp32 MFMA with float64 stage sums (gemm.hip PRECISE):
        // log_assignment = 2 sim - lse_row - lse_col needs sim to a fraction of an fp32 ulp to stay within 1e-4 of the exact value
        GemmArgs g = lin_args(fproj, X, 256, (int)T, MD, 256);
        g.scale = 0.25f;  // / 256^(1/4), exact
        g.scale_cols = 256;
        g.precise = 1;
        launch_gemm(g, s);
    }
    {
        ProfScope ps(h, s, "lg.gemm.sim");
        GemmArgs g;  // sim[b] (M x N) = mdesc0[b] mdesc1[b]^T
        g.A0 = MD; g.lda0 = 256; g.ksplit = 256;
        g.W = MD + T0 * 256; g.w_rowmajor = 1; g.ldw = 256;
        g.C = SIM; g.ldc = Nc; g.M = Mc; g.N = Nc; g.K = 256;
        g.batch = B; g.strideA = (long long)Mc * 256; g.strideW = (long long)Nc * 256; g.strideC = (long long)Mc * Nc;
        g.precise = 1;
        launch_gemm(g, s);
    }
    // the adaptive variant assigns in the surviving points' index space and scatters back to the original one afterwards
    long long* am0 = reinterpret_cast<long long*>(matches0);
    long long* am1 = reinterpret_cast<long long*>(matches1);
    float *as0 = mscores0, *as1 = mscores1;
    if (do_prune) {
        am0 = reinterpret_cast<long long*>(h->buf("lg.am", 2 * (T + 2)));
        as0 = h->buf("lg.as", T + 2);
        if (!am0 || !as0) return fail(MG_ERR_HIP, "workspace allocation failed");
        am1 = am0 + T0;
        as1 = as0 + T0;
    }
    {
        ProfScope ps(h, s, "lg.assign");
        launch_matchability(X, (long long)T, mw, mbias, Z, s);
        AssignArgs a;
        a.sim = SIM; a.z0 = Z; a.z1 = Z + T0; a.B = B; a.M = Mc; a.N = Nc; a.threshold = conf->filter_threshold;
        a.row_mx = ST; a.row_ls = ST + T0; a.row_best = ST + 2 * T0;
        a.row_arg = reinterpret_cast<int*>(ST + 3 * T0);
        float* cbase = ST + 4 * T0;
        a.col_mx = cbase; a.col_ls = cbase + T1; a.col_best = cbase + 2 * T1;
        a.col_arg = reinterpret_cast<int*>(cbase + 3 * T1);
        a.ls0 = ST + 4 * T; a.ls1 = a.ls0 + T0; a.lsn0 = ST + 5 * T; a.lsn1 = a.lsn0 + T0;
        a.log_assignment = log_assignment;
        a.m0 = am0; a.m1 = am1;
        a.ms0 = as0; a.ms1 = as1;
        launch_assignment(a, s);
    }
    if (do_prune)  // :500-508: back to the original indices; pruned points are unmatched with score 0
        launch_lg_scatter_matches(am0, am1, as0, as1, IND, Mc, Nc, M, N, reinterpret_cast<long long*>(matches0), reinterpret_cast<long long*>(matches1),
                                  mscores0, mscores1, s);
    if (ad && ad->out_dims) { ad->out_dims[0] = Mc; ad->out_dims[1] = Nc; ad->out_dims[2] = last + 1; }
    if (ref_desc0) HIPCHK(hipMemcpyAsync(ref_desc0, X, T0 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (ref_desc1) HIPCHK(hipMemcpyAsync(ref_desc1, X + T0 * 256, T1 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipGetLastError());
    return 0;
}

int mg_lg_forward(mg_handle h, const mg_lg_conf* conf, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                  const float* size0, const float* size1, int B, int M, int N, int64_t* matches0, int64_t* matches1, float* mscores0,
                  float* mscores1, float* log_assignment, float* ref_desc0, float* ref_desc1, void* stream) {
    return lg_run(h, conf, kpts0, kpts1, desc0, desc1, size0, size1, B, M, N, matches0, matches1, mscores0, mscores1, log_assignment, ref_desc0,
                  ref_desc1, stream, nullptr);
}

int mg_lg_forward_adaptive(mg_handle h, const mg_lg_conf* conf, double depth_confidence, double width_confidence, const float* kpts0,
                           const float* kpts1, const float* desc0, const float* desc1, const float* size0, const float* size1, int M, int N,
                           int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* log_assignment, float* ref_desc0,
                           float* ref_desc1, int32_t* prune0, int32_t* prune1, int32_t* out_dims, void* stream) {
    if (!prune0 || !prune1 || !out_dims) return fail(MG_ERR_ARG, "NULL argument");
    LgAdaptive ad;
    ad.depth = depth_confidence; ad.width = width_confidence; ad.prune0 = prune0; ad.prune1 = prune1; ad.out_dims = out_dims;
    return lg_run(h, conf, kpts0, kpts1, desc0, desc1, size0, size1, 1, M, N, matches0, matches1, mscores0, mscores1, log_assignment, ref_desc0,
                  ref_desc1, stream, &ad);
}

// ------------------------------------------------------------------------------------------------
// Shared by SuperGlue and GlueStick: one dispatcher for both arithmetic modes over the (T, .) token matrix.
// "act" buffers hold SH8 rows in f16x3 mode and fp32 rows otherwise; X is the fp32 residual stream, XS its SH8 copy.
struct GnnRun {
    mg_context* h;
    hipStream_t s;
    bool f16;
    int B, M, N;
    size_t T0, T;
    float *X, *XS, *QKV, *CTX, *HID, *P0;
    struct Opt { int relu = 0; float scale = 1.f; int scale_cols = 0; const float* residual = nullptr; float* c32 = nullptr; int ldc32 = 0; };

    float* xa() const { return f16 ? XS : X; }  // the GEMM-input view of the descriptor stream

    void linear(const LinLayer& L, size_t rows, const float* A, int lda, const float* A1, int lda1, float* act_out, int ld_out, const Opt& o) const {
        if (f16) {
            GemmF16Args g = lin_args_f16(L, A, lda, (int)rows);
            g.generic_only = h->taps.count("lg.gemm_generic") ? 1 : 0;
            if (A1) { g.A1 = A1; g.lda1 = lda1; g.ksplit = lda; }
            g.Cs = act_out; g.ldcs = ld_out; g.C = o.c32; g.ldc = o.ldc32;
            g.relu = o.relu; g.scale = o.scale; g.scale_cols = o.scale_cols; g.residual = o.residual; g.ldr = 256;
            launch_gemm_f16x3(g, s);
        } else {
            float* out = o.c32 ? o.c32 : act_out;
            GemmArgs g = lin_args(L, A, lda, (int)rows, out, o.c32 ? o.ldc32 : ld_out);
            if (A1) { g.A1 = A1; g.lda1 = lda1; g.ksplit = lda; }
            g.relu = o.relu; g.scale = o.scale; g.scale_cols = o.scale_cols; g.residual = o.residual; g.ldr = 256;
            launch_gemm(g, s);
        }
    }
    // MLP([32 (padded input) -> 32, 64, 128, 256, 256]) on `rows` rows; in32 = fp32 input rows, in_sh8 = room for their SH8
    // copy, t0 / t1 = ping-pong activations (>= rows x 256), the last layer goes where `last` / last_act say
    void encoder(const LinLayer* enc, size_t rows, float* in32, float* in_sh8, float* t0, float* t1, float* last_act, const Opt& last) const {
        const float* a = in32;
        if (f16) { launch_to_sh8(in32, in_sh8, (long long)rows * 32, 1.f, s); a = in_sh8; }
        Opt relu; relu.relu = 1;
        linear(enc[0], rows, a, 32, nullptr, 0, t0, 32, relu);
        linear(enc[1], rows, t0, 32, nullptr, 0, t1, 64, relu);
        linear(enc[2], rows, t1, 64, nullptr, 0, t0, 128, relu);
        linear(enc[3], rows, t0, 128, nullptr, 0, t1, 256, relu);
        linear(enc[4], rows, t1, 256, nullptr, 0, last_act, 256, last);
    }
    // desc += MLP([x, y, score]) for every keypoint (superglue.py:89-103, :275-276; gluestick.py:199-200)
    void keypoint_encoder(const LinLayer* enc, const float* kpts0, const float* kpts1, const float* scores0, const float* scores1,
                          const float* size0, const float* size1) const {
        float* KIN = HID;  // (T,32) fp32 followed by its SH8 copy
        launch_sg_kenc_input(kpts0, kpts1, scores0, scores1, size0, size1, B, M, N, KIN, s);
        Opt last; last.residual = X; last.c32 = X; last.ldc32 = 256;
        encoder(enc, T, KIN, HID + T * 32, P0, CTX, XS, last);
    }
    // one AttentionalPropagation block (superglue.py:118-141): x += MLP([x, merge(attention(x, source))]) with the merge
    // projection folded into mlp.0 at upload; cross = attend to the other image of the pair
    void layer(const mg_context::SgLayer& Lr, int cross, const char* tag) const {
        const std::string t(tag);
        {
            ProfScope ps(h, s, t + ".gemm.qkv");
            Opt o; o.scale = f16 ? 0.125f * 1.4426950408889634f : 0.125f; o.scale_cols = 256;  // / sqrt(64) (x log2 e in f16x3 mode)
            linear(Lr.proj, T, xa(), 256, nullptr, 0, QKV, 768, o);
        }
        {
            ProfScope ps(h, s, t + (cross ? ".attn.cross" : ".attn.self"));
            if (f16) {
                AttnF16Args a;
                a.q = QKV; a.k = QKV; a.v = QKV; a.ld = 768; a.q_col = 0; a.k_col = 256; a.v_col = 512;
                a.out = CTX; a.ldo = 256; a.B = B; a.M = M; a.N = N; a.heads = 4; a.cross = cross; a.variant = attn_variant(h);
                launch_attention_f16x3(a, s);
            } else {
                AttnArgs a;
                a.q = QKV; a.k = QKV; a.v = QKV; a.ld = 768; a.q_col = 0; a.k_col = 256; a.v_col = 512;
                a.out = CTX; a.ldo = 256; a.B = B; a.M = M; a.N = N; a.heads = 4; a.cross = cross;
                launch_attention(a, s);
            }
        }
        const bool unfused_ffn = h->taps.count("lg.unfused_ffn") != 0;
        if (f16 && !unfused_ffn) {
            ProfScope ps(h, s, t + ".ffn");
            FfnF16Args a;
            a.A0 = XS; a.A1 = CTX; a.W0 = Lr.mlp0f.Wh; a.winv0 = Lr.mlp0f.winv; a.b0 = Lr.mlp0f.bias; a.W3 = Lr.mlp3.Whp; a.winv3 = Lr.mlp3.winv; a.b3 = Lr.mlp3.bias;
            a.X = X; a.XS = XS; a.T = (int)T; a.act = 1;
            launch_ffn_f16x3(a, s);
            return;
        }
        {
            ProfScope ps(h, s, t + ".gemm.mlp0");
            Opt o; o.relu = 1;
            linear(Lr.mlp0f, T, xa(), 256, CTX, 256, HID, 512, o);
        }
        {
            ProfScope ps(h, s, t + ".gemm.mlp3");
            Opt o; o.residual = X; o.c32 = X; o.ldc32 = 256;
            linear(Lr.mlp3, T, HID, 512, nullptr, 0, XS, 256, o);
        }
    }
    // The final projection and the similarity behind the assignment, in both arithmetic modes on the fp32 MFMA with float64 stage
    // sums (gemm.hip PRECISE, as LightGlue's): the Sinkhorn iterations / double soft-max amplify every error of the costs.
    // MD = fp32 projected tokens (image-0 rows, then image-1 rows), computed from the fp32 descriptor stream X.
    void final_linear(const LinLayer& L, size_t rows, float* MD) const {
        GemmArgs g = lin_args(L, X, 256, (int)rows, MD, 256);
        g.precise = 1;
        launch_gemm(g, s);
    }
    // out[b] = MD0[b, :rowsA] . MD1[b, :rowsW]^T / 16
    void similarity(const float* MD, size_t rowsA, size_t rowsW, float* out) const {
        GemmArgs g;
        g.A0 = MD; g.lda0 = 256; g.ksplit = 256;
        g.W = MD + T0 * 256; g.w_rowmajor = 1; g.ldw = 256;
        g.C = out; g.ldc = (int)rowsW; g.M = (int)rowsA; g.N = (int)rowsW; g.K = 256;
        g.scale = 0.0625f; g.scale_cols = (int)rowsW;
        g.batch = B; g.strideA = (long long)M * 256; g.strideW = (long long)N * 256; g.strideC = (long long)rowsA * rowsW;
        g.precise = 1;
        launch_gemm(g, s);
    }
};

// ------------------------------------------------------------------------------------------------
int mg_sg_forward(mg_handle h, const mg_sg_conf* conf, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                  const float* scores0, const float* scores1, const float* size0, const float* size1, int B, int M, int N,
                  int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* log_assignment, float* sinkhorn_cost,
                  void* stream) {
    if (!h || !conf || !kpts0 || !kpts1 || !desc0 || !desc1 || !scores0 || !scores1 || !size0 || !size1 || !matches0 || !matches1 ||
        !mscores0 || !mscores1)
        return fail(MG_ERR_ARG, "NULL argument (SuperGlue needs keypoint scores and image sizes)");
    if (B < 1 || M < 1 || N < 1 || conf->n_layers < 1 || conf->sinkhorn_iterations < 0) return fail(MG_ERR_ARG, "bad shape / configuration");
    HIPCHK(hipSetDevice(h->device));
    if (int rc = prepare_sg(h, conf->n_layers)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool f16 = h->precision == MG_PREC_F16X3;

    const size_t T0 = (size_t)B * M, T1 = (size_t)B * N, T = T0 + T1;
    float* X = h->buf("lg.x", T * 256);
    float* XS = h->buf("lg.xs", T * 256);
    float* QKV = h->buf("lg.qkv", T * 768);
    float* CTX = h->buf("lg.ctx", T * 256);
    float* HID = h->buf("lg.hid", T * 512);
    float* P0 = h->buf("lg.msg", T * 256);
    float* SIM = h->buf("lg.sim", (size_t)B * M * N);
    float* ST = h->buf("lg.stats", 6 * T + 16);
    float* UV = h->buf("sg.uv", sinkhorn_work_floats(B, M, N));
    if (!X || !XS || !QKV || !CTX || !HID || !P0 || !SIM || !ST || !UV) return fail(MG_ERR_HIP, "workspace allocation failed");

    const GnnRun g{h, s, f16, B, M, N, T0, T, X, XS, QKV, CTX, HID, P0};
    HIPCHK(hipMemcpyAsync(X, desc0, T0 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(X + T0 * 256, desc1, T1 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    {
        ProfScope ps(h, s, "sg.kenc");
        g.keypoint_encoder(h->sg_kenc, kpts0, kpts1, scores0, scores1, size0, size1);
    }
    if (int rc = tap_copy(h, "sg.kenc", X, T * 256, s)) return rc;
    for (int l = 0; l < conf->n_layers; ++l) {
        g.layer(h->sg_gnn[l], l & 1, "sg");  // GNN_layers = ["self", "cross"] * 9
        if (int rc = tap_copy(h, "sg.layer" + std::to_string(l), X, T * 256, s)) return rc;
    }
    float* MD = QKV;
    {
        ProfScope ps(h, s, "sg.gemm.final_proj");
        g.final_linear(h->sg_final, T, MD);
    }
    {
        ProfScope ps(h, s, "sg.gemm.cost");  // scores / sqrt(256), superglue.py:283-284
        g.similarity(MD, M, N, SIM);
    }
    if (sinkhorn_cost) HIPCHK(hipMemcpyAsync(sinkhorn_cost, SIM, (size_t)B * M * N * sizeof(float), hipMemcpyDeviceToDevice, s));
    {
        ProfScope ps(h, s, "sg.sinkhorn");
        SinkhornArgs k;
        k.cost = SIM; k.alpha = h->sg_bin; k.B = B; k.M = M; k.N = N; k.iters = conf->sinkhorn_iterations;
        k.work = UV;
        k.log_assignment = log_assignment;
        k.row_best = ST + 2 * T0; k.row_arg = reinterpret_cast<int*>(ST + 3 * T0);
        float* cbase = ST + 4 * T0;
        k.col_best = cbase + 2 * T1; k.col_arg = reinterpret_cast<int*>(cbase + 3 * T1);
        launch_sinkhorn(k, s);
        AssignArgs a;
        a.B = B; a.M = M; a.N = N; a.threshold = conf->filter_threshold;
        a.row_best = k.row_best; a.row_arg = k.row_arg; a.col_best = k.col_best; a.col_arg = k.col_arg;
        a.m0 = reinterpret_cast<long long*>(matches0); a.m1 = reinterpret_cast<long long*>(matches1);
        a.ms0 = mscores0; a.ms1 = mscores1;
        launch_assign_filter(a, s);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
int mg_gs_forward(mg_handle h, const mg_gs_conf* conf, const float* kpts0, const float* kpts1, const float* desc0, const float* desc1,
                  const float* scores0, const float* scores1, const float* size0, const float* size1, const float* lines0, const float* lines1,
                  const int64_t* junc_idx0, const int64_t* junc_idx1, const float* line_scores0, const float* line_scores1, int B, int M, int N,
                  int n0, int n1, int64_t* matches0, int64_t* matches1, float* mscores0, float* mscores1, float* log_assignment,
                  int64_t* line_matches0, int64_t* line_matches1, float* line_mscores0, float* line_mscores1, float* line_log_assignment,
                  float* raw_line_scores, void* stream) {
    if (!h || !conf || !kpts0 || !kpts1 || !desc0 || !desc1 || !scores0 || !scores1 || !size0 || !size1 || !matches0 || !matches1 ||
        !mscores0 || !mscores1)
        return fail(MG_ERR_ARG, "NULL argument (GlueStick needs keypoint scores and image sizes)");
    if (B < 1 || M < 1 || N < 1 || n0 < 0 || n1 < 0 || conf->n_layers < 1) return fail(MG_ERR_ARG, "bad shape / configuration");
    const bool lines = n0 > 0 && n1 > 0;  // gluestick.py:203, :735-739: the line branch needs lines in both images
    if (lines) {
        if (!lines0 || !lines1 || !junc_idx0 || !junc_idx1 || !line_scores0 || !line_scores1 || !line_matches0 || !line_matches1 ||
            !line_mscores0 || !line_mscores1)
            return fail(MG_ERR_ARG, "NULL line argument");
        if (2 * n0 > M || 2 * n1 > N)
            return fail(MG_ERR_ARG, "line junctions must be the first 2*n_lines keypoints (gluestick.py:262-263 slices them off the front)");
    }
    HIPCHK(hipSetDevice(h->device));
    if (int rc = prepare_gs(h, conf->n_layers)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool f16 = h->precision == MG_PREC_F16X3;

    const size_t T0 = (size_t)B * M, T1 = (size_t)B * N, T = T0 + T1;
    const size_t E0 = (size_t)B * 2 * n0, E = lines ? E0 + (size_t)B * 2 * n1 : 0;
    float* X = h->buf("lg.x", T * 256);
    float* XS = h->buf("lg.xs", T * 256);
    float* QKV = h->buf("lg.qkv", T * 768);
    float* CTX = h->buf("lg.ctx", T * 256);
    float* HID = h->buf("lg.hid", T * 512);
    float* P0 = h->buf("lg.msg", T * 256);
    float* SIM = h->buf("lg.sim", (size_t)B * M * N);
    float* ST = h->buf("lg.stats", 6 * T + 16);
    if (!X || !XS || !QKV || !CTX || !HID || !P0 || !SIM || !ST) return fail(MG_ERR_HIP, "workspace allocation failed");
    float *ENC = nullptr, *MSG = nullptr, *LH = nullptr, *UPD = nullptr, *LT = nullptr;
    if (lines) {
        ENC = h->buf("gs.enc", E * 256);
        MSG = h->buf("gs.msg", E * 768);
        LH = h->buf("gs.hid", E * 512);
        UPD = h->buf("gs.upd", E * 256);
        LT = h->buf("gs.tmp", E * 256 + E * 64);
        if (!ENC || !MSG || !LH || !UPD || !LT) return fail(MG_ERR_HIP, "workspace allocation failed");
    }

    const GnnRun g{h, s, f16, B, M, N, T0, T, X, XS, QKV, CTX, HID, P0};
    using Opt = GnnRun::Opt;
    float* XA = g.xa();

    HIPCHK(hipMemcpyAsync(X, desc0, T0 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(X + T0 * 256, desc1, T1 * 256 * sizeof(float), hipMemcpyDeviceToDevice, s));
    {
        ProfScope ps(h, s, "gs.kenc");
        g.keypoint_encoder(h->gs_kenc, kpts0, kpts1, scores0, scores1, size0, size1);
    }
    if (int rc = tap_copy(h, "gs.kenc", X, T * 256, s)) return rc;
    if (lines) {  // endpoint encodings (gluestick.py:204-212), kept for every line layer
        ProfScope ps(h, s, "gs.lenc");
        float* LIN = LT;  // (E,32) fp32 followed by its SH8 copy
        launch_gs_lenc_input(lines0, lines1, line_scores0, line_scores1, size0, size1, B, n0, n1, LIN, s);
        g.encoder(h->gs_lenc, E, LIN, LIN + E * 32, MSG, LH, ENC, Opt());
    }

    for (int l = 0; l < conf->n_layers; ++l) {
        const int cross = l & 1;  // GNN_layers = ["self", "cross"] * 9
        g.layer(h->gs_gnn[l], cross, "gs");
        if (!cross && lines) {  // line layer after every self layer (gluestick.py:719-741), num_line_iterations = 1
            ProfScope ps(h, s, "gs.line_layer");
            const mg_context::GsLine& Ll = h->gs_line[l / 2];
            GsLineArgs a;
            a.B = B; a.M = M; a.N = N; a.n0 = n0; a.n1 = n1;
            a.idx0 = reinterpret_cast<const long long*>(junc_idx0); a.idx1 = reinterpret_cast<const long long*>(junc_idx1);
            a.x = XA; a.enc = ENC; a.msg = MSG; a.upd = UPD; a.x32 = X; a.xs = f16 ? XS : nullptr;
            launch_gs_gather_msg(a, s);
            Opt o; o.relu = 1;
            g.linear(Ll.mlp0, E, MSG, 768, nullptr, 0, LH, 512, o);
            Opt u; u.c32 = UPD; u.ldc32 = 256;
            g.linear(Ll.mlp3, E, LH, 512, nullptr, 0, nullptr, 0, u);
            launch_gs_scatter_mean(a, s);
        }
        if (int rc = tap_copy(h, "gs.layer" + std::to_string(l), X, T * 256, s)) return rc;
    }

    auto assign = [&](const float* scores, float bin, int m, int n, float* la, int64_t* mt0, int64_t* mt1, float* sc0, float* sc1) {
        const size_t r0 = (size_t)B * m, r1 = (size_t)B * n;
        GsSoftmaxArgs d;
        d.scores = scores; d.bin = bin; d.B = B; d.M = m; d.N = n; d.out = la;
        d.row_lse = ST; d.col_lse = ST + r0;
        d.row_best = ST + r0 + r1; d.row_arg = reinterpret_cast<int*>(ST + 2 * r0 + r1);
        d.col_best = ST + 3 * r0 + r1; d.col_arg = reinterpret_cast<int*>(ST + 3 * r0 + 2 * r1);
        launch_gs_double_softmax(d, s);
        AssignArgs a;
        a.B = B; a.M = m; a.N = n; a.threshold = conf->filter_threshold;
        a.row_best = d.row_best; a.row_arg = d.row_arg; a.col_best = d.col_best; a.col_arg = d.col_arg;
        a.m0 = reinterpret_cast<long long*>(mt0); a.m1 = reinterpret_cast<long long*>(mt1); a.ms0 = sc0; a.ms1 = sc1;
        launch_assign_filter(a, s);
    };

    float* MD = QKV;
    {
        ProfScope ps(h, s, "gs.match");
        g.final_linear(h->gs_final, T, MD);
        g.similarity(MD, M, N, SIM);
        assign(SIM, h->gs_bin, M, N, log_assignment, matches0, matches1, mscores0, mscores1);
    }
    if (lines) {
        ProfScope ps(h, s, "gs.line_match");
        float* JS = h->buf("gs.js", (size_t)B * 4 * n0 * n1);
        float* RAW = raw_line_scores ? raw_line_scores : h->buf("gs.raw", (size_t)B * n0 * n1);
        if (!JS || !RAW) return fail(MG_ERR_HIP, "workspace allocation failed");
        g.final_linear(h->gs_final_line, T, MD);  // all tokens; only the leading 2*n rows of a pair are used
        g.similarity(MD, 2 * (size_t)n0, 2 * (size_t)n1, JS);
        launch_gs_line_scores(JS, reinterpret_cast<const long long*>(junc_idx0), reinterpret_cast<const long long*>(junc_idx1), B, n0, n1, RAW, s);
        assign(RAW, h->gs_line_bin, n0, n1, line_log_assignment, line_matches0, line_matches1, line_mscores0, line_mscores1);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
int mg_wf_filter(mg_handle h, const float* keypoints, const float* scores, const float* descriptors, const int32_t* num_valid,
                 const float* endpoints, int B, int K, int E, float radius, float* out_keypoints, float* out_scores, float* out_descriptors,
                 int32_t* out_count, void* stream) {
    if (!h || !keypoints || !scores || !descriptors || !out_keypoints || !out_scores || !out_descriptors || !out_count || (E > 0 && !endpoints))
        return fail(MG_ERR_ARG, "NULL argument");
    if (B < 1 || K < 1 || E < 0) return fail(MG_ERR_ARG, "bad shape");
    HIPCHK(hipSetDevice(h->device));
    WfFilterArgs a;
    a.keypoints = keypoints; a.scores = scores; a.descriptors = descriptors; a.num_valid = num_valid; a.endpoints = endpoints;
    a.B = B; a.K = K; a.E = E; a.radius = radius;
    a.out_keypoints = out_keypoints; a.out_scores = out_scores; a.out_descriptors = out_descriptors; a.out_count = out_count;
    a.src_row = reinterpret_cast<int*>(h->buf("wf.src", (size_t)B * K));
    if (!a.src_row) return fail(MG_ERR_HIP, "workspace allocation failed");
    launch_wf_filter(a, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
int mg_nn_forward(mg_handle h, const mg_nn_conf* conf, const float* desc0, const float* desc1, int B, int M, int N, int D, int64_t* matches0,
                  int64_t* matches1, float* mscores0, float* mscores1, float* similarity, float* log_assignment, void* stream) {
    if (!h || !conf || !desc0 || !desc1 || !matches0 || !matches1 || !mscores0 || !mscores1) return fail(MG_ERR_ARG, "NULL argument");
    if (B < 1 || M < 1 || N < 1 || D < 1) return fail(MG_ERR_ARG, "bad shape");
    if (conf->ratio_thresh > 0.f && (M < 2 || N < 2))
        return fail(MG_ERR_ARG, "ratio test needs two candidates per side (the reference's topk(2) raises as well)");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool f16 = h->precision == MG_PREC_F16X3;
    const size_t T0 = (size_t)B * M, T1 = (size_t)B * N;
    float* SIM = similarity ? similarity : h->buf("lg.sim", (size_t)B * M * N);
    float* ST = h->buf("lg.stats", 6 * (T0 + T1) + 16);
    if (!SIM || !ST) return fail(MG_ERR_HIP, "workspace allocation failed");
    if (D & 15) {  // the GEMMs contract in steps of 16: zero-pad the descriptor rows (zeros add nothing to a dot product)
        const int Dp = (D + 15) & ~15;
        float* PD = h->buf("nn.pad", (T0 + T1) * Dp);
        if (!PD) return fail(MG_ERR_HIP, "workspace allocation failed");
        launch_zero_words(PD, (size_t)(T0 + T1) * Dp, s);  // (a kernel, not a memset node: kernels.h)
        HIPCHK(hipMemcpy2DAsync(PD, Dp * sizeof(float), desc0, D * sizeof(float), D * sizeof(float), T0, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpy2DAsync(PD + T0 * Dp, Dp * sizeof(float), desc1, D * sizeof(float), D * sizeof(float), T1, hipMemcpyDeviceToDevice, s));
        desc0 = PD;
        desc1 = PD + T0 * Dp;
        D = Dp;
    }
    {
        ProfScope ps(h, s, "nn.gemm.sim");
        if (f16) {
            float* DS = h->buf("nn.desc", (T0 + T1) * D);
            if (!DS) return fail(MG_ERR_HIP, "workspace allocation failed");
            launch_to_sh8(desc0, DS, (long long)T0 * D, 1.f, s);
            launch_to_sh8(desc1, DS + T0 * D, (long long)T1 * D, 1.f, s);
            GemmF16Args g;
            g.A0 = DS; g.lda0 = D; g.ksplit = D;
            g.W = DS + T0 * D; g.w_rowmajor = 1; g.ldw = D;
            g.C = SIM; g.ldc = N; g.M = M; g.N = N; g.K = D;
            g.batch = B; g.strideA = (long long)M * D; g.strideW = (long long)N * D; g.strideC = (long long)M * N;
            launch_gemm_f16x3(g, s);
        } else {
            GemmArgs g;
            g.A0 = desc0; g.lda0 = D; g.ksplit = D;
            g.W = desc1; g.w_rowmajor = 1; g.ldw = D;
            g.C = SIM; g.ldc = N; g.M = M; g.N = N; g.K = D;
            g.batch = B; g.strideA = (long long)M * D; g.strideW = (long long)N * D; g.strideC = (long long)M * N;
            launch_gemm(g, s);
        }
    }
    {
        ProfScope ps(h, s, "nn.match");
        NnArgs a;
        a.sim = SIM; a.B = B; a.M = M; a.N = N;
        a.use_ratio = conf->ratio_thresh > 0.f; a.use_dist = conf->distance_thresh > 0.f; a.mutual = conf->mutual_check != 0;
        a.ratio2 = (float)((double)conf->ratio_thresh * (double)conf->ratio_thresh);
        a.dist2 = (float)((double)conf->distance_thresh * (double)conf->distance_thresh);
        a.row_lse = ST; a.col_lse = ST + T0;
        a.cand0 = reinterpret_cast<int*>(ST + T0 + T1); a.cand1 = a.cand0 + T0;
        a.m0 = reinterpret_cast<long long*>(matches0); a.m1 = reinterpret_cast<long long*>(matches1);
        a.ms0 = mscores0; a.ms1 = mscores1; a.la = log_assignment;
        launch_nn_match(a, s);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
int mg_test_linear(mg_handle h, const float* A, const float* W_host, const float* bias_host, int M, int N, int K, float* C, void* stream) {
    if (!h || !A || !W_host || !C || M < 1 || N < 1 || K < 32 || (K % 32)) return fail(MG_ERR_ARG, "bad argument (K must be a multiple of 32)");
    HIPCHK(hipSetDevice(h->device));
    if (int rc = begin_prepare(h, 4)) return rc;
    LinLayer L;
    std::vector<const float*> rows(N);
    for (int n = 0; n < N; ++n) rows[n] = W_host + (size_t)n * K;
    std::vector<float> bias;
    if (bias_host) bias.assign(bias_host, bias_host + N);
    if (int rc = prep_linear_rows(h, rows, bias, K, &L)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->precision == MG_PREC_F16X3) {
        float* AS = h->buf("test.a", (size_t)M * K);
        if (!AS) return fail(MG_ERR_HIP, "workspace allocation failed");
        launch_to_sh8(A, AS, (long long)M * K, 1.f, s);
        GemmF16Args g = lin_args_f16(L, AS, K, M);
        g.C = C; g.ldc = N;
        launch_gemm_f16x3(g, s);
    } else {
        GemmArgs g = lin_args(L, A, K, M, C, N);
        launch_gemm(g, s);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int mg_test_attention(mg_handle h, const float* q, const float* k, const float* v, int B, int M, int N, int cross, float* out, void* stream) {
    if (!h || !q || !k || !v || !out || B < 1 || M < 1 || N < 1) return fail(MG_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t T = (size_t)B * (M + N);
    if (h->precision == MG_PREC_F16X3) {
        float* QS = h->buf("test.q", T * 256);
        float* KS = h->buf("test.k", T * 256);
        float* VS = h->buf("test.v", T * 256);
        float* OS = h->buf("test.o", T * 256);
        if (!QS || !KS || !VS || !OS) return fail(MG_ERR_HIP, "workspace allocation failed");
        launch_to_sh8(q, QS, (long long)T * 256, 1.4426950408889634f, s);  // the kernel exponentiates in base 2
        launch_to_sh8(k, KS, (long long)T * 256, 1.f, s);
        launch_to_sh8(v, VS, (long long)T * 256, 1.f, s);
        AttnF16Args a;
        a.q = QS; a.k = KS; a.v = VS; a.ld = 256; a.q_col = 0; a.k_col = 0; a.v_col = 0;
        a.out = OS; a.ldo = 256; a.B = B; a.M = M; a.N = N; a.heads = 4; a.cross = cross; a.variant = attn_variant(h);
        launch_attention_f16x3(a, s);
        launch_from_sh8(OS, out, (long long)T * 256, s);
    } else {
        AttnArgs a;
        a.q = q; a.k = k; a.v = v; a.ld = 256; a.q_col = 0; a.k_col = 0; a.v_col = 0;
        a.out = out; a.ldo = 256; a.B = B; a.M = M; a.N = N; a.heads = 4; a.cross = cross;
        launch_attention(a, s);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
