"""CPU oracle for the SuperPoint + LightGlue forward path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package
(`glue-factory_amd/`).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may use it, and only as the checker.

This is a functional restatement (PyTorch-CPU operators, no reference imports)
of the reference's hot path.  It computes in the dtype of its inputs: fp32
tensors give the reference's own CPU arithmetic (that is what the goldens pin);
the same code fed float64 weights and inputs (`cast(..., torch.float64)`) is the
fp64 ANCHOR of oracle/anchor.py -- the independent truth against which both the
HIP path's and the fp32 CPU path's float errors are measured.  Integer decisions
of a run (keypoint pixels, pruning / stopping) can be replayed into another run
(`force_yx`, `replay`) so that floats of two dtypes are compared element by
element on the same decisions.  Paths restated:

  * SuperPoint (MagicLeap layout)  /root/reference/gluefactory_nonfree/superpoint.py:202-350
  * SuperPoint-open                /root/reference/gluefactory/models/extractors/superpoint_open.py:123-217
  * LightGlue (in-tree)            /root/reference/gluefactory/models/matchers/lightglue.py:399-525
  * TwoViewPipeline key plumbing   /root/reference/gluefactory/models/two_view_pipeline.py:62-90
  * SuperGlue                      /root/reference/gluefactory_nonfree/superglue.py:253-307
  * GlueStick                      /root/reference/gluefactory/models/matchers/gluestick.py:136-312
  * nearest-neighbour matcher      /root/reference/gluefactory/models/matchers/nearest_neighbor_matcher.py:15-72

Parity pin: `tests/golden/*.npz` were produced by importing the reference
itself in the build container (`oracle/tools/make_goldens.py`); the oracle is
checked against them bit-for-bit on CPU in `tests/test_oracle_golden.py`.

Weights are plain `dict[str, Tensor]` using the reference's state_dict key
names, so a reference checkpoint can be fed in unchanged.
"""
import math

import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

GRAY_WEIGHTS = (0.299, 0.587, 0.114)


def cast(obj, dtype):
    """Recursively cast the floating tensors of a dict / list / tuple (weights, data dicts) to `dtype`."""
    if isinstance(obj, torch.Tensor):
        return obj.to(dtype) if obj.is_floating_point() else obj
    if isinstance(obj, dict):
        return {k: cast(v, dtype) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(cast(v, dtype) for v in obj)
    return obj


# --------------------------------------------------------------------------
# SuperPoint
# --------------------------------------------------------------------------
def to_gray(image):
    """superpoint.py:204-206 -- weighted channel sum when the input is RGB."""
    if image.shape[1] == 3:
        w = image.new_tensor(GRAY_WEIGHTS).view(1, 3, 1, 1)
        image = (image * w).sum(1, keepdim=True)
    return image


def _conv_relu(x, W, name):
    return F.relu(F.conv2d(x, W[name + ".weight"], W[name + ".bias"], padding=W[name + ".weight"].shape[-1] // 2))


def sp_backbone(image, W):
    """superpoint.py:209-219 -- 8x (conv3x3 + ReLU), 2x2 max-pool after blocks 1..3."""
    x = image
    for blk in (1, 2, 3, 4):
        x = _conv_relu(x, W, f"conv{blk}a")
        x = _conv_relu(x, W, f"conv{blk}b")
        if blk < 4:
            x = F.max_pool2d(x, 2, 2)
    return x


def _open_block(x, W, prefix, relu=True):
    """superpoint_open.py:59-75 -- Conv -> (ReLU) -> BatchNorm(eps=1e-3), eval-mode statistics."""
    w = W[prefix + ".conv.weight"]
    x = F.conv2d(x, w, W[prefix + ".conv.bias"], padding=(w.shape[-1] - 1) // 2)
    if relu:
        x = F.relu(x)
    return F.batch_norm(
        x,
        W[prefix + ".bn.running_mean"],
        W[prefix + ".bn.running_var"],
        W[prefix + ".bn.weight"],
        W[prefix + ".bn.bias"],
        training=False,
        eps=1e-3,
    )


def sp_open_backbone(image, W):
    """superpoint_open.py:94-105."""
    x = image
    for i in range(4):
        x = _open_block(x, W, f"backbone.{i}.0")
        x = _open_block(x, W, f"backbone.{i}.1")
        if i < 3:
            x = F.max_pool2d(x, 2, 2)
    return x


def logits_to_heatmap(logits, cell=8):
    """superpoint.py:226-230 -- softmax over 65, drop the dustbin, depth-to-space."""
    p = torch.softmax(logits, 1)[:, :-1]
    b, _, h, w = p.shape
    p = p.permute(0, 2, 3, 1).reshape(b, h, w, cell, cell)
    return p.permute(0, 1, 3, 2, 4).reshape(b, h * cell, w * cell)


def sp_dense(image, W, variant="magicleap"):
    """Dense stage: returns (heatmap (B,8h,8w), dense descriptors (B,256,h,w))."""
    image = to_gray(image)
    if variant == "magicleap":
        feat = sp_backbone(image, W)
        logits = F.conv2d(_conv_relu(feat, W, "convPa"), W["convPb.weight"], W["convPb.bias"])
        raw = F.conv2d(_conv_relu(feat, W, "convDa"), W["convDb.weight"], W["convDb.bias"])
    elif variant == "open":
        feat = sp_open_backbone(image, W)
        logits = _open_block(_open_block(feat, W, "detector.0"), W, "detector.1", relu=False)
        raw = _open_block(_open_block(feat, W, "descriptor.0"), W, "descriptor.1", relu=False)
    else:
        raise ValueError(variant)
    dense_desc = F.normalize(raw, p=2, dim=1)
    return logits_to_heatmap(logits), dense_desc


def nms_heatmap(scores, radius):
    """superpoint.py:60-80 (= superpoint_open.py:34-49).

    Keeps local maxima of the (2r+1)^2 window, then twice re-admits maxima of
    the not-yet-suppressed remainder.  Equal neighbours are all kept.
    """
    k = 2 * radius + 1

    def pool(t):
        return F.max_pool2d(t, kernel_size=k, stride=1, padding=radius)

    zero = torch.zeros_like(scores)
    keep = scores == pool(scores)
    for _ in range(2):
        suppressed = pool(keep.to(scores.dtype)) > 0
        rest = torch.where(suppressed, zero, scores)
        fresh = rest == pool(rest)
        keep = keep | (fresh & ~suppressed)
    return torch.where(keep, scores, zero)


def mask_borders(scores, border, image_size=None, honour_image_size=True):
    """superpoint.py:243-254 / superpoint_open.py:147-152 -- score = -1 within `border` px of the frame.

    `image_size` is (B,2) as (w,h); the MagicLeap variant masks relative to it,
    the open variant always relative to the tensor.
    """
    if not border:
        return scores
    scores = scores.clone()
    scores[:, :border] = -1
    scores[:, :, :border] = -1
    if image_size is not None and honour_image_size:
        for i in range(scores.shape[0]):
            w, h = image_size[i]
            scores[i, int(h.item()) - border:] = -1
            scores[i, :, int(w.item()) - border:] = -1
    else:
        scores[:, -border:] = -1
        scores[:, :, -border:] = -1
    return scores


def select_keypoints(scores, threshold, max_kp):
    """superpoint.py:257-293, :83-87 -- per image: row-major candidates > threshold, then top-k.

    Returns a list of (yx int64 (n,2), score (n,)).  When n <= k the candidates
    come back in row-major order, otherwise sorted by descending score.
    """
    out = []
    for s in scores:
        yx = torch.nonzero(s > threshold)  # row-major
        v = s[yx[:, 0], yx[:, 1]]
        if max_kp is not None and max_kp > 0 and max_kp < len(yx):
            v, idx = torch.topk(v, max_kp, dim=0, sorted=True)
            yx = yx[idx]
        out.append((yx, v))
    return out


def sample_descriptors(kp_xy, dense, mode, cell=8):
    """Bilinear descriptor lookup at integer keypoints (x,y), then L2-normalise.

    mode 'legacy' : superpoint.py:117-133 (align_corners=True, shifted grid)
    mode 'fixed'  : superpoint.py:138-149
    mode 'open'   : superpoint_open.py:20-31
    kp_xy (B,K,2) float, dense (B,C,h,w)  ->  (B,C,K)
    """
    b, c, h, w = dense.shape
    kp = kp_xy.clone()
    if mode == "legacy":
        kp = kp - cell / 2 + 0.5
        kp = kp / torch.tensor([(w * cell - cell / 2 - 0.5), (h * cell - cell / 2 - 0.5)]).to(kp)[None]
        align = True
    elif mode == "fixed":
        kp = kp / (kp.new_tensor([w, h]) * cell)
        align = False
    elif mode == "open":
        kp = (kp + 0.5) / (kp.new_tensor([w, h]) * cell)
        align = False
    else:
        raise ValueError(mode)
    kp = kp * 2 - 1
    d = F.grid_sample(dense, kp.view(b, 1, -1, 2), mode="bilinear", align_corners=align)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


SP_DEFAULTS = dict(
    variant="magicleap",
    nms_radius=4,
    detection_threshold=0.005,
    max_num_keypoints=-1,
    remove_borders=4,
    legacy_sampling=True,
)


def superpoint_forward(data, W, force_yx=None, dense=None, **conf):
    """Sparse SuperPoint forward.  Returns per-image lists plus stacked tensors when counts agree.

    `force_yx`: per-image list of (n,2) int64 (y,x) pixels that replaces the detector's own selection (the scores are
    read from this run's heat map at those pixels) -- used to evaluate the float outputs of another run's decisions.
    `dense`: a (heatmap, dense descriptors) pair already computed by `sp_dense` for this input (skips the dense stage).

    Output keys: keypoints (x,y)+0.5, keypoint_scores, descriptors (K,256) per image;
    `heatmap`, `dense_descriptors` always returned for stage-wise checks.
    """
    cfg = {**SP_DEFAULTS, **conf}
    variant = cfg["variant"]
    heat, dense = sp_dense(data["image"], W, variant) if dense is None else dense
    s = nms_heatmap(heat, cfg["nms_radius"])
    s = mask_borders(s, cfg["remove_borders"], data.get("image_size"), honour_image_size=(variant == "magicleap"))
    k = cfg["max_num_keypoints"]
    if force_yx is not None:
        sel = [(yx, heat[i][yx[:, 0], yx[:, 1]]) for i, yx in enumerate(force_yx)]  # the raw map: a forced pixel need not survive THIS run's NMS
    else:
        sel = select_keypoints(s, cfg["detection_threshold"], None if (k is None or k <= 0) else k)
    if variant == "open":
        mode = "open"
    else:
        mode = "legacy" if cfg["legacy_sampling"] else "fixed"
    kps, scs, descs = [], [], []
    for i, (yx, v) in enumerate(sel):
        xy = torch.flip(yx, [1]).to(heat.dtype)
        d = sample_descriptors(xy[None], dense[i:i + 1], mode)[0]
        kps.append(xy + 0.5)
        scs.append(v)
        descs.append(d.t().contiguous())
    out = {
        "keypoints_list": kps,
        "keypoint_scores_list": scs,
        "descriptors_list": descs,
        "heatmap": heat,
        "nms_scores": s,
        "dense_descriptors": dense,
    }
    if len({len(x) for x in kps}) == 1:
        out["keypoints"] = torch.stack(kps)
        out["keypoint_scores"] = torch.stack(scs)
        out["descriptors"] = torch.stack(descs)
    return out


# --------------------------------------------------------------------------
# LightGlue
# --------------------------------------------------------------------------
def normalize_keypoints(kpts, size=None):
    """lightglue.py:22-33 -- centre on the image and divide by half the longer side."""
    if size is None:
        size = 1 + kpts.max(-2).values - kpts.min(-2).values
    size = torch.as_tensor(size).to(kpts)
    if size.dim() == 1:
        size = size[None].expand(kpts.shape[0], -1)
    shift = size / 2
    scale = size.max(-1).values / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def fourier_encoding(kpts, Wr):
    """lightglue.py:46-59 -- (2,B,1,K,64): cos/sin of Wr*kp, each frequency repeated twice."""
    proj = F.linear(kpts, Wr)
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


def _rot_pairs(t):
    """lightglue.py:36-39 -- (x0,x1) -> (-x1,x0) on consecutive pairs."""
    a, b = t.unflatten(-1, (-1, 2)).unbind(-1)
    return torch.stack((-b, a), -1).flatten(-2)


def apply_rotary(enc, t):
    return t * enc[0] + _rot_pairs(t) * enc[1]


def _ffn(x, msg, W, p):
    """lightglue.py:137-142 -- Linear(512,512) -> LayerNorm -> GELU(erf) -> Linear(512,256)."""
    h = F.linear(torch.cat([x, msg], -1), W[p + ".ffn.0.weight"], W[p + ".ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), W[p + ".ffn.1.weight"], W[p + ".ffn.1.bias"], 1e-5)
    h = F.gelu(h)
    return F.linear(h, W[p + ".ffn.3.weight"], W[p + ".ffn.3.bias"])


def self_block(x, enc, W, p, heads=4):
    """lightglue.py:144-157.  Wqkv output feature = head*192 + dim*3 + {q,k,v}."""
    qkv = F.linear(x, W[p + ".Wqkv.weight"], W[p + ".Wqkv.bias"])
    qkv = qkv.unflatten(-1, (heads, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q, k = apply_rotary(enc, q), apply_rotary(enc, k)
    # lightglue.py:101-102 -- constructing the reference with `flash: False` calls
    # torch.backends.cuda.enable_flash_sdp(False) process-wide, so its CPU run takes SDPA's MATH path.
    with sdpa_kernel(SDPBackend.MATH):
        ctx = F.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous())
    msg = F.linear(ctx.transpose(1, 2).flatten(-2), W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])
    return x + _ffn(x, msg, W, p)


def cross_block(x0, x1, W, p, heads=4):
    """lightglue.py:186-215 (einsum branch :200-210): one similarity, softmax along both axes."""
    def split(t):
        return t.unflatten(-1, (heads, -1)).transpose(1, 2)

    qk0 = split(F.linear(x0, W[p + ".to_qk.weight"], W[p + ".to_qk.bias"]))
    qk1 = split(F.linear(x1, W[p + ".to_qk.weight"], W[p + ".to_qk.bias"]))
    v0 = split(F.linear(x0, W[p + ".to_v.weight"], W[p + ".to_v.bias"]))
    v1 = split(F.linear(x1, W[p + ".to_v.weight"], W[p + ".to_v.bias"]))
    s = (qk0.shape[-1] ** -0.5) ** 0.5
    qk0, qk1 = qk0 * s, qk1 * s
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    a01 = torch.softmax(sim, -1)
    a10 = torch.softmax(sim.transpose(-2, -1).contiguous(), -1)
    m0 = torch.einsum("bhij,bhjd->bhid", a01, v1)
    m1 = torch.einsum("bhji,bhjd->bhid", a10.transpose(-2, -1), v0)
    m0 = F.linear(m0.transpose(1, 2).flatten(-2), W[p + ".to_out.weight"], W[p + ".to_out.bias"])
    m1 = F.linear(m1.transpose(1, 2).flatten(-2), W[p + ".to_out.weight"], W[p + ".to_out.bias"])
    return x0 + _ffn(x0, m0, W, p), x1 + _ffn(x1, m1, W, p)


def log_assignment(d0, d1, W, p):
    """lightglue.py:250-284 -- (B,M+1,N+1) log assignment with matchability bins."""
    dim = d0.shape[-1]
    m0 = F.linear(d0, W[p + ".final_proj.weight"], W[p + ".final_proj.bias"]) / dim ** 0.25
    m1 = F.linear(d1, W[p + ".final_proj.weight"], W[p + ".final_proj.bias"]) / dim ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", m0, m1)
    z0 = F.linear(d0, W[p + ".matchability.weight"], W[p + ".matchability.bias"])
    z1 = F.linear(d1, W[p + ".matchability.weight"], W[p + ".matchability.bias"])
    b, m, n = sim.shape
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    rows = F.log_softmax(sim, 2)
    cols = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    S = sim.new_zeros((b, m + 1, n + 1))
    S[:, :m, :n] = rows + cols + cert
    S[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    S[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return S, sim


def filter_matches(S, th):
    """lightglue.py:287-303 -- mutual arg-max over the inner (M,N) block, exp(score) > th."""
    inner = S[:, :-1, :-1]
    r, c = inner.max(2), inner.max(1)
    m0, m1 = r.indices, c.indices
    i0 = torch.arange(m0.shape[1])[None]
    i1 = torch.arange(m1.shape[1])[None]
    mut0 = i0 == m1.gather(1, m0)
    mut1 = i1 == m0.gather(1, m1)
    e = r.values.exp()
    zero = e.new_tensor(0)
    ms0 = torch.where(mut0, e, zero)
    ms1 = torch.where(mut1, ms0.gather(1, m1), zero)
    ok0 = mut0 & (ms0 > th)
    ok1 = mut1 & ok0.gather(1, m1)
    return torch.where(ok0, m0, -1), torch.where(ok1, m1, -1), ms0, ms1, r.indices, c.indices


LG_DEFAULTS = dict(n_layers=9, num_heads=4, filter_threshold=0.0, depth_confidence=-1, width_confidence=-1)


def lg_confidence_threshold(layer_index, n_layers):
    """lightglue.py:536-539."""
    return float(min(1.0, max(0.0, 0.8 + 0.1 * math.exp(-4.0 * layer_index / n_layers))))


def lightglue_forward(data, W, collect_layers=(), replay=None, **conf):
    """lightglue.py:399-525.  With depth_confidence / width_confidence <= 0 (every shipped config) the plain 9-layer pass.

    width_confidence > 0 restates the point pruning of :480-497 / :541-549 / :500-508 (it runs in the reference; pinned by
    tests/golden/lightglue_adaptive.npz).  depth_confidence > 0 restates the early stopping of :474-479 / :551-560 with the
    thresholds of `confidence_threshold()`: the fork reads an undefined `self.confidence_thresholds` there and raises, so this
    half follows the evident intent (= upstream LightGlue) and has no reference output to be pinned to.  Both need batch 1.
    `replay`: the `_decisions` list of another run (per layer: stop flag, kept indices of image 0 / 1); the adaptive decisions are
    then taken from it instead of from this run's own confidences (fp64 anchor on the fp32 run's decisions)."""
    cfg = {**LG_DEFAULTS, **conf}
    k0, k1 = data["keypoints0"], data["keypoints1"]
    size0 = size1 = None
    if "view0" in data and "view1" in data:
        size0 = data["view0"].get("image_size")
        size1 = data["view1"].get("image_size")
    k0 = normalize_keypoints(k0, size0)
    k1 = normalize_keypoints(k1, size1)
    d0 = data["descriptors0"].contiguous()
    d1 = data["descriptors1"].contiguous()
    if "input_proj.weight" in W:
        d0 = F.linear(d0, W["input_proj.weight"], W["input_proj.bias"])
        d1 = F.linear(d1, W["input_proj.weight"], W["input_proj.bias"])
    e0 = fourier_encoding(k0, W["posenc.Wr.weight"])
    e1 = fourier_encoding(k1, W["posenc.Wr.weight"])
    L, H = cfg["n_layers"], cfg["num_heads"]
    do_stop, do_prune = cfg["depth_confidence"] > 0, cfg["width_confidence"] > 0
    b, m, n = d0.shape[0], d0.shape[1], d1.shape[1]
    if do_stop or do_prune:
        assert b == 1
    ind0, ind1 = torch.arange(m)[None], torch.arange(n)[None]
    prune0, prune1 = torch.ones_like(ind0), torch.ones_like(ind1)
    layers = {}
    decisions = []
    last = L - 1
    for i in range(L):
        p = f"transformers.{i}"
        d0 = self_block(d0, e0, W, p + ".self_attn", H)
        d1 = self_block(d1, e1, W, p + ".self_attn", H)
        if ("self", i) in collect_layers:
            layers[("self", i)] = (d0.clone(), d1.clone())
        d0, d1 = cross_block(d0, d1, W, p + ".cross_attn", H)
        if i in collect_layers:
            layers[i] = (d0.clone(), d1.clone())
        if i == L - 1 or not (do_stop or do_prune):
            continue
        thr = lg_confidence_threshold(i, L)
        t0 = t1 = None
        if do_stop:  # :62-72, :551-560
            tw, tb = W[f"token_confidence.{i}.token.0.weight"], W[f"token_confidence.{i}.token.0.bias"]
            t0, t1 = torch.sigmoid(F.linear(d0, tw, tb)).squeeze(-1), torch.sigmoid(F.linear(d1, tw, tb)).squeeze(-1)
            conf_all = torch.cat([t0, t1], -1)
            stop = bool(1.0 - (conf_all < thr).float().sum() / (m + n) > cfg["depth_confidence"])
            if replay is not None:
                stop = replay[i][0]
            if stop:
                decisions.append((True, None, None))
                last = i
                break
        dec = [False, None, None]
        if do_prune:  # :480-497
            mw, mb = W[f"log_assignment.{i}.matchability.weight"], W[f"log_assignment.{i}.matchability.bias"]
            for side in (0, 1):
                d, t = (d0, t0) if side == 0 else (d1, t1)
                keep = torch.sigmoid(F.linear(d, mw, mb)).squeeze(-1) > (1 - cfg["width_confidence"])
                if t is not None:
                    keep = keep | (t <= thr)
                kidx = torch.where(keep)[1]
                if replay is not None:
                    kidx = replay[i][1 + side]
                dec[1 + side] = kidx
                if side == 0:
                    ind0, d0, e0 = ind0.index_select(1, kidx), d0.index_select(1, kidx), e0.index_select(-2, kidx)
                    prune0[:, ind0] += 1
                else:
                    ind1, d1, e1 = ind1.index_select(1, kidx), d1.index_select(1, kidx), e1.index_select(-2, kidx)
                    prune1[:, ind1] += 1
        decisions.append(tuple(dec))
    S, sim = log_assignment(d0, d1, W, f"log_assignment.{last}")
    m0, m1, ms0, ms1, am0, am1 = filter_matches(S, cfg["filter_threshold"])
    if do_prune:  # :500-508
        m0_ = torch.full((b, m), -1, dtype=m0.dtype)
        m1_ = torch.full((b, n), -1, dtype=m1.dtype)
        m0_[:, ind0] = torch.where(m0 == -1, -1, ind1.gather(1, m0.clamp(min=0)))
        m1_[:, ind1] = torch.where(m1 == -1, -1, ind0.gather(1, m1.clamp(min=0)))
        s0_, s1_ = ms0.new_zeros(b, m), ms0.new_zeros(b, n)
        s0_[:, ind0] = ms0
        s1_[:, ind1] = ms1
        m0, m1, ms0, ms1 = m0_, m1_, s0_, s1_
    else:
        prune0, prune1 = torch.ones_like(ms0) * L, torch.ones_like(ms1) * L
    return {
        "matches0": m0,
        "matches1": m1,
        "matching_scores0": ms0,
        "matching_scores1": ms1,
        "ref_descriptors0": d0[:, None],
        "ref_descriptors1": d1[:, None],
        "log_assignment": S,
        "prune0": prune0,
        "prune1": prune1,
        "_argmax0": am0,
        "_argmax1": am1,
        "_sim": sim,
        "_layers": layers,
        "_layers_run": last + 1,
        "_decisions": decisions,
        "_ind0": ind0,
        "_ind1": ind1,
    }


# --------------------------------------------------------------------------
# Two-view composite
# --------------------------------------------------------------------------
def two_view_forward(data, W_sp, W_lg, sp_conf, lg_conf, with_maps=False, force_yx=None, dense=None, matcher=None):
    """two_view_pipeline.py:62-90 -- extractor per view, keys suffixed 0/1, then the matcher.
    `with_maps` also returns the score maps behind the keypoint decisions (`_nms_scores0/1`, `_heatmap0/1`) for
    the tie-aware index checks of the tests; they are not part of the reference's pred dict.
    `force_yx` = (list per image of view 0, list per image of view 1) of (n,2) (y,x) pixels, `dense` = (dense stage of view 0,
    of view 1): see superpoint_forward.  `matcher`: the matcher's forward (default lightglue_forward; superglue_forward for
    BASELINE.json configs[3]) -- called as matcher({**data, **pred}, W_lg, **lg_conf)."""
    pred, maps = {}, {}
    for i in ("0", "1"):
        p = superpoint_forward(data["view" + i], W_sp, force_yx=None if force_yx is None else force_yx[int(i)],
                               dense=None if dense is None else dense[int(i)], **sp_conf)
        for k in ("keypoints", "keypoint_scores", "descriptors"):
            pred[k + i] = p[k]
        if sp_conf.get("variant") == "open":
            pred["heatmap" + i] = p["heatmap"]
        maps["_nms_scores" + i], maps["_heatmap" + i] = p["nms_scores"], p["heatmap"]
    lg = (matcher or lightglue_forward)({**data, **pred}, W_lg, **lg_conf)
    pred.update({k: v for k, v in lg.items() if not k.startswith("_")})
    if with_maps:
        pred.update(maps)
        if "_sim" in lg:
            pred["_sim"] = lg["_sim"]
    return pred


# --------------------------------------------------------------------------
# Seeded weights (same key names / shapes as the reference modules)
# --------------------------------------------------------------------------
def _conv_init(g, cout, cin, k, gain=1.4, bias_scale=0.05):
    """Zero-sum filters (no DC response) with He-style gain: keeps activations O(1), ~50 % sparse and
    spatially diverse through 10 layers, so seeded weights give peaky heatmaps and distinct descriptors."""
    w = torch.randn(cout, cin, k, k, generator=g) * gain / math.sqrt(cin * k * k)
    w = w - w.mean(dim=(1, 2, 3), keepdim=True)
    b = torch.randn(cout, generator=g) * bias_scale
    return w, b


def make_sp_weights(seed=0, variant="magicleap"):
    """Seeded SuperPoint weights under the reference's state_dict key names."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    chans = [(1, 64), (64, 64), (64, 64), (64, 64), (64, 128), (128, 128), (128, 128), (128, 128)]
    if variant == "magicleap":
        names = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"]
        for n, (ci, co) in zip(names, chans):
            W[n + ".weight"], W[n + ".bias"] = _conv_init(g, co, ci, 3)
        for n, (ci, co, k) in {"convPa": (128, 256, 3), "convPb": (256, 65, 1), "convDa": (128, 256, 3), "convDb": (256, 256, 1)}.items():
            W[n + ".weight"], W[n + ".bias"] = _conv_init(g, co, ci, k)
        W["conv1a.weight"] = W["conv1a.weight"] * 8.0  # images are in [0,1]
        W["convPb.weight"] = W["convPb.weight"] * 4.0  # spread the detector logits -> real peaks
    else:
        def bn(prefix, c):
            W[prefix + ".bn.weight"] = 0.75 + 0.5 * torch.rand(c, generator=g)
            W[prefix + ".bn.bias"] = 0.2 * (torch.rand(c, generator=g) - 0.5)
            W[prefix + ".bn.running_mean"] = 0.2 * (torch.rand(c, generator=g) - 0.3)
            W[prefix + ".bn.running_var"] = 0.5 + torch.rand(c, generator=g)

        for i in range(4):
            for j in range(2):
                ci, co = chans[2 * i + j]
                p = f"backbone.{i}.{j}"
                W[p + ".conv.weight"], W[p + ".conv.bias"] = _conv_init(g, co, ci, 3)
                bn(p, co)
        for head, cout in (("detector", 65), ("descriptor", 256)):
            W[head + ".0.conv.weight"], W[head + ".0.conv.bias"] = _conv_init(g, 256, 128, 3)
            bn(head + ".0", 256)
            W[head + ".1.conv.weight"], W[head + ".1.conv.bias"] = _conv_init(g, cout, 256, 1)
            bn(head + ".1", cout)
        W["backbone.0.0.conv.weight"] = W["backbone.0.0.conv.weight"] * 8.0
        W["detector.1.conv.weight"] = W["detector.1.conv.weight"] * 4.0
    return W


def make_lg_weights(seed=0, n_layers=9, dim=256, structured=True, final_gain=4.0, match_bias=0.0):
    """Seeded LightGlue weights.

    `structured=True` scales the transformer so the residual stream keeps the
    input descriptors recognisable and sets final_proj = I, which makes the
    assignment non-degenerate (random weights yield ~1 mutual match in 512).
    """
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f, gain=1.0):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound * gain
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound * gain
        return w, b

    gain = 0.3 if structured else 1.0
    W = {"posenc.Wr.weight": torch.randn(32, 2, generator=g)}
    for i in range(n_layers):
        for blk, names in (("self_attn", (("Wqkv", 3 * dim, dim), ("out_proj", dim, dim))),
                           ("cross_attn", (("to_qk", dim, dim), ("to_v", dim, dim), ("to_out", dim, dim)))):
            p = f"transformers.{i}.{blk}"
            for n, o, ii in names:
                W[f"{p}.{n}.weight"], W[f"{p}.{n}.bias"] = lin(o, ii, 1.0 if n in ("Wqkv", "to_qk") else gain)
            W[p + ".ffn.0.weight"], W[p + ".ffn.0.bias"] = lin(2 * dim, 2 * dim)
            W[p + ".ffn.1.weight"] = 0.75 + 0.5 * torch.rand(2 * dim, generator=g)
            W[p + ".ffn.1.bias"] = 0.1 * (torch.rand(2 * dim, generator=g) - 0.5)
            W[p + ".ffn.3.weight"], W[p + ".ffn.3.bias"] = lin(dim, 2 * dim, gain)
        p = f"log_assignment.{i}"
        W[p + ".matchability.weight"], W[p + ".matchability.bias"] = lin(1, dim)
        if structured:
            W[p + ".final_proj.weight"] = torch.eye(dim) * final_gain
            W[p + ".matchability.bias"] = W[p + ".matchability.bias"] + match_bias
            W[p + ".final_proj.bias"] = torch.zeros(dim)
        else:
            W[p + ".final_proj.weight"], W[p + ".final_proj.bias"] = lin(dim, dim)
        if i < n_layers - 1:
            W[f"token_confidence.{i}.token.0.weight"], W[f"token_confidence.{i}.token.0.bias"] = lin(1, dim)
    return W


# --------------------------------------------------------------------------
# SuperGlue (gluefactory_nonfree/superglue.py:208-307)
# --------------------------------------------------------------------------
def sg_normalize_keypoints(kpts, size):
    """superglue.py:74-86 -- centre on the image, divide by 0.7 * longer side."""
    size = torch.as_tensor(size).to(kpts)
    if size.dim() == 1:
        size = size[None]
    shift = size / 2
    scale = size.max(1).values * 0.7
    return (kpts - shift[:, None]) / scale[:, None, None]


def _sg_mlp(x, W, prefix, n_layers):
    """superglue.py:63-72 -- Conv1d(k=1) [+ BatchNorm1d + ReLU] chain on (B,C,N); last layer is plain."""
    idx = 0
    for i in range(n_layers):
        x = F.conv1d(x, W[f"{prefix}.{idx}.weight"], W[f"{prefix}.{idx}.bias"])
        idx += 1
        if i < n_layers - 1:
            x = F.batch_norm(x, W[f"{prefix}.{idx}.running_mean"], W[f"{prefix}.{idx}.running_var"],
                             W[f"{prefix}.{idx}.weight"], W[f"{prefix}.{idx}.bias"], training=False, eps=1e-5)
            x = F.relu(x)
            idx += 2
    return x


def _sg_propagate(x, src, W, p, heads=4):
    """superglue.py:105-141 -- multi-head attention (channel = dim*heads + head) + merge + MLP([512,512,256])."""
    b = x.shape[0]
    dim = x.shape[1] // heads
    q = F.conv1d(x, W[p + ".attn.proj.0.weight"], W[p + ".attn.proj.0.bias"]).view(b, dim, heads, -1)
    k = F.conv1d(src, W[p + ".attn.proj.1.weight"], W[p + ".attn.proj.1.bias"]).view(b, dim, heads, -1)
    v = F.conv1d(src, W[p + ".attn.proj.2.weight"], W[p + ".attn.proj.2.bias"]).view(b, dim, heads, -1)
    scores = torch.einsum("bdhn,bdhm->bhnm", q, k) / dim ** 0.5
    prob = torch.softmax(scores, dim=-1)
    msg = torch.einsum("bhnm,bdhm->bdhn", prob, v).contiguous().view(b, dim * heads, -1)
    msg = F.conv1d(msg, W[p + ".attn.merge.weight"], W[p + ".attn.merge.bias"])
    return _sg_mlp(torch.cat([x, msg], 1), W, p + ".mlp", 2)


def log_optimal_transport(scores, alpha, iters):
    """superglue.py:173-201 -- log-domain Sinkhorn with a dustbin row/column."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one), (n * one)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    al = alpha.expand(b, 1, 1)
    Z = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, al], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1) - norm


SG_DEFAULTS = dict(n_layers=18, num_sinkhorn_iterations=50, filter_threshold=0.2)


def superglue_forward(data, W, **conf):
    """superglue.py:253-307 (use_scores=True, keypoint_encoder [32,64,128,256], descriptor_dim 256)."""
    cfg = {**SG_DEFAULTS, **conf}
    d0 = data["descriptors0"].transpose(-1, -2)
    d1 = data["descriptors1"].transpose(-1, -2)
    k0 = sg_normalize_keypoints(data["keypoints0"], data["view0"]["image_size"])
    k1 = sg_normalize_keypoints(data["keypoints1"], data["view1"]["image_size"])
    enc0 = _sg_mlp(torch.cat([k0.transpose(1, 2), data["keypoint_scores0"].unsqueeze(1)], 1), W, "kenc.encoder", 5)
    enc1 = _sg_mlp(torch.cat([k1.transpose(1, 2), data["keypoint_scores1"].unsqueeze(1)], 1), W, "kenc.encoder", 5)
    d0, d1 = d0 + enc0, d1 + enc1
    taps = {"kenc": (d0.clone(), d1.clone())}
    for i in range(cfg["n_layers"]):
        p = f"gnn.layers.{i}"
        if i % 2 == 0:
            e0, e1 = _sg_propagate(d0, d0, W, p), _sg_propagate(d1, d1, W, p)
        else:
            e0, e1 = _sg_propagate(d0, d1, W, p), _sg_propagate(d1, d0, W, p)
        d0, d1 = d0 + e0, d1 + e1
        if i in (0, 1, cfg["n_layers"] - 1):
            taps[i] = (d0.clone(), d1.clone())
    m0 = F.conv1d(d0, W["final_proj.weight"], W["final_proj.bias"])
    m1 = F.conv1d(d1, W["final_proj.weight"], W["final_proj.bias"])
    cost = torch.einsum("bdn,bdm->bnm", m0, m1) / 256 ** 0.5
    S = log_optimal_transport(cost, W["bin_score"], cfg["num_sinkhorn_iterations"])
    mt0, mt1, ms0, ms1, am0, am1 = filter_matches(S, cfg["filter_threshold"])
    return {"sinkhorn_cost": cost, "log_assignment": S, "matches0": mt0, "matches1": mt1,
            "matching_scores0": ms0, "matching_scores1": ms1, "_taps": taps}


def make_sg_weights(seed=0, n_layers=18, final_gain=6.0, bin_score=1.0):
    """Seeded SuperGlue weights under the reference's state_dict key names (superglue.py:208-250)."""
    g = torch.Generator().manual_seed(seed)
    W = {}

    def conv(name, co, ci, gain=1.0, zero_bias=False):
        W[name + ".weight"] = torch.randn(co, ci, 1, generator=g) * gain / math.sqrt(ci)
        W[name + ".bias"] = torch.zeros(co) if zero_bias else torch.randn(co, generator=g) * 0.05

    def bn(name, c):
        W[name + ".weight"] = 0.75 + 0.5 * torch.rand(c, generator=g)
        W[name + ".bias"] = 0.2 * (torch.rand(c, generator=g) - 0.5)
        W[name + ".running_mean"] = 0.2 * (torch.rand(c, generator=g) - 0.5)
        W[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        W[name + ".num_batches_tracked"] = torch.tensor(0)

    chans = [3, 32, 64, 128, 256, 256]
    idx = 0
    for i in range(5):
        conv(f"kenc.encoder.{idx}", chans[i + 1], chans[i], gain=0.3 if i == 4 else 1.4, zero_bias=(i == 4))
        idx += 1
        if i < 4:
            bn(f"kenc.encoder.{idx}", chans[i + 1])
            idx += 2
    for l in range(n_layers):
        p = f"gnn.layers.{l}"
        conv(p + ".attn.merge", 256, 256, gain=0.5)
        for j in range(3):
            conv(p + f".attn.proj.{j}", 256, 256, gain=1.0)
        conv(p + ".mlp.0", 512, 512, gain=1.4)
        bn(p + ".mlp.1", 512)
        conv(p + ".mlp.3", 256, 512, gain=0.3, zero_bias=True)
    W["final_proj.weight"] = (torch.eye(256) * final_gain)[:, :, None].contiguous()
    W["final_proj.bias"] = torch.zeros(256)
    W["bin_score"] = torch.tensor(float(bin_score))
    return W


# --------------------------------------------------------------------------
# Nearest-neighbour matcher
# --------------------------------------------------------------------------
def _nn_find(sim, ratio_thresh, distance_thresh):
    """nearest_neighbor_matcher.py:15-26."""
    sim_nn, ind_nn = sim.topk(2 if ratio_thresh else 1, dim=-1, largest=True)
    dist_nn = 2 * (1 - sim_nn)
    mask = torch.ones(ind_nn.shape[:-1], dtype=torch.bool)
    if ratio_thresh:
        mask = mask & (dist_nn[..., 0] <= (ratio_thresh ** 2) * dist_nn[..., 1])
    if distance_thresh:
        mask = mask & (dist_nn[..., 0] <= distance_thresh ** 2)
    return torch.where(mask, ind_nn[..., 0], ind_nn.new_tensor(-1))


def nn_forward(data, ratio_thresh=None, distance_thresh=None, mutual_check=True):
    """nearest_neighbor_matcher.py:51-72 (mutual check :29-36)."""
    sim = torch.einsum("bnd,bmd->bnm", data["descriptors0"], data["descriptors1"])
    m0 = _nn_find(sim, ratio_thresh, distance_thresh)
    m1 = _nn_find(sim.transpose(1, 2), ratio_thresh, distance_thresh)
    if mutual_check:
        i0 = torch.arange(m0.shape[-1])
        i1 = torch.arange(m1.shape[-1])
        loop0 = torch.gather(m1, -1, torch.where(m0 > -1, m0, m0.new_tensor(0)))
        loop1 = torch.gather(m0, -1, torch.where(m1 > -1, m1, m1.new_tensor(0)))
        m0, m1 = (torch.where((m0 > -1) & (i0 == loop0), m0, m0.new_tensor(-1)),
                  torch.where((m1 > -1) & (i1 == loop1), m1, m1.new_tensor(-1)))
    b, m, n = sim.shape
    la = sim.new_zeros(b, m + 1, n + 1)
    la[:, :-1, :-1] = F.log_softmax(sim, -1) + F.log_softmax(sim, -2)
    return {"matches0": m0, "matches1": m1, "matching_scores0": (m0 > -1).float(), "matching_scores1": (m1 > -1).float(),
            "similarity": sim, "log_assignment": la}


# --------------------------------------------------------------------------
# GlueStick: points + line endpoints in one graph
# --------------------------------------------------------------------------
def gs_log_double_softmax(scores, bin_score):
    """gluestick.py:758-770 -- mean of the row / column log-softmax, each with its own dustbin."""
    b, m, n = scores.shape
    bin_ = bin_score[None, None, None]
    s0 = F.log_softmax(torch.cat([scores, bin_.expand(b, m, 1)], 2), 2)
    s1 = F.log_softmax(torch.cat([scores, bin_.expand(b, 1, n)], 1), 1)
    out = scores.new_full((b, m + 1, n + 1), 0)
    out[:, :m, :n] = (s0[:, :, :n] + s1[:, :m, :]) / 2
    out[:, :-1, -1] = s0[:, :, -1]
    out[:, -1, :-1] = s1[:, -1, :]
    return out


def _gs_endpoint_encoding(lines, scores, W):
    """gluestick.py:498-517 -- per endpoint [x, y, dx, dy, score] -> MLP; lines (B,n,2,2) normalised."""
    b, n = lines.shape[:2]
    off = (lines[:, :, 1] - lines[:, :, 0]).unsqueeze(2)
    off = torch.cat([off, -off], 2).reshape(b, 2 * n, 2).transpose(1, 2)
    x = torch.cat([lines.flatten(1, 2).transpose(1, 2), off, scores.repeat(1, 2).unsqueeze(1)], 1)
    return _sg_mlp(x, W, "lenc.encoder", 5)


def _gs_line_layer(d, enc, idx, W, p):
    """gluestick.py:581-612, 656-689 (line_attention=False): message per endpoint = [own junction, the line's other
    junction, endpoint encoding] -> MLP([768,512,256]) -> mean over the endpoints that share a junction."""
    b, dim = d.shape[:2]
    ld = torch.gather(d, 2, idx[:, None].repeat(1, dim, 1))
    ld2 = ld.reshape(b, dim, -1, 2).flip([-1]).flatten(2, 3)
    upd = _sg_mlp(torch.cat([ld, ld2, enc], 1), W, p + ".mlp", 2)
    out = torch.zeros_like(d).scatter_reduce_(2, idx[:, None].repeat(1, dim, 1), upd, reduce="mean", include_self=False)
    return d + out


def _gs_line_scores(ld0, ld1, idx0, idx1, W, proj):
    """gluestick.py:331-373 -- endpoint similarity, gathered per line, best of the two endpoint orders."""
    m0 = F.conv1d(ld0, W[proj + ".weight"], W[proj + ".bias"])
    m1 = F.conv1d(ld1, W[proj + ".weight"], W[proj + ".bias"])
    s = torch.einsum("bdn,bdm->bnm", m0, m1) / 256 ** 0.5
    n0, n1 = idx0.shape[1], idx1.shape[1]
    s = torch.gather(s, 2, idx1[:, None, :].repeat(1, s.shape[1], 1))
    s = torch.gather(s, 1, idx0[:, :, None].repeat(1, 1, n1))
    s = s.reshape(-1, n0 // 2, 2, n1 // 2, 2)
    return 0.5 * torch.maximum(s[:, :, 0, :, 0] + s[:, :, 1, :, 1], s[:, :, 0, :, 1] + s[:, :, 1, :, 0])


GS_DEFAULTS = dict(n_layers=18, filter_threshold=0.2)


def gluestick_forward(data, W, **conf):
    """gluestick.py:136-312 (input_dim == descriptor_dim == 256, num_line_iterations 1, no line attention,
    no intermediate supervision) for inputs with at least one keypoint and one line per image."""
    cfg = {**GS_DEFAULTS, **conf}
    d0 = data["descriptors0"].transpose(-1, -2)
    d1 = data["descriptors1"].transpose(-1, -2)
    b = d0.shape[0]
    s0, s1 = data["view0"]["image_size"], data["view1"]["image_size"]
    n0, n1 = data["lines0"].shape[1], data["lines1"].shape[1]
    idx0, idx1 = data["lines_junc_idx0"].flatten(1, 2), data["lines_junc_idx1"].flatten(1, 2)
    k0 = sg_normalize_keypoints(data["keypoints0"], s0)
    k1 = sg_normalize_keypoints(data["keypoints1"], s1)
    d0 = d0 + _sg_mlp(torch.cat([k0.transpose(1, 2), data["keypoint_scores0"].unsqueeze(1)], 1), W, "kenc.encoder", 5)
    d1 = d1 + _sg_mlp(torch.cat([k1.transpose(1, 2), data["keypoint_scores1"].unsqueeze(1)], 1), W, "kenc.encoder", 5)
    l0 = sg_normalize_keypoints(data["lines0"].flatten(1, 2), s0).reshape(b, n0, 2, 2)
    l1 = sg_normalize_keypoints(data["lines1"].flatten(1, 2), s1).reshape(b, n1, 2, 2)
    e0 = _gs_endpoint_encoding(l0, data["line_scores0"], W)
    e1 = _gs_endpoint_encoding(l1, data["line_scores1"], W)
    taps = {"kenc": (d0.clone(), d1.clone()), "lenc": (e0.clone(), e1.clone())}
    for i in range(cfg["n_layers"]):
        p = f"gnn.layers.{i}.update"
        if i % 2 == 0:
            u0, u1 = _sg_propagate(d0, d0, W, p), _sg_propagate(d1, d1, W, p)
        else:
            u0, u1 = _sg_propagate(d0, d1, W, p), _sg_propagate(d1, d0, W, p)
        d0, d1 = d0 + u0, d1 + u1
        if i % 2 == 0:
            q = f"gnn.line_layers.{i // 2}"
            d0, d1 = _gs_line_layer(d0, e0, idx0, W, q), _gs_line_layer(d1, e1, idx1, W, q)
        if i in (0, 1, cfg["n_layers"] - 1):
            taps[i] = (d0.clone(), d1.clone())
    m0 = F.conv1d(d0, W["final_proj.weight"], W["final_proj.bias"])
    m1 = F.conv1d(d1, W["final_proj.weight"], W["final_proj.bias"])
    S = gs_log_double_softmax(torch.einsum("bdn,bdm->bnm", m0, m1) / 256 ** 0.5, W["bin_score"])
    mt0, mt1, ms0, ms1, _, _ = filter_matches(S, cfg["filter_threshold"])
    raw = _gs_line_scores(d0[:, :, : 2 * n0], d1[:, :, : 2 * n1], idx0, idx1, W, "final_line_proj")
    LS = gs_log_double_softmax(raw, W["line_bin_score"])
    lt0, lt1, ls0, ls1, _, _ = filter_matches(LS, cfg["filter_threshold"])
    return {"log_assignment": S, "matches0": mt0, "matches1": mt1, "matching_scores0": ms0, "matching_scores1": ms1,
            "line_log_assignment": LS, "line_matches0": lt0, "line_matches1": lt1, "line_matching_scores0": ls0,
            "line_matching_scores1": ls1, "raw_line_scores": raw, "_taps": taps}


# --------------------------------------------------------------------------
# Wireframe extractor (the side feeding GlueStick)
# --------------------------------------------------------------------------
def corner_sample(points, dense, s=8):
    """wireframe.py:8-20 `sample_descriptors_corner_conv`: bilinear sample of the dense map at raw pixel coordinates
    (align_corners=False), L2-normalised.  points (B,n,2), dense (B,C,h,w) -> (B,n,C)."""
    b, c, h, w = dense.shape
    g = points / (points.new_tensor([w, h]) * s) * 2 - 1
    d = F.grid_sample(dense, g.view(b, 1, -1, 2), mode="bilinear", align_corners=False)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1).transpose(1, 2)


def wireframe_forward(sp_pred, lines, line_scores, nms_radius=3, merge_points=True, merge_line_endpoints=True):
    """wireframe.py:163-306 for one image (batch 1, no forced padding): `sp_pred` holds the point extractor's
    `keypoints`, `keypoint_scores`, `descriptors`, `dense_descriptors`; lines (1,n,2,2), line_scores (1,n)."""
    from sklearn.cluster import DBSCAN  # the reference's own CPU dependency for this step (wireframe.py:2,58)

    kp, sc, ds, dense = sp_pred["keypoints"], sp_pred["keypoint_scores"], sp_pred["descriptors"], sp_pred["dense_descriptors"]
    out = {"lines": lines, "line_scores": line_scores}
    n_lines = lines.shape[1]
    if n_lines:
        out["line_scores"] = line_scores / (line_scores.max(dim=1)[0][:, None] + 1e-8)
    if merge_points:  # :184-216
        ends = lines.reshape(1, -1, 2)
        dist = torch.norm(kp[:, :, None] - ends[:, None], dim=-1)
        rm = torch.any(dist < nms_radius, dim=2)[0]
        kp, sc, ds = kp[0][~rm][None], sc[0][~rm][None], ds[0][~rm][None]
    orig = lines.clone()
    if merge_line_endpoints and n_lines:  # :222-268 with lines_to_wireframe (:23-127)
        ends = lines[0].reshape(-1, 2)
        labels = torch.tensor(DBSCAN(eps=nms_radius, min_samples=1).fit(ends.numpy()).labels_, dtype=torch.long)
        n = len(set(labels.tolist()))
        junc = torch.zeros(n, 2).scatter_reduce_(0, labels[:, None].repeat(1, 2), ends, reduce="mean", include_self=False)
        jsc = torch.zeros(n).scatter_reduce_(0, labels, torch.repeat_interleave(out["line_scores"][0], 2), reduce="mean", include_self=False)
        out["lines"] = junc[labels].reshape(1, -1, 2, 2)
        idx = labels.reshape(1, -1, 2)
        jd = corner_sample(junc[None], dense)
        pts, scores, descs = torch.cat([junc[None], kp], 1), torch.cat([jsc[None], sc], 1), torch.cat([jd, ds], 1)
        assoc = torch.eye(pts.shape[1], dtype=torch.bool)
        assoc[idx[0][:, 0], idx[0][:, 1]] = True
        assoc[idx[0][:, 1], idx[0][:, 0]] = True
        nj = n
    else:  # :269-296
        ends = lines.reshape(1, -1, 2)
        pts = torch.cat([ends, kp], 1)
        scores = torch.cat([torch.repeat_interleave(out["line_scores"], 2, dim=1), sc], 1)
        descs = torch.cat([corner_sample(ends, dense), ds], 1)
        assoc = torch.eye(pts.shape[1], dtype=torch.bool)
        idx = torch.arange(n_lines * 2).reshape(1, -1, 2)
        nj = n_lines * 2
    out.update(keypoints=pts, keypoint_scores=scores, descriptors=descs, pl_associativity=assoc[None], num_junctions=torch.tensor([nj]),
               orig_lines=orig, lines_junc_idx=idx)
    return out


# the weight set of tests/golden/gluestick.npz: points and lines of the synthetic pairs really match (see make_gs_weights)
GS_GOLDEN = dict(structured=True, final_gain=10.0, bin_score=0.0, line_bin_score=0.0)


def make_gs_weights(seed=0, n_layers=18, final_gain=6.0, bin_score=1.0, line_bin_score=0.5, structured=False):
    """Seeded GlueStick weights under the reference's state_dict key names (gluestick.py:62-100, 459-470, 561-700).

    `structured=True` keeps the residual stream close to the input descriptors (small gains on everything that is
    ADDED to it: the last encoder layer, `mlp.3` of the GNN and line layers) so that corresponding points and lines of
    the synthetic pairs really match -- the plain random set matches 1-2 % of them, which makes an "agreement" figure
    mostly agreement on -1."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    add_gain = 0.02 if structured else 0.3

    def conv(name, co, ci, gain=1.0, zero_bias=False):
        W[name + ".weight"] = torch.randn(co, ci, 1, generator=g) * gain / math.sqrt(ci)
        W[name + ".bias"] = torch.zeros(co) if zero_bias else torch.randn(co, generator=g) * 0.05

    def bn(name, c):
        W[name + ".weight"] = 0.75 + 0.5 * torch.rand(c, generator=g)
        W[name + ".bias"] = 0.2 * (torch.rand(c, generator=g) - 0.5)
        W[name + ".running_mean"] = 0.2 * (torch.rand(c, generator=g) - 0.5)
        W[name + ".running_var"] = 0.5 + torch.rand(c, generator=g)
        W[name + ".num_batches_tracked"] = torch.tensor(0)

    for enc, cin in (("kenc", 3), ("lenc", 5)):
        chans = [cin, 32, 64, 128, 256, 256]
        idx = 0
        for i in range(5):
            conv(f"{enc}.encoder.{idx}", chans[i + 1], chans[i], gain=add_gain if i == 4 else 1.4, zero_bias=(i == 4))
            idx += 1
            if i < 4:
                bn(f"{enc}.encoder.{idx}", chans[i + 1])
                idx += 2
    for l in range(n_layers):
        p = f"gnn.layers.{l}.update"
        conv(p + ".attn.merge", 256, 256, gain=0.5)
        for j in range(3):
            conv(p + f".attn.proj.{j}", 256, 256, gain=1.0)
        conv(p + ".mlp.0", 512, 512, gain=1.4)
        bn(p + ".mlp.1", 512)
        conv(p + ".mlp.3", 256, 512, gain=add_gain, zero_bias=True)
    for l in range(n_layers // 2):
        p = f"gnn.line_layers.{l}"
        conv(p + ".mlp.0", 512, 768, gain=1.4)
        bn(p + ".mlp.1", 512)
        conv(p + ".mlp.3", 256, 512, gain=add_gain, zero_bias=True)
    eye = torch.eye(256)[:, :, None]
    W["final_proj.weight"] = final_gain * eye + 0.05 * torch.randn(256, 256, 1, generator=g)
    W["final_proj.bias"] = torch.zeros(256)
    W["final_line_proj.weight"] = final_gain * eye + 0.05 * torch.randn(256, 256, 1, generator=g)
    W["final_line_proj.bias"] = torch.zeros(256)
    W["bin_score"] = torch.tensor(float(bin_score))
    W["line_bin_score"] = torch.tensor(float(line_bin_score))
    return W
