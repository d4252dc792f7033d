#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF on CPU.

Runs only in the build container (needs /root/reference).  The reference's
modules are imported unmodified; two pieces of tooling make that possible
offline, neither of which touches arithmetic:
  * `oracle/tools/_shim/omegaconf` -- config-merge stand-in (package not installed);
  * `torch.hub.load_state_dict_from_url` is pointed at a local seeded
    state_dict, because gluefactory_nonfree/superpoint.py:198-200 downloads its
    checkpoint unconditionally and there is no network.
Weights are the seeded sets from `oracle/sp_lg_oracle.py` (same key names as
the reference), loaded with the reference's own `load_state_dict`.

Usage:  python oracle/tools/make_goldens.py   (writes tests/golden/)
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import sp_lg_oracle as O  # noqa: E402  (weight generators only)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)
torch.set_grad_enabled(False)


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrs)} arrays")


def load_boat():
    from PIL import Image

    out = []
    for n in ("boat1.png", "boat2.png"):
        im = Image.open(f"/root/reference/assets/{n}").convert("RGB")
        out.append(np.asarray(im))
    return out


# ---------------------------------------------------------------- reference models
import gluefactory_nonfree.superpoint as ref_spn  # noqa: E402
from gluefactory.models import get_model  # noqa: E402
from gluefactory.models.extractors import superpoint_open as ref_spo  # noqa: E402
from gluefactory.models.two_view_pipeline import TwoViewPipeline  # noqa: E402

RefLG = get_model("matchers.lightglue")


def ref_sp_magicleap(conf, W):
    old = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: W
    try:
        m = ref_spn.SuperPoint(conf).eval()
    finally:
        torch.hub.load_state_dict_from_url = old
    return m


def ref_sp_open(conf, W, tmp="/tmp/_spo_seed.pth"):
    torch.save(W, tmp)
    return ref_spo.SuperPoint({**conf, "weights": tmp}).eval()


def ref_lg(conf, W):
    m = RefLG({**conf, "weights": None}).eval()
    missing, unexpected = m.load_state_dict(W, strict=False)
    assert not unexpected and all("loss" in k for k in missing), (missing, unexpected)
    return m


# ---------------------------------------------------------------- (1) nms + top-k
def gold_nms_topk():
    g = torch.Generator().manual_seed(11)
    s = torch.rand(2, 64, 96, generator=g)
    # exact ties and plateaus
    s[0, 10:13, 20:23] = 0.99
    s[0, 40, 50] = s[0, 40, 57] = 0.97
    s[1] = (s[1] * 16).floor() / 16  # heavy quantisation -> many ties
    s[1, 30:34, 60:70] = 1.0
    arrs = {"scores": s}
    for r in (3, 4):
        arrs[f"nms_r{r}"] = ref_spn.simple_nms(s, r)
        arrs[f"nms_open_r{r}"] = ref_spo.batched_nms(s, r)
    # top-k on the NMS'd, border-masked map of image 0 (distinct values) and image 1 (ties)
    for b in (0, 1):
        m = arrs["nms_r3"][b].clone()
        m[:4] = -1; m[:, :4] = -1; m[-4:] = -1; m[:, -4:] = -1
        yx = torch.nonzero(m > 0.0)
        v = m[yx[:, 0], yx[:, 1]]
        arrs[f"cand_yx_{b}"] = yx
        arrs[f"cand_v_{b}"] = v
        for k in (16, 10 ** 6):
            kk, vv = ref_spn.top_k_keypoints(yx, v, k)
            arrs[f"top{k}_yx_{b}"] = kk
            arrs[f"top{k}_v_{b}"] = vv
    save("nms_topk", **arrs)


# ---------------------------------------------------------------- (2,3) SuperPoint forward
def gold_superpoint(boat):
    img = torch.from_numpy(boat[0]).float().permute(2, 0, 1)[None] / 255.0  # (1,3,680,850)
    gray = (img * img.new_tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1)).sum(1, keepdim=True)
    crops = {
        "g64x96": gray[:, :, 200:264, 300:396],
        "rgb64x96": img[:, :, 200:264, 300:396],
        "g68x85": gray[:, :, 100:168, 400:485],
        "b2_g48x64": torch.cat([gray[:, :, 200:248, 300:364], gray[:, :, 300:348, 100:164]], 0),
    }
    W_ml = O.make_sp_weights(0, "magicleap")
    W_op = O.make_sp_weights(0, "open")
    arrs = {}
    for cname, im in crops.items():
        arrs[f"{cname}.image"] = im
        for variant, W, mk in (("ml", W_ml, ref_sp_magicleap), ("open", W_op, ref_sp_open)):
            for r, k, th in ((3, 24, 0.0), (4, -1, 0.005)):
                if im.shape[0] > 1 and k < 0:
                    continue  # ragged counts cannot be stacked by the reference
                if im.shape[0] > 1:
                    k = 12  # the reference only batches with force_num_keypoints (its list path calls .transpose on a list)
                conf = {"nms_radius": r, "detection_threshold": th, "max_num_keypoints": k if (k > 0 or variant == "ml") else None,
                        "remove_borders": 4, "dense_outputs": True, "force_num_keypoints": im.shape[0] > 1}
                m = mk(conf, W)
                p = m({"image": im})
                tag = f"{cname}.{variant}.r{r}k{k}"
                arrs[tag + ".keypoints"] = p["keypoints"]
                arrs[tag + ".keypoint_scores"] = p["keypoint_scores"]
                arrs[tag + ".descriptors"] = p["descriptors"]
                arrs[f"{cname}.{variant}.dense_descriptors"] = p["dense_descriptors"]  # independent of r,k
                if "heatmap" in p:
                    arrs[tag + ".heatmap"] = p["heatmap"]
    # image_size handling of the MagicLeap variant (square-padded input, config 3 style)
    im = torch.zeros(1, 1, 64, 96)
    im[:, :, :48, :80] = crops["g64x96"][:, :, :48, :80]
    m = ref_sp_magicleap({"nms_radius": 3, "detection_threshold": 0.0, "max_num_keypoints": 16, "dense_outputs": True}, W_ml)
    p = m({"image": im, "image_size": torch.tensor([[80, 48]])})
    arrs["padded.image"] = im
    for k in ("keypoints", "keypoint_scores", "descriptors"):
        arrs["padded.ml.r3k16." + k] = p[k]
    # non-legacy sampling
    m = ref_sp_magicleap({"nms_radius": 3, "detection_threshold": 0.0, "max_num_keypoints": 24, "legacy_sampling": False}, W_ml)
    p = m({"image": crops["g64x96"]})
    for k in ("keypoints", "keypoint_scores", "descriptors"):
        arrs["g64x96.ml_fixed.r3k24." + k] = p[k]
    save("superpoint", **arrs)

    # (3) sampling functions on their own
    g = torch.Generator().manual_seed(5)
    dense = torch.nn.functional.normalize(torch.randn(2, 256, 8, 12, generator=g), dim=1)
    kp = torch.stack([torch.randint(0, 96, (2, 40), generator=g), torch.randint(0, 64, (2, 40), generator=g)], -1).float()
    kp[0, 0] = torch.tensor([0.0, 0.0]); kp[0, 1] = torch.tensor([95.0, 63.0]); kp[0, 2] = torch.tensor([95.0, 0.0])
    save(
        "sampling",
        dense=dense,
        kp=kp,
        legacy=ref_spn.sample_descriptors(kp.clone(), dense, 8),
        fixed=ref_spn.sample_descriptors_fix_sampling(kp.clone(), dense, 8),
        open=ref_spo.sample_descriptors(kp.clone(), dense, 8),
    )


# ---------------------------------------------------------------- (4) LightGlue
def gold_lightglue():
    arrs = {}
    for tag, structured, M, N, seed in (("s32", True, 32, 32, 1), ("s33", True, 33, 29, 2), ("r32", False, 32, 32, 3), ("s64", True, 64, 48, 4)):
        g = torch.Generator().manual_seed(100 + seed)
        W = O.make_lg_weights(seed, structured=structured)
        B = 2
        size0 = torch.tensor([[640.0, 480.0]] * B)
        size1 = torch.tensor([[600.0, 480.0]] * B)
        k0 = torch.rand(B, M, 2, generator=g) * size0[:, None] * 0.98 + 0.5
        d0 = torch.nn.functional.normalize(torch.randn(B, M, 256, generator=g), dim=-1)
        # image 1 = permuted, perturbed subset of image 0 so real matches exist
        perm = torch.stack([torch.randperm(M, generator=g)[:N] for _ in range(B)])
        k1 = torch.gather(k0, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 3.0 * torch.randn(B, N, 2, generator=g)
        d1 = torch.gather(d0, 1, perm[..., None].expand(-1, -1, 256)) + 0.05 * torch.randn(B, N, 256, generator=g)
        d1 = torch.nn.functional.normalize(d1, dim=-1)
        data = {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1,
                "view0": {"image_size": size0}, "view1": {"image_size": size1}}
        for k, v in (("keypoints0", k0), ("keypoints1", k1), ("descriptors0", d0), ("descriptors1", d1), ("size0", size0), ("size1", size1)):
            arrs[f"{tag}.{k}"] = v
        for th in (0.0, 0.1):
            m = ref_lg({"filter_threshold": th}, W)
            hooks = {}
            if th == 0.0:
                def mk(i):
                    def hook(mod, inp, out):
                        hooks[i] = out
                    return hook
                hs = [m.transformers[i].register_forward_hook(mk(i)) for i in (0, 8)]
                hs.append(m.transformers[0].self_attn.register_forward_hook(lambda mod, inp, out: hooks.setdefault("self0", []).append(out)))
            p = m(data)
            if th == 0.0:
                for h in hs:
                    h.remove()
                for i in (0, 8):
                    arrs[f"{tag}.layer{i}.desc0"], arrs[f"{tag}.layer{i}.desc1"] = hooks[i]
                arrs[f"{tag}.self0.desc0"], arrs[f"{tag}.self0.desc1"] = hooks["self0"]
                arrs[f"{tag}.log_assignment"] = p["log_assignment"]
                inner = p["log_assignment"][:, :-1, :-1]
                arrs[f"{tag}.argmax0"] = inner.max(2).indices
                arrs[f"{tag}.argmax1"] = inner.max(1).indices
                arrs[f"{tag}.ref_descriptors0"] = p["ref_descriptors0"]
                arrs[f"{tag}.prune0"] = p["prune0"]
                # positional encoding on its own
                from gluefactory.models.matchers.lightglue import normalize_keypoints
                arrs[f"{tag}.kn0"] = normalize_keypoints(k0, size0)
                arrs[f"{tag}.enc0"] = m.posenc(normalize_keypoints(k0, size0))
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
                arrs[f"{tag}.th{th}.{k}"] = p[k]
    save("lightglue", **arrs)


# ---------------------------------------------------------------- (4b) SuperGlue
def lg_adaptive_pair(M, N, seed):
    g = torch.Generator().manual_seed(seed)
    size = torch.tensor([[640.0, 480.0]])
    k0 = torch.rand(1, M, 2, generator=g) * size * 0.98 + 0.5
    d0 = torch.nn.functional.normalize(torch.randn(1, M, 256, generator=g), dim=-1)
    perm = (torch.randperm(max(M, N), generator=g)[:N] % M)[None]
    k1 = torch.gather(k0, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 2.0 * torch.randn(1, N, 2, generator=g)
    d1 = torch.nn.functional.normalize(torch.gather(d0, 1, perm[..., None].expand(-1, -1, 256)) + 0.03 * torch.randn(1, N, 256, generator=g), dim=-1)
    return {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1, "view0": {"image_size": size}, "view1": {"image_size": size}}


def gold_lightglue_adaptive():
    """Point pruning (`width_confidence` > 0, lightglue.py:480-508) of the in-tree LightGlue run as is.  (`depth_confidence` > 0
    raises AttributeError in this fork -- `self.confidence_thresholds` is never defined, :556 -- so early stopping has no
    reference output; it is restated with the thresholds of `confidence_threshold()` and tested against the oracle only.)"""
    Ref = get_model("matchers.lightglue")
    arrs = {}
    for tag, M, N, wseed, dseed, wc in (("a", 200, 180, 11, 5, 0.57), ("b", 200, 180, 12, 5, 0.55), ("c", 97, 130, 13, 6, 0.58)):
        W = O.make_lg_weights(wseed, structured=True, final_gain=3.0, match_bias=0.0)
        data = lg_adaptive_pair(M, N, dseed)
        m = Ref({"weights": None, "width_confidence": wc, "filter_threshold": 0.1}).eval()
        missing, unexpected = m.load_state_dict(W, strict=False)
        assert not missing and not unexpected
        p = m(data)
        for k in ("keypoints0", "keypoints1", "descriptors0", "descriptors1"):
            arrs[f"{tag}.{k}"] = data[k]
        arrs[f"{tag}.size"] = data["view0"]["image_size"]
        arrs[f"{tag}.width_confidence"] = torch.tensor(wc)
        for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "log_assignment", "prune0", "prune1", "ref_descriptors0", "ref_descriptors1"):
            arrs[f"{tag}.{k}"] = p[k]
        try:
            Ref({"weights": None, "depth_confidence": 0.95, "filter_threshold": 0.1}).eval()(data)
            raise SystemExit("the reference's early stopping ran: pin it with a golden instead of the restatement")
        except AttributeError as e:
            assert "confidence_thresholds" in str(e)
        print("lightglue adaptive", tag, "survivors", tuple(p["log_assignment"].shape), "matches", int((p["matches0"] > -1).sum()),
              "prune0 histogram", torch.bincount(p["prune0"][0]).tolist())
    save("lightglue_adaptive", **arrs)


def gold_superglue():
    import gluefactory_nonfree.superglue as ref_sg

    arrs = {}
    for tag, B, M, N, seed in (("a", 2, 40, 33, 3), ("b", 1, 64, 64, 4), ("c", 2, 17, 100, 5)):
        g = torch.Generator().manual_seed(200 + seed)
        W = O.make_sg_weights(seed)
        size0 = torch.tensor([[640.0, 480.0]] * B)
        size1 = torch.tensor([[600.0, 480.0]] * B)
        K = max(M, N)
        kk = torch.rand(B, K, 2, generator=g) * size1[:, None] * 0.98 + 0.5
        dd = torch.nn.functional.normalize(torch.randn(B, K, 256, generator=g), dim=-1)
        k0, d0 = kk[:, :M].contiguous(), dd[:, :M].contiguous()
        perm = torch.stack([torch.randperm(K, generator=g)[:N] for _ in range(B)])
        k1 = (torch.gather(kk, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 3.0 * torch.randn(B, N, 2, generator=g)).clamp(1.0, 479.0)
        d1 = torch.nn.functional.normalize(torch.gather(dd, 1, perm[..., None].expand(-1, -1, 256)) + 0.05 * torch.randn(B, N, 256, generator=g), dim=-1)
        s0, s1 = torch.rand(B, M, generator=g), torch.rand(B, N, generator=g)
        data = {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1, "keypoint_scores0": s0, "keypoint_scores1": s1,
                "view0": {"image_size": size0, "image": torch.zeros(B, 1, 480, 640)}, "view1": {"image_size": size1, "image": torch.zeros(B, 1, 480, 600)}}
        for k in ("keypoints0", "keypoints1", "descriptors0", "descriptors1", "keypoint_scores0", "keypoint_scores1"):
            arrs[f"{tag}.{k}"] = data[k]
        arrs[f"{tag}.size0"], arrs[f"{tag}.size1"] = size0, size1
        for th in (0.0, 0.2):
            m = ref_sg.SuperGlue({"weights": None, "filter_threshold": th}).eval()
            missing, unexpected = m.load_state_dict(W, strict=False)
            assert not missing and not unexpected
            p = m(data)
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
                arrs[f"{tag}.th{th}.{k}"] = p[k]
            if th == 0.0:
                arrs[f"{tag}.log_assignment"] = p["log_assignment"]
                arrs[f"{tag}.sinkhorn_cost"] = p["sinkhorn_cost"]
        print("superglue", tag, "matches", int((p["matches0"] > -1).sum()), "of", B * M)
    save("superglue", **arrs)


# ---------------------------------------------------------------- (5) two-view pipeline on the boat pair
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gs_inputs import gs_inputs  # noqa: E402  (the synthetic wireframe-style pair generator, shared with the tests)


def gold_gluestick():
    Ref = get_model("matchers.gluestick")
    arrs = {}
    for tag, B, M, N, n0, n1, seed in (("a", 2, 48, 40, 10, 8, 1), ("b", 1, 64, 64, 16, 16, 2), ("c", 2, 33, 70, 5, 12, 3)):
        W = O.make_gs_weights(seed, **O.GS_GOLDEN)
        data = gs_inputs(B, M, N, n0, n1, seed, corresponding=True)
        for k, v in data.items():
            if k.startswith("view"):
                arrs[f"{tag}.size{k[-1]}"] = v["image_size"]
            else:
                arrs[f"{tag}.{k}"] = v
        for th in (0.0, 0.2):
            m = Ref({"weights": None, "filter_threshold": th}).eval()
            missing, unexpected = m.load_state_dict(W, strict=False)
            assert not missing and not unexpected, (missing, unexpected)
            p = m(data)
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "line_matches0", "line_matches1",
                      "line_matching_scores0", "line_matching_scores1"):
                arrs[f"{tag}.th{th}.{k}"] = p[k]
            if th == 0.0:
                for k in ("log_assignment", "line_log_assignment", "raw_line_scores"):
                    arrs[f"{tag}.{k}"] = p[k]
        print("gluestick", tag, "matches", int((p["matches0"] > -1).sum()), "of", B * M, f"(possible {B * min(M, N)})", "line matches",
              int((p["line_matches0"] > -1).sum()), "of", B * n0, f"(possible {B * min(n0, n1)})")
    save("gluestick", **arrs)


def wireframe_segments(kp, n, seed, w, h):
    """Synthetic segments for the wireframe fixture (LSD is a CPU dependency and not part of the path): endpoints near
    detected keypoints (so the distance filter has something to remove), endpoints shared or 1-2 px apart between
    segments (so DBSCAN merges them), and free ones.  No keypoint-endpoint distance falls within 1e-3 of the radius."""
    g = torch.Generator().manual_seed(seed)
    ends = torch.rand(n, 2, 2, generator=g) * torch.tensor([w - 9.0, h - 9.0]) + 4.0
    k = kp[torch.randperm(len(kp), generator=g)[: n // 2]]
    ends[: n // 2, 0] = k + (torch.rand(n // 2, 2, generator=g) - 0.5) * 3.0      # within ~2 px of a keypoint
    ends[1: n // 3, 1] = ends[: n // 3 - 1, 0] + (torch.rand(n // 3 - 1, 2, generator=g) - 0.5) * 2.0  # chains: next segment starts where one ends
    ends[n // 2, 0] = ends[0, 1]                                                    # an exactly shared endpoint
    return ends[None].contiguous(), (torch.rand(1, n, generator=g) * 5 + 0.1)


def gold_wireframe(boat):
    """gluefactory/models/lines/wireframe.py run as is on fixed extractor outputs: `WireframeExtractor._forward` (and through it
    `lines_to_wireframe`, `sample_descriptors_corner_conv`) with its two sub-extractors replaced by callables that return the
    SuperPoint outputs of a boat crop and synthetic segments -- everything downstream of the extractors is the reference's code."""
    from omegaconf import OmegaConf

    from gluefactory.models.lines import wireframe as ref_wf

    gray = torch.from_numpy(np.asarray(boat[0]).astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)) / 255.0
    img = gray[200:320, 300:460][None, None].contiguous()
    sp = O.superpoint_forward({"image": img}, O.make_sp_weights(0), nms_radius=3, detection_threshold=0.0, max_num_keypoints=96)
    sp_pred = {k: sp[k] for k in ("keypoints", "keypoint_scores", "descriptors", "dense_descriptors")}
    arrs = {"image": img, **{"sp." + k: v for k, v in sp_pred.items()}}
    for tag, n, merge_ends in (("merge", 24, True), ("plain", 10, False), ("nolines", 0, True)):
        lines, scores = wireframe_segments(sp_pred["keypoints"][0], max(n, 8), 7 + n, 160, 120)
        lines, scores = lines[:, :n], scores[:, :n]
        d = torch.norm(sp_pred["keypoints"][:, :, None] - lines.reshape(1, -1, 2)[:, None], dim=-1)
        assert not bool(((d - 3.0).abs() < 1e-3).any()), "a keypoint sits on the filter radius: re-seed the fixture"
        ext = object.__new__(ref_wf.WireframeExtractor)
        torch.nn.Module.__init__(ext)
        ext.conf = OmegaConf.create({"point_extractor": {"force_num_keypoints": False}, "line_extractor": {"force_num_lines": False, "max_num_lines": None},
                                     "wireframe_params": {"merge_points": True, "merge_line_endpoints": merge_ends, "nms_radius": 3}})
        ext.point_extractor = lambda data: {k: v.clone() for k, v in sp_pred.items()}
        ext.line_extractor = lambda data: {"lines": lines.clone(), "line_scores": scores.clone()}
        out = ext._forward({"image": img})
        arrs[tag + ".in.lines"], arrs[tag + ".in.line_scores"] = lines, scores
        for k, v in out.items():
            arrs[f"{tag}.{k}"] = v.to(torch.uint8) if v.dtype == torch.bool else v
        print("wireframe", tag, "keypoints kept", out["keypoints"].shape[1] - int(out["num_junctions"][0]), "of", sp_pred["keypoints"].shape[1],
              "junctions", int(out["num_junctions"][0]), "of", 2 * n, "endpoints")
    save("wireframe", **arrs)


def export_fixture(n_pairs=4, K=12, seed=41):
    """Deterministic fake loader + model for the export test: every pair detects a different number of keypoints."""
    g = torch.Generator().manual_seed(seed)
    pairs = []
    for i in range(n_pairs):
        n0, n1 = K - (i % 3), K - ((i + 1) % 4)
        item = {}
        for sfx, n in (("0", n0), ("1", n1)):
            item["keypoints" + sfx] = torch.rand(n, 2, generator=g) * 100
            item["keypoint_scores" + sfx] = torch.rand(n, generator=g).sort(descending=True).values * 0.9 + 0.05
            item["descriptors" + sfx] = torch.randn(n, 8, generator=g)
        m0 = torch.randint(-1, n1, (n0,), generator=g)
        item["matches0"], item["matching_scores0"] = m0, (m0 > -1).float() * 0.5
        m1 = torch.randint(-1, n0, (n1,), generator=g)
        item["matches1"], item["matching_scores1"] = m1, (m1 > -1).float() * 0.5
        scales0, scales1 = torch.rand(2, generator=g) + 0.5, torch.rand(2, generator=g) + 0.5
        pairs.append((f"seq{i // 2}/pair{i}.ppm", item, scales0, scales1))
    return pairs


def gold_export():
    """gluefactory/utils/export_predictions.py run as is (batch 1, as it only supports) on the fixture; h5py is the
    in-memory stand-in of oracle/tools/_shim."""
    import h5py
    from gluefactory.utils.export_predictions import export_predictions

    pairs = export_fixture()
    loader = [{"name": [name], "idx": torch.tensor([i]), "view0": {"scales": s0[None]}, "view1": {"scales": s1[None]}}
              for i, (name, _, s0, s1) in enumerate(pairs)]

    class Model(torch.nn.Module):
        def forward(self, data):
            return {k: v[None] for k, v in pairs[int(data["idx"][0])][1].items()}

    arrs = {}
    for half in (False, True):
        out = f"/tmp/export_gold_{int(half)}.h5"
        export_predictions(loader, Model(), out, as_half=half)
        for name, grp in h5py.FILES[out].items():
            for k, ds in grp.items():
                arrs[f"half{int(half)}|{name}|{k}"] = ds.data
        # Round trip through the consumer: THIS build's batched exporter (zero-padded batches of 3) writes the fixture, the
        # reference's own CacheLoader (models/cache_loader.py:59-141) reads both files back; what it returns must be the same
        # tensors, and is kept as the fixture of the read-back test.
        sys.path.insert(0, os.path.join(ROOT, "glue-factory_amd"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_export_h5 import _batched_loader, _model

        from gluefactory.models.cache_loader import CacheLoader
        from gluefactory_amd.export import export_predictions as export_batched

        ours = f"/tmp/export_ours_{int(half)}.h5"
        export_batched(_batched_loader(pairs, 3), _model, ours, as_half=half, h5=h5py)
        for i, (name, _, s0, s1) in enumerate(pairs):
            data = {"name": [name], "view0": {"scales": s0[None]}, "view1": {"scales": s1[None]}}
            got = {}
            for path in (out, ours):
                cl = CacheLoader({"path": path, "add_data_path": False, "collate": False})
                got[path] = cl(data)
            assert sorted(got[out]) == sorted(got[ours]), (name, sorted(got[out]), sorted(got[ours]))
            for k in got[out]:
                assert got[out][k].dtype == got[ours][k].dtype and torch.equal(got[out][k], got[ours][k]), (name, k)
                arrs[f"cache{int(half)}|{name}|{k}"] = got[ours][k]
        print(f"export (half={half}): the reference CacheLoader reads {len(pairs)} pairs x {len(got[ours])} keys identically from the reference's and from the batched export")
    save("export_h5", **arrs)


def gold_nn():
    """gluefactory/models/matchers/nearest_neighbor_matcher.py run as is; descriptors with clear margins
    (noisy permuted copies) so that the arg-maxes do not hinge on the last bit of a dot product."""
    Ref = get_model("matchers.nearest_neighbor_matcher")
    arrs = {}
    for tag, B, M, N, D, seed in (("a", 2, 48, 40, 256, 21), ("b", 1, 100, 131, 128, 22), ("c", 3, 17, 64, 64, 23)):
        g = torch.Generator().manual_seed(seed)
        d0 = torch.nn.functional.normalize(torch.randn(B, M, D, generator=g), dim=-1)
        perm = torch.stack([torch.randperm(max(M, N), generator=g)[:N] % M for _ in range(B)])
        d1 = torch.gather(d0, 1, perm[..., None].expand(-1, -1, D)) + (0.7 / D ** 0.5) * torch.randn(B, N, D, generator=g)
        d1 = torch.nn.functional.normalize(d1, dim=-1)
        arrs[f"{tag}.descriptors0"], arrs[f"{tag}.descriptors1"] = d0, d1
        for ci, conf in enumerate(({}, {"ratio_thresh": 0.9}, {"distance_thresh": 0.8}, {"ratio_thresh": 0.95, "distance_thresh": 0.9, "mutual_check": False})):
            p = Ref(conf).eval()({"descriptors0": d0, "descriptors1": d1})
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
                arrs[f"{tag}.c{ci}.{k}"] = p[k]
            if ci == 0:
                arrs[f"{tag}.similarity"] = p["similarity"]
                arrs[f"{tag}.log_assignment"] = p["log_assignment"]
    save("nn_matcher", **arrs)


def gold_two_view(boat):
    g0 = np.round(boat[0].astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)).astype(np.uint8)
    g1 = np.round(boat[1].astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)).astype(np.uint8)
    arrs = {"gray0": g0, "gray1": g1}
    im0 = torch.from_numpy(g0).float()[None, None] / 255.0
    im1 = torch.from_numpy(g1).float()[None, None] / 255.0
    W_lg = O.make_lg_weights(7, structured=True, final_gain=8.0, match_bias=2.0)
    for variant in ("open", "ml"):
        W_sp = O.make_sp_weights(0, "open" if variant == "open" else "magicleap")
        if variant == "open":
            torch.save(W_sp, "/tmp/_spo_seed.pth")
            ext = {"name": "extractors.superpoint_open", "max_num_keypoints": 512, "nms_radius": 3, "detection_threshold": 0.0,
                   "weights": "/tmp/_spo_seed.pth"}
        else:
            ext = {"name": "gluefactory_nonfree.superpoint", "max_num_keypoints": 512, "nms_radius": 3, "detection_threshold": 0.0}
        conf = {"extractor": ext, "matcher": {"name": "matchers.lightglue", "filter_threshold": 0.1, "weights": None}}
        old = torch.hub.load_state_dict_from_url
        torch.hub.load_state_dict_from_url = lambda *a, **k: W_sp
        try:
            pipe = TwoViewPipeline(conf).eval()
        finally:
            torch.hub.load_state_dict_from_url = old
        pipe.matcher.load_state_dict(W_lg, strict=False)
        size = torch.tensor([[im0.shape[-1], im0.shape[-2]]]).float()
        data = {"view0": {"image": im0, "image_size": size}, "view1": {"image": im1, "image_size": size}}
        p = pipe(data)
        arrs[f"{variant}.keys"] = np.array(sorted(p.keys()))
        for k in ("keypoints0", "keypoints1", "keypoint_scores0", "keypoint_scores1", "matches0", "matches1",
                  "matching_scores0", "matching_scores1", "prune0"):
            arrs[f"{variant}.{k}"] = p[k]
        arrs[f"{variant}.descriptors0_rows"] = p["descriptors0"][0, ::8]
        arrs[f"{variant}.descriptors1_rows"] = p["descriptors1"][0, ::8]
        arrs[f"{variant}.desc0_sum"] = p["descriptors0"].double().sum(-1)
        arrs[f"{variant}.log_assignment_rowsum"] = p["log_assignment"].double().sum(-1)
        arrs[f"{variant}.log_assignment_diag"] = p["log_assignment"][0, ::16, ::16]
        print(variant, "matches:", int((p["matches0"] > -1).sum()), "of", p["matches0"].shape[1])
        # shifted self-pair: view1 = view0 translated by (24,16) px -> SuperPoint is equivariant to
        # multiples of 8 px, so genuine correspondences exist even with seeded weights
        a = im0[:, :, 0:640, 0:800].contiguous()
        b = im0[:, :, 16:656, 24:824].contiguous()
        size = torch.tensor([[800.0, 640.0]])
        p = pipe({"view0": {"image": a, "image_size": size}, "view1": {"image": b, "image_size": size}})
        for k in ("keypoints0", "keypoints1", "matches0", "matches1", "matching_scores0", "matching_scores1"):
            arrs[f"{variant}.shift.{k}"] = p[k]
        print(variant, "shifted-pair matches:", int((p["matches0"] > -1).sum()), "of", p["matches0"].shape[1])
    save("two_view_boat", **arrs)


if __name__ == "__main__":
    boat = load_boat()
    gold_nms_topk()
    gold_superpoint(boat)
    gold_lightglue()
    gold_lightglue_adaptive()
    gold_superglue()
    gold_nn()
    gold_export()
    gold_gluestick()
    gold_wireframe(boat)
    gold_two_view(boat)
