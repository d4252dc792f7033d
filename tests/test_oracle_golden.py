"""Pin the CPU oracle (oracle/sp_lg_oracle.py) to golden vectors produced by the reference itself.

The goldens were generated in the build container by oracle/tools/make_goldens.py, which imports
/root/reference and runs its modules on CPU.  Same torch build + same operators => the oracle is
expected to agree bit-for-bit; a small float tolerance is allowed only where thread count may
change the reduction order on another host (conv / linear), indices stay exact.
"""
import torch

from oracle import sp_lg_oracle as O

torch.set_grad_enabled(False)
FTOL = dict(rtol=0, atol=2e-6)


def test_nms_matches_reference(golden):
    g = golden("nms_topk")
    s = g["scores"]
    for r in (3, 4):
        out = O.nms_heatmap(s, r)
        assert torch.equal(out, g[f"nms_r{r}"])
        assert torch.equal(out, g[f"nms_open_r{r}"])  # both reference variants agree


def test_select_topk_matches_reference(golden):
    g = golden("nms_topk")
    for b in (0, 1):
        m = O.mask_borders(g["nms_r3"][b:b + 1], 4)
        for k in (16, 10 ** 6):
            (yx, v), = O.select_keypoints(m, 0.0, k)
            assert torch.equal(v, g[f"top{k}_v_{b}"])
            if b == 0 or k > 16:  # image 1 has exact score ties: order among ties is torch's
                assert torch.equal(yx, g[f"top{k}_yx_{b}"])
            else:
                assert torch.equal(yx, g[f"top{k}_yx_{b}"])


def test_sampling_modes_match_reference(golden):
    g = golden("sampling")
    for mode in ("legacy", "fixed", "open"):
        out = O.sample_descriptors(g["kp"], g["dense"], mode)
        assert torch.equal(out, g[mode]), mode


def _sp_cases(g):
    for key in g.keys():
        if key.endswith(".keypoints"):
            yield key[: -len(".keypoints")]


def test_superpoint_matches_reference(golden):
    g = golden("superpoint")
    W = {"ml": O.make_sp_weights(0, "magicleap"), "open": O.make_sp_weights(0, "open")}
    n = 0
    for tag in _sp_cases(g):
        crop, variant, rk = tag.split(".")
        r = int(rk[1])
        k = int(rk.split("k")[1])
        conf = dict(
            variant="open" if variant == "open" else "magicleap",
            nms_radius=r,
            detection_threshold=0.0 if r == 3 else 0.005,
            max_num_keypoints=k,
            legacy_sampling=(variant != "ml_fixed"),
        )
        data = {"image": g[crop + ".image"]}
        if crop == "padded":
            data["image_size"] = torch.tensor([[80, 48]])
        p = O.superpoint_forward(data, W["open" if variant == "open" else "ml"], **conf)
        assert torch.equal(p["keypoints"], g[tag + ".keypoints"]), tag
        torch.testing.assert_close(p["keypoint_scores"], g[tag + ".keypoint_scores"], **FTOL)
        torch.testing.assert_close(p["descriptors"], g[tag + ".descriptors"], **FTOL)
        dkey = f"{crop}.{variant}.dense_descriptors"
        if dkey in g:
            torch.testing.assert_close(p["dense_descriptors"], g[dkey], **FTOL)
        if tag + ".heatmap" in g:
            torch.testing.assert_close(p["heatmap"], g[tag + ".heatmap"], **FTOL)
        n += 1
    assert n >= 14


def test_lightglue_matches_reference(golden):
    g = golden("lightglue")
    for tag, structured, seed in (("s32", True, 1), ("s33", True, 2), ("r32", False, 3), ("s64", True, 4)):
        W = O.make_lg_weights(seed, structured=structured)
        data = {
            "keypoints0": g[tag + ".keypoints0"],
            "keypoints1": g[tag + ".keypoints1"],
            "descriptors0": g[tag + ".descriptors0"],
            "descriptors1": g[tag + ".descriptors1"],
            "view0": {"image_size": g[tag + ".size0"]},
            "view1": {"image_size": g[tag + ".size1"]},
        }
        torch.testing.assert_close(O.normalize_keypoints(data["keypoints0"], g[tag + ".size0"]), g[tag + ".kn0"], **FTOL)
        torch.testing.assert_close(O.fourier_encoding(g[tag + ".kn0"], W["posenc.Wr.weight"]), g[tag + ".enc0"], **FTOL)
        for th in (0.0, 0.1):
            p = O.lightglue_forward(data, W, collect_layers=(0, 8, ("self", 0)), filter_threshold=th)
            for k in ("matches0", "matches1"):
                assert torch.equal(p[k], g[f"{tag}.th{th}.{k}"]), (tag, th, k)
            for k in ("matching_scores0", "matching_scores1"):
                torch.testing.assert_close(p[k], g[f"{tag}.th{th}.{k}"], rtol=0, atol=1e-5)
        torch.testing.assert_close(p["log_assignment"], g[tag + ".log_assignment"], rtol=0, atol=5e-5)
        assert torch.equal(p["_argmax0"], g[tag + ".argmax0"])
        assert torch.equal(p["_argmax1"], g[tag + ".argmax1"])
        for i in (0, 8):
            torch.testing.assert_close(p["_layers"][i][0], g[f"{tag}.layer{i}.desc0"], rtol=0, atol=2e-5)
            torch.testing.assert_close(p["_layers"][i][1], g[f"{tag}.layer{i}.desc1"], rtol=0, atol=2e-5)
        torch.testing.assert_close(p["_layers"][("self", 0)][0], g[tag + ".self0.desc0"], rtol=0, atol=2e-5)
        torch.testing.assert_close(p["ref_descriptors0"], g[tag + ".ref_descriptors0"], rtol=0, atol=2e-5)
        assert torch.equal(p["prune0"], g[tag + ".prune0"])


def test_two_view_boat_matches_reference(golden):
    """Config 1 of BASELINE.json: boat1 <-> boat2 (reference test asset), 512 keypoints, CPU."""
    g = golden("two_view_boat")
    im0 = g["gray0"].float()[None, None] / 255.0
    im1 = g["gray1"].float()[None, None] / 255.0
    W_lg = O.make_lg_weights(7, structured=True, final_gain=8.0, match_bias=2.0)
    for variant in ("open", "ml"):
        W_sp = O.make_sp_weights(0, "open" if variant == "open" else "magicleap")
        sp_conf = dict(variant="open" if variant == "open" else "magicleap", nms_radius=3, detection_threshold=0.0, max_num_keypoints=512)
        size = torch.tensor([[im0.shape[-1], im0.shape[-2]]]).float()
        data = {"view0": {"image": im0, "image_size": size}, "view1": {"image": im1, "image_size": size}}
        p = O.two_view_forward(data, W_sp, W_lg, sp_conf, dict(filter_threshold=0.1))
        assert sorted(p.keys()) == list(g[variant + ".keys"])
        for k in ("keypoints0", "keypoints1", "matches0", "matches1", "prune0"):
            assert torch.equal(p[k], g[f"{variant}.{k}"]), (variant, k)
        for k in ("keypoint_scores0", "keypoint_scores1", "matching_scores0", "matching_scores1"):
            torch.testing.assert_close(p[k], g[f"{variant}.{k}"], rtol=0, atol=1e-5)
        torch.testing.assert_close(p["descriptors0"][0, ::8], g[variant + ".descriptors0_rows"], **FTOL)
        torch.testing.assert_close(p["log_assignment"][0, ::16, ::16], g[variant + ".log_assignment_diag"], rtol=0, atol=1e-4)
        # shifted self-pair (real correspondences)
        a = im0[:, :, 0:640, 0:800].contiguous()
        b = im0[:, :, 16:656, 24:824].contiguous()
        size = torch.tensor([[800.0, 640.0]])
        data = {"view0": {"image": a, "image_size": size}, "view1": {"image": b, "image_size": size}}
        p = O.two_view_forward(data, W_sp, W_lg, sp_conf, dict(filter_threshold=0.1))
        for k in ("keypoints0", "keypoints1", "matches0", "matches1"):
            assert torch.equal(p[k], g[f"{variant}.shift.{k}"]), (variant, k)
        assert int((p["matches0"] > -1).sum()) > 400


def _sg_data(g, tag):
    return {
        "keypoints0": g[tag + ".keypoints0"], "keypoints1": g[tag + ".keypoints1"],
        "descriptors0": g[tag + ".descriptors0"], "descriptors1": g[tag + ".descriptors1"],
        "keypoint_scores0": g[tag + ".keypoint_scores0"], "keypoint_scores1": g[tag + ".keypoint_scores1"],
        "view0": {"image_size": g[tag + ".size0"]}, "view1": {"image_size": g[tag + ".size1"]},
    }


def test_superglue_matches_reference(golden):
    """gluefactory_nonfree/superglue.py run by oracle/tools/make_goldens.py."""
    g = golden("superglue")
    for tag, seed in (("a", 3), ("b", 4), ("c", 5)):
        W = O.make_sg_weights(seed)
        for th in (0.0, 0.2):
            p = O.superglue_forward(_sg_data(g, tag), W, filter_threshold=th)
            for k in ("matches0", "matches1"):
                assert torch.equal(p[k], g[f"{tag}.th{th}.{k}"]), (tag, th, k)
            for k in ("matching_scores0", "matching_scores1"):
                torch.testing.assert_close(p[k], g[f"{tag}.th{th}.{k}"], rtol=0, atol=1e-5)
        torch.testing.assert_close(p["sinkhorn_cost"], g[tag + ".sinkhorn_cost"], rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(p["log_assignment"], g[tag + ".log_assignment"], rtol=1e-5, atol=1e-4)


NN_CONFS = ({}, {"ratio_thresh": 0.9}, {"distance_thresh": 0.8}, {"ratio_thresh": 0.95, "distance_thresh": 0.9, "mutual_check": False})


def test_nn_matcher_matches_reference(golden):
    """gluefactory/models/matchers/nearest_neighbor_matcher.py run by oracle/tools/make_goldens.py."""
    g = golden("nn_matcher")
    for tag in ("a", "b", "c"):
        data = {"descriptors0": g[tag + ".descriptors0"], "descriptors1": g[tag + ".descriptors1"]}
        for ci, conf in enumerate(NN_CONFS):
            p = O.nn_forward(data, **conf)
            for k in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
                assert torch.equal(p[k], g[f"{tag}.c{ci}.{k}"]), (tag, ci, k)
        torch.testing.assert_close(p["similarity"], g[tag + ".similarity"], rtol=0, atol=2e-6)
        torch.testing.assert_close(p["log_assignment"], g[tag + ".log_assignment"], rtol=0, atol=1e-5)


def _gs_data(g, tag):
    d = {k: g[f"{tag}.{k}"] for k in ("keypoints0", "keypoints1", "descriptors0", "descriptors1", "keypoint_scores0", "keypoint_scores1",
                                     "lines0", "lines1", "lines_junc_idx0", "lines_junc_idx1", "line_scores0", "line_scores1")}
    d["view0"], d["view1"] = {"image_size": g[tag + ".size0"]}, {"image_size": g[tag + ".size1"]}
    return d


GS_INT = ("matches0", "matches1", "line_matches0", "line_matches1")
GS_FLT = ("matching_scores0", "matching_scores1", "line_matching_scores0", "line_matching_scores1")


def test_gluestick_matches_reference(golden):
    """gluefactory/models/matchers/gluestick.py run by oracle/tools/make_goldens.py."""
    g = golden("gluestick")
    for tag, seed in (("a", 1), ("b", 2), ("c", 3)):
        W = O.make_gs_weights(seed, **O.GS_GOLDEN)
        for th in (0.0, 0.2):
            p = O.gluestick_forward(_gs_data(g, tag), W, filter_threshold=th)
            for k in GS_INT:
                assert torch.equal(p[k], g[f"{tag}.th{th}.{k}"]), (tag, th, k)
            for k in GS_FLT:
                torch.testing.assert_close(p[k], g[f"{tag}.th{th}.{k}"], rtol=0, atol=1e-5)
        for k in ("log_assignment", "line_log_assignment", "raw_line_scores"):
            torch.testing.assert_close(p[k], g[f"{tag}.{k}"], rtol=1e-5, atol=1e-4)


WF_KEYS = ("keypoints", "keypoint_scores", "descriptors", "lines", "line_scores", "orig_lines", "lines_junc_idx", "num_junctions", "pl_associativity")


def test_wireframe_matches_reference(golden):
    """SURVEY 8f rank 3, wireframe side: gluefactory/models/lines/wireframe.py (`WireframeExtractor._forward`,
    `lines_to_wireframe`, `sample_descriptors_corner_conv`) run by oracle/tools/make_goldens.py on fixed extractor outputs."""
    g = golden("wireframe")
    sp = {k: g["sp." + k] for k in ("keypoints", "keypoint_scores", "descriptors", "dense_descriptors")}
    for tag, merge_ends in (("merge", True), ("plain", False), ("nolines", True)):
        p = O.wireframe_forward(sp, g[tag + ".in.lines"], g[tag + ".in.line_scores"], nms_radius=3, merge_line_endpoints=merge_ends)
        for k in WF_KEYS:
            want = g[f"{tag}.{k}"]
            got = p[k].to(torch.uint8) if p[k].dtype == torch.bool else p[k]
            assert got.shape == want.shape, (tag, k, got.shape, want.shape)
            if got.dtype.is_floating_point:
                torch.testing.assert_close(got, want, rtol=0, atol=1e-6, msg=lambda s: f"{tag} {k}: {s}")
            else:
                assert torch.equal(got, want), (tag, k)
    # the fixture exercises both GPU steps: keypoints are removed, endpoints are merged
    assert g["merge.keypoints"].shape[1] - int(g["merge.num_junctions"][0]) < sp["keypoints"].shape[1]
    assert int(g["merge.num_junctions"][0]) < 2 * g["merge.in.lines"].shape[1]


def _lga_data(g, tag):
    return {"keypoints0": g[tag + ".keypoints0"], "keypoints1": g[tag + ".keypoints1"], "descriptors0": g[tag + ".descriptors0"],
            "descriptors1": g[tag + ".descriptors1"], "view0": {"image_size": g[tag + ".size"]}, "view1": {"image_size": g[tag + ".size"]}}


def test_lightglue_point_pruning_matches_reference(golden):
    """`width_confidence` > 0 of the in-tree LightGlue (lightglue.py:480-508), run by oracle/tools/make_goldens.py."""
    g = golden("lightglue_adaptive")
    for tag, wseed in (("a", 11), ("b", 12), ("c", 13)):
        W = O.make_lg_weights(wseed, structured=True, final_gain=3.0, match_bias=0.0)
        p = O.lightglue_forward(_lga_data(g, tag), W, filter_threshold=0.1, width_confidence=float(g[tag + ".width_confidence"]))
        for k in ("matches0", "matches1", "prune0", "prune1"):
            assert torch.equal(p[k], g[f"{tag}.{k}"]), (tag, k)
        for k in ("matching_scores0", "matching_scores1", "log_assignment", "ref_descriptors0", "ref_descriptors1"):
            torch.testing.assert_close(p[k], g[f"{tag}.{k}"], rtol=0, atol=1e-6)
        assert p["log_assignment"].shape[1] - 1 < p["matches0"].shape[1]  # points were really pruned
