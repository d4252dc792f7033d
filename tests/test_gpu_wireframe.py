"""GPU parity of the wireframe side of the GlueStick path (gluefactory_amd.wireframe: mg_wf_filter + corner sampling) against
golden vectors produced by the reference's own wireframe.py, and the whole SuperPoint -> wireframe -> GlueStick chain."""
import pytest
import torch

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
WF_KEYS = ("keypoints", "keypoint_scores", "descriptors", "lines", "line_scores", "orig_lines", "lines_junc_idx", "num_junctions", "pl_associativity")


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _wf(dev, merge_ends, **pe):
    from gluefactory_amd import get_model

    m = get_model("gluefactory_amd.wireframe")({"point_extractor": {"name": "gluefactory_nonfree.superpoint", "nms_radius": 3, "detection_threshold": 0.0,
                                                                    "max_num_keypoints": 96, **pe},
                                                "wireframe_params": {"merge_line_endpoints": merge_ends}}).eval()
    m.point_extractor.load_state_dict(O.make_sp_weights(0))
    return m


def test_wireframe_golden(dev, golden):
    """The extractor outputs are pinned to the fixture's (so integer outputs must be exact); filter, compaction, junction
    descriptors and the assembly run on the GPU."""
    g = golden("wireframe")
    sp = {k: g["sp." + k].to(dev) for k in ("keypoints", "keypoint_scores", "descriptors", "dense_descriptors")}
    for tag, merge_ends in (("merge", True), ("plain", False), ("nolines", True)):
        m = _wf(dev, merge_ends)
        m.point_extractor._forward = lambda data: {k: v.clone() for k, v in sp.items()}
        p = m({"image": g["image"].to(dev), "lines": g[tag + ".in.lines"].to(dev), "line_scores": g[tag + ".in.line_scores"].to(dev)})
        for k in WF_KEYS:
            want = g[f"{tag}.{k}"]
            got = p[k].cpu()
            got = got.to(torch.uint8) if got.dtype == torch.bool else got
            assert got.shape == want.shape, (tag, k, got.shape, want.shape)
            if k == "descriptors":
                torch.testing.assert_close(got, want, rtol=0, atol=1e-6, msg=lambda s: f"{tag} {k}: {s}")
            elif got.dtype.is_floating_point:
                assert torch.equal(got, want), (tag, k, (got - want).abs().max())
            else:
                assert torch.equal(got, want), (tag, k)


@pytest.mark.parametrize("K,E,r", [(96, 48, 3.0), (1000, 500, 4.0), (4096, 1000, 2.0), (17, 0, 3.0), (300, 1, 50.0)])
def test_wf_filter_vs_torch(dev, K, E, r):
    """mg_wf_filter against the reference's formulation (torch.norm(kp - e) < r, any over endpoints, boolean mask) on random
    points, including rows that are padding already."""
    from gluefactory_amd import _native

    g = torch.Generator().manual_seed(K + E)
    B = 3
    kp = torch.rand(B, K, 2, generator=g) * 200
    ep = torch.rand(B, E, 2, generator=g) * 200
    sc = torch.rand(B, K, generator=g)
    ds = torch.randn(B, K, 256, generator=g)
    nv = torch.tensor([K, max(K // 2, 1), K])
    eng = _native.Engine(dev)
    okp, osc, ods, cnt = (t.cpu() for t in eng.wf_filter(kp, sc, ds, ep, r, num_valid=nv))
    for b in range(B):
        dist = torch.norm(kp[b, :, None] - ep[b][None], dim=-1)
        valid = torch.arange(K) < nv[b]
        keep = ~torch.any(dist < r, dim=1) & valid
        # a keypoint whose nearest endpoint sits within 1e-5 of the radius may fall either way (sqrt / sum rounding); with millions of
        # pairs a few do: they are excluded from the exact comparison, everything else must agree row for row
        border = (torch.any((dist - r).abs() < 1e-5, dim=1) if E else torch.zeros(K, dtype=torch.bool)) & valid
        n = int(cnt[b])
        assert int((keep & ~border).sum()) <= n <= int((keep | border).sum())
        got_rows = {tuple(v) for v in okp[b, :n].tolist()}
        for k in torch.nonzero(valid & ~border).flatten().tolist():
            assert (tuple(kp[b, k].tolist()) in got_rows) == bool(keep[k]), (b, k)
        if not bool(border.any()):
            assert torch.equal(okp[b, :n], kp[b][keep]) and torch.equal(osc[b, :n], sc[b][keep]) and torch.equal(ods[b, :n], ds[b][keep])
        else:  # order preserved, rows travel together
            src = [int(torch.nonzero((kp[b] == okp[b, i]).all(-1))[0]) for i in range(n)]
            assert src == sorted(src)
            assert torch.equal(osc[b, :n], sc[b][src]) and torch.equal(ods[b, :n], ds[b][src])
        assert bool((okp[b, n:] == 0).all()) and bool((osc[b, n:] == 0).all()) and bool((ods[b, n:] == 0).all())


def test_superpoint_wireframe_gluestick_chain(dev, golden):
    """configs[4] end to end on the GPU: SuperPoint (dense outputs) -> wireframe (synthetic segments between detected keypoints; LSD
    itself is a CPU dependency) -> GlueStick, for an image and an 8-pixel-multiple shift of it.  Checked against the same chain of
    oracle functions fed the GPU extractor's outputs (the extractor's own parity is covered in test_gpu_parity.py)."""
    from gluefactory_amd import get_model

    z = golden("two_view_boat")
    im = (z["gray0"].float()[None, None] / 255.0)
    a, b = im[:, :, 40:360, 100:580].contiguous(), im[:, :, 56:376, 124:604].contiguous()  # shift (24, 16)
    wf = _wf(dev, True, max_num_keypoints=400)
    Wg = O.make_gs_weights(4, **O.GS_GOLDEN)
    gs = get_model("gluefactory_amd.gluestick")({"filter_threshold": 0.2}).eval()
    gs.load_state_dict(Wg)
    g = torch.Generator().manual_seed(9)
    preds, refs = [], []
    base_kp = None
    for i, (img, off) in enumerate(((a, (0.0, 0.0)), (b, (24.0, 16.0)))):
        spp = wf.point_extractor({"image": img.to(dev)})
        kp = spp["keypoints"][0].cpu()
        if i == 0:  # 40 segments between detected keypoints (moved 4 px off them so that the keypoints themselves are filtered out)
            sel = torch.randperm(len(kp), generator=g)[:80].reshape(40, 2)
            base_kp = kp[sel] + 0.25
        lines = (base_kp - torch.tensor(off))[None] if i else base_kp[None]
        scores = torch.rand(1, 40, generator=torch.Generator().manual_seed(3)) + 0.5
        p = wf({"image": img.to(dev), "lines": lines.to(dev), "line_scores": scores.to(dev)})
        ref = O.wireframe_forward({k: v.cpu() for k, v in spp.items()}, lines, scores, nms_radius=3)
        for k in ("keypoints", "keypoint_scores", "lines", "lines_junc_idx", "num_junctions"):
            assert torch.equal(p[k].cpu(), ref[k]), (i, k)
        torch.testing.assert_close(p["descriptors"].cpu(), ref["descriptors"], rtol=0, atol=1e-6)
        assert int(p["num_junctions"][0]) <= 80 and p["keypoints"].shape[1] < 400 + 80
        preds.append(p)
        refs.append(ref)
    size = torch.tensor([[480.0, 320.0]], device=dev)
    data = {"view0": {"image_size": size}, "view1": {"image_size": size}}
    for i, p in enumerate(preds):
        for k in ("keypoints", "keypoint_scores", "descriptors", "lines", "lines_junc_idx", "line_scores"):
            data[k + str(i)] = p[k]
    out = gs(data)
    rdata = {"view0": {"image_size": size.cpu()}, "view1": {"image_size": size.cpu()}}
    for i, p in enumerate(preds):
        for k in ("keypoints", "keypoint_scores", "descriptors", "lines", "lines_junc_idx", "line_scores"):
            rdata[k + str(i)] = p[k].cpu()
    ref = O.gluestick_forward(rdata, Wg, filter_threshold=0.2)
    import tie_aware as TA

    for pre in ("", "line_"):
        S = ref[pre + "log_assignment"][0]
        eps = TA.la_eps(out[pre + "log_assignment"][0].cpu(), S)
        r = TA.check_matches(out[pre + "matches0"][0].cpu(), ref[pre + "matches0"][0], S, 0.2, eps)
        n = int((ref[pre + "matches0"] > -1).sum())
        print(f"chain: {pre}matches0 {n} of {ref[pre + 'matches0'].numel()} matched by the oracle, {r['differing']} differ (near-ties, eps {eps:.1e})")
    assert int((out["matches0"] > -1).sum()) > 100 and int((out["line_matches0"] > -1).sum()) > 10
