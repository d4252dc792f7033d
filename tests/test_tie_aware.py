"""The tie-aware comparison (oracle/tie_aware.py) has teeth: it accepts near-tie differences of the oracle's own
decisions and nothing else.  CPU only: the "other implementation" is the oracle with its floats perturbed."""
import pytest
import torch

import tie_aware as TA
from oracle import sp_lg_oracle as O

torch.set_grad_enabled(False)


def _sp_case(k=96):
    g = torch.Generator().manual_seed(11)
    img = torch.nn.functional.avg_pool2d(torch.rand(1, 1, 124, 164, generator=g), 5, 1)
    img = (img - img.amin()) / (img.amax() - img.amin())
    W = O.make_sp_weights(0)
    ref = O.superpoint_forward({"image": img}, W, nms_radius=3, detection_threshold=0.0, max_num_keypoints=k)
    return ref, k, g


def _detect(heat, k):
    s = O.mask_borders(O.nms_heatmap(heat, 3), 4, None)
    (yx, v), = O.select_keypoints(s, 0.0, k)
    return torch.flip(yx, [1]).float() + 0.5, v


def test_keypoints_accepts_float_noise_rejects_real_errors():
    ref, k, g = _sp_case()
    rk, rs, nms = ref["keypoints_list"][0], ref["keypoint_scores_list"][0], ref["nms_scores"][0]
    # identical lists
    r = TA.check_keypoints(rk, rk, rs, nms, k_max=k)
    assert r["rank_diffs"] == 0 and r["boundary_swaps"] == 0 and torch.equal(r["perm"], torch.arange(len(rk)))
    # float noise of the size the HIP path shows (<= EPS_HEAT per score): whatever moves is accepted as a near-tie
    moved = 0
    for seed in range(20):
        noise = (torch.rand(ref["heatmap"].shape, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * 3e-6
        gk, gs = _detect(ref["heatmap"] + noise, k)
        r = TA.check_keypoints(gk, rk, rs, nms, got_scores=gs, k_max=k)
        assert r["eps"] <= 6.1e-6  # the near-tie width follows the measured score error of the image, not the global cap
        moved += r["rank_diffs"] + r["boundary_swaps"]
    # two keypoints with clearly different scores swapped: rejected
    gaps = (rs[:-1] - rs[1:])
    i = int(gaps.argmax())
    assert gaps[i] > 2 * TA.EPS_KP
    bad = rk.clone()
    bad[[i, i + 1]] = bad[[i + 1, i]]
    with pytest.raises(AssertionError, match="not a near-tie"):
        TA.check_keypoints(bad, rk, rs, nms, k_max=k)
    # a pixel that NMS suppressed: rejected as a wrong pixel
    bad = rk.clone()
    bad[3, 0] += 1.0
    with pytest.raises(AssertionError, match="NMS survivor"):
        TA.check_keypoints(bad, rk, rs, nms, k_max=k)
    # a duplicate
    bad = rk.clone()
    bad[5] = bad[4]
    with pytest.raises(AssertionError, match="duplicate"):
        TA.check_keypoints(bad, rk, rs, nms, k_max=k)
    # the (K+1)-th candidate instead of the K-th although their scores are far apart: rejected
    (yx, v), = O.select_keypoints(nms[None], 0.0, k + 1)
    if float(v[k - 1] - v[k]) > TA.EPS_KP:
        bad = rk.clone()
        bad[k - 1] = torch.flip(yx[k], [0]).float() + 0.5
        with pytest.raises(AssertionError):
            TA.check_keypoints(bad, rk, rs, nms, k_max=k)


def test_matches_accepts_near_ties_rejects_real_errors():
    g = torch.Generator().manual_seed(5)
    M = N = 40
    W = O.make_lg_weights(5, structured=True, final_gain=8.0, match_bias=2.0)
    size = torch.tensor([[640.0, 480.0]])
    k0 = torch.rand(1, M, 2, generator=g) * size * 0.98 + 0.5
    d0 = torch.nn.functional.normalize(torch.randn(1, M, 256, generator=g), dim=-1)
    perm = torch.randperm(M, generator=g)[None]
    k1 = torch.gather(k0, 1, perm[..., None].expand(-1, -1, 2)) * 0.9
    d1 = torch.nn.functional.normalize(torch.gather(d0, 1, perm[..., None].expand(-1, -1, 256)) + 0.03 * torch.randn(1, N, 256, generator=g), dim=-1)
    data = {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1, "view0": {"image_size": size}, "view1": {"image_size": size}}
    ref = O.lightglue_forward(data, W, filter_threshold=0.1)
    m0, S = ref["matches0"][0], ref["log_assignment"][0]
    assert int((m0 > -1).sum()) > 10
    r = TA.check_matches(m0, m0, S, 0.1, 1e-5)
    assert r["differing"] == 0 and r["compared"] == M
    # the same result seen through a permutation of both keypoint lists (what a near-tie rank swap does)
    p0, p1 = torch.randperm(M, generator=g), torch.randperm(N, generator=g)  # HIP rank -> oracle rank
    inv1 = torch.empty_like(p1)
    inv1[p1] = torch.arange(N)
    got = torch.empty_like(m0)
    for i_h in range(M):
        j = int(m0[p0[i_h]])
        got[i_h] = -1 if j < 0 else inv1[j]
    r = TA.check_matches(got, m0, S, 0.1, 1e-5, perm0=p0, perm1=p1)
    assert r["differing"] == 0
    la_h = S[torch.cat([p0, torch.tensor([M])])][:, torch.cat([p1, torch.tensor([N])])]
    assert torch.equal(TA.to_oracle_order(la_h, p0, p1), S)
    # a confident match redirected to another column / dropped: rejected
    i = int(torch.nonzero(m0 > -1)[0])
    bad = m0.clone()
    bad[i] = (m0[i] + 1) % N
    with pytest.raises(AssertionError, match="not a near-tie"):
        TA.check_matches(bad, m0, S, 0.1, 1e-5)
    bad = m0.clone()
    bad[i] = -1
    with pytest.raises(AssertionError, match="not a near-tie"):
        TA.check_matches(bad, m0, S, 0.1, 1e-5)
    # a constructed near-tie: two columns of one row 1e-7 apart -> either answer is accepted
    S2 = S.clone()
    j = int(m0[i])
    j2 = (j + 1) % N
    S2[:M, j2] = S2[:M, j2].clamp(max=float(S2[i, j]) - 5.0)  # make (i, j2) the column maximum as well
    S2[i, j2] = S2[i, j] - 1e-7
    alt = m0.clone()
    alt[i] = j2
    assert TA.check_matches(alt, m0, S2, 0.1, 1e-5)["differing"] == 1
