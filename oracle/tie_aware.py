"""Tie-aware exact comparison of integer outputs (keypoint indices, match assignments) with the CPU oracle.

north_star asks for bit-exact indices.  Floats of the HIP path and of the oracle differ by summation order
(heat map: <= EPS_HEAT, asserted in the dense tests), so an index can legitimately differ only where the ORACLE's own
decision was a near-tie.  These helpers accept a difference only when they can prove that from the oracle's numbers,
and fail otherwise -- no percentage thresholds:

  keypoints  every HIP keypoint is an NMS survivor of the oracle's score map; at every rank where the two lists hold a
             different pixel, the oracle scores of the two pixels differ by < eps; the two sets are equal except for
             swaps across rank K between pixels whose oracle scores are within eps of the oracle's K-th score.
  matches    compared through keypoint COORDINATES (so a near-tie swap of two ranks is not a match error); a
             differing assignment is accepted only if the oracle's own decision for the row had a margin < eps_la:
             row arg-max margin, column arg-max margin (mutual check) or distance of the score to the filter threshold
             (lightglue.py:287-303).

Both return the counts they found so the tests can print them (DESIGN.md section 4b quotes them).
"""
import torch

# Bound on |score_hip - score_oracle| of a heat-map value, asserted wherever scores are compared.  Measured on the GPU
# (gpurun_out/r02_*_tests.log): 2.3e-6 (f32) / 3.1e-6 (f16x3) on smooth noise, 6.7e-6 on the boat images (peakier soft-max),
# 1.1e-5 for the open variant (BatchNorm affine after every ReLU) -- fp32 summation order over a 10-layer conv chain.
EPS_HEAT = 1.5e-5
# Near-tie width for keypoint ranks: two pixels can swap order only if their ORACLE scores are closer than the sum of the two
# score errors.  check_keypoints measures the score error of the very image at hand (HIP keypoint scores vs the oracle's map
# at the same pixels) and uses eps = 2 x that, so the rule is as tight as the floats of that image allow; EPS_KP caps it.
EPS_KP = 2 * EPS_HEAT


def _pix(xy):
    return (xy - 0.5).round().long()


def check_keypoints(got_xy, ref_xy, ref_scores, nms_map, got_scores=None, eps=None, threshold=0.0, k_max=None):
    """got_xy / ref_xy: (n,2) (x,y)+0.5 rows (valid rows only); ref_scores: (n,) oracle scores in oracle order;
    nms_map: (H,W) the oracle's score map after NMS and border removal (superpoint.py:240-254); got_scores: (n,) HIP scores.
    eps: near-tie width; default 2 x the measured score error of this image (needs got_scores), at most EPS_KP.
    Returns dict(n, rank_diffs, boundary_swaps, max_gap, score_err, eps, perm); perm[i] = oracle rank of HIP keypoint i (-1: not in the oracle's set)."""
    assert got_xy.shape == ref_xy.shape, f"keypoint counts differ: {tuple(got_xy.shape)} vs {tuple(ref_xy.shape)}"
    n = len(ref_xy)
    gp, rp = _pix(got_xy), _pix(ref_xy)
    H, W = nms_map.shape
    assert bool(((gp[:, 0] >= 0) & (gp[:, 0] < W) & (gp[:, 1] >= 0) & (gp[:, 1] < H)).all()), "keypoint outside the image"
    got_lin, ref_lin = gp[:, 1] * W + gp[:, 0], rp[:, 1] * W + rp[:, 0]
    assert len(torch.unique(got_lin)) == n, "duplicate keypoints"
    g_or = nms_map.reshape(-1)[got_lin]  # the oracle's score at the pixels the HIP path chose
    assert bool((g_or > threshold).all()), "a HIP keypoint is not an NMS survivor / above threshold in the oracle's map (wrong pixel, not a tie)"
    score_err = None
    if got_scores is not None:
        score_err = float((got_scores - g_or).abs().max()) if n else 0.0
        assert score_err <= EPS_HEAT, f"keypoint scores are off by {score_err:.2e} (> {EPS_HEAT:.1e})"
        if n > 1:
            assert bool((got_scores[:-1] >= got_scores[1:]).all()) or n < (k_max or n + 1), "HIP keypoints are not sorted by their own scores"
    if eps is None:
        eps = min(EPS_KP, max(2 * score_err, 1e-7)) if score_err is not None else EPS_KP
    diff = got_lin != ref_lin
    gap = (g_or - ref_scores).abs()
    max_gap = float(gap[diff].max()) if bool(diff.any()) else 0.0
    assert max_gap <= eps, f"keypoint order differs at a rank where the oracle scores are {max_gap:.3e} apart (eps {eps:.1e}): not a near-tie"
    # set equality, up to swaps across the rank-K boundary between near-ties of the K-th score
    ref_pos = {int(v): i for i, v in enumerate(ref_lin.tolist())}
    perm = torch.tensor([ref_pos.get(int(v), -1) for v in got_lin.tolist()])
    extra = perm < 0
    n_extra = int(extra.sum())
    if n_extra:
        assert k_max is not None and n == k_max, "sets differ although fewer than max_num_keypoints were detected"
        kth = float(ref_scores.min())
        assert bool(((g_or[extra] - kth).abs() <= eps).all()), "a keypoint outside the oracle's set is not a near-tie of the oracle's K-th score"
        got_set = set(got_lin.tolist())
        missing = torch.tensor([v not in got_set for v in ref_lin.tolist()])
        assert int(missing.sum()) == n_extra
        assert bool(((ref_scores[missing] - kth).abs() <= eps).all()), "an oracle keypoint missing from the HIP set is not a near-tie of the K-th score"
    return {"n": n, "rank_diffs": int(diff.sum()), "boundary_swaps": n_extra, "max_gap": max_gap, "score_err": score_err, "eps": eps, "perm": perm}


def match_margins(S, th):
    """Per-entry decision margins of filter_matches (lightglue.py:287-303) on the oracle's log_assignment S (M+1,N+1):
    how far entry (i,j) is from changing the row arg-max, the column arg-max, or its side of the score threshold."""
    A = S[:-1, :-1]
    M, N = A.shape
    r2 = torch.topk(A, min(2, N), dim=1).values
    c2 = torch.topk(A, min(2, M), dim=0).values
    row_best, row_second = r2[:, :1], (r2[:, 1:2] if N > 1 else torch.full((M, 1), -float("inf")))
    col_best, col_second = c2[:1], (c2[1:2] if M > 1 else torch.full((1, N), -float("inf")))
    # margin to the best OTHER entry of the row / column
    row_other = torch.where(A == row_best, row_second, row_best)
    col_other = torch.where(A == col_best, col_second, col_best)
    m_row = (A - row_other).abs()
    m_col = (A - col_other).abs()
    m_th = (A.exp() - th).abs()
    return m_row, m_col, m_th


def check_matches(got_m0, ref_m0, ref_S, th, eps_la, perm0=None, perm1=None, eps_th=None):
    """got_m0: (M,) HIP matches0 (indices into the HIP keypoint order of image 1); ref_m0: (M,) oracle matches0;
    ref_S: (M+1,N+1) oracle log_assignment; perm0/perm1: HIP rank -> oracle rank for image 0 / 1 (check_keypoints), None = identity.
    Rows whose keypoint (or whose HIP match) is not in the oracle's set are skipped and counted.
    Returns dict(rows, compared, differing, skipped, max_margin)."""
    M, N = ref_S.shape[0] - 1, ref_S.shape[1] - 1
    perm0 = torch.arange(M) if perm0 is None else perm0
    perm1 = torch.arange(N) if perm1 is None else perm1
    eps_th = eps_la if eps_th is None else eps_th
    m_row, m_col, m_th = match_margins(ref_S, th)
    compared = differing = skipped = 0
    max_margin = 0.0
    for i_hip in range(len(got_m0)):
        i = int(perm0[i_hip])
        j_hip = int(got_m0[i_hip])
        if i < 0 or (j_hip >= 0 and int(perm1[j_hip]) < 0):
            skipped += 1
            continue
        j_got = int(perm1[j_hip]) if j_hip >= 0 else -1
        j_ref = int(ref_m0[i])
        compared += 1
        if j_got == j_ref:
            continue
        differing += 1
        # every entry whose acceptance changed must have been a near-tie decision of the oracle
        for j in {j_got, j_ref} - {-1}:
            margin = min(float(m_row[i, j]), float(m_col[i, j]))
            near = margin < eps_la or float(m_th[i, j]) < eps_th
            max_margin = max(max_margin, margin if margin < eps_la else float(m_th[i, j]))
            assert near, (f"match of oracle keypoint {i} differs (HIP -> {j_got}, oracle -> {j_ref}) although the oracle's decision margins for "
                          f"({i},{j}) are row {float(m_row[i, j]):.3e} / column {float(m_col[i, j]):.3e} (eps {eps_la:.1e}) and "
                          f"|score - th| {float(m_th[i, j]):.3e} (eps {eps_th:.1e}): not a near-tie")
    return {"rows": len(got_m0), "compared": compared, "differing": differing, "skipped": skipped, "max_margin": max_margin}


def to_oracle_order(got_la, perm0, perm1):
    """HIP log_assignment (M+1,N+1) re-indexed to the oracle's keypoint order (perm from check_keypoints; sets must be equal)."""
    assert bool((perm0 >= 0).all()) and bool((perm1 >= 0).all())
    M, N = len(perm0), len(perm1)
    r = torch.cat([perm0, torch.tensor([M])])
    c = torch.cat([perm1, torch.tensor([N])])
    out = torch.empty_like(got_la)
    out[r[:, None], c[None, :]] = got_la
    return out


def la_eps(got_la, ref_la, floor=1e-5):
    """Near-tie width for the assignment: twice the measured float error of log_assignment (two entries, each off by
    that much, can swap).  The error itself is bounded by the float tolerance the same test asserts."""
    return max(2.0 * float((got_la - ref_la).abs().max()), floor)


def check_two_view(p, ref, b, k_max, th, tag="", la_floor=1e-5, anchor=None):
    """One pair of a two-view result against the oracle's result for that pair (O.two_view_forward(..., with_maps=True), B = 1).

    p: HIP pred dict (CPU tensors, batch index b is compared); ref: oracle pred for the single pair.
    Integers: keypoints of both images under the tie-aware rule; matches0 / matches1 through keypoint coordinates under the
    near-tie rule with eps = 2 x the measured log_assignment difference between the two runs.
    Floats (keypoint scores, descriptors, log_assignment, matching scores): north_star's 1e-4.  Without `anchor` that is a plain
    1e-4 against the fp32 oracle.  With `anchor` (oracle.anchor.TwoViewAnchor of the same input: the oracle in float64) the HIP
    floats are compared with the fp64 evaluation of the HIP path's OWN keypoints, the oracle's floats with the fp64 evaluation of
    ITS keypoints, and the bound is 1e-4 where the fp32 CPU path itself is within 1e-4 of fp64 and 2 x the CPU path's own error
    elsewhere (oracle/anchor.py) -- needed with the seeded stress weights, whose |sim| ~ 3e2 puts the fp32 CPU log_assignment
    itself 1e-3 from the exact value.  Returns the counts and errors (and prints them)."""
    from . import anchor as AN

    out = {}
    perms = []
    for i in ("0", "1"):
        n = ref["keypoints" + i].shape[1]
        if p["keypoints" + i].shape[1] > n:  # fixed-K rows: everything past the detected count is zero padding
            assert bool((p["keypoint_scores" + i][b, n:] == 0).all()), "rows past the oracle's keypoint count are not padding"
        r = check_keypoints(p["keypoints" + i][b, :n], ref["keypoints" + i][0], ref["keypoint_scores" + i][0], ref["_nms_scores" + i][0],
                            got_scores=p["keypoint_scores" + i][b, :n], k_max=k_max)
        perms.append(r["perm"])
        out["kp" + i] = {k: v for k, v in r.items() if k != "perm"}
    M, N = len(perms[0]), len(perms[1])
    sets_equal = bool((perms[0] >= 0).all()) and bool((perms[1] >= 0).all())
    S = ref["log_assignment"][0]
    have_la = "log_assignment" in p
    if have_la:
        la_hip = p["log_assignment"][b][: M + 1, : N + 1] if p["log_assignment"].shape[1] == M + 1 else _crop_la(p["log_assignment"][b], M, N)
    # ---- floats
    name = f"{tag} pair {b}".strip()
    if anchor is not None:
        a_hip = anchor.at(p["keypoints0"][b, :M], p["keypoints1"][b, :N], b)  # fp64 on the HIP path's own keypoints, HIP order
        a_cpu = anchor.at(ref["keypoints0"][0], ref["keypoints1"][0], b)  # fp64 on the oracle's keypoints, oracle order
        fl = {}
        for k in ("keypoint_scores0", "keypoint_scores1", "descriptors0", "descriptors1"):
            n = M if k.endswith("0") else N
            fl[k] = AN.anchored_errs(AN.max_err(p[k][b, :n], a_hip[k][0]), AN.max_err(ref[k][0], a_cpu[k][0]), f"{name} {k}", log=None)
        if have_la:
            fl["log_assignment"] = AN.anchored_errs(AN.max_err(la_hip, a_hip["log_assignment"][0]), AN.max_err(S, a_cpu["log_assignment"][0]),
                                                    f"{name} log_assignment", log=None)
            fl["matching_scores0"] = AN.anchored_errs(AN.score_rows(la_hip, p["matching_scores0"][b, :M], a_hip["log_assignment"][0]),
                                                      AN.score_rows(S, ref["matching_scores0"][0], a_cpu["log_assignment"][0]),
                                                      f"{name} matching_scores0", log=None)
            fl["matching_scores1"] = AN.anchored_errs(AN.score_rows(la_hip.t(), p["matching_scores1"][b, :N], a_hip["log_assignment"][0].t()),
                                                      AN.score_rows(S.t(), ref["matching_scores1"][0], a_cpu["log_assignment"][0].t()),
                                                      f"{name} matching_scores1", log=None)
        out["floats"] = fl
        print(f"[fp64 anchor {name}] |HIP - fp64| / |CPU-fp32 - fp64| (bound): " +
              ", ".join(f"{k} {v['err_hip']:.1e} / {v['err_cpu']:.1e} ({v['bound']:.0e})" for k, v in fl.items()))
    eps = la_floor
    if sets_equal and have_la:
        got_la = to_oracle_order(la_hip, perms[0], perms[1])
        out["la_err"] = float((got_la - S).abs().max())  # HIP vs CPU-fp32: the width of a near-tie between the two runs
        if anchor is None:
            assert out["la_err"] <= 1e-4, f"log_assignment differs from the fp32 oracle by {out['la_err']:.2e} (> 1e-4); pass an fp64 anchor if the oracle itself is ill-conditioned here"
        else:  # triangle inequality: both runs were just held to their bounds against fp64
            lim = out["floats"]["log_assignment"]["bound"] + out["floats"]["log_assignment"]["err_cpu"]
            assert out["la_err"] <= lim, (out["la_err"], lim)
        eps = max(2 * out["la_err"], la_floor)
    out["eps_la"] = eps
    out["m0"] = check_matches(p["matches0"][b, :M], ref["matches0"][0], S, th, eps, perms[0], perms[1])
    out["m1"] = check_matches(p["matches1"][b, :N], ref["matches1"][0], S.t(), th, eps, perms[1], perms[0])
    # scores where both sides made the same assignment.  The reference keeps exp(row max) as the score of a row that is
    # mutual but below the filter threshold (match -1, lightglue.py:296-300) and 0 for a non-mutual row, so two unmatched
    # rows may still differ in score when the MUTUAL check itself was a near-tie: accepted only with that proof.
    m_row, m_col, _ = match_margins(S, th)
    worst, worst_at, mutual_flips = 0.0, None, 0
    for i_h in range(M):
        i = int(perms[0][i_h])
        if i < 0:
            continue
        j_h = int(p["matches0"][b, i_h])
        j_ref = int(ref["matches0"][0, i])
        if (j_h < 0) != (j_ref < 0) or (j_h >= 0 and int(perms[1][j_h]) != j_ref):
            continue
        s_h, s_r = float(p["matching_scores0"][b, i_h]), float(ref["matching_scores0"][0, i])
        if j_ref < 0 and (s_h == 0.0) != (s_r == 0.0):
            js = int(S[i, :N].argmax())
            assert min(float(m_row[i, js]), float(m_col[i, js])) < eps, f"row {i}: mutual check differs (scores {s_h} vs {s_r}) without a near-tie"
            mutual_flips += 1
            continue
        d = abs(s_h - s_r)
        if d > worst:
            worst, worst_at = d, (i, j_ref, s_h, s_r)
    out["mutual_flips"] = mutual_flips
    # HIP vs CPU-fp32 on rows with the same assignment: 1e-4, or (anchored) what the two fp64 bounds allow together
    score_lim = 1e-4 if anchor is None or not have_la else out["floats"]["matching_scores0"]["bound"] + out["floats"]["matching_scores0"]["err_cpu"]
    assert worst <= score_lim, (f"matching_scores0 differ by {worst:.2e} (> {score_lim:.1e}) where the assignment agrees: (row, col, hip, oracle) = {worst_at}, "
                                f"log_assignment difference {out.get('la_err')}")
    out["score_err"] = worst
    print(f"[tie-aware{' ' + tag if tag else ''}] pair {b}: keypoints0 {out['kp0']}, keypoints1 {out['kp1']}; "
          f"matches0 {out['m0']}, matches1 {out['m1']} (eps_la {eps:.2e}, HIP-vs-CPU32 log_assignment {out.get('la_err', float('nan')):.2e}, scores {worst:.2e})")
    return out


def _crop_la(la, M, N):
    """(Mk+1, Nk+1) log_assignment of fixed-K rows -> the (M+1, N+1) block of the valid rows plus the dustbins."""
    r = torch.cat([torch.arange(M), torch.tensor([la.shape[0] - 1])])
    c = torch.cat([torch.arange(N), torch.tensor([la.shape[1] - 1])])
    return la[r][:, c]
