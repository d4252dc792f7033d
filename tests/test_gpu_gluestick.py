"""GPU parity of the native GlueStick path (mg_gs_forward) -- SURVEY 8f rank 3.

Golden vectors come from gluefactory/models/matchers/gluestick.py itself (oracle/tools/make_goldens.py); a larger case is
checked against the CPU oracle and through oracle-independent properties of the double-softmax assignment.
"""
import pytest
import torch

from oracle import anchor as AN
from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
ATOL = 1e-4
KEYS = ("keypoints0", "keypoints1", "descriptors0", "descriptors1", "keypoint_scores0", "keypoint_scores1",
        "lines0", "lines1", "lines_junc_idx0", "lines_junc_idx1", "line_scores0", "line_scores1")
GS_INT = ("matches0", "matches1", "line_matches0", "line_matches1")
GS_FLT = ("matching_scores0", "matching_scores1", "line_matching_scores0", "line_matching_scores1")
GS_SCORES = (("log_assignment", "matching_scores0", "matching_scores1"), ("line_log_assignment", "line_matching_scores0", "line_matching_scores1"))


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _module(W, th, precision, **kw):
    from gluefactory_amd import get_model

    m = get_model("gluefactory_amd.gluestick")({"filter_threshold": th, "precision": precision, **kw}).eval()
    m.load_state_dict(W)
    return m


def _to(x, dev):
    return {k: (_to(v, dev) if isinstance(v, dict) else v.to(dev)) for k, v in x.items()}


def _data(g, tag):
    d = {k: g[f"{tag}.{k}"] for k in KEYS}
    d["view0"], d["view1"] = {"image_size": g[tag + ".size0"]}, {"image_size": g[tag + ".size1"]}
    return d


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_gluestick_golden(dev, golden, precision):
    g = golden("gluestick")
    for tag, seed in (("a", 1), ("b", 2), ("c", 3)):
        W = O.make_gs_weights(seed, **O.GS_GOLDEN)
        data = _data(g, tag)
        ref = O.gluestick_forward(data, W, filter_threshold=0.0)
        B, M = data["keypoints0"].shape[:2]
        N = data["keypoints1"].shape[1]
        T = B * (M + N)
        r64 = None
        for th in (0.0, 0.2):
            m = _module(W, th, precision)
            eng = m.engine(dev)
            for tap in ("gs.kenc", "gs.layer0", "gs.layer1", "gs.layer17"):
                eng.debug_request(tap)
            p = m(_to(data, dev))
            for tap, key in (("gs.kenc", "kenc"), ("gs.layer0", 0), ("gs.layer1", 1), ("gs.layer17", 17)):
                x = eng.debug_read(tap, T * 256).cpu().reshape(T, 256)
                r0, r1 = ref["_taps"][key]  # (B,256,K)
                torch.testing.assert_close(x[: B * M].reshape(B, M, 256), r0.transpose(1, 2), rtol=0, atol=ATOL, msg=lambda s: f"{tag} {tap}: {s}")
                torch.testing.assert_close(x[B * M:].reshape(B, N, 256), r1.transpose(1, 2), rtol=0, atol=ATOL, msg=lambda s: f"{tag} {tap}: {s}")
            gold = {k: g[f"{tag}.{k}"] for k in ("log_assignment", "line_log_assignment", "raw_line_scores")}
            gold.update({k: g[f"{tag}.th{th}.{k}"] for k in GS_FLT})
            _, r64 = AN.matcher_floats(O.gluestick_forward, p, gold, data, W, f"GS golden {tag} th{th} {precision}", ("log_assignment", "line_log_assignment", "raw_line_scores"),
                                       dict(filter_threshold=th), r64=r64, scores=GS_SCORES)
            for k in GS_INT:
                assert torch.equal(p[k].cpu(), g[f"{tag}.th{th}.{k}"]), (tag, th, k)
                assert p[k].dtype == torch.int64
            for k in GS_FLT:
                torch.testing.assert_close(p[k].cpu(), g[f"{tag}.th{th}.{k}"], rtol=0, atol=ATOL)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_gluestick_vs_oracle_large(dev, precision):
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("gs_inputs_src", os.path.join(os.path.dirname(__file__), "gs_inputs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    B, M, N, n0, n1 = 2, 1000, 1100, 250, 230
    data = mod.gs_inputs(B, M, N, n0, n1, 7, corresponding=True)  # lines of image 1 are the counterparts of image 0's: matches exist
    W = O.make_gs_weights(4, **O.GS_GOLDEN)
    ref = AN.memo(("gslarge.ref",), lambda: O.gluestick_forward(data, W, filter_threshold=0.2))  # shared by the two arithmetic modes
    p = _module(W, 0.2, precision)(_to(data, dev))
    # floats of both runs against the oracle in float64: 1e-4, or 2 x the fp32 CPU path's own error where that exceeds 1e-4 (oracle/anchor.py)
    r64 = AN.memo(("gslarge.r64",), lambda: O.gluestick_forward(O.cast(data, torch.float64), O.cast(W, torch.float64), filter_threshold=0.2))
    AN.matcher_floats(O.gluestick_forward, p, ref, data, W, f"GS large {precision}", ("log_assignment", "line_log_assignment", "raw_line_scores"),
                      dict(filter_threshold=0.2), scores=GS_SCORES, r64=r64)
    # assignments identical except at proven near-ties of the oracle's own decision (oracle/tie_aware.py); no percentage
    import tie_aware as TA

    for pre in ("", "line_"):
        for b in range(B):
            S = ref[pre + "log_assignment"][b]
            eps = TA.la_eps(p[pre + "log_assignment"][b].cpu(), S)
            r0 = TA.check_matches(p[pre + "matches0"][b].cpu(), ref[pre + "matches0"][b], S, 0.2, eps)
            r1 = TA.check_matches(p[pre + "matches1"][b].cpu(), ref[pre + "matches1"][b], S.t(), 0.2, eps)
            print(f"[{precision}] {pre}matches pair {b}: {r0['differing']}/{r0['compared']} and {r1['differing']}/{r1['compared']} differ (near-ties, eps {eps:.1e}); "
                  f"matched {(ref[pre + 'matches0'][b] > -1).float().mean().item():.3f}")
    # oracle-independent: mutual consistency of both assignments, rows of exp(S) are sub-stochastic
    for pre, m in (("", M), ("line_", n0)):
        m0, m1 = p[pre + "matches0"], p[pre + "matches1"]
        ok = m0 > -1
        idx = torch.arange(m, device=dev)[None].expand(B, -1)
        assert torch.all(torch.gather(m1, 1, m0.clamp(min=0))[ok] == idx[ok])
        assert torch.all(p[pre + "matching_scores0"][ok] > 0.2)


def test_gluestick_contract(dev):
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("gs_inputs_src", os.path.join(os.path.dirname(__file__), "gs_inputs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    W = O.make_gs_weights(2)
    m = _module(W, 0.2, "f16x3")
    data = _to(mod.gs_inputs(1, 64, 48, 8, 6, 5), dev)
    with pytest.raises(AssertionError):
        m({k: v for k, v in data.items() if k != "line_scores0"})
    a = m(data)
    # no lines in one image: point matching still runs, line outputs are empty / -1 (gluestick.py:291-312)
    nl = dict(data, lines1=data["lines1"][:, :0], lines_junc_idx1=data["lines_junc_idx1"][:, :0], line_scores1=data["line_scores1"][:, :0])
    p = m(nl)
    assert p["line_matches0"].shape == (1, 8) and torch.all(p["line_matches0"] == -1) and p["line_matches1"].shape == (1, 0)
    assert p["matches0"].shape == (1, 64) and p["raw_line_scores"].shape == (1, 8, 0)
    ref = O.gluestick_forward.__doc__ is not None
    assert ref and torch.equal(m(data)["log_assignment"], a["log_assignment"])  # deterministic (ordered junction means)
    with pytest.raises(RuntimeError):  # junctions must be the leading keypoints
        m(dict(data, keypoints0=data["keypoints0"][:, :10], descriptors0=data["descriptors0"][:, :10], keypoint_scores0=data["keypoint_scores0"][:, :10]))
