"""The weight-stationary Linear of the QKV projections (csrc/gemm_ws_f16.hip: weights in registers, persistent workgroups, the
epilogue of tile t-1 issued in the MFMA shadows of tile t) against the generic tiled GEMM it replaces for those launches
(csrc/gemm_f16.hip, still in the library behind the debug request "lg.gemm_generic").  Same products in the same order, the same
epilogue arithmetic operation for operation: every output of the matcher must be BIT-IDENTICAL -- taps after the first self block,
the first and the last layer, the assignment matrix, the matches.  Shapes: full tiles, ragged token counts (rows past M are dropped
by the buffer range check), fewer tokens than one 128-token tile, more tiles than persistent workgroups per feature group."""
import pytest
import torch

from oracle import sp_lg_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _data(B, M, N, seed):
    g = torch.Generator().manual_seed(seed)
    size = torch.tensor([[640.0, 480.0]] * B)
    k0 = torch.rand(B, M, 2, generator=g) * size[:, None] * 0.98 + 0.5
    d0 = torch.nn.functional.normalize(torch.randn(B, M, 256, generator=g), dim=-1)
    perm = torch.stack([torch.randperm(max(M, N), generator=g)[:N] % M for _ in range(B)])
    k1 = torch.gather(k0, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 2.0 * torch.randn(B, N, 2, generator=g)
    d1 = torch.nn.functional.normalize(torch.gather(d0, 1, perm[..., None].expand(-1, -1, 256)) + 0.03 * torch.randn(B, N, 256, generator=g), dim=-1)
    return {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1, "view0": {"image_size": size}, "view1": {"image_size": size}}


def _bits(a, b):
    return torch.equal(a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)) if a.dtype.is_floating_point else torch.equal(a, b)


@pytest.mark.parametrize("B,M,N", [(2, 1024, 1024), (1, 130, 61), (3, 129, 517), (1, 1, 5), (24, 1024, 1024)])
def test_weight_stationary_qkv_equals_generic_gemm(B, M, N):
    from gluefactory_amd import get_model

    dev = torch.device("cuda:0")
    W = O.make_lg_weights(5, structured=True, final_gain=3.0, match_bias=2.0)
    data = _data(B, M, N, B + M)
    to = lambda x: {k: (to(v) if isinstance(v, dict) else v.to(dev)) for k, v in x.items()}
    outs = []
    for generic in (False, True):
        m = get_model("gluefactory_amd.lightglue")({"filter_threshold": 0.1, "precision": "f16x3"}).eval()
        m.load_state_dict(W)
        eng = m.engine(dev)
        for tap in ("lg.self0", "lg.layer0", "lg.layer8", "lg.qkv0"):
            eng.debug_request(tap)
        if generic:
            eng.debug_request("lg.gemm_generic")
        p = m(to(data))
        T = B * (M + N)
        taps = {t: eng.debug_read(t, T * 256).cpu() for t in ("lg.self0", "lg.layer0", "lg.layer8")}
        taps["lg.qkv0"] = eng.debug_read("lg.qkv0", T * 768).cpu()
        outs.append(({k: v.cpu() for k, v in p.items() if torch.is_tensor(v)}, taps))
    (pw, tw), (pg, tg) = outs
    # the raw SH8 words of the first projection, column block by column block (q: rotary + scale, k: rotary, v: neither)
    qw, qg = tw["lg.qkv0"].view(torch.int32).view(-1, 768), tg["lg.qkv0"].view(torch.int32).view(-1, 768)
    for name, c0 in (("q", 0), ("k", 256), ("v", 512)):
        bad = (qw[:, c0:c0 + 256] != qg[:, c0:c0 + 256])
        if bool(bad.any()):
            r, c = [int(x[0]) for x in torch.nonzero(bad, as_tuple=True)]
            print(f"{name}: {int(bad.sum())} of {bad.numel()} words differ; first at row {r}, word {c} (column block of 8: {c // 8}, word-in-block {c % 8}): "
                  f"ws 0x{int(qw[r, c0 + c]) & 0xffffffff:08x} generic 0x{int(qg[r, c0 + c]) & 0xffffffff:08x}; rows affected {int(bad.any(1).sum())}, columns affected {sorted(set((torch.nonzero(bad.any(0)).flatten() // 32).tolist()))}")
    for t in tw:
        d = (tw[t] - tg[t]).abs().max().item()
        assert _bits(tw[t], tg[t]), f"{t}: weight-stationary differs from the generic GEMM by up to {d:.3e}"
    for k in sorted(pg):
        assert _bits(pw[k], pg[k]), k
    assert bool(torch.isfinite(pw["log_assignment"]).all())
