"""Synthetic wireframe-style GlueStick inputs (the generator oracle/tools/make_goldens.py uses for tests/golden/gluestick.npz)."""
import torch


def gs_inputs(B, M, N, n0, n1, seed, share=0.6, corresponding=False):
    """Synthetic wireframe-style inputs: the first junctions of the keypoint list are line endpoints
    (`lines_junc_idx` < 2 * n_lines, several endpoints share a junction), image 1 is a noisy re-ordering of image 0.

    `corresponding=True` makes the LINES correspond as well: the leading junction keypoints of image 1 are the
    counterparts of image 0's, and the first min(n0, n1) segments of image 1 join the counterparts of the junctions
    the same segment joins in image 0 (endpoint order flipped for every other one) -- so that point AND line matches
    exist to be found (the independent random segments of the plain generator have no counterparts)."""
    g = torch.Generator().manual_seed(300 + seed)
    size0 = torch.tensor([[640.0, 480.0]] * B)
    size1 = torch.tensor([[600.0, 480.0]] * B)
    K = max(M, N)
    J0, J1 = max(2, int(2 * n0 * share)), max(2, int(2 * n1 * share))  # junctions actually used (< 2n: endpoints share junctions)
    Jc = min(J0, J1, M, N)
    kk = torch.rand(B, K, 2, generator=g) * size1[:, None] * 0.98 + 0.5
    dd = torch.nn.functional.normalize(torch.randn(B, K, 256, generator=g), dim=-1)
    k0, d0 = kk[:, :M].contiguous(), dd[:, :M].contiguous()
    if corresponding:
        perm = torch.stack([torch.cat([torch.randperm(Jc, generator=g), Jc + torch.randperm(K - Jc, generator=g)])[:N] for _ in range(B)])
    else:
        perm = torch.stack([torch.randperm(K, generator=g)[:N] for _ in range(B)])
    k1 = (torch.gather(kk, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 3.0 * torch.randn(B, N, 2, generator=g)).clamp(1.0, 479.0)
    d1 = torch.nn.functional.normalize(torch.gather(dd, 1, perm[..., None].expand(-1, -1, 256)) + 0.05 * torch.randn(B, N, 256, generator=g), dim=-1)
    out = {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1,
           "keypoint_scores0": torch.rand(B, M, generator=g), "keypoint_scores1": torch.rand(B, N, generator=g),
           "view0": {"image_size": size0}, "view1": {"image_size": size1}}
    idx = {}
    for side, n, J in (("0", n0, J0), ("1", n1, J1)):
        ii = torch.randint(0, J, (B, n, 2), generator=g)
        ii[:, :, 1] = torch.where(ii[:, :, 1] == ii[:, :, 0], (ii[:, :, 1] + 1) % J, ii[:, :, 1])
        idx[side] = ii
    if corresponding:
        nc = min(n0, n1)
        a = torch.randint(0, Jc, (B, nc, 2), generator=g)
        a[:, :, 1] = torch.where(a[:, :, 1] == a[:, :, 0], (a[:, :, 1] + 1) % Jc, a[:, :, 1])
        idx["0"][:, :nc] = a
        inv = torch.empty(B, Jc, dtype=torch.long)
        for b in range(B):
            inv[b, perm[b, :Jc]] = torch.arange(Jc)  # image-0 junction i is image-1 keypoint inv[i]
        c = torch.gather(inv, 1, a.reshape(B, -1)).reshape(B, nc, 2)
        c[:, 1::2] = torch.flip(c[:, 1::2], [-1])
        idx["1"][:, :nc] = c
    for side, n, kp in (("0", n0, k0), ("1", n1, k1)):
        out["lines_junc_idx" + side] = idx[side]
        out["lines" + side] = torch.gather(kp, 1, idx[side].reshape(B, 2 * n)[..., None].expand(-1, -1, 2)).reshape(B, n, 2, 2)
        out["line_scores" + side] = torch.rand(B, n, generator=g)
    return out
