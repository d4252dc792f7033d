"""Synthetic wireframe-style GlueStick inputs (same generator as oracle/tools/make_goldens.py)."""
import torch


def gs_inputs(B, M, N, n0, n1, seed, share=0.6):
    """Synthetic wireframe-style inputs: the first junctions of the keypoint list are line endpoints
    (`lines_junc_idx` < 2 * n_lines, several endpoints share a junction), image 1 is a noisy re-ordering of image 0."""
    g = torch.Generator().manual_seed(300 + seed)
    size0 = torch.tensor([[640.0, 480.0]] * B)
    size1 = torch.tensor([[600.0, 480.0]] * B)
    K = max(M, N)
    kk = torch.rand(B, K, 2, generator=g) * size1[:, None] * 0.98 + 0.5
    dd = torch.nn.functional.normalize(torch.randn(B, K, 256, generator=g), dim=-1)
    k0, d0 = kk[:, :M].contiguous(), dd[:, :M].contiguous()
    perm = torch.stack([torch.randperm(K, generator=g)[:N] for _ in range(B)])
    k1 = (torch.gather(kk, 1, perm[..., None].expand(-1, -1, 2)) * 0.9 + 3.0 * torch.randn(B, N, 2, generator=g)).clamp(1.0, 479.0)
    d1 = torch.nn.functional.normalize(torch.gather(dd, 1, perm[..., None].expand(-1, -1, 256)) + 0.05 * torch.randn(B, N, 256, generator=g), dim=-1)
    out = {"keypoints0": k0, "keypoints1": k1, "descriptors0": d0, "descriptors1": d1,
           "keypoint_scores0": torch.rand(B, M, generator=g), "keypoint_scores1": torch.rand(B, N, generator=g),
           "view0": {"image_size": size0}, "view1": {"image_size": size1}}
    for side, n, kp in (("0", n0, k0), ("1", n1, k1)):
        J = max(2, int(2 * n * share))  # junctions actually used (< 2n: endpoints share junctions)
        idx = torch.randint(0, J, (B, n, 2), generator=g)
        idx[:, :, 1] = torch.where(idx[:, :, 1] == idx[:, :, 0], (idx[:, :, 1] + 1) % J, idx[:, :, 1])
        out["lines_junc_idx" + side] = idx
        out["lines" + side] = torch.gather(kp, 1, idx.reshape(B, 2 * n)[..., None].expand(-1, -1, 2)).reshape(B, n, 2, 2)
        out["line_scores" + side] = torch.rand(B, n, generator=g)
    return out
