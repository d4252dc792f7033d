"""Deterministic fake loader / model outputs for the export test (same generator as oracle/tools/make_goldens.py)."""
import torch


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
