"""Data-parallel sharding of image pairs over the GPUs of one node.

Pairs are independent units (no exchange inside the path), so each rank runs the whole
SuperPoint + LightGlue forward on a contiguous block of B/G pairs with replicated weights; the only
collective is ONE all-gather of the fixed-shape match outputs (RCCL over xGMI when the process group
backend is "nccl", gloo in the CPU tests).  Indices travel as int32 and are widened to the
reference's int64 on return; descriptors are not gathered (134 MB/rank at config 3) unless asked.

The reference has no inference-time collective (its only distributed code is DDP training,
gluefactory/train.py:235-257); this is the MI355X-native addition named by BASELINE.json.
"""
import torch
import torch.distributed as dist

GATHER_KEYS = ("matches0", "matches1", "matching_scores0", "matching_scores1")
OPTIONAL_KEYS = ("keypoints0", "keypoints1", "keypoint_scores0", "keypoint_scores1")


def shard_range(num_pairs, rank, world_size):
    """Contiguous block of pairs owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(num_pairs, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(data, rank, world_size):
    """Slice every batched tensor of a (possibly nested) two-view data dict to this rank's pairs."""
    n = data["view0"]["image"].shape[0]
    lo, hi = shard_range(n, rank, world_size)

    def cut(x):
        if isinstance(x, dict):
            return {k: cut(v) for k, v in x.items()}
        if torch.is_tensor(x) and x.dim() > 0 and x.shape[0] == n:
            return x[lo:hi]
        return x

    return cut(data)


def all_gather_matches(pred, group=None, with_keypoints=False, force=False, num_pairs=None, stats=None):
    """All-gather the match outputs of every rank -> full-batch tensors (pair order = rank order) on all ranks.

    One packed buffer per rank -> a single `all_gather_into_tensor` call.  Shards may differ in size
    (`shard_range` hands out blocks that differ by one pair when B % world != 0): every rank pads its buffer to the
    largest shard and the padding rows are dropped on return.  The shard sizes come from `num_pairs` (the global
    batch, split by `shard_range`) when given -- no extra traffic, no host synchronisation: what bench.py and any caller that
    knows its global batch uses -- and from one tiny all-gather of the local row counts otherwise (that branch reads the counts
    back to the host, one D2H sync per call: the output's row count has to be known to allocate it).  `force` runs the collective even in a one-rank group (the single-GPU test of the RCCL path).
    `stats`, if a dict, receives `bytes_sent` / `bytes_received` of this call.
    """
    keys = [k for k in GATHER_KEYS + (OPTIONAL_KEYS if with_keypoints else ()) if k in pred]
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        if stats is not None:
            stats.update(bytes_sent=0, bytes_received=0)
        return {k: pred[k] for k in keys}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts, meta = [], []
    for k in keys:
        t = pred[k]
        if t.dtype == torch.int64:
            w = t.contiguous().to(torch.int32).view(torch.float32)
        else:
            w = t.float()
        meta.append((k, t.shape, t.dtype))
        parts.append(w.reshape(t.shape[0], -1))
    packed = torch.cat(parts, 1).contiguous()  # (b_local, F) fp32 words
    b_local, F = packed.shape
    if num_pairs is not None:
        counts = [hi - lo for lo, hi in (shard_range(num_pairs, r, world) for r in range(world))]
        if counts[rank] != b_local:
            raise ValueError(f"rank {rank} holds {b_local} pairs but shard_range({num_pairs}, {rank}, {world}) gives {counts[rank]}")
    else:
        c = torch.tensor([b_local], dtype=torch.int64, device=packed.device)
        cs = torch.empty(world, dtype=torch.int64, device=packed.device)
        dist.all_gather_into_tensor(cs, c, group=group)
        counts = [int(v) for v in cs.tolist()]
    b_max = max(counts)
    if b_local < b_max:  # equal send counts are required by the collective (RCCL: undefined behaviour otherwise)
        packed = torch.cat([packed, packed.new_zeros(b_max - b_local, F)], 0)
    out = torch.empty((world * b_max, F), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    if stats is not None:
        stats.update(bytes_sent=packed.numel() * 4, bytes_received=out.numel() * 4)
    if min(counts) != b_max:
        out = torch.cat([out[r * b_max: r * b_max + n] for r, n in enumerate(counts)], 0)
    res, col = {}, 0
    for k, shape, dtype in meta:
        f = 1
        for d in shape[1:]:
            f *= int(d)
        chunk = out[:, col: col + f]
        col += f
        if dtype == torch.int64:
            chunk = chunk.contiguous().view(torch.int32).to(torch.int64)
        res[k] = chunk.reshape((out.shape[0],) + tuple(shape[1:]))
    return res
