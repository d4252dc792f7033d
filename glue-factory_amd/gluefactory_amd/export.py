"""Batched prediction export in the reference's h5 layout.

`gluefactory/utils/export_predictions.py:17-81` runs the model on one pair at a time and writes `v[0]` of every
prediction into an h5 group named after the pair (datasets keyed by prediction name, keypoints / lines rescaled by
`1 / scales`, optional fp16).  This exporter writes the SAME groups and datasets from batches of any size, so the
reference's `run_eval` / `CacheLoader` (`models/cache_loader.py:59-141`) read GPU-batched results unmodified.

With batching the extractor emits a fixed number of rows per image; `trim_padding=True` (default) cuts each image's
tensors back to its detected keypoints (rows with a positive score come first; switch it off when the detection
threshold is negative and zero scores are legitimate), which is what the one-pair-at-a-time reference writes.  `h5py` is imported lazily (it is not installed in the build image); pass `h5=` to inject it.
"""
from pathlib import Path

import numpy as np
import torch

_PER_VIEW = ("keypoints", "keypoint_scores", "descriptors", "matches", "matching_scores")


def _to_device(x, device):
    if isinstance(x, dict):
        return {k: _to_device(v, device) for k, v in x.items()}
    if torch.is_tensor(x):
        return x.to(device, non_blocking=True)
    return x


def _scales(data, idx):
    s = data["scales"] if len(idx) == 0 else data[f"view{idx}"]["scales"]
    return 1.0 / s


def _trim(item, suffixes):
    """Cut the fixed-size rows of one pair back to the detected keypoints of each view."""
    counts = {}
    for sfx in suffixes:
        sc = item.get("keypoint_scores" + sfx)
        if sc is not None:
            counts[sfx] = int((sc > 0).sum())
    for sfx, n in counts.items():
        for base in _PER_VIEW:
            k = base + sfx
            if k in item and item[k].shape[0] >= n:
                item[k] = item[k][:n]
    if "0" in counts and "1" in counts:  # a match into the other view's padding is no match
        for a, b in (("0", "1"), ("1", "0")):
            k = "matches" + a
            if k in item:
                m = item[k].copy()
                m[m >= counts[b]] = -1
                item[k] = m
    return item


@torch.no_grad()
def export_predictions(loader, model, output_file, as_half=False, keys="*", callback_fn=None, optional_keys=(), trim_padding=True, h5=None):
    assert keys == "*" or isinstance(keys, (tuple, list))
    if h5 is None:
        import h5py as h5  # noqa: N813
    Path(output_file).parent.mkdir(exist_ok=True, parents=True)
    hfile = h5.File(str(output_file), "w")
    device = "cuda" if torch.cuda.is_available() else "cpu"
    model = model.to(device).eval() if hasattr(model, "to") else model
    for data_ in loader:
        data = _to_device(data_, device)
        pred = model(data)
        if callback_fn is not None:
            pred = {**callback_fn(pred, data), **pred}
        if keys != "*":
            if len(set(keys) - set(pred.keys())) > 0:
                raise ValueError(f"Missing key {set(keys) - set(pred.keys())}")
            pred = {k: v for k, v in pred.items() if k in list(keys) + list(optional_keys)}
        assert len(pred) > 0
        # renormalisation (export_predictions.py:42-62), per item of the batch
        for k in list(pred.keys()):
            for prefix in ("keypoints", "orig_lines", "lines"):
                if k.startswith(prefix) and not (prefix == "lines" and k.startswith("lines_junc")):
                    sc = _scales(data, k.replace(prefix, ""))  # (B,2)
                    pred[k] = pred[k] * sc.reshape((sc.shape[0],) + (1,) * (pred[k].dim() - 2) + (2,))
                    break
        names = data["name"]
        B = len(names)
        host = {k: v.cpu().numpy() for k, v in pred.items()}
        suffixes = sorted({k[len("keypoint_scores"):] for k in host if k.startswith("keypoint_scores")})
        for b in range(B):
            item = {k: v[b] for k, v in host.items()}
            if trim_padding:
                item = _trim(item, suffixes)
            if as_half:
                for k in item:
                    if item[k].dtype == np.float32:
                        item[k] = item[k].astype(np.float16)
            try:
                grp = hfile.create_group(names[b])
                for k, v in item.items():
                    grp.create_dataset(k, data=v)
            except RuntimeError:
                continue
        del pred
    hfile.close()
    return output_file
