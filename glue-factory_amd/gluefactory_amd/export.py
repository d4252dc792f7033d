"""Batched prediction export / read-back in the reference's h5 layout.

Layout (what `gluefactory/utils/export_predictions.py:17-81` writes and `models/cache_loader.py:59-141` reads): one h5
group per pair name, one dataset per prediction key holding ONE pair's tensor, coordinates (`keypoints*`, `lines*`,
`orig_lines*`) stored in original-image pixels, i.e. divided by the view's `scales`; optionally fp16.

The reference runs the model on one pair at a time and writes `v[0]`.  `PredictionWriter` takes batches of any size
from the GPU path: it undoes the fixed-K padding of a batch item by item (`trim_padding`: rows with a positive keypoint
score come first; matches into the other view's padding become -1) and writes the same groups and datasets, so the
reference's `run_eval` / `CacheLoader` consume GPU-batched results unmodified (asserted in oracle/tools/make_goldens.py by
opening both files with the reference's own loader).  `read_predictions` is the inverse used to feed exported features
back through `TwoViewPipeline`'s `cache` path.  `h5py` is imported lazily (not installed in the build image); `h5=` injects it.
"""
from pathlib import Path

import numpy as np
import torch

PIXEL_KEYS = ("keypoints", "orig_lines", "lines")  # prefixes of datasets held in original-image pixels
_ROWS_PER_KEYPOINT = ("keypoints", "keypoint_scores", "descriptors", "matches", "matching_scores")


def _h5(h5):
    if h5 is not None:
        return h5
    import h5py

    return h5py


def _view_of(key):
    """'keypoints1' -> ('keypoints', '1'); None for keys that carry no pixel coordinates ('lines_junc_idx0' is an index table)."""
    if key.startswith("lines_junc"):
        return None
    for prefix in PIXEL_KEYS:
        if key.startswith(prefix):
            return prefix, key[len(prefix):]
    return None


def _view_scales(data, view):
    return data["scales"] if view == "" else data["view" + view]["scales"]


class PredictionWriter:
    def __init__(self, path, as_half=False, trim_padding=True, h5=None):
        Path(path).parent.mkdir(exist_ok=True, parents=True)
        self.file = _h5(h5).File(str(path), "w")
        self.as_half, self.trim_padding = as_half, trim_padding

    def close(self):
        self.file.close()

    @staticmethod
    def _unpad(item):
        """One pair's arrays cut back from the batch's fixed row count to the keypoints each view really has."""
        found = {k[len("keypoint_scores"):]: int((v > 0).sum()) for k, v in item.items() if k.startswith("keypoint_scores")}
        for view, n in found.items():
            for base in _ROWS_PER_KEYPOINT:
                key = base + view
                if key in item and item[key].shape[0] >= n:
                    item[key] = item[key][:n]
        for view, other in (("0", "1"), ("1", "0")):
            if "matches" + view in item and view in found and other in found:
                m = item["matches" + view]
                item["matches" + view] = np.where(m >= found[other], -1, m).astype(m.dtype)
        return item

    def add_batch(self, names, pred, data):
        """pred: dict of (B, ...) tensors on any device; data: the batch (for the views' `scales`)."""
        host = {}
        for key, t in pred.items():
            kv = _view_of(key)
            if kv is not None:  # back to original-image pixels, per item
                inv = 1.0 / _view_scales(data, kv[1]).to(t.device)
                t = t * inv.reshape((inv.shape[0],) + (1,) * (t.dim() - 2) + (2,))
            host[key] = t.cpu().numpy()
        for b, name in enumerate(names):
            item = {k: v[b] for k, v in host.items()}
            if self.trim_padding:
                item = self._unpad(item)
            try:
                grp = self.file.create_group(name)
            except RuntimeError:  # a pair exported twice keeps its first record, as in the reference
                continue
            for k, v in item.items():
                grp.create_dataset(k, data=v.astype(np.float16) if self.as_half and v.dtype == np.float32 else v)


def _to_device(x, device):
    if isinstance(x, dict):
        return {k: _to_device(v, device) for k, v in x.items()}
    return x.to(device, non_blocking=True) if torch.is_tensor(x) else x


@torch.no_grad()
def export_predictions(loader, model, output_file, as_half=False, keys="*", callback_fn=None, optional_keys=(), trim_padding=True, h5=None):
    """Same call as the reference's `export_predictions` (utils/export_predictions.py:17-27), any batch size."""
    assert keys == "*" or isinstance(keys, (tuple, list))
    writer = PredictionWriter(output_file, as_half, trim_padding, h5)
    device = "cuda" if torch.cuda.is_available() else "cpu"
    if hasattr(model, "to"):
        model = model.to(device).eval()
    for batch in loader:
        data = _to_device(batch, device)
        pred = model(data)
        if callback_fn is not None:
            pred = {**callback_fn(pred, data), **pred}
        if keys != "*":
            missing = set(keys) - set(pred)
            if missing:
                raise ValueError(f"Missing key {missing}")
            wanted = set(keys) | set(optional_keys)
            pred = {k: v for k, v in pred.items() if k in wanted}
        assert len(pred) > 0
        writer.add_batch(data["name"], pred, data)
    writer.close()
    return output_file


def read_predictions(path, names, data=None, keys=None, dtype=torch.float32, device="cpu", h5=None):
    """Inverse of `PredictionWriter`: per pair name a dict of tensors; floating datasets cast to `dtype`, pixel-coordinate
    datasets multiplied back by the views' `scales` taken from `data` (row i of `data` belongs to names[i]; None = leave them
    in original-image pixels).  Returns a list with one dict per name."""
    f = _h5(h5).File(str(path), "r")
    out = []
    for i, name in enumerate(names):
        grp = f[name]
        item = {}
        for key in (keys if keys is not None else list(grp.keys())):
            t = torch.from_numpy(np.asarray(grp[key])).to(device)
            if t.is_floating_point() and dtype is not None:
                t = t.to(dtype)
            kv = _view_of(key)
            if kv is not None and data is not None:
                t = t * _view_scales(data, kv[1])[i].to(t.device)
            item[key] = t
        out.append(item)
    f.close()
    return out
