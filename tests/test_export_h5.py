"""Batched h5 export against the layout the reference's own export_predictions wrote.

tests/golden/export_h5.npz holds every group / dataset that gluefactory/utils/export_predictions.py produced (batch 1, the
only size it supports) for the fixture; the batched exporter must write the same groups, dataset names, dtypes and
values from zero-padded batches of 4 and of 1.  h5py is not installed here: an in-memory stand-in records the calls.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(__file__))
from export_fixture import export_fixture  # noqa: E402


class _Group(dict):
    def create_group(self, name):
        if name in self:
            raise RuntimeError("exists")
        self[name] = _Group()
        return self[name]

    def create_dataset(self, name, data=None):
        self[name] = np.asarray(data)


class _H5:
    """In-memory stand-in for the h5py calls the exporter and the cache loader make (h5py is not installed in this image)."""
    last = None
    files = {}

    class File(_Group):
        def __init__(self, path, mode):
            super().__init__()
            if mode == "r":
                self.update(_H5.files[str(path)])
            else:
                _H5.files[str(path)] = self
                _H5.last = self

        def close(self):
            pass


def _batched_loader(pairs, bs, K=12):
    """Zero-padded fixed-K batches, the shape the native extractor / matcher emit."""
    batches = []
    for i in range(0, len(pairs), bs):
        chunk = pairs[i: i + bs]
        pred = {}
        for key in chunk[0][1]:
            rows = []
            for _, item, _, _ in chunk:
                t = item[key]
                fill = -1 if key.startswith("matches") else 0
                pad = torch.full((K - t.shape[0],) + tuple(t.shape[1:]), fill, dtype=t.dtype)
                rows.append(torch.cat([t, pad], 0) if bs > 1 else t)
            pred[key] = torch.stack(rows)
        data = {"name": [c[0] for c in chunk], "view0": {"scales": torch.stack([c[2] for c in chunk])},
                "view1": {"scales": torch.stack([c[3] for c in chunk])}, "_pred": pred}
        batches.append(data)
    return batches


def _model(data):
    return dict(data["_pred"])


def test_batched_export_matches_reference_layout(golden, tmp_path):
    from gluefactory_amd.export import export_predictions

    g = golden("export_h5")
    pairs = export_fixture()
    for half in (False, True):
        want = {k[len(f"half{int(half)}|"):]: g[k] for k in g.keys() if k.startswith(f"half{int(half)}|")}
        for bs in (1, 4, 3):
            export_predictions(_batched_loader(pairs, bs), _model, tmp_path / "pred.h5", as_half=half, h5=_H5)
            got = {f"{name}|{k}": v for name, grp in _H5.last.items() for k, v in grp.items()}
            assert sorted(got) == sorted(want), (bs, half)
            for k in want:
                w = want[k].numpy()
                assert got[k].dtype == w.dtype and got[k].shape == w.shape, (bs, half, k, got[k].dtype, w.dtype, got[k].shape, w.shape)
                np.testing.assert_array_equal(got[k], w, err_msg=f"{bs} {half} {k}")


def test_export_key_filter_and_errors(tmp_path):
    from gluefactory_amd.export import export_predictions

    pairs = export_fixture()
    loader = _batched_loader(pairs, 2)
    export_predictions(loader, _model, tmp_path / "p.h5", keys=["keypoints0", "matches0"], optional_keys=["matching_scores0"], h5=_H5)
    grp = next(iter(_H5.last.values()))
    assert sorted(grp) == ["keypoints0", "matches0", "matching_scores0"]
    try:
        export_predictions(loader, _model, tmp_path / "p.h5", keys=["keypoints0", "nope"], h5=_H5)
        raise AssertionError("missing key must raise")
    except ValueError:
        pass


def test_read_back_equals_what_the_reference_loader_reads(golden, tmp_path):
    """Batched export -> gluefactory_amd.export.read_predictions == what the reference's own CacheLoader returned for the same file
    (recorded by oracle/tools/make_goldens.py after checking that it reads the reference's and the batched export identically)."""
    from gluefactory_amd.export import export_predictions, read_predictions

    g = golden("export_h5")
    pairs = export_fixture()
    for half in (False, True):
        path = tmp_path / f"pred{int(half)}.h5"
        export_predictions(_batched_loader(pairs, 3), _model, path, as_half=half, h5=_H5)
        for name, _, s0, s1 in pairs:
            got = read_predictions(path, [name], {"view0": {"scales": s0[None]}, "view1": {"scales": s1[None]}}, h5=_H5)[0]
            want = {k.split("|", 2)[2]: g[k] for k in g.keys() if k.startswith(f"cache{int(half)}|{name}|")}
            assert sorted(got) == sorted(want) and len(want) == 10
            for k in want:
                assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), (half, name, k)
    # a key subset of several pairs, coordinates left in original-image pixels
    out = read_predictions(path, [pairs[0][0], pairs[3][0]], keys=["matching_scores0", "keypoints0"], h5=_H5)
    assert len(out) == 2 and sorted(out[0]) == ["keypoints0", "matching_scores0"]
