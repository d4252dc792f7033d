"""Reader of the h5 prediction cache, mirror of `gluefactory.models.cache_loader.CacheLoader`
(gluefactory/models/cache_loader.py:59-141): same conf keys, same behaviour for the keys the hot path uses -- one h5
group per pair name, one dataset per prediction key, floating-point datasets cast to `numeric_type`, `keypoints*` /
`lines*` / `orig_lines*` multiplied back by the view's `scales`.  Host code only (no arithmetic beyond that multiply);
it exists so that features exported from GPU batches (`gluefactory_amd.export`) can be fed back through
`TwoViewPipeline`'s `cache` path (two_view_pipeline.py:62-70) without the reference package.

Not built: `padding_fn` (the reference pads with random points, models/utils/misc.py:43-54) and DATA_PATH-relative paths
(`add_data_path`): pass the path as it is.  `h5py` is imported lazily; pass `h5=` to inject a stand-in.
"""
import string

import torch

from .base_model import BaseModel


def _load_group(grp, keys):
    out = {}
    for k in keys:
        v = grp[k]
        if hasattr(v, "keys") and not hasattr(v, "__array__"):
            out[k] = _load_group(v, list(v.keys()))
        else:
            out[k] = torch.from_numpy(v.__array__())
    return out


class CacheLoader(BaseModel):
    default_conf = {
        "path": "???",  # may be a format string like exports/{scene}.h5
        "data_keys": None,
        "device": None,
        "trainable": False,
        "add_data_path": False,
        "collate": True,
        "scale": ["keypoints", "lines", "orig_lines"],
        "padding_fn": None,
        "padding_length": None,
        "numeric_type": "float32",
    }
    required_data_keys = ["name"]

    def __init__(self, conf=None, h5=None):
        self._h5 = h5
        super().__init__(conf)

    def _init(self, conf):
        if conf.padding_fn is not None:
            raise NotImplementedError("padding_fn (random padding points) is not built; cached features come fixed-size from the batched exporter")
        if conf.add_data_path:
            raise NotImplementedError("add_data_path: pass the full path of the h5 file")
        self.numeric_dtype = {None: None, "float16": torch.float16, "float32": torch.float32, "float64": torch.float64}[conf.numeric_type]
        self.set_initialized()

    def _forward(self, data):
        h5 = self._h5
        if h5 is None:
            import h5py as h5  # noqa: N813
        device = self.conf.device
        if not device:
            devices = {v.device for v in data.values() if isinstance(v, torch.Tensor)}
            assert len(devices) <= 1
            device = devices.pop() if devices else "cpu"
        var_names = [x[1] for x in string.Formatter().parse(self.conf.path) if x[1]]
        preds = []
        for i, name in enumerate(data["name"]):
            fpath = self.conf.path.format(**{k: data[k][i] for k in var_names})
            hfile = h5.File(str(fpath), "r")
            grp = hfile[name]
            keys = self.conf.data_keys if self.conf.data_keys is not None else list(grp.keys())
            pred = _load_group(grp, keys)
            if self.numeric_dtype is not None:
                pred = {k: (v.to(self.numeric_dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in pred.items()}
            pred = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in pred.items()}
            for k in list(pred):
                for pattern in self.conf.scale:
                    if k.startswith(pattern):
                        view_idx = k.replace(pattern, "")
                        scales = data["scales"] if len(view_idx) == 0 else data[f"view{view_idx}"]["scales"]
                        pred[k] = pred[k] * scales[i].to(pred[k].device)
            preds.append(pred)
            hfile.close()
        if self.conf.collate:
            return {k: torch.stack([p[k] for p in preds]) for k in preds[0]}
        assert len(preds) == 1
        return preds[0]

    def loss(self, pred, data):
        raise NotImplementedError


__main_model__ = CacheLoader
