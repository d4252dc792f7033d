"""Host-side mirror of the reference's model plug-in contract.

Reference: gluefactory/models/base_model.py:25-157 (`BaseModel`): `__init__(conf)` merges the
`default_conf` chain with the user conf and calls `_init`; `forward(data)` checks
`required_data_keys` then calls `_forward`; `is_initialized()` / `set_initialized()` gate the eval
harness (gluefactory/eval/io.py:87-97).  This class reproduces that contract without importing the
reference or omegaconf: `conf` may be a plain dict or an OmegaConf DictConfig (anything Mapping-like).

Weights live in a tree of *buffers* named exactly like the reference state_dict keys, so
`load_state_dict` accepts reference checkpoints unchanged and the composite's `is_initialized`
(which counts *parameters* of foreign children) treats these modules as ready.
"""
from collections.abc import Mapping

import torch
from torch import nn


class Conf(dict):
    """Read-only attribute dict (the subset of DictConfig behaviour the model layer uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"Missing key {k} in configuration")

    def __setattr__(self, k, v):
        raise TypeError("configuration is read-only")


def _to_plain(c):
    if isinstance(c, Mapping):
        return {k: _to_plain(v) for k, v in c.items()}
    if isinstance(c, (list, tuple)) or type(c).__name__ == "ListConfig":
        return [_to_plain(v) for v in c]
    return c


def _freeze(c):
    if isinstance(c, dict):
        return Conf({k: _freeze(v) for k, v in c.items()})
    return c


def merge_conf(*confs):
    out = {}
    for c in confs:
        for k, v in _to_plain(c or {}).items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = merge_conf(out[k], v)
            else:
                out[k] = v
    return out


class WeightTree(nn.Module):
    """Nested container of buffers addressed by dotted reference keys.

    The owner (`NativeModel`) emits and ingests these tensors under the reference's key names itself, so the
    tree takes no part in `nn.Module.load_state_dict`'s recursion (its keys would carry a `weights.` infix).
    """

    def set(self, key, tensor):
        head, _, rest = key.partition(".")
        if not rest:
            if head in self._buffers:
                self._buffers[head] = tensor
            else:
                self.register_buffer(head, tensor)
            return
        if head not in self._modules:
            self.add_module(head, WeightTree())
        self._modules[head].set(rest, tensor)

    def _load_from_state_dict(self, *args, **kwargs):
        return


class BaseModel(nn.Module):
    base_default_conf = {
        "name": None,
        "trainable": True,
        "freeze_batch_normalization": False,
        "timeit": False,
    }
    default_conf = {}
    required_data_keys = []
    strict_conf = False
    are_weights_initialized = False

    def __init__(self, conf=None):
        super().__init__()
        chain = [BaseModel.base_default_conf]
        for klass in reversed(type(self).__mro__):
            dc = klass.__dict__.get("default_conf")
            if isinstance(dc, dict):
                chain.append(dc)
        default = merge_conf(*chain)
        user = _to_plain(conf or {})
        if self.strict_conf:
            unknown = set(user) - set(default)
            if unknown:
                raise KeyError(f"unknown configuration entries: {sorted(unknown)}")
        self.conf = _freeze(merge_conf(default, user))
        self.required_data_keys = list(self.required_data_keys)
        self._init(self.conf)

    # -- reference contract ----------------------------------------------------------------------
    def forward(self, data):
        def check(expected, given):
            for key in expected:
                assert key in given, f"Missing key {key} in data"
                if isinstance(expected, dict):
                    check(expected[key], given[key])

        check(self.required_data_keys, data)
        return self._forward(data)

    def _init(self, conf):
        raise NotImplementedError

    def _forward(self, data):
        raise NotImplementedError

    def loss(self, pred, data):
        raise NotImplementedError

    def load_state_dict(self, *args, **kwargs):
        """base_model.py:132-136: load, then mark the model (and its BaseModel children) initialized."""
        ret = super().load_state_dict(*args, **kwargs)
        self.set_initialized()
        return ret

    def is_initialized(self):
        ok = True
        for _, child in self.named_children():
            if isinstance(child, BaseModel):
                ok = ok and child.is_initialized()
            elif isinstance(child, WeightTree):
                ok = ok and self.are_weights_initialized
            else:
                n = len(list(child.parameters()))
                ok = ok and (n == 0 or self.are_weights_initialized)
        return ok

    def set_initialized(self, to=True):
        self.are_weights_initialized = to
        for child in self.children():
            if isinstance(child, BaseModel):
                child.set_initialized(to)


class NativeModel(BaseModel):
    """A BaseModel whose arithmetic runs in libmatchglue; holds the weights and the native handle."""

    which = 0  # 0 SuperPoint, 1 LightGlue, 2 SuperGlue, 3 GlueStick (mg_set_weight `which`)

    # weights -------------------------------------------------------------------------------------
    def _init_weights(self, conf):
        self._engine = None
        self._uploaded_version = -1
        self._weights_version = 0
        self.weights = WeightTree()
        # One empty parameter while no weights are loaded: the reference's composite counts the *parameters* of foreign
        # children to decide `is_initialized()` (base_model.py:138-151, guard eval/io.py:92-96); it disappears with the
        # first successful load, so a pipeline built without `weights:` is rejected there instead of at the first forward.
        self.weights_not_loaded = nn.Parameter(torch.zeros(0), requires_grad=False)
        path = conf.get("weights")
        if path:
            import os

            if not os.path.exists(str(path)):
                raise FileNotFoundError(
                    f"weights '{path}' not found: this build has no network access and takes a local "
                    "state_dict file (reference key names)"
                )
            self.load_state_dict(torch.load(str(path), map_location="cpu"))

    # Loading composes with nn.Module's recursion: a composite (this package's TwoViewPipeline or the reference's own,
    # gluefactory/models/two_view_pipeline.py) calls `_load_from_state_dict` of every child with the child's prefix --
    # the reference's checkpoint path, utils/experiments.py:90 `model.load_state_dict(ckpt["model"], strict=False)`.
    def _ingest(self, sd):
        """Take the tensors this model needs out of a (prefix-free) state_dict; returns (loaded, unexpected) key lists."""
        sd = self._rename_keys(dict(sd))
        loaded, unexpected = [], []
        for k, v in sd.items():
            if k == "weights_not_loaded":
                continue
            if k.startswith("weights."):  # a state_dict written through the plain nn.Module path of an older build
                k = k[len("weights."):]
            if torch.is_tensor(v) and v.dtype.is_floating_point and self._wants(k):
                self.weights.set(k, v.detach().clone().float())
                loaded.append(k)
            elif torch.is_tensor(v) and v.dtype.is_floating_point:
                unexpected.append(k)
        if loaded:
            self._weights_version += 1
            self._parameters.pop("weights_not_loaded", None)
            self.are_weights_initialized = True
        return loaded, unexpected

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        own = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
        had = set(self.weights.state_dict())
        loaded, unexpected = self._ingest(own)
        if strict:
            unexpected_keys.extend(prefix + k for k in unexpected)
            missing_keys.extend(prefix + k for k in self._missing_after(had, loaded))

    def _missing_after(self, had, loaded):
        """Keys still absent after a load: against the set the model NEEDS where the subclass states it (`_required_keys`), so a first,
        partial load reports what is missing; otherwise against what it held before."""
        need = self._required_keys()
        if need is not None:
            return sorted(set(need) - set(self.weights.state_dict()))
        return sorted(had - set(loaded)) if had else ([] if loaded else ["*"])

    def _required_keys(self):
        return None

    def load_state_dict(self, state_dict, strict=False, **kw):
        had = set(self.weights.state_dict())
        loaded, unexpected = self._ingest(state_dict)
        if not loaded:
            raise RuntimeError("state_dict holds none of the tensors this model needs")
        missing = self._missing_after(had, loaded)
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict(strict=True): missing {missing}, unexpected {unexpected}")
        from torch.nn.modules.module import _IncompatibleKeys

        return _IncompatibleKeys(missing, unexpected)

    def state_dict(self, *a, **kw):
        return self.weights.state_dict(*a, **kw)

    def _has_weights(self):
        return len(self.weights._modules) + len(self.weights._buffers) > 0

    def is_initialized(self):
        return bool(self.are_weights_initialized) and self._has_weights()

    def set_initialized(self, to=True):
        # a composite's load_state_dict(strict=False) marks every child afterwards (base_model.py:132-136), also one that received no
        # tensors (an extractor-only checkpoint): such a child stays uninitialized instead of uploading an empty weight set later
        self.are_weights_initialized = bool(to) and self._has_weights()

    def _rename_keys(self, sd):
        return sd

    def _wants(self, key):
        return True

    def _variant_or_layers(self):
        return 0

    def engine(self, device):
        from . import _native

        device = torch.device(device)
        if device.type != "cuda":
            _native.Engine(device)  # raises: there is no CPU path
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device.index != idx:
            if self._engine is not None:
                self._engine.close()
            self._engine = _native.Engine(torch.device("cuda", idx))
            self._engine.set_precision(self.conf.get("precision", "f16x3"))
            self._uploaded_version = -1
        if self._uploaded_version != self._weights_version:
            if not self.is_initialized():
                raise RuntimeError(f"{type(self).__name__}: weights are not loaded (pass conf.weights=<path> or call load_state_dict)")
            self._engine.set_weights(self.which, self.weights.state_dict(), self._variant_or_layers())
            self._uploaded_version = self._weights_version
        return self._engine
