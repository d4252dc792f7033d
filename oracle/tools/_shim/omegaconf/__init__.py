"""Minimal attribute-dict stand-in for the `omegaconf` package.

TEST TOOLING ONLY.  The reference (/root/reference) imports omegaconf at module
scope for *configuration merging*; the package is not installed in this image
and there is no network.  This file implements just the handful of entry points
the reference's model layer touches (create / merge / to_container / set_struct
/ set_readonly / read_write / open_dict) so that `oracle/tools/make_goldens.py`
can import the reference's SuperPoint / LightGlue / TwoViewPipeline modules and
run them on CPU to produce golden vectors.  No arithmetic goes through here.
It is never imported by the product package, by tests, or on the GPU box.
"""
from contextlib import contextmanager


class MissingMandatoryValue(Exception):
    pass


class DictConfig(dict):
    """dict with attribute access; nested dicts are wrapped on the way in."""

    def __init__(self, content=None):
        super().__init__()
        for k, v in (content or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def pop(self, k, *a):
        return super().pop(k, *a)


class ListConfig(list):
    pass


def _wrap(v):
    if isinstance(v, DictConfig):
        return v
    if isinstance(v, dict):
        return DictConfig(v)
    if isinstance(v, (list, tuple)) and not isinstance(v, ListConfig):
        return ListConfig(_wrap(x) for x in v)
    return v


def _unwrap(v):
    if isinstance(v, dict):
        return {k: _unwrap(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_unwrap(x) for x in v]
    return v


def _merge_into(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_into(dst[k], v)
        else:
            dst[k] = _wrap(_unwrap(v)) if isinstance(v, (dict, list)) else v


class OmegaConf:
    @staticmethod
    def create(content=None):
        return DictConfig(_unwrap(content) if content else {})

    @staticmethod
    def merge(*confs):
        out = DictConfig()
        for c in confs:
            if c is None:
                continue
            _merge_into(out, c if isinstance(c, dict) else dict(c))
        return out

    @staticmethod
    def to_container(conf, resolve=False, **kw):
        return _unwrap(conf)

    @staticmethod
    def set_struct(conf, flag):
        return None

    @staticmethod
    def set_readonly(conf, flag):
        return None

    @staticmethod
    def load(path):
        import yaml

        with open(path) as f:
            return DictConfig(yaml.safe_load(f))

    @staticmethod
    def from_cli(args=None):
        return DictConfig()


@contextmanager
def read_write(conf):
    yield conf


@contextmanager
def open_dict(conf):
    yield conf
