"""MI355X-native SuperPoint + LightGlue forward path behind glue-factory's model plug-in API.

Module paths usable as `model.extractor.name` / `model.matcher.name` in a glue-factory YAML
(resolved by gluefactory/models/__init__.py:7-30 through `__main_model__`):

    gluefactory_amd.superpoint        MagicLeap-layout SuperPoint   (gluefactory_nonfree/superpoint.py)
    gluefactory_amd.superpoint_open   open SuperPoint               (models/extractors/superpoint_open.py)
    gluefactory_amd.lightglue         LightGlue matcher             (models/matchers/lightglue.py)
    gluefactory_amd.lightglue_pretrained  the same behind the `lightglue` package's conf (models/matchers/lightglue_pretrained.py)
    gluefactory_amd.superglue         SuperGlue matcher             (gluefactory_nonfree/superglue.py)
    gluefactory_amd.gluestick         GlueStick point+line matcher  (models/matchers/gluestick.py)
    gluefactory_amd.wireframe         points + segments -> wireframe (models/lines/wireframe.py)
    gluefactory_amd.nearest_neighbor_matcher  mutual NN matcher     (models/matchers/nearest_neighbor_matcher.py)
    gluefactory_amd.two_view_pipeline extractor x2 -> matcher       (models/two_view_pipeline.py)

All arithmetic runs in hand-written HIP kernels for gfx950 (csrc/, libmatchglue.so).
"""
import importlib

from .base_model import BaseModel  # noqa: F401


def get_model(name):
    """Resolve a model class the way gluefactory.models.get_model does (models/__init__.py:7-30)."""
    for path in (name, f"{__name__}.{name}", f"{__name__}.{name.split('.')[-1]}"):
        try:
            mod = importlib.import_module(path)
        except ModuleNotFoundError:
            continue
        if hasattr(mod, "__main_model__"):
            return mod.__main_model__
    raise RuntimeError(f"Model {name} not found")
