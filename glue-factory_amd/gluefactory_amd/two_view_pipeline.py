"""Two-view composite: extractor on both views -> suffix keys with 0/1 -> matcher.

Mirror of gluefactory/models/two_view_pipeline.py:21-114 for the components on the hot path
(extractor + matcher; `filter`, `solver`, `ground_truth` are CPU stages outside it and are rejected).
Reproduces the `cache` / `allow_no_extract` skip (:62-70), the key suffixing (:75-78) and the merge
order (:80-81).  When both views have the same shape the two extractor calls are fused into ONE
native launch over 2B images (result-identical: every image is processed independently).
"""
import torch

from . import get_model
from .base_model import BaseModel

_ALIASES = {
    "gluefactory_nonfree.superpoint": "gluefactory_amd.superpoint",
    "extractors.superpoint_open": "gluefactory_amd.superpoint_open",
    "gluefactory.models.extractors.superpoint_open": "gluefactory_amd.superpoint_open",
    "matchers.lightglue": "gluefactory_amd.lightglue",
    "gluefactory.models.matchers.lightglue": "gluefactory_amd.lightglue",
    "matchers.lightglue_pretrained": "gluefactory_amd.lightglue_pretrained",  # configs/eval/superpoint+lightglue-official.yaml:9
    "gluefactory.models.matchers.lightglue_pretrained": "gluefactory_amd.lightglue_pretrained",
    "gluefactory_nonfree.superglue": "gluefactory_amd.superglue",
    "matchers.nearest_neighbor_matcher": "gluefactory_amd.nearest_neighbor_matcher",
    "matchers.gluestick": "gluefactory_amd.gluestick",
    "gluefactory.models.matchers.gluestick": "gluefactory_amd.gluestick",
    "gluefactory.models.matchers.nearest_neighbor_matcher": "gluefactory_amd.nearest_neighbor_matcher",
    "lines.wireframe": "gluefactory_amd.wireframe",
    "gluefactory.models.lines.wireframe": "gluefactory_amd.wireframe",
}


class TwoViewPipeline(BaseModel):
    default_conf = {
        "extractor": {"name": None, "trainable": False},
        "matcher": {"name": None},
        "filter": {"name": None},
        "solver": {"name": None},
        "ground_truth": {"name": None},
        "allow_no_extract": False,
        "run_gt_in_forward": False,
        "fuse_views": True,  # build-specific: one extractor launch for both views when shapes agree
    }
    required_data_keys = ["view0", "view1"]
    strict_conf = False
    components = ["extractor", "matcher", "filter", "solver", "ground_truth"]

    def _init(self, conf):
        for k in ("filter", "solver", "ground_truth"):
            if conf[k].get("name"):
                raise NotImplementedError(f"'{k}' runs on the CPU after the hot path; keep it in the reference pipeline")
        if conf.extractor.name:
            self.extractor = get_model(_ALIASES.get(conf.extractor.name, conf.extractor.name))(dict(conf.extractor))
        if conf.matcher.name:
            self.matcher = get_model(_ALIASES.get(conf.matcher.name, conf.matcher.name))(dict(conf.matcher))

    def extract_view(self, data, i):
        data_i = data[f"view{i}"]
        pred_i = data_i.get("cache", {})
        skip_extract = len(pred_i) > 0 and self.conf.allow_no_extract
        if self.conf.extractor.name and not skip_extract:
            pred_i = {**pred_i, **self.extractor(data_i)}
        elif self.conf.extractor.name and not self.conf.allow_no_extract:
            pred_i = {**pred_i, **self.extractor({**data_i, **pred_i})}
        return pred_i

    def _can_fuse(self, data):
        if not (self.conf.fuse_views and self.conf.extractor.name):
            return False
        v0, v1 = data["view0"], data["view1"]
        if v0.get("cache") or v1.get("cache"):
            return False
        if v0["image"].shape != v1["image"].shape or ("image_size" in v0) != ("image_size" in v1):
            return False
        return True

    def _forward(self, data):
        if self._can_fuse(data):
            v0, v1 = data["view0"], data["view1"]
            b = v0["image"].shape[0]
            both = {"image": torch.cat([v0["image"], v1["image"]], 0)}
            if "image_size" in v0:
                both["image_size"] = torch.cat([v0["image_size"], v1["image_size"]], 0)
            p = self.extractor(both)
            pred0 = {k: v[:b] for k, v in p.items()}
            pred1 = {k: v[b:] for k, v in p.items()}
        else:
            pred0 = self.extract_view(data, "0")
            pred1 = self.extract_view(data, "1")
        pred = {
            **{k + "0": v for k, v in pred0.items()},
            **{k + "1": v for k, v in pred1.items()},
        }
        if self.conf.matcher.name:
            pred = {**pred, **self.matcher({**data, **pred})}
        return pred

    def loss(self, pred, data):
        raise NotImplementedError("training losses are outside the inference hot path")


__main_model__ = TwoViewPipeline
