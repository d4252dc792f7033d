"""`matchers.lightglue_pretrained` on the MI355X-native path -- the matcher name of the north-star config
(gluefactory/configs/eval/superpoint+lightglue-official.yaml:9-13: `features: superpoint`, `depth_confidence: -1`,
`width_confidence: -1`, `filter_threshold: 0.1`).

The reference module (gluefactory/models/matchers/lightglue_pretrained.py:7-36) is a shell around the third-party `lightglue`
package (pyproject.toml:34, unpinned, source not in the reference tree): `LightGlue_(features, **conf)` downloads the released
checkpoint of that feature type and runs the same 9-layer architecture the in-tree `matchers/lightglue.py` defines (which loads the
very same checkpoint names, lightglue.py:331,369-376).  Here the shell sits on the native LightGlue: same conf keys (`features` plus the
package's own: its defaults differ from the in-tree module's -- `filter_threshold` 0.1, adaptive depth / width switched ON at 0.95 /
0.99, which the official config switches off), same required inputs (:9-16: `view0/1`, `keypoints0/1`, `descriptors0/1`), the package's
pred keys (`matches0/1`, `matching_scores0/1`, `prune0/1`, `stop`; plus `log_assignment` on request).  There is no network: `weights`
must be the local path of the released `superpoint_lightglue` state_dict (or call `load_state_dict`).  PARITY UNPINNED at the numeric
level for this entry (the package is absent and the reference's only test of it needs the download); the arithmetic is the
in-tree model's, which IS pinned.
"""
import torch

from .lightglue import LightGlue as NativeLightGlue

# input dimensions the package knows per feature type (its `features` table); only SuperPoint's 256-d descriptors are on this path
_FEATURE_DIMS = {"superpoint": 256, "disk": 128, "aliked": 128, "sift": 128, "doghardnet": 128}


class LightGlue(NativeLightGlue):
    default_conf = {
        **NativeLightGlue.default_conf,
        "features": "superpoint",
        "depth_confidence": 0.95,
        "width_confidence": 0.99,
        "filter_threshold": 0.1,
        "flash": True,  # accepted and ignored: the attention kernels are this build's own
        "return_log_assignment": False,  # the package does not return the matrix; skipping it saves the (B,M+1,N+1) write
    }
    required_data_keys = ["view0", "keypoints0", "descriptors0", "view1", "keypoints1", "descriptors1"]

    def _init(self, conf):
        if conf.features not in _FEATURE_DIMS:
            raise ValueError(f"unknown feature type '{conf.features}' (the lightglue package knows {sorted(_FEATURE_DIMS)})")
        if conf.features != "superpoint":
            raise NotImplementedError(f"features: {conf.features} -- only the SuperPoint + LightGlue path is built (256-d descriptors, no scale / orientation inputs)")
        super()._init(conf)

    def _forward(self, data):
        # the package normalises keypoints by `image_size` when the view carries it and by the image tensor's shape otherwise
        views = {}
        for v in ("view0", "view1"):
            view = data[v]
            if "image_size" not in view and "image" in view:
                h, w = view["image"].shape[-2:]
                view = {**view, "image_size": torch.tensor([[float(w), float(h)]], device=data["keypoints0"].device).expand(data["keypoints0"].shape[0], -1)}
            views[v] = view
        pred = super()._forward({**data, **views})
        out = {k: pred[k] for k in ("matches0", "matches1", "matching_scores0", "matching_scores1", "prune0", "prune1")}
        out["stop"] = int(getattr(self, "last_layers_run", self.conf.n_layers))
        if "log_assignment" in pred and self.conf.return_log_assignment:
            out["log_assignment"] = pred["log_assignment"]
        return out


__main_model__ = LightGlue
