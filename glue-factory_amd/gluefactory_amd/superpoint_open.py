"""Open (MIT) SuperPoint on the MI355X-native path.

Drop-in for `gluefactory.models.extractors.superpoint_open`
(gluefactory/models/extractors/superpoint_open.py:78-220): Conv -> ReLU -> BatchNorm(eps=1e-3) blocks
(:59-75; the BatchNorm runs as a per-channel affine in the conv epilogue), border removal relative to
the tensor (not `image_size`, :147-152), `(kp+0.5)/(8w)` descriptor sampling (:20-31), and the extra
`heatmap` output (:208-213).  State_dict keys: `backbone.{i}.{j}.{conv,bn}.*`, `detector.*`, `descriptor.*`.
"""
from . import _native
from .superpoint import SuperPoint as _SuperPointBase


class SuperPoint(_SuperPointBase):
    default_conf = {
        "descriptor_dim": 256,
        "nms_radius": 4,
        "max_num_keypoints": None,
        "force_num_keypoints": False,
        "detection_threshold": 0.005,
        "remove_borders": 4,
        "channels": [64, 64, 128, 128, 256],
        "dense_outputs": None,
        "weights": None,
        "precision": "f16x3",
    }
    variant = _native.MG_SP_OPEN
    _prefixes = ("backbone", "detector", "descriptor")

    def _init(self, conf):
        if list(conf.channels) != [64, 64, 128, 128, 256]:
            raise NotImplementedError("only the published channel layout [64,64,128,128,256] is built")
        super()._init(conf)

    def _native_conf(self):
        nc = super()._native_conf()
        nc.sampling = _native.MG_SAMPLE_OPEN
        return nc

    def _wants_heatmap(self):
        return True


__main_model__ = SuperPoint
