"""SuperPoint extractor (MagicLeap weight layout) on the MI355X-native path.

Drop-in for `gluefactory_nonfree.superpoint` (gluefactory_nonfree/superpoint.py:152-356): same conf
keys (:153-169), same input dict (`image` (B,C,H,W) in [0,1], optional `image_size` (B,2) (w,h)),
same pred dict (`keypoints` (x,y)+0.5, `keypoint_scores`, `descriptors` (B,K,256),
`dense_descriptors` when `dense_outputs`), same state_dict key names (`conv1a.weight` ... `convDb.bias`).
The whole forward is one call into libmatchglue (`mg_sp_forward`).

Deliberate differences (documented in DESIGN.md):
  * weights are never downloaded -- `conf.weights` is a local state_dict path (or `load_state_dict`);
  * the output holds exactly `max_num_keypoints` rows per image; images with fewer detections are
    zero-padded (the reference either fails to stack or pads with random points, :304-317).  For
    B == 1 the rows are trimmed to the true count, as the reference returns them;
  * `max_num_keypoints` <= 0 means "all", capped at 4096; training-only options are rejected.
"""
import torch

from . import _native
from .base_model import NativeModel


class SuperPoint(NativeModel):
    default_conf = {
        "has_detector": True,
        "has_descriptor": True,
        "descriptor_dim": 256,
        "sparse_outputs": True,
        "dense_outputs": False,
        "nms_radius": 4,
        "refinement_radius": 0,
        "detection_threshold": 0.005,
        "max_num_keypoints": -1,
        "max_num_keypoints_val": None,
        "force_num_keypoints": False,
        "randomize_keypoints_training": False,
        "remove_borders": 4,
        "legacy_sampling": True,
        "weights": None,  # local path of a state_dict (this build never downloads)
        "precision": "f16x3",  # build-specific: "f32" = fp32-input MFMA, "f16x3" = split-fp16 on the f16 MFMA
    }
    required_data_keys = ["image"]
    which = 0
    variant = _native.MG_SP_MAGICLEAP
    _prefixes = ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb")

    def _init(self, conf):
        if not (conf.has_detector and conf.has_descriptor and conf.sparse_outputs):
            raise NotImplementedError("only the sparse detector+descriptor path is built (superpoint.py:238-239)")
        if conf.refinement_radius:
            raise NotImplementedError("soft-argmax refinement is off in every shipped eval config and is not built")
        if conf.descriptor_dim != 256:
            raise NotImplementedError("descriptor_dim must be 256")
        self._init_weights(conf)

    def _wants(self, key):
        return key.split(".")[0] in self._prefixes

    def _required_keys(self):
        if self.variant == _native.MG_SP_MAGICLEAP:  # superpoint.py:179-196
            return [f"{n}.{p}" for n in self._prefixes for p in ("weight", "bias")]
        return None

    def _variant_or_layers(self):
        return self.variant

    def _native_conf(self):
        conf = self.conf
        k = conf.max_num_keypoints
        if not self.training and conf.max_num_keypoints_val is not None:
            k = conf.max_num_keypoints_val
        if k is None or k <= 0:
            k = 4096
        sampling = _native.MG_SAMPLE_LEGACY if conf.get("legacy_sampling", True) else _native.MG_SAMPLE_FIXED
        return _native.SpConf(self.variant, int(conf.nms_radius), int(conf.remove_borders or 0), int(k), sampling,
                              float(conf.detection_threshold))

    def _forward(self, data):
        if self.training and self.conf.randomize_keypoints_training:
            raise NotImplementedError("randomized keypoint sampling is a training-only option")
        image = data["image"]
        eng = self.engine(image.device)
        nc = self._native_conf()
        kp, sc, ds, nv, heat, dense = eng.sp_forward(
            nc, image, data.get("image_size"), want_heatmap=self._wants_heatmap(), want_dense=bool(self.conf.dense_outputs)
        )
        # `force_num_keypoints` keeps K rows per image whatever was detected (the reference pads to K, :304-317), so
        # nothing depends on the counts and the forward stays free of host synchronisation
        if not self.conf.force_num_keypoints and not self.conf.get("sync_free", False):
            counts = nv.cpu()
            n_min = int(counts.min())
            if n_min < nc.max_num_keypoints:
                if image.shape[0] > 1 and int(counts.max()) != n_min:
                    raise RuntimeError(
                        f"images of one batch produced different keypoint counts {counts.tolist()} "
                        "(the reference cannot stack them either, superpoint.py:319-320); set force_num_keypoints"
                    )
                kp, sc, ds = kp[:, :n_min], sc[:, :n_min], ds[:, :n_min]
        pred = {"keypoints": kp, "keypoint_scores": sc, "descriptors": ds}
        if self.conf.dense_outputs:
            pred["dense_descriptors"] = dense
        if heat is not None:
            pred["heatmap"] = heat
        return pred

    def _wants_heatmap(self):
        return False

    def loss(self, pred, data):
        raise NotImplementedError

    def metrics(self, pred, data):
        raise NotImplementedError


__main_model__ = SuperPoint
