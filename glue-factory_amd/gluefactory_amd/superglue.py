"""SuperGlue matcher on the MI355X-native path.

Drop-in for `gluefactory_nonfree.superglue` (gluefactory_nonfree/superglue.py:208-307): same conf keys
(:209-220), same required inputs (:221-230 -- `view0/view1`, `keypoints0/1`, `descriptors0/1`,
`keypoint_scores0/1`), same pred dict (:300-307: `sinkhorn_cost` (B,M,N), `log_assignment` (B,M+1,N+1),
`matches0/1` int64 with -1 = unmatched, `matching_scores0/1`) and the reference's state_dict key names
(`kenc.encoder.{0..12}`, `gnn.layers.{i}.attn.{merge,proj.{0,1,2}}`, `gnn.layers.{i}.mlp.{0,1,3}`,
`final_proj`, `bin_score`).  The forward is one call into libmatchglue (`mg_sg_forward`).

`weights` takes a local state_dict path: the reference downloads `superglue_{indoor,outdoor}.pth`
(:232-240), which this build cannot (no network) -- point it at the same file on disk.
"""
import torch

from . import _native
from .base_model import NativeModel


class SuperGlue(NativeModel):
    default_conf = {
        "name": "superglue",
        "descriptor_dim": 256,
        "weights": None,
        "keypoint_encoder": [32, 64, 128, 256],
        "GNN_layers": ["self", "cross"] * 9,
        "num_sinkhorn_iterations": 50,
        "filter_threshold": 0.2,
        "use_scores": True,
        "loss": {"nll_balancing": 0.5},
        "return_log_assignment": True,  # build-specific: skip the (B,M+1,N+1) export when False
        "precision": "f16x3",  # build-specific, see lightglue.py
    }
    required_data_keys = [
        "view0", "view1", "keypoints0", "keypoints1", "descriptors0", "descriptors1", "keypoint_scores0", "keypoint_scores1",
    ]
    which = 2

    def _init(self, conf):
        if conf.descriptor_dim != 256 or list(conf.keypoint_encoder) != [32, 64, 128, 256]:
            raise NotImplementedError("built for the released SuperGlue shape: 256-d descriptors, encoder [32,64,128,256]")
        layers = list(conf.GNN_layers)
        if any(t != ("self", "cross")[i % 2] for i, t in enumerate(layers)) or not layers:
            raise NotImplementedError("GNN_layers must alternate self/cross starting with self (every released model does)")
        if not conf.use_scores:
            raise NotImplementedError("use_scores=False is never used by the reference's configs")
        if conf.weights in ("indoor", "outdoor"):
            raise FileNotFoundError(
                f"weights='{conf.weights}' is a download in the reference (superglue.py:232-240); this build has no "
                "network -- pass the local path of superglue_" + str(conf.weights) + ".pth"
            )
        self._init_weights(conf)

    def _variant_or_layers(self):
        return len(self.conf.GNN_layers)

    @staticmethod
    def _size(view, B, like):
        # superglue.py:32-47: image_size if given, else the image tensor's (w, h)
        size = view.get("image_size")
        if size is None:
            h, w = view["image"].shape[-2:]
            size = like.new_tensor([[w, h]], dtype=torch.float32)
        return size

    def _forward(self, data):
        k0, k1 = data["keypoints0"], data["keypoints1"]
        if k0.shape[1] == 0 or k1.shape[1] == 0:  # superglue.py:257-265
            s0, s1 = k0.shape[:-1], k1.shape[:-1]
            return {
                "matches0": k0.new_full(s0, -1, dtype=torch.int),
                "matches1": k1.new_full(s1, -1, dtype=torch.int),
                "matching_scores0": k0.new_zeros(s0),
                "matching_scores1": k1.new_zeros(s1),
            }
        B = k0.shape[0]
        size0 = self._size(data["view0"], B, k0)
        size1 = self._size(data["view1"], B, k1)
        eng = self.engine(k0.device)
        nc = _native.SgConf(len(self.conf.GNN_layers), int(self.conf.num_sinkhorn_iterations), float(self.conf.filter_threshold))
        m0, m1, ms0, ms1, la, cost = eng.sg_forward(
            nc, k0, k1, data["descriptors0"], data["descriptors1"], data["keypoint_scores0"], data["keypoint_scores1"], size0, size1,
            want_log_assignment=bool(self.conf.return_log_assignment), want_cost=True,
        )
        pred = {"sinkhorn_cost": cost, "matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1}
        if la is not None:
            pred["log_assignment"] = la
        return pred

    def loss(self, pred, data):
        raise NotImplementedError("training losses are outside the inference hot path")


__main_model__ = SuperGlue
