"""GlueStick point + line matcher on the MI355X-native path.

Drop-in for `gluefactory.models.matchers.gluestick` (gluefactory/models/matchers/gluestick.py:18-312): same conf keys
(:19-36), same required inputs (:37-52 -- keypoints / descriptors / scores, `lines0/1` (B,n,2,2), `lines_junc_idx0/1`
(B,n,2), `line_scores0/1`, `view0/view1`), same pred dict (:229-312: `matches0/1`, `matching_scores0/1`,
`log_assignment`, and the `line_*` counterparts plus `raw_line_scores`) and the reference's state_dict key names
(`kenc`, `lenc`, `gnn.layers.{i}.update.{attn,mlp}`, `gnn.line_layers.{i}.mlp`, `final_proj`, `final_line_proj`,
`bin_score`, `line_bin_score`).  The forward is one call into libmatchglue (`mg_gs_forward`).

Built for the released shape: `input_dim == descriptor_dim == 256`, `num_line_iterations == 1`, `line_attention`
off, no `inter_supervision`.  `weights` is a local state_dict path (the reference downloads a release tarball, :57-60,
:107-123); a checkpoint with a `model` entry is unwrapped the way the reference does (:125-133).
"""
import torch

from . import _native
from .base_model import NativeModel


class GlueStick(NativeModel):
    default_conf = {
        "name": "gluestick",
        "input_dim": 256,
        "descriptor_dim": 256,
        "weights": None,
        "version": "v0.1_arxiv",
        "keypoint_encoder": [32, 64, 128, 256],
        "GNN_layers": ["self", "cross"] * 9,
        "num_line_iterations": 1,
        "line_attention": False,
        "filter_threshold": 0.2,
        "checkpointed": False,
        "skip_init": False,
        "inter_supervision": None,
        "loss": {"nll_weight": 1.0, "nll_balancing": 0.5, "inter_supervision": [0.3, 0.6]},
        "return_log_assignment": True,  # build-specific: skip the (B,M+1,N+1) export when False
        "precision": "f16x3",  # build-specific, see lightglue.py
    }
    required_data_keys = [
        "view0", "view1", "keypoints0", "keypoints1", "descriptors0", "descriptors1", "keypoint_scores0", "keypoint_scores1",
        "lines0", "lines1", "lines_junc_idx0", "lines_junc_idx1", "line_scores0", "line_scores1",
    ]
    which = 3

    def _init(self, conf):
        if conf.input_dim != 256 or conf.descriptor_dim != 256 or list(conf.keypoint_encoder) != [32, 64, 128, 256]:
            raise NotImplementedError("built for the released GlueStick shape: 256-d descriptors, encoder [32,64,128,256]")
        layers = list(conf.GNN_layers)
        if any(t != ("self", "cross")[i % 2] for i, t in enumerate(layers)) or not layers:
            raise NotImplementedError("GNN_layers must alternate self/cross starting with self (the released model does)")
        if conf.num_line_iterations != 1 or conf.line_attention or conf.inter_supervision or conf.skip_init:
            raise NotImplementedError("num_line_iterations != 1, line_attention, inter_supervision and skip_init are not used by the released model")
        self._init_weights(conf)

    def _rename_keys(self, sd):
        if "model" in sd:  # gluestick.py:125-133
            sd = {k.replace("matcher.", ""): v for k, v in sd["model"].items() if "matcher." in k}
            sd = {k.replace("module.", ""): v for k, v in sd.items()}
        return sd

    def _variant_or_layers(self):
        return len(self.conf.GNN_layers)

    @staticmethod
    def _size(view, like):
        if "image_size" in view:  # gluestick.py:139-148
            return view["image_size"]
        h, w = view["image"].shape[-2:]
        return like.new_tensor([[w, h]], dtype=torch.float32)

    def _forward(self, data):
        k0, k1 = data["keypoints0"], data["keypoints1"]
        dev = k0.device
        B, M, N = k0.shape[0], k0.shape[1], k1.shape[1]
        n0, n1 = data["lines0"].shape[1], data["lines1"].shape[1]

        def empty_side(m, n, prefix):  # gluestick.py:158-191 / :291-304
            return {
                prefix + "log_assignment": torch.zeros(B, m, n, dtype=torch.float, device=dev),
                prefix + "matches0": torch.full((B, m), -1, device=dev, dtype=torch.int64),
                prefix + "matches1": torch.full((B, n), -1, device=dev, dtype=torch.int64),
                prefix + "matching_scores0": torch.zeros((B, m), device=dev, dtype=torch.float32),
                prefix + "matching_scores1": torch.zeros((B, n), device=dev, dtype=torch.float32),
            }

        if M == 0 or N == 0:
            pred = {**empty_side(M, N, ""), **empty_side(n0, n1, "line_")}
            pred["line_matching_scores1"] = torch.zeros((B, N), device=dev, dtype=torch.float32)  # sic, gluestick.py:188-190
            return pred
        eng = self.engine(dev)
        nc = _native.GsConf(len(self.conf.GNN_layers), float(self.conf.filter_threshold))
        (m0, m1, ms0, ms1, la), (lm0, lm1, lms0, lms1, lla, raw) = eng.gs_forward(
            nc, k0, k1, data["descriptors0"], data["descriptors1"], data["keypoint_scores0"], data["keypoint_scores1"],
            self._size(data["view0"], k0), self._size(data["view1"], k1), data["lines0"], data["lines1"],
            data["lines_junc_idx0"], data["lines_junc_idx1"], data["line_scores0"], data["line_scores1"],
            want_log_assignment=bool(self.conf.return_log_assignment),
        )
        pred = {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1}
        if la is not None:
            pred["log_assignment"] = la
        if n0 > 0 and n1 > 0:
            pred.update({"line_log_assignment": lla, "line_matches0": lm0, "line_matches1": lm1, "line_matching_scores0": lms0,
                         "line_matching_scores1": lms1, "raw_line_scores": raw})
        else:  # gluestick.py:291-312
            pred.update(empty_side(n0, n1, "line_"))
            pred["raw_line_scores"] = torch.zeros(B, n0, n1, dtype=torch.float, device=dev)
        return pred

    def loss(self, pred, data):
        raise NotImplementedError("training losses are outside the inference hot path")


__main_model__ = GlueStick
