"""LightGlue matcher on the MI355X-native path.

Drop-in for `gluefactory.models.matchers.lightglue` (gluefactory/models/matchers/lightglue.py:306-609):
same conf keys (:307-327), same required inputs (`keypoints0/1`, `descriptors0/1`, optional
`view0/view1: {image_size}`), same pred dict (:513-523: `matches0/1` int64 with -1 = unmatched,
`matching_scores0/1`, `ref_descriptors0/1` (B,1,K,256), `log_assignment` (B,M+1,N+1), `prune0/1`),
same state_dict keys (incl. the legacy `self_attn.{i}` -> `transformers.{i}.self_attn` rename, :378-385).
The forward is one call into libmatchglue (`mg_lg_forward`).

Adaptive depth / width (:468-508): `width_confidence` > 0 (point pruning) runs in the reference and is reproduced
(`mg_lg_forward_adaptive`, pinned by tests/golden/lightglue_adaptive.npz); `depth_confidence` > 0 (early stopping)
raises AttributeError in the reference fork (undefined `self.confidence_thresholds`, :556) -- it is built with the
evident intent (= upstream LightGlue: thresholds from `confidence_threshold()`).  Both need batch 1, as in the reference.
"""
import torch

from . import _native
from .base_model import NativeModel


class LightGlue(NativeModel):
    default_conf = {
        "name": "lightglue",
        "input_dim": 256,
        "add_scale_ori": False,
        "descriptor_dim": 256,
        "n_layers": 9,
        "num_heads": 4,
        "flash": False,
        "mp": False,
        "depth_confidence": -1,
        "width_confidence": -1,
        "filter_threshold": 0.0,
        "checkpointed": False,
        "weights": None,
        "weights_from_version": "v0.1_arxiv",
        "loss": {"gamma": 1.0, "fn": "nll", "nll_balancing": 0.5},
        "return_log_assignment": True,  # build-specific: skip the (B,M+1,N+1) export when False
        "precision": "f16x3",  # build-specific: "f32" = fp32-input MFMA, "f16x3" = split-fp16 on the f16 MFMA
    }
    required_data_keys = ["keypoints0", "keypoints1", "descriptors0", "descriptors1"]
    which = 1

    def _init(self, conf):
        if conf.input_dim != 256 or conf.descriptor_dim != 256 or conf.num_heads != 4:
            raise NotImplementedError("built for 256-d descriptors and 4 heads (SuperPoint + LightGlue)")
        if conf.add_scale_ori:
            raise NotImplementedError("add_scale_ori (SIFT/DISK inputs) is outside the SuperPoint+LightGlue path")
        self._init_weights(conf)

    def _rename_keys(self, sd):
        for i in range(self.conf.n_layers):
            sd = {k.replace(f"self_attn.{i}", f"transformers.{i}.self_attn") if not k.startswith("transformers.") else k: v for k, v in sd.items()}
            sd = {k.replace(f"cross_attn.{i}", f"transformers.{i}.cross_attn") if not k.startswith("transformers.") else k: v for k, v in sd.items()}
        return sd

    def _variant_or_layers(self):
        return int(self.conf.n_layers)

    def _required_keys(self):
        """state_dict keys of lightglue.py:340-357 that the plain forward reads (the `token_confidence.{i}` heads only with adaptive depth)."""
        keys = ["posenc.Wr.weight"]
        for i in range(int(self.conf.n_layers)):
            for blk, lins in (("self_attn", ("Wqkv", "out_proj")), ("cross_attn", ("to_qk", "to_v", "to_out"))):
                keys += [f"transformers.{i}.{blk}.{n}.{p}" for n in lins + ("ffn.0", "ffn.1", "ffn.3") for p in ("weight", "bias")]
            keys += [f"log_assignment.{i}.{n}.{p}" for n in ("matchability", "final_proj") for p in ("weight", "bias")]
        return keys

    def _forward(self, data):
        k0, k1 = data["keypoints0"], data["keypoints1"]
        size0 = size1 = None
        if "view0" in data and "view1" in data:
            size0 = data["view0"].get("image_size")
            size1 = data["view1"].get("image_size")
        eng = self.engine(k0.device)
        nc = _native.LgConf(int(self.conf.n_layers), int(self.conf.num_heads), float(self.conf.filter_threshold))
        if not self.training and (self.conf.depth_confidence > 0 or self.conf.width_confidence > 0):
            assert k0.shape[0] == 1  # lightglue.py:476,:481
            m0, m1, ms0, ms1, la, r0, r1, p0, p1, layers = eng.lg_forward_adaptive(
                nc, float(self.conf.depth_confidence), float(self.conf.width_confidence), k0, k1, data["descriptors0"], data["descriptors1"],
                size0, size1, want_log_assignment=True)
            pruning = self.conf.width_confidence > 0
            pred = {
                "matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
                "ref_descriptors0": r0[:, None], "ref_descriptors1": r1[:, None], "log_assignment": la,
                # :463-466 int64 counters when pruning, :509-511 n_layers (float) otherwise
                "prune0": p0.to(torch.int64) if pruning else torch.full_like(ms0, float(self.conf.n_layers)),
                "prune1": p1.to(torch.int64) if pruning else torch.full_like(ms1, float(self.conf.n_layers)),
            }
            self.last_layers_run = layers
            return pred
        m0, m1, ms0, ms1, la, r0, r1 = eng.lg_forward(
            nc, k0, k1, data["descriptors0"], data["descriptors1"], size0, size1,
            want_log_assignment=bool(self.conf.return_log_assignment), want_ref=True,
        )
        L = float(self.conf.n_layers)
        pred = {
            "matches0": m0,
            "matches1": m1,
            "matching_scores0": ms0,
            "matching_scores1": ms1,
            "ref_descriptors0": r0[:, None],
            "ref_descriptors1": r1[:, None],
            "prune0": torch.full_like(ms0, L),
            "prune1": torch.full_like(ms1, L),
        }
        if la is not None:
            pred["log_assignment"] = la
        return pred

    def loss(self, pred, data):
        raise NotImplementedError("training losses are outside the inference hot path")


__main_model__ = LightGlue
