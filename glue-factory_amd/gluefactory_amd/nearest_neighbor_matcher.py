"""Nearest-neighbour matcher on the MI355X-native path.

Drop-in for `gluefactory.models.matchers.nearest_neighbor_matcher`
(gluefactory/models/matchers/nearest_neighbor_matcher.py:39-72): same conf keys (`ratio_thresh`,
`distance_thresh`, `mutual_check`, `loss`), same required inputs (`descriptors0/1`, L2-normalised), same pred dict
(`matches0/1` int64 with -1 = unmatched, `matching_scores0/1` in {0,1}, `similarity` (B,M,N), `log_assignment`
(B,M+1,N+1)).  No weights; the forward is one call into libmatchglue (`mg_nn_forward`).

Among exactly equal similarities the lowest index wins (torch.topk leaves that order unspecified); descriptors of any
width are accepted (rows are zero-padded to the GEMM step inside the library).
"""
import torch

from . import _native
from .base_model import BaseModel


class NearestNeighborMatcher(BaseModel):
    default_conf = {
        "ratio_thresh": None,
        "distance_thresh": None,
        "mutual_check": True,
        "loss": None,
        "return_log_assignment": True,  # build-specific: skip the (B,M+1,N+1) export when False
        "precision": "f16x3",  # build-specific, see lightglue.py
    }
    required_data_keys = ["descriptors0", "descriptors1"]

    def _init(self, conf):
        if conf.loss is not None:
            raise NotImplementedError("training losses are outside the inference hot path")
        self._engine = None
        self.set_initialized()

    def _forward(self, data):
        d0, d1 = data["descriptors0"], data["descriptors1"]
        dev = d0.device
        if dev.type != "cuda":
            _native.Engine(dev)  # raises: there is no CPU path
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device.index != idx:
            self._engine = _native.Engine(torch.device("cuda", idx))
            self._engine.set_precision(self.conf.get("precision", "f16x3"))
        nc = _native.NnConf(float(self.conf.ratio_thresh or 0.0), float(self.conf.distance_thresh or 0.0), int(bool(self.conf.mutual_check)))
        m0, m1, ms0, ms1, sim, la = self._engine.nn_forward(nc, d0, d1, True, bool(self.conf.return_log_assignment))
        pred = {"matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1, "similarity": sim}
        if la is not None:
            pred["log_assignment"] = la
        return pred

    def loss(self, pred, data):
        raise NotImplementedError("training losses are outside the inference hot path")


__main_model__ = NearestNeighborMatcher
