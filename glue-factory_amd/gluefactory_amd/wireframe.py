"""Wireframe extractor (points + line segments -> junction graph) on the MI355X-native path.

Drop-in for `gluefactory.models.lines.wireframe.WireframeExtractor` (gluefactory/models/lines/wireframe.py:130-306):
same conf keys (:131-151), same pred dict (`keypoints` = junctions then keypoints, `keypoint_scores`, `descriptors`,
`lines`, `orig_lines`, `lines_junc_idx`, `line_scores`, `pl_associativity`, `num_junctions`).

What runs where:
  * keypoints, dense descriptors: the point extractor of this package (SuperPoint, one native launch);
  * keypoint / line-endpoint distance filter + compaction (:184-216): `mg_wf_filter` (csrc/wf_misc.hip);
  * junction descriptors, `sample_descriptors_corner_conv` (:8-20,118,283-287): `mg_sp_sample`, MG_SAMPLE_CORNER;
  * line segments: a CPU dependency in the reference too (pytlsd LSD, models/lines/lsd.py) -- either the module named
    by `line_extractor.name` (any module returning `lines` (B,n,2,2) and `line_scores` (B,n), e.g. the reference's
    `lines.lsd`), or precomputed segments passed in the data dict (`lines`, `line_scores`) when no name is given;
  * DBSCAN merge of close-by endpoints (:58-63, sklearn) and the per-cluster means: CPU, as in the reference.

Deliberate difference: with `force_num_keypoints` / `force_num_lines` the reference pads with RANDOM points; this build
(like its SuperPoint) keeps fixed-size rows with zero padding behind the valid ones.  Non-padded inputs are batch 1, as
in the reference (:172-176).
"""
import numpy as np
import torch

from . import _native, get_model
from .base_model import BaseModel


def cluster_endpoints(endpoints, line_scores, radius):
    """wireframe.py:57-92 for one image, on the CPU: DBSCAN(eps=radius, min_samples=1) labels of the endpoints, the mean
    position and mean score of every cluster (fp32 sums in endpoint order, as scatter_reduce("mean") adds them)."""
    from sklearn.cluster import DBSCAN

    pts = endpoints.detach().cpu().float().numpy()
    labels = DBSCAN(eps=radius, min_samples=1).fit(pts).labels_.astype(np.int64)
    n = int(labels.max()) + 1 if len(labels) else 0
    sums = np.zeros((n, 2), np.float32)
    ssum = np.zeros(n, np.float32)
    cnt = np.zeros(n, np.float32)
    sc = np.repeat(line_scores.detach().cpu().float().numpy(), 2)
    for i, c in enumerate(labels):  # sequential fp32 adds in index order
        sums[c] += pts[i]
        ssum[c] += sc[i]
        cnt[c] += 1.0
    return labels, sums / cnt[:, None], ssum / cnt


class WireframeExtractor(BaseModel):
    default_conf = {
        "point_extractor": {
            "name": "gluefactory_amd.superpoint",
            "trainable": False,
            "dense_outputs": True,
            "max_num_keypoints": None,
            "force_num_keypoints": False,
        },
        "line_extractor": {
            "name": None,  # None: precomputed `lines` / `line_scores` arrive in the data dict
            "trainable": False,
            "max_num_lines": None,
            "force_num_lines": False,
            "min_length": 15,
        },
        "wireframe_params": {
            "merge_points": True,
            "merge_line_endpoints": True,
            "nms_radius": 3,
        },
    }
    required_data_keys = ["image"]

    def _init(self, conf):
        from .two_view_pipeline import _ALIASES

        pe = dict(conf.point_extractor)
        if not pe.get("dense_outputs", True):
            raise ValueError("the point extractor must return dense descriptors (wireframe.py:178-180)")
        pe["dense_outputs"] = True
        self.point_extractor = get_model(_ALIASES.get(pe["name"], pe["name"]))(pe)
        self.line_extractor = None
        if conf.line_extractor.name:
            self.line_extractor = get_model(conf.line_extractor.name)(dict(conf.line_extractor))
        if conf.point_extractor.force_num_keypoints or conf.line_extractor.force_num_lines:
            raise NotImplementedError("padded batches (random padding points, wireframe.py:95-117,195-210) are not built: run pairs one at a time")

    def _lines(self, data):
        if self.line_extractor is not None:
            p = self.line_extractor(data)
            return p["lines"], p["line_scores"]
        assert "lines" in data and "line_scores" in data, "Missing key lines / line_scores in data (no line_extractor configured)"
        return data["lines"], data["line_scores"]

    def _forward(self, data):
        image = data["image"]
        b_size, _, h, w = image.shape
        dev = image.device
        assert b_size == 1, "Only batch size of 1 accepted for non padded inputs"
        lines, line_scores = self._lines(data)
        lines = lines.to(dev).float()
        line_scores = line_scores.to(dev).float()
        pred = {"lines": lines, "line_scores": line_scores}
        if line_scores.shape[-1] != 0:  # wireframe.py:180-181
            pred["line_scores"] = line_scores / (line_scores.max(dim=1)[0][:, None] + 1e-8)
        pp = self.point_extractor(data)
        assert "dense_descriptors" in pp, "The KP extractor should return dense descriptors"
        dense = pp["dense_descriptors"]
        if h // dense.shape[2] != 8:
            raise RuntimeError("the descriptor map must be at 1/8 of the image resolution")
        eng = self.point_extractor.engine(dev)
        kp, sc, ds = pp["keypoints"], pp["keypoint_scores"], pp["descriptors"]
        rad = float(self.conf.wireframe_params.nms_radius)
        n_lines = lines.shape[1]

        if self.conf.wireframe_params.merge_points and n_lines > 0 and kp.shape[1] > 0:
            okp, osc, ods, cnt = eng.wf_filter(kp, sc, ds, lines.reshape(b_size, -1, 2), rad)
            n = int(cnt[0])  # batch 1: the reference removes the rows (:211-216)
            kp, sc, ds = okp[:, :n], osc[:, :n], ods[:, :n]

        orig_lines = lines.clone()
        corner = _native.SpConf(0, 0, 0, 1, _native.MG_SAMPLE_CORNER, 0.0)
        if self.conf.wireframe_params.merge_line_endpoints and n_lines > 0:
            labels, junc, jscore = cluster_endpoints(lines[0].reshape(-1, 2), pred["line_scores"][0], rad)
            clusters = torch.from_numpy(labels).to(dev)
            junctions = torch.from_numpy(junc).to(dev)[None]
            junc_scores = torch.from_numpy(jscore).to(dev)[None]
            n_junc = junctions.shape[1]
            pred["lines"] = junctions[0][clusters].reshape(1, -1, 2, 2)
            lines_junc_idx = clusters.reshape(1, -1, 2)
            junc_descs = eng.sp_sample(corner, junctions, dense)
            all_points = torch.cat([junctions, kp], 1)
            all_scores = torch.cat([junc_scores, sc], 1)
            all_descs = torch.cat([junc_descs, ds], 1)
            assoc = torch.eye(all_points.shape[1], dtype=torch.bool, device=dev)
            pairs = lines_junc_idx[0]
            assoc[pairs[:, 0], pairs[:, 1]] = True
            assoc[pairs[:, 1], pairs[:, 0]] = True
            pl_associativity = assoc[None]
            n_true = [n_junc]
        else:  # lines are independent (:269-296)
            ends = lines.reshape(b_size, -1, 2)
            all_points = torch.cat([ends, kp], 1)
            all_scores = torch.cat([torch.repeat_interleave(pred["line_scores"], 2, dim=1), sc], 1)
            line_descs = eng.sp_sample(corner, ends, dense) if n_lines > 0 else ds[:, :0]
            all_descs = torch.cat([line_descs, ds], 1)
            pl_associativity = torch.eye(all_points.shape[1], dtype=torch.bool, device=dev)[None]
            n_true = [n_lines * 2]
            lines_junc_idx = torch.arange(n_lines * 2, device=dev).reshape(1, -1, 2)

        pred["keypoints"] = all_points
        pred["keypoint_scores"] = all_scores
        pred["descriptors"] = all_descs
        pred["pl_associativity"] = pl_associativity
        pred["num_junctions"] = torch.tensor(n_true)
        pred["orig_lines"] = orig_lines
        pred["lines_junc_idx"] = lines_junc_idx
        return pred

    def loss(self, pred, data):
        raise NotImplementedError

    def metrics(self, _pred, _data):
        return {}


__main_model__ = WireframeExtractor
