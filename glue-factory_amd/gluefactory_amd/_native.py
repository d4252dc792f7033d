"""ctypes binding of libmatchglue.so (include/matchglue.h).

This is the only place the product touches native code.  There is NO fallback: if the shared
library is missing or a call fails, a RuntimeError is raised -- the product path never routes
through PyTorch ops or the CPU oracle.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
# MG_LIB_PATH lets a benchmark A/B two builds of the library in one process tree (same device, same clocks)
LIB_PATH = os.environ.get("MG_LIB_PATH") or os.path.join(os.path.dirname(_PKG), "lib", "libmatchglue.so")

MG_SP_MAGICLEAP, MG_SP_OPEN = 0, 1
MG_SAMPLE_LEGACY, MG_SAMPLE_FIXED, MG_SAMPLE_OPEN, MG_SAMPLE_CORNER = 0, 1, 2, 3

# every symbol include/matchglue.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mg_create", "mg_destroy", "mg_last_error", "mg_abi_version", "mg_workspace_generation", "mg_set_precision", "mg_set_weight", "mg_weights_ready",
    "mg_sp_forward", "mg_sp_detect", "mg_sp_sample", "mg_lg_forward", "mg_lg_forward_adaptive", "mg_sg_forward", "mg_gs_forward", "mg_wf_filter", "mg_nn_forward", "mg_debug_request", "mg_debug_read",
    "mg_profile_enable", "mg_profile_filter", "mg_profile_read", "mg_test_linear", "mg_test_attention",
]


class SpConf(Structure):
    _fields_ = [
        ("variant", c_int32),
        ("nms_radius", c_int32),
        ("remove_borders", c_int32),
        ("max_num_keypoints", c_int32),
        ("sampling", c_int32),
        ("detection_threshold", c_float),
    ]


class LgConf(Structure):
    _fields_ = [("n_layers", c_int32), ("num_heads", c_int32), ("filter_threshold", c_float)]


class SgConf(Structure):
    _fields_ = [("n_layers", c_int32), ("sinkhorn_iterations", c_int32), ("filter_threshold", c_float)]


class GsConf(Structure):
    _fields_ = [("n_layers", c_int32), ("filter_threshold", c_float)]


class NnConf(Structure):
    _fields_ = [("ratio_thresh", c_float), ("distance_thresh", c_float), ("mutual_check", c_int32)]


_lib = None


def lib():
    """Load the shared library once; fail loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libmatchglue.so not found at {LIB_PATH}: build it with `python glue-factory_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU/PyTorch fallback for this path."
        )
    L = ctypes.CDLL(LIB_PATH)
    L.mg_last_error.restype = c_char_p
    L.mg_workspace_generation.restype = ctypes.c_ulonglong
    L.mg_workspace_generation.argtypes = []
    L.mg_create.argtypes = [c_int, POINTER(c_void_p)]
    L.mg_destroy.argtypes = [c_void_p]
    L.mg_set_precision.argtypes = [c_void_p, c_int]
    L.mg_set_weight.argtypes = [c_void_p, c_int, c_char_p, c_void_p, POINTER(c_int64), c_int]
    L.mg_weights_ready.argtypes = [c_void_p, c_int, c_int]
    L.mg_sp_forward.argtypes = [c_void_p, POINTER(SpConf), c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.mg_sp_detect.argtypes = [c_void_p, POINTER(SpConf), c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p]
    L.mg_sp_sample.argtypes = [c_void_p, POINTER(SpConf), c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    L.mg_lg_forward.argtypes = [c_void_p, POINTER(LgConf)] + [c_void_p] * 6 + [c_int] * 3 + [c_void_p] * 8
    L.mg_lg_forward_adaptive.argtypes = [c_void_p, POINTER(LgConf), c_double, c_double] + [c_void_p] * 6 + [c_int] * 2 + [c_void_p] * 9 + [POINTER(c_int32), c_void_p]
    L.mg_sg_forward.argtypes = [c_void_p, POINTER(SgConf)] + [c_void_p] * 8 + [c_int] * 3 + [c_void_p] * 7
    L.mg_gs_forward.argtypes = [c_void_p, POINTER(GsConf)] + [c_void_p] * 14 + [c_int] * 5 + [c_void_p] * 12
    L.mg_wf_filter.argtypes = [c_void_p] + [c_void_p] * 5 + [c_int] * 3 + [c_float] + [c_void_p] * 5
    L.mg_nn_forward.argtypes = [c_void_p, POINTER(NnConf), c_void_p, c_void_p] + [c_int] * 4 + [c_void_p] * 7
    L.mg_debug_request.argtypes = [c_void_p, c_char_p]
    L.mg_debug_read.argtypes = [c_void_p, c_char_p, c_void_p, c_size_t, POINTER(c_size_t), c_void_p]
    L.mg_profile_enable.argtypes = [c_void_p, c_int]
    L.mg_profile_filter.argtypes = [c_void_p, c_char_p]
    L.mg_profile_read.argtypes = [c_void_p, c_void_p, POINTER(c_float), c_int]
    L.mg_test_linear.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    L.mg_test_attention.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"libmatchglue error {rc}: {lib().mg_last_error().decode()}")


def _ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def _stream(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _f32(t, device):
    return t.to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """One native handle (weights + workspace) bound to one HIP device."""

    def __init__(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                f"the MI355X path runs on a HIP device only (got '{device}'); there is no CPU fallback -- "
                "move the module and its inputs to 'cuda'"
            )
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        self._h = c_void_p()
        _check(lib().mg_create(self.device.index, ctypes.byref(self._h)))
        self._keep = []

    def close(self):
        if self._h:
            lib().mg_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_precision(self, mode):
        """'f32' (fp32-input MFMA) or 'f16x3' (split-fp16 operands on the f16 MFMA)."""
        code = {"f32": 0, "f16x3": 1}.get(mode)
        if code is None:
            raise ValueError(f"precision must be 'f32' or 'f16x3', got {mode!r}")
        _check(lib().mg_set_precision(self._h, code))

    # -- weights ---------------------------------------------------------------------------------
    def set_weights(self, which, state_dict, variant_or_layers):
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            t = v.detach().to("cpu", torch.float32).contiguous()
            shape = (c_int64 * max(t.dim(), 1))(*t.shape)
            _check(lib().mg_set_weight(self._h, which, k.encode(), c_void_p(t.data_ptr()), shape, t.dim()))
        _check(lib().mg_weights_ready(self._h, which, variant_or_layers))

    # -- SuperPoint ------------------------------------------------------------------------------
    def sp_forward(self, conf, image, image_size=None, want_heatmap=False, want_dense=False):
        dev = self.device
        image = _f32(image, dev)
        B, C, H, W = image.shape
        K = conf.max_num_keypoints
        isz = None if image_size is None else _f32(image_size, dev)
        kp = torch.empty(B, K, 2, device=dev)
        sc = torch.empty(B, K, device=dev)
        ds = torch.empty(B, K, 256, device=dev)
        nv = torch.empty(B, dtype=torch.int32, device=dev)
        hc, wc = H // 8, W // 8
        heat = torch.empty(B, hc * 8, wc * 8, device=dev) if want_heatmap else None
        dense = torch.empty(B, 256, hc, wc, device=dev) if want_dense else None
        _check(lib().mg_sp_forward(self._h, ctypes.byref(conf), _ptr(image), B, C, H, W, _ptr(isz), _ptr(kp), _ptr(sc), _ptr(ds),
                                   _ptr(nv), _ptr(heat), _ptr(dense), _stream(dev)))
        return kp, sc, ds, nv, heat, dense

    def sp_detect(self, conf, heatmap, image_size=None, want_nms=False):
        dev = self.device
        heatmap = _f32(heatmap, dev)
        B, Hs, Ws = heatmap.shape
        K = conf.max_num_keypoints
        isz = None if image_size is None else _f32(image_size, dev)
        kp = torch.empty(B, K, 2, device=dev)
        sc = torch.empty(B, K, device=dev)
        nv = torch.empty(B, dtype=torch.int32, device=dev)
        nms = torch.empty(B, Hs, Ws, device=dev) if want_nms else None
        _check(lib().mg_sp_detect(self._h, ctypes.byref(conf), _ptr(heatmap), B, Hs, Ws, _ptr(isz), _ptr(kp), _ptr(sc), _ptr(nv),
                                  _ptr(nms), _stream(dev)))
        return kp, sc, nv, nms

    def sp_sample(self, conf, keypoints, dense_nchw):
        dev = self.device
        keypoints = _f32(keypoints, dev)
        dense_nchw = _f32(dense_nchw, dev)
        B, K, _ = keypoints.shape
        _, Cc, hc, wc = dense_nchw.shape
        assert Cc == 256
        out = torch.empty(B, K, 256, device=dev)
        _check(lib().mg_sp_sample(self._h, ctypes.byref(conf), _ptr(keypoints), _ptr(dense_nchw), B, K, hc, wc, _ptr(out), _stream(dev)))
        return out

    # -- LightGlue -------------------------------------------------------------------------------
    def lg_forward(self, conf, k0, k1, d0, d1, size0=None, size1=None, want_log_assignment=True, want_ref=True):
        dev = self.device
        k0, k1, d0, d1 = (_f32(t, dev) for t in (k0, k1, d0, d1))
        B, M, _ = k0.shape
        N = k1.shape[1]
        if d0.shape != (B, M, 256) or d1.shape != (B, N, 256):
            raise RuntimeError(f"descriptors must be (B,K,256); got {tuple(d0.shape)} / {tuple(d1.shape)}")
        s0 = None if size0 is None else _f32(size0, dev).reshape(-1, 2).expand(B, 2).contiguous()
        s1 = None if size1 is None else _f32(size1, dev).reshape(-1, 2).expand(B, 2).contiguous()
        m0 = torch.empty(B, M, dtype=torch.int64, device=dev)
        m1 = torch.empty(B, N, dtype=torch.int64, device=dev)
        ms0 = torch.empty(B, M, device=dev)
        ms1 = torch.empty(B, N, device=dev)
        la = torch.empty(B, M + 1, N + 1, device=dev) if want_log_assignment else None
        r0 = torch.empty(B, M, 256, device=dev) if want_ref else None
        r1 = torch.empty(B, N, 256, device=dev) if want_ref else None
        _check(lib().mg_lg_forward(self._h, ctypes.byref(conf), _ptr(k0), _ptr(k1), _ptr(d0), _ptr(d1), _ptr(s0), _ptr(s1), B, M, N,
                                   _ptr(m0), _ptr(m1), _ptr(ms0), _ptr(ms1), _ptr(la), _ptr(r0), _ptr(r1), _stream(dev)))
        return m0, m1, ms0, ms1, la, r0, r1

    def lg_forward_adaptive(self, conf, depth_confidence, width_confidence, k0, k1, d0, d1, size0=None, size1=None, want_log_assignment=True):
        """Adaptive depth / width LightGlue, batch 1.  Returns matches / scores in the original indices, the log assignment and the
        final descriptors of the surviving points, the prune counters and (M', N', layers run)."""
        dev = self.device
        k0, k1, d0, d1 = (_f32(t, dev) for t in (k0, k1, d0, d1))
        B, M, _ = k0.shape
        N = k1.shape[1]
        if B != 1:
            raise RuntimeError("adaptive depth / width needs batch 1 (lightglue.py:476,:481)")
        if d0.shape != (1, M, 256) or d1.shape != (1, N, 256):
            raise RuntimeError(f"descriptors must be (1,K,256); got {tuple(d0.shape)} / {tuple(d1.shape)}")
        s0 = None if size0 is None else _f32(size0, dev).reshape(-1, 2)[:1].contiguous()
        s1 = None if size1 is None else _f32(size1, dev).reshape(-1, 2)[:1].contiguous()
        m0 = torch.empty(1, M, dtype=torch.int64, device=dev)
        m1 = torch.empty(1, N, dtype=torch.int64, device=dev)
        ms0 = torch.empty(1, M, device=dev)
        ms1 = torch.empty(1, N, device=dev)
        la = torch.empty((M + 1) * (N + 1), device=dev) if want_log_assignment else None
        r0 = torch.empty(M, 256, device=dev)
        r1 = torch.empty(N, 256, device=dev)
        p0 = torch.empty(1, M, dtype=torch.int32, device=dev)
        p1 = torch.empty(1, N, dtype=torch.int32, device=dev)
        dims = (c_int32 * 3)()
        _check(lib().mg_lg_forward_adaptive(self._h, ctypes.byref(conf), float(depth_confidence), float(width_confidence), _ptr(k0), _ptr(k1), _ptr(d0),
                                            _ptr(d1), _ptr(s0), _ptr(s1), M, N, _ptr(m0), _ptr(m1), _ptr(ms0), _ptr(ms1), _ptr(la), _ptr(r0), _ptr(r1),
                                            _ptr(p0), _ptr(p1), dims, _stream(dev)))
        Mc, Nc, layers = int(dims[0]), int(dims[1]), int(dims[2])
        if la is not None:
            la = la[: (Mc + 1) * (Nc + 1)].reshape(1, Mc + 1, Nc + 1)
        return m0, m1, ms0, ms1, la, r0[None, :Mc], r1[None, :Nc], p0, p1, layers

    # -- SuperGlue -------------------------------------------------------------------------------
    def sg_forward(self, conf, k0, k1, d0, d1, sc0, sc1, size0, size1, want_log_assignment=True, want_cost=True):
        dev = self.device
        k0, k1, d0, d1, sc0, sc1 = (_f32(t, dev) for t in (k0, k1, d0, d1, sc0, sc1))
        B, M, _ = k0.shape
        N = k1.shape[1]
        if d0.shape != (B, M, 256) or d1.shape != (B, N, 256):
            raise RuntimeError(f"descriptors must be (B,K,256); got {tuple(d0.shape)} / {tuple(d1.shape)}")
        if sc0.shape != (B, M) or sc1.shape != (B, N):
            raise RuntimeError(f"keypoint scores must be (B,K); got {tuple(sc0.shape)} / {tuple(sc1.shape)}")
        s0 = _f32(size0, dev).reshape(-1, 2).expand(B, 2).contiguous()
        s1 = _f32(size1, dev).reshape(-1, 2).expand(B, 2).contiguous()
        m0 = torch.empty(B, M, dtype=torch.int64, device=dev)
        m1 = torch.empty(B, N, dtype=torch.int64, device=dev)
        ms0 = torch.empty(B, M, device=dev)
        ms1 = torch.empty(B, N, device=dev)
        la = torch.empty(B, M + 1, N + 1, device=dev) if want_log_assignment else None
        cost = torch.empty(B, M, N, device=dev) if want_cost else None
        _check(lib().mg_sg_forward(self._h, ctypes.byref(conf), _ptr(k0), _ptr(k1), _ptr(d0), _ptr(d1), _ptr(sc0), _ptr(sc1), _ptr(s0),
                                   _ptr(s1), B, M, N, _ptr(m0), _ptr(m1), _ptr(ms0), _ptr(ms1), _ptr(la), _ptr(cost), _stream(dev)))
        return m0, m1, ms0, ms1, la, cost

    # -- GlueStick -------------------------------------------------------------------------------
    def gs_forward(self, conf, k0, k1, d0, d1, sc0, sc1, size0, size1, lines0, lines1, idx0, idx1, ls0, ls1, want_log_assignment=True):
        dev = self.device
        k0, k1, d0, d1, sc0, sc1 = (_f32(t, dev) for t in (k0, k1, d0, d1, sc0, sc1))
        B, M, _ = k0.shape
        N = k1.shape[1]
        n0, n1 = lines0.shape[1], lines1.shape[1]
        if d0.shape != (B, M, 256) or d1.shape != (B, N, 256):
            raise RuntimeError(f"descriptors must be (B,K,256); got {tuple(d0.shape)} / {tuple(d1.shape)}")
        s0 = _f32(size0, dev).reshape(-1, 2).expand(B, 2).contiguous()
        s1 = _f32(size1, dev).reshape(-1, 2).expand(B, 2).contiguous()
        m0 = torch.empty(B, M, dtype=torch.int64, device=dev)
        m1 = torch.empty(B, N, dtype=torch.int64, device=dev)
        ms0 = torch.empty(B, M, device=dev)
        ms1 = torch.empty(B, N, device=dev)
        la = torch.empty(B, M + 1, N + 1, device=dev) if want_log_assignment else None
        lines = n0 > 0 and n1 > 0
        L0 = L1 = I0 = I1 = S0 = S1 = lm0 = lm1 = lms0 = lms1 = lla = raw = None
        if lines:
            L0, L1, S0, S1 = _f32(lines0, dev), _f32(lines1, dev), _f32(ls0, dev), _f32(ls1, dev)
            I0 = idx0.to(device=dev, dtype=torch.int64).contiguous()
            I1 = idx1.to(device=dev, dtype=torch.int64).contiguous()
            if L0.shape != (B, n0, 2, 2) or L1.shape != (B, n1, 2, 2) or I0.shape != (B, n0, 2) or I1.shape != (B, n1, 2):
                raise RuntimeError("lines must be (B,n,2,2) and lines_junc_idx (B,n,2)")
            lm0 = torch.empty(B, n0, dtype=torch.int64, device=dev)
            lm1 = torch.empty(B, n1, dtype=torch.int64, device=dev)
            lms0 = torch.empty(B, n0, device=dev)
            lms1 = torch.empty(B, n1, device=dev)
            lla = torch.empty(B, n0 + 1, n1 + 1, device=dev)
            raw = torch.empty(B, n0, n1, device=dev)
        _check(lib().mg_gs_forward(self._h, ctypes.byref(conf), _ptr(k0), _ptr(k1), _ptr(d0), _ptr(d1), _ptr(sc0), _ptr(sc1), _ptr(s0), _ptr(s1),
                                   _ptr(L0), _ptr(L1), _ptr(I0), _ptr(I1), _ptr(S0), _ptr(S1), B, M, N, n0, n1, _ptr(m0), _ptr(m1), _ptr(ms0),
                                   _ptr(ms1), _ptr(la), _ptr(lm0), _ptr(lm1), _ptr(lms0), _ptr(lms1), _ptr(lla), _ptr(raw), _stream(dev)))
        return (m0, m1, ms0, ms1, la), (lm0, lm1, lms0, lms1, lla, raw)

    # -- wireframe ------------------------------------------------------------------------------------
    def wf_filter(self, keypoints, scores, descriptors, endpoints, radius, num_valid=None):
        """Drop the keypoints closer than `radius` to a line endpoint (wireframe.py:184-216); survivors first, original order."""
        dev = self.device
        kp, sc, ds, ep = (_f32(t, dev) for t in (keypoints, scores, descriptors, endpoints))
        B, K, _ = kp.shape
        E = ep.shape[1]
        if sc.shape != (B, K) or ds.shape != (B, K, 256) or ep.shape != (B, E, 2):
            raise RuntimeError("keypoints (B,K,2), scores (B,K), descriptors (B,K,256), endpoints (B,E,2) expected")
        nv = None if num_valid is None else num_valid.to(device=dev, dtype=torch.int32).contiguous()
        okp, osc, ods = torch.empty_like(kp), torch.empty_like(sc), torch.empty_like(ds)
        cnt = torch.empty(B, dtype=torch.int32, device=dev)
        _check(lib().mg_wf_filter(self._h, _ptr(kp), _ptr(sc), _ptr(ds), _ptr(nv), _ptr(ep), B, K, E, float(radius), _ptr(okp), _ptr(osc),
                                  _ptr(ods), _ptr(cnt), _stream(dev)))
        return okp, osc, ods, cnt

    # -- nearest-neighbour matcher ----------------------------------------------------------------
    def nn_forward(self, conf, d0, d1, want_similarity=True, want_log_assignment=True):
        dev = self.device
        d0, d1 = _f32(d0, dev), _f32(d1, dev)
        B, M, D = d0.shape
        N = d1.shape[1]
        if d1.shape[0] != B or d1.shape[2] != D:
            raise RuntimeError(f"descriptor shapes do not agree: {tuple(d0.shape)} / {tuple(d1.shape)}")
        m0 = torch.empty(B, M, dtype=torch.int64, device=dev)
        m1 = torch.empty(B, N, dtype=torch.int64, device=dev)
        ms0 = torch.empty(B, M, device=dev)
        ms1 = torch.empty(B, N, device=dev)
        sim = torch.empty(B, M, N, device=dev) if want_similarity else None
        la = torch.empty(B, M + 1, N + 1, device=dev) if want_log_assignment else None
        _check(lib().mg_nn_forward(self._h, ctypes.byref(conf), _ptr(d0), _ptr(d1), B, M, N, D, _ptr(m0), _ptr(m1), _ptr(ms0), _ptr(ms1),
                                   _ptr(sim), _ptr(la), _stream(dev)))
        return m0, m1, ms0, ms1, sim, la

    # -- introspection ---------------------------------------------------------------------------
    def debug_request(self, tap):
        _check(lib().mg_debug_request(self._h, tap.encode()))

    def debug_read(self, tap, numel):
        out = torch.empty(numel, device=self.device)
        n = c_size_t(0)
        _check(lib().mg_debug_read(self._h, tap.encode(), _ptr(out), numel, ctypes.byref(n), _stream(self.device)))
        return out[: n.value]

    def profile_enable(self, on=True):
        _check(lib().mg_profile_enable(self._h, 1 if on else 0))

    def profile_filter(self, name=None):
        """Time only the scope called `name` (None: every scope).  Two HIP events per timed launch cost ~10 us of stream time."""
        _check(lib().mg_profile_filter(self._h, name.encode() if name else None))

    def profile_read(self, cap=4096):
        names = ctypes.create_string_buffer(48 * cap)
        ms = (c_float * cap)()
        n = lib().mg_profile_read(self._h, names, ms, cap)
        if n < 0:
            _check(n)
        out = []
        for i in range(min(n, cap)):
            out.append((names.raw[48 * i: 48 * (i + 1)].split(b"\0", 1)[0].decode(), float(ms[i])))
        return out

    def test_linear(self, A, W, bias=None):
        A = _f32(A, self.device)
        Wc = W.detach().to("cpu", torch.float32).contiguous()
        bc = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
        M, K = A.shape
        N = Wc.shape[0]
        C = torch.empty(M, N, device=self.device)
        _check(lib().mg_test_linear(self._h, _ptr(A), c_void_p(Wc.data_ptr()), None if bc is None else c_void_p(bc.data_ptr()), M, N, K,
                                    _ptr(C), _stream(self.device)))
        return C

    def test_attention(self, q, k, v, B, M, N, cross=False):
        """softmax(q k^T) v per head on (B*(M+N), 256) token rows (image-0 rows first), in the engine's precision mode."""
        q, k, v = (_f32(t, self.device) for t in (q, k, v))
        T = B * (M + N)
        if q.shape != (T, 256) or k.shape != (T, 256) or v.shape != (T, 256):
            raise RuntimeError("q, k, v must be (B*(M+N), 256)")
        out = torch.empty(T, 256, device=self.device)
        _check(lib().mg_test_attention(self._h, _ptr(q), _ptr(k), _ptr(v), B, M, N, 1 if cross else 0, _ptr(out), _stream(self.device)))
        return out
