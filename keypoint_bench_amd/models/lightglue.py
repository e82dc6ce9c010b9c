"""Drop-in for the reference's models/lightglue.py: ``LightGlue(features=..., weight_path=...)`` with
``.match(pts0, pts1, desc_map_0, desc_map_1, {'w': W, 'h': H}) -> (pts0_matched, pts1_matched)``
(lightglue.py:447-477), computed by csrc/lightglue.hip through libkpb.so.

Behaviour follows the reference's fp32 CPU path: early stopping and point pruning are both active
(pruning_keypoint_thresholds['cpu'] = -1, lightglue.py:351-357); pass ``prune_min_kpts=1024`` (or 1536) to
mimic what the reference does on CUDA instead."""
import ctypes
import os

import torch

from .. import weights as _weights
from .._lib import Context, LgParams, c_void_p, ptr

_FEATURES = {"superpoint": dict(weights="superpoint_lightglue", input_dim=256, desc_scale=8),     # lightglue.py:361-385, 405-408
             "disk": dict(weights="disk_lightglue", input_dim=128, desc_scale=1)}


class LightGlue:
    default_conf = dict(depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1)       # lightglue.py:335-348

    def __init__(self, features="superpoint", weight_path="", desc_scale=None, prune_min_kpts=-1, attention="fp32", **conf):
        """attention: "fp32" (default) = the fp32 result of the reference's CPU branch (lightglue.py:135-137: what the fixtures pin), as split-f16
        MFMA triples; "f16" = what the reference runs on a GPU (lightglue.py:129-134: q.half(), k.half(), v.half() through SDPA)."""
        if features is not None and features not in _FEATURES:
            raise ValueError("Unsupported features: %r (this build: %s)" % (features, ", ".join(_FEATURES)))
        if attention not in ("fp32", "f16"):
            raise ValueError("attention must be 'fp32' or 'f16', not %r" % (attention,))
        self.attention = attention
        self.conf = dict(self.default_conf, **{k: v for k, v in conf.items() if k in self.default_conf})
        self.prune_min_kpts = int(prune_min_kpts)
        self.desc_scale = desc_scale
        self._blob = None
        self._handle = None
        self._ctx = None
        self._device = None
        if features is not None:
            f = _FEATURES[features]
            self.desc_scale = f["desc_scale"]
            path = os.path.join(str(weight_path), "weights", f["weights"] + ".pth")                 # lightglue.py:421-424
            self.load_state_dict(torch.load(path, map_location="cpu"))

    def load_state_dict(self, state_dict, strict=False):
        self._blob = _weights.pack(_weights.tensors_lightglue(state_dict), _weights.ARCH_LIGHTGLUE)
        self._release()
        return "<All keys matched successfully>"

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def _release(self):
        if self._handle is not None:
            self._ctx.lib.kpb_lg_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _ensure(self, device):
        ctx = Context.get(device)       # follow torch's current stream on every call
        if self._handle is not None and self._device == device:
            return
        if self._blob is None:
            raise RuntimeError("LightGlue: load_state_dict() must be called first")
        if self.desc_scale is None:
            raise RuntimeError("LightGlue(features=None): set desc_scale (8 for SuperPoint maps, 1 for DISK)")
        self._release()
        self._ctx = ctx
        h = c_void_p()
        self._ctx.check(self._ctx.lib.kpb_lg_create(self._ctx.handle, self._blob, len(self._blob), float(self.desc_scale), ctypes.byref(h)))
        self._handle, self._device = h, device
        if self.attention == "f16":
            self._ctx.check(self._ctx.lib.kpb_lg_set_attention(h, 1))

    def match_indices(self, pts0, pts1, desc_map_0, desc_map_1, params):
        """Returns (pairs [K,2] int64, scores [K] float32, layers_run)."""
        if not pts0.is_cuda:
            raise RuntimeError("keypoint_bench_amd.LightGlue needs CUDA/HIP tensors (MI355X); there is no CPU path")
        dev = pts0.device
        self._ensure(dev)
        p0 = pts0.detach().to(torch.float32).contiguous()
        p1 = pts1.detach().to(torch.float32).contiguous()
        if p0.shape[1] != 3 or p1.shape[1] != 3:
            raise ValueError("LightGlue.match expects (x, y, score) rows (lightglue.py:451-452)")
        n0, n1 = p0.shape[0], p1.shape[0]
        K = max(n0, n1, 1)
        pad = lambda p, n: torch.cat([p, torch.zeros((K - n, 3), device=dev)], 0) if n < K else p
        d0, d1 = desc_map_0.detach().float(), desc_map_1.detach().float()
        if d0.stride() != d1.stride() or d0.shape != d1.shape:
            d0, d1 = d0.contiguous(), d1.contiguous()
        _, C, Hd, Wd = d0.shape
        sb, sc, sh, sw = d0.stride()
        nn0 = torch.tensor([n0], dtype=torch.int32, device=dev)
        nn1 = torch.tensor([n1], dtype=torch.int32, device=dev)
        pairs = torch.empty((K, 2), dtype=torch.int32, device=dev)
        scores = torch.empty((K,), dtype=torch.float32, device=dev)
        k = torch.zeros((1,), dtype=torch.int32, device=dev)
        stop = torch.zeros((1,), dtype=torch.int32, device=dev)
        prm = LgParams(float(self.conf["depth_confidence"]), float(self.conf["width_confidence"]), float(self.conf["filter_threshold"]),
                       self.prune_min_kpts)
        ctx = self._ctx
        ctx.check(ctx.lib.kpb_lg_match(self._handle, ptr(pad(p0, n0)), ptr(pad(p1, n1)), ptr(nn0), ptr(nn1), 1, K, ptr(d0), ptr(d1), C, Hd, Wd,
                                       sb, sc, sh, sw, int(params["w"]), int(params["h"]), ctypes.byref(prm), ptr(pairs), ptr(scores), ptr(k), ptr(stop)))
        kk = int(k.item())
        return pairs[:kk].to(torch.int64), scores[:kk].clone(), int(stop.item())

    def match(self, pts0, pts1, desc_map_0, desc_map_1, params=None):
        """lightglue.py:447-477."""
        pairs, _, _ = self.match_indices(pts0, pts1, desc_map_0, desc_map_1, params)
        return pts0[pairs[:, 0]], pts1[pairs[:, 1]]
