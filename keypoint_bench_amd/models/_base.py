"""Shared torch.nn.Module-shaped surface of the HIP-backed extractor nets (model_interface.py:43-86 uses
load_state_dict / eval / __call__)."""
import ctypes

import torch

from .._lib import Context, c_void_p, ptr


class HipNet:
    ARCH = 0

    def __init__(self):
        self._handle = None
        self._ctx = None
        self._device = None
        self._blob = None
        self._forward_count = 0
        self.training = False

    # ---- torch.nn.Module surface
    def load_packed(self, blob: bytes):
        self._blob = bytes(blob)
        self._release()
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def parameters(self):
        return iter(())

    def _ensure(self, device):
        ctx = Context.get(device)       # every forward: the context follows torch's CURRENT stream (torch.cuda.stream(s))
        if self._handle is not None and self._device == device:
            return
        if self._blob is None:
            raise RuntimeError("%s: load_state_dict() / load_packed() must be called before forward" % type(self).__name__)
        self._release()
        self._ctx = ctx
        h = c_void_p()
        self._ctx.check(self._ctx.lib.kpb_net_create(self._ctx.handle, self.ARCH, self._blob, len(self._blob), ctypes.byref(h)))
        self._handle, self._device = h, device
        self.dim = self._ctx.lib.kpb_net_desc_dim(h)
        self.desc_div = self._ctx.lib.kpb_net_desc_div(h)

    def _release(self):
        if self._handle is not None:
            self._ctx.lib.kpb_net_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _run(self, image: torch.Tensor, want_desc=True):
        if not image.is_cuda:
            raise RuntimeError("keypoint_bench_amd.%s needs a CUDA/HIP tensor (MI355X); there is no CPU path" % type(self).__name__)
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError("image must be B x 3 x H x W")
        x = image.detach().to(torch.float32).contiguous()
        B, _, H, W = x.shape
        self._ensure(x.device)
        score = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        desc = None
        if want_desc:
            desc = torch.empty((B, H // self.desc_div, W // self.desc_div, self.dim), dtype=torch.float32, device=x.device)
        self._ctx.check(self._ctx.lib.kpb_net_forward(self._handle, ptr(x), B, H, W, ptr(score), ptr(desc)))
        self._forward_count += 1
        return score, desc

    def forward(self, image):
        score, desc = self._run(image)
        return score, desc.permute(0, 3, 1, 2)     # [B, C, H/div, W/div] view, channels-last storage

    __call__ = forward
