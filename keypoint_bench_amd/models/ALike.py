"""Drop-in for the reference's models/ALike.py: ``ALNet(param)`` with ``load_state_dict`` / ``eval`` /
``__call__(image) -> (scores_map, descriptor_map)``, computed by csrc/alike.hip through libkpb.so.
No torch.nn forward runs: tensors are containers for the image, the weights and the two outputs.

    net = ALNet({'c1': 8, 'c2': 16, 'c3': 32, 'c4': 64, 'dim': 64})      # Alike_params of the YAML configs
    net.load_state_dict(torch.load('weights/alike-t.pth'))               # model_interface.py:43-45
    score_map, desc_map = net.eval()(img)                                # model_interface.py:205-212

``descriptor_map`` is the reference's [B, dim, H, W] tensor stored channels-last (a valid torch
memory format; ``.shape[2:4]``, ``.detach()`` and ``grid_sample`` all behave).  With
``dense_descriptors=False`` the 78.6 MB/image dense map is not materialised: a ``LazyDescriptors``
handle is returned instead, which ``brute_force_matcher`` samples at the keypoints (same values: the
1x1 head and the bilinear sampling are both linear).
"""
import ctypes

import torch

from .. import weights as _weights
from .._lib import Context, c_void_p, ptr


class LazyDescriptors:
    """Stands where the dense descriptor map would be; only the matcher ever looks inside it
    (tasks pass desc_map through opaquely: MHA.py:38-39, AUC.py:119-120)."""

    def __init__(self, net, shape, slot, batch_index=0):
        self._net, self.shape, self._b, self._slot = net, torch.Size(shape), batch_index, slot
        self._stamp = net._slot_stamp[slot]

    def detach(self):
        return self

    @property
    def device(self):
        return self._net._device

    def sample(self, pts: torch.Tensor) -> torch.Tensor:
        """Descriptors at pts [N, >=2] (x, y normalised): what grid_sample on the dense map returns."""
        net = self._net
        if self._stamp != net._slot_stamp[self._slot]:
            raise RuntimeError("LazyDescriptors used after %d later forwards of the same net; its features are gone "
                               "(construct the net with dense_descriptors=True to keep maps alive)" % net.KEEP)
        return net._desc_at(pts, self._slot, self._b)


class ALNet:
    """models/ALike.py:84-164."""

    KEEP = 2    # dense_descriptors=False: the features of the last KEEP forwards stay alive (the reference's pattern is
                # forward(img0), forward(img1), then the matcher: model_interface.py:205-212 -> tasks/MHA.py:38-39)

    def __init__(self, param=None, dense_descriptors=True):
        if param is None:
            param = dict(c1=32, c2=64, c3=128, c4=128, dim=128)   # ALike.py:87-92
        self.param = dict(c1=param["c1"], c2=param["c2"], c3=param["c3"], c4=param["c4"], dim=param["dim"])
        if (self.param["c1"], self.param["c2"], self.param["c3"], self.param["c4"], self.param["dim"]) != (8, 16, 32, 64, 64):
            raise NotImplementedError("this build carries kernels for ALIKE-t (c1..c4 = 8,16,32,64, dim = 64) only")
        self.dense_descriptors = dense_descriptors
        self._handle = None
        self._extra = []            # the other KEEP-1 native nets of the keypoint-only mode (own activations each)
        self._slot_stamp = [0] * self.KEEP
        self._ctx = None
        self._device = None
        self._blob = None
        self._forward_count = 0
        self.training = False
        self.dim, self.desc_div = self.param["dim"], 1

    # ---- torch.nn.Module surface used by model_interface.py:43-86
    def load_state_dict(self, state_dict, strict=True):
        self._blob = _weights.pack(_weights.fold_alike(state_dict), _weights.ARCH_ALIKE)
        self._release()
        return "<All keys matched successfully>"

    def load_packed(self, blob: bytes):
        """Load an already folded .kpbw blob (keypoint_bench_amd/weights/alike-t.kpbw)."""
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

    # ---- forward
    def _ensure(self, device):
        ctx = Context.get(device)       # every forward: the context follows torch's CURRENT stream (torch.cuda.stream(s))
        if self._handle is not None and self._device == device:
            return
        if self._blob is None:
            raise RuntimeError("ALNet: load_state_dict() / load_packed() must be called before forward")
        self._release()
        self._ctx = ctx
        hs = []
        for _ in range(1 if self.dense_descriptors else self.KEEP):
            h = c_void_p()
            self._ctx.check(self._ctx.lib.kpb_net_create(self._ctx.handle, _weights.ARCH_ALIKE, self._blob, len(self._blob),
                                                         ctypes.byref(h)))
            hs.append(h)
        self._handle, self._extra, self._device = hs[0], hs[1:], device

    def _release(self):
        if self._handle is not None:
            for h in [self._handle] + self._extra:
                self._ctx.lib.kpb_net_destroy(h)
            self._handle, self._extra = None, []

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def forward(self, image: torch.Tensor):
        if not image.is_cuda:
            raise RuntimeError("keypoint_bench_amd.ALNet needs a CUDA/HIP tensor (MI355X); there is no CPU path")
        if image.dim() != 4 or image.shape[1] != 3:
            raise ValueError("image must be B x 3 x H x W")
        x = image.detach().to(torch.float32).contiguous()
        B, _, H, W = x.shape
        self._ensure(x.device)
        score = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        desc = None
        if self.dense_descriptors:
            desc = torch.empty((B, H, W, self.param["dim"]), dtype=torch.float32, device=x.device)
        slot = 0 if self.dense_descriptors else self._forward_count % self.KEEP
        handle = ([self._handle] + self._extra)[slot]
        self._ctx.check(self._ctx.lib.kpb_net_forward(handle, ptr(x), B, H, W, ptr(score), ptr(desc)))
        self._forward_count += 1
        self._slot_stamp[slot] = self._forward_count
        if desc is not None:
            return score, desc.permute(0, 3, 1, 2)   # [B, dim, H, W] view, channels-last storage
        return score, LazyDescriptors(self, (B, self.param["dim"], H, W), slot)

    __call__ = forward

    def _desc_at(self, pts: torch.Tensor, slot: int = 0, batch_index: int = 0):
        p = pts.detach().to(torch.float32).contiguous()
        n = p.shape[0]
        out = torch.empty((n, self.param["dim"]), dtype=torch.float32, device=p.device)
        if n == 0:
            return out
        if batch_index != 0 or self._last_batch() != 1:
            raise NotImplementedError("LazyDescriptors.sample: batch_size 1 only (config/config_MHA.yaml:10); "
                                      "use keypoint_bench_amd.pipeline for batched pairs")
        self._ctx.check(self._ctx.lib.kpb_net_desc_at(([self._handle] + self._extra)[slot], ptr(p), p.shape[1], n, ptr(None), ptr(out)))
        return out

    def _last_batch(self):
        return 1


def alike_t(device=None, dense_descriptors=True) -> "ALNet":
    """ALIKE-t with the weights shipped in keypoint_bench_amd/weights/alike-t.kpbw."""
    import os
    net = ALNet(dict(c1=8, c2=16, c3=32, c4=64, dim=64), dense_descriptors=dense_descriptors)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights", "alike-t.kpbw")
    with open(path, "rb") as f:
        net.load_packed(f.read())
    return net
