"""Device-side version of the per-image transform the reference's datasets apply after decoding
(datasets/hpatches.py:47-69: BGR -> RGB, / 255, cv2.resize to image_size, HWC -> CHW; datasets/megadepth.py:312-313:
transforms.ToTensor), computed by csrc/preprocess.hip.  Decoding stays on the host."""
import numpy as np
import torch

from .._lib import Context, ptr


def to_tensor_resized(img_u8, size=None, bgr=False, device="cuda:0"):
    """img_u8: uint8 [H, W, 3] or [B, H, W, 3] (numpy or torch), as decoded; size: None (keep), int (square, as
    hpatches.py:66-67) or (height, width).  Returns fp32 [B, 3, Hd, Wd] in [0, 1] on `device`."""
    t = torch.as_tensor(np.ascontiguousarray(img_u8) if isinstance(img_u8, np.ndarray) else img_u8)
    if t.dtype != torch.uint8 or t.shape[-1] != 3 or t.dim() not in (3, 4):
        raise ValueError("expected a uint8 [H, W, 3] or [B, H, W, 3] image")
    if t.dim() == 3:
        t = t[None]
    t = t.to(device).contiguous()
    if not t.is_cuda:
        raise RuntimeError("keypoint_bench_amd needs a CUDA/HIP device; there is no CPU path")
    B, Hs, Ws, _ = t.shape
    Hd, Wd = (Hs, Ws) if size is None else ((size, size) if isinstance(size, int) else size)
    out = torch.empty((B, 3, Hd, Wd), dtype=torch.float32, device=t.device)
    ctx = Context.get(t.device)
    ctx.check(ctx.lib.kpb_preprocess(ctx.handle, ptr(t), B, Hs, Ws, 1 if bgr else 0, Hd, Wd, ptr(out)))
    return out
