"""torch-fp32 restatement of models/ALike.py (ALNet.forward, lines 136-164) -- TEST INFRASTRUCTURE.

Works from the BN-folded tensors of a .kpbw blob (the same bytes the HIP path consumes), so checking
this against the reference's golden outputs also pins the folding done in keypoint_bench_amd/weights.py.
Eval-mode BatchNorm (model_interface.py:86) is an affine map, folded as
    w' = w * gamma / sqrt(var + eps),  b' = beta - mean * gamma / sqrt(var + eps).
"""
import torch
import torch.nn.functional as F


def _conv_block(x, t, p):  # ConvBlock.forward, ALike.py:25-28
    x = F.relu(F.conv2d(x, t[p + "c1.w"], t[p + "c1.b"], padding=1))
    x = F.relu(F.conv2d(x, t[p + "c2.w"], t[p + "c2.b"], padding=1))
    return x


def _res_block(x, t, p):  # ResBlock.forward, ALike.py:65-81
    out = F.relu(F.conv2d(x, t[p + "c1.w"], t[p + "c1.b"], padding=1))
    out = F.conv2d(out, t[p + "c2.w"], t[p + "c2.b"], padding=1)
    cout, cin = t[p + "ds.w"].shape[:2]
    identity = F.conv2d(x, t[p + "ds.w"].reshape(cout, cin, 1, 1), t[p + "ds.b"])
    return F.relu(out + identity)


def alnet_forward(image, t, return_intermediates=False):
    """image [B,3,H,W] fp32 in [0,1]; t = dict of folded tensors (torch).  Returns (score, desc)."""
    up = lambda x, s: F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=True)
    c11 = lambda x, w: F.conv2d(x, w.reshape(w.shape[0], w.shape[1], 1, 1))
    x1 = _conv_block(image, t, "b1")                      # ALike.py:138
    x2 = _res_block(F.max_pool2d(x1, 2, 2), t, "b2")      # 139-140
    x3 = _res_block(F.max_pool2d(x2, 4, 4), t, "b3")      # 141-142
    x4 = _res_block(F.max_pool2d(x3, 4, 4), t, "b4")      # 143-144
    a1 = F.relu(c11(x1, t["agg1.w"]))                     # 147-150
    a2 = F.relu(c11(x2, t["agg2.w"]))
    a3 = F.relu(c11(x3, t["agg3.w"]))
    a4 = F.relu(c11(x4, t["agg4.w"]))
    x1234 = torch.cat([a1, up(a2, 2), up(a3, 8), up(a4, 32)], dim=1)  # 151-154
    x = c11(x1234, t["head.w"])                           # 159
    desc = x[:, :-1]                                      # 161
    score = torch.sigmoid(x[:, -1]).unsqueeze(1)          # 162
    if return_intermediates:
        return score, desc, dict(x1=x1, x2=x2, x3=x3, x4=x4, a2=a2, a3=a3, a4=a4)
    return score, desc
