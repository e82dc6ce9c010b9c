"""torch-fp32 restatement of models/XFeat.py (XFeatModel.forward, lines 112-140) from the BN-folded tensors
of keypoint_bench_amd.weights.fold_xfeat -- TEST INFRASTRUCTURE."""
import torch
import torch.nn.functional as F


def xfeat_forward(image, t):
    def basic(x, n, stride=1):
        k = t[n + ".w"].shape[-1]
        return F.relu(F.conv2d(x, t[n + ".w"], t[n + ".b"], stride=stride, padding=k // 2))

    x = image.mean(dim=1, keepdim=True)                                            # :122
    x = F.instance_norm(x, eps=1e-5)                                               # :123
    x1 = basic(basic(basic(basic(x, "block1.0"), "block1.1", 2), "block1.2"), "block1.3", 2)
    skip = F.avg_pool2d(x, 4, 4) * t["skip1.w"].reshape(1, 24, 1, 1) + t["skip1.b"].reshape(1, 24, 1, 1)   # :27-28
    x2 = basic(basic(x1 + skip, "block2.0"), "block2.1")                           # :127
    x3 = basic(basic(basic(x2, "block3.0", 2), "block3.1"), "block3.2")
    x4 = basic(basic(basic(x3, "block4.0", 2), "block4.1"), "block4.2")
    x5 = basic(basic(basic(basic(x4, "block5.0", 2), "block5.1"), "block5.2"), "block5.3")
    x4u = F.interpolate(x4, (x3.shape[-2], x3.shape[-1]), mode="bilinear")         # :133-134
    x5u = F.interpolate(x5, (x3.shape[-2], x3.shape[-1]), mode="bilinear")
    f = basic(basic(x3 + x4u + x5u, "block_fusion.0"), "block_fusion.1")
    feats = F.normalize(F.conv2d(f, t["block_fusion.2.w"], t["block_fusion.2.b"]), dim=1)   # :135-136
    B, C, H, W = x.shape
    u = x.unfold(2, 8, 8).unfold(3, 8, 8).reshape(B, C, H // 8, W // 8, 64).permute(0, 1, 4, 2, 3).reshape(B, -1, H // 8, W // 8)
    k = basic(basic(basic(u, "keypoint_head.0"), "keypoint_head.1"), "keypoint_head.2")
    k = F.conv2d(k, t["keypoint_head.3.w"], t["keypoint_head.3.b"])
    scores = F.softmax(k, 1)[:, :64]                                               # :106-110
    heat = scores.permute(0, 2, 3, 1).reshape(B, H // 8, W // 8, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, 1, H, W)
    return heat, feats
