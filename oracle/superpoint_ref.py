"""torch-fp32 restatement of models/SuperPoint.py (SuperPointNet.forward, lines 30-71) -- TEST INFRASTRUCTURE."""
import torch
import torch.nn.functional as F


def superpoint_forward(image, t):
    """image [B,3,H,W]; t: dict name -> torch tensor ('conv1a.weight', ...).  Returns (heatmap, desc)."""
    c = lambda x, n, pad: F.conv2d(x, t[n + ".weight"], t[n + ".bias"], padding=pad)
    B, _, H, W = image.shape
    Hc, Wc = H // 8, W // 8
    x = torch.sum(image, dim=1, keepdim=True)                                   # :42
    x = F.relu(c(x, "conv1a", 1)); x = F.relu(c(x, "conv1b", 1)); x = F.max_pool2d(x, 2, 2)   # :44-46
    x = F.relu(c(x, "conv2a", 1)); x = F.relu(c(x, "conv2b", 1)); x = F.max_pool2d(x, 2, 2)   # :47-49
    x = F.relu(c(x, "conv3a", 1)); x = F.relu(c(x, "conv3b", 1)); x = F.max_pool2d(x, 2, 2)   # :50-52
    x = F.relu(c(x, "conv4a", 1)); x = F.relu(c(x, "conv4b", 1))                               # :53-54
    semi = c(F.relu(c(x, "convPa", 1)), "convPb", 0)                            # :56-57
    desc = c(F.relu(c(x, "convDa", 1)), "convDb", 0)                            # :59-60
    desc = desc.div(torch.unsqueeze(torch.norm(desc, p=2, dim=1), 1))           # :61-62
    dense = torch.softmax(semi, dim=1)                                          # :65
    nodust = dense[:, :-1].permute(0, 2, 3, 1)
    heat = torch.reshape(nodust, [B, Hc, Wc, 8, 8]).permute(0, 1, 3, 2, 4)
    return torch.reshape(heat, [B, 1, Hc * 8, Wc * 8]), desc                    # :66-71
