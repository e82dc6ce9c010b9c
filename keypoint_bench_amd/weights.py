"""Weight container (.kpbw) shared by the HIP path and the parity oracle.

Layout (little endian): 16-byte header {magic "KPBWGT1\\0", u32 arch, u32 n}, n records of
{char name[40]; u32 ndim; u32 dims[4]; u32 offset_in_floats}, then the fp32 payload.
Tensors keep the reference's OIHW order; kernels repack at kpb_net_create time.

``fold_alike`` turns a state_dict of models/ALike.py (ALNet) into the folded tensor set:
eval-mode BatchNorm (model_interface.py:86) is folded into the preceding conv.
"""
import struct

import numpy as np

MAGIC = b"KPBWGT1\0"
ARCH_ALIKE = 1
ARCH_SUPERPOINT = 2
ARCH_XFEAT = 3
ARCH_DISK = 4
ARCH_LIGHTGLUE = 5
_REC = struct.Struct("<40sI4II")


def pack(tensors: dict, arch: int) -> bytes:
    names = list(tensors)
    recs, payload, off = [], [], 0
    for n in names:
        a = np.ascontiguousarray(np.asarray(tensors[n], dtype=np.float32))
        dims = list(a.shape) + [1] * (4 - a.ndim)
        assert len(n.encode()) <= 40, "tensor name longer than the 40-byte field: " + n
        recs.append(_REC.pack(n.encode(), a.ndim, *dims, off))
        payload.append(a.tobytes())
        off += a.size
    return MAGIC + struct.pack("<II", arch, len(names)) + b"".join(recs) + b"".join(payload)


def unpack(blob: bytes):
    assert blob[:8] == MAGIC, "not a .kpbw blob"
    arch, n = struct.unpack_from("<II", blob, 8)
    base = 16 + n * _REC.size
    out = {}
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off = _REC.unpack_from(blob, 16 + i * _REC.size)
        shape = (d0, d1, d2, d3)[:ndim]
        cnt = int(np.prod(shape)) if ndim else 1
        out[name.rstrip(b"\0").decode()] = np.frombuffer(blob, np.float32, cnt, base + 4 * off).reshape(shape).copy()
    return arch, out


def _np(v):
    return v.detach().cpu().numpy().astype(np.float64) if hasattr(v, "detach") else np.asarray(v, np.float64)


def _fold(sd, conv, bn, eps=1e-5):
    w = _np(sd[conv + ".weight"])
    g, b = _np(sd[bn + ".weight"]), _np(sd[bn + ".bias"])
    mu, var = _np(sd[bn + ".running_mean"]), _np(sd[bn + ".running_var"])
    s = g / np.sqrt(var + eps)
    return (w * s[:, None, None, None]).astype(np.float32), (b - mu * s).astype(np.float32)


def fold_alike(sd) -> dict:
    """state_dict of ALNet (models/ALike.py:84-134) -> folded tensors named as csrc/alike.hip expects."""
    t = {}
    t["b1c1.w"], t["b1c1.b"] = _fold(sd, "block1.conv1", "block1.bn1")
    t["b1c2.w"], t["b1c2.b"] = _fold(sd, "block1.conv2", "block1.bn2")
    for i in (2, 3, 4):
        p, q = "block%d" % i, "b%d" % i
        t[q + "c1.w"], t[q + "c1.b"] = _fold(sd, p + ".conv1", p + ".bn1")
        t[q + "c2.w"], t[q + "c2.b"] = _fold(sd, p + ".conv2", p + ".bn2")
        w = _np(sd[p + ".downsample.weight"]).astype(np.float32)
        t[q + "ds.w"] = w.reshape(w.shape[0], w.shape[1])
        t[q + "ds.b"] = _np(sd[p + ".downsample.bias"]).astype(np.float32)
    for i in (1, 2, 3, 4):
        w = _np(sd["conv%d.weight" % i]).astype(np.float32)
        t["agg%d.w" % i] = w.reshape(w.shape[0], w.shape[1])
    w = _np(sd["convhead2.weight"]).astype(np.float32)
    t["head.w"] = w.reshape(w.shape[0], w.shape[1])
    return t


def load_alike_t():
    """The ALIKE-t tensors shipped with the package (folded from the reference's weights/alike-t.pth)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights", "alike-t.kpbw")
    with open(path, "rb") as f:
        return unpack(f.read())[1]


SUPERPOINT_LAYERS = ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b",
                     "convPa", "convPb", "convDa", "convDb")


def tensors_superpoint(sd) -> dict:
    """state_dict of SuperPointNet (models/SuperPoint.py:9-28) -> fp32 tensors under the same names."""
    t = {}
    for n in SUPERPOINT_LAYERS:
        t[n + ".weight"] = _np(sd[n + ".weight"]).astype(np.float32)
        t[n + ".bias"] = _np(sd[n + ".bias"]).astype(np.float32)
    return t


def random_superpoint(seed: int) -> dict:
    """Seeded stand-in for the absent superpoint_v1.pth (.MISSING_LARGE_BLOBS): He-uniform conv weights, small
    biases, drawn from numpy so the same seed gives the same tensors on every machine."""
    rng = np.random.default_rng(seed)
    shapes = dict(conv1a=(64, 1, 3), conv1b=(64, 64, 3), conv2a=(64, 64, 3), conv2b=(64, 64, 3), conv3a=(128, 64, 3),
                  conv3b=(128, 128, 3), conv4a=(128, 128, 3), conv4b=(128, 128, 3), convPa=(256, 128, 3),
                  convPb=(65, 256, 1), convDa=(256, 128, 3), convDb=(256, 256, 1))
    t = {}
    for n in SUPERPOINT_LAYERS:
        co, ci, k = shapes[n]
        bound = np.sqrt(6.0 / (ci * k * k))
        t[n + ".weight"] = rng.uniform(-bound, bound, size=(co, ci, k, k)).astype(np.float32)
        t[n + ".bias"] = rng.uniform(-0.05, 0.05, size=(co,)).astype(np.float32)
    return t


XFEAT_BASIC = ("block1.0", "block1.1", "block1.2", "block1.3", "block2.0", "block2.1", "block3.0", "block3.1", "block3.2",
               "block4.0", "block4.1", "block4.2", "block5.0", "block5.1", "block5.2", "block5.3",
               "block_fusion.0", "block_fusion.1", "keypoint_head.0", "keypoint_head.1", "keypoint_head.2")
XFEAT_SHAPES = {"block1.0": (4, 1, 3), "block1.1": (8, 4, 3), "block1.2": (8, 8, 3), "block1.3": (24, 8, 3),
                "block2.0": (24, 24, 3), "block2.1": (24, 24, 3), "block3.0": (64, 24, 3), "block3.1": (64, 64, 3),
                "block3.2": (64, 64, 1), "block4.0": (64, 64, 3), "block4.1": (64, 64, 3), "block4.2": (64, 64, 3),
                "block5.0": (128, 64, 3), "block5.1": (128, 128, 3), "block5.2": (128, 128, 3), "block5.3": (64, 128, 1),
                "block_fusion.0": (64, 64, 3), "block_fusion.1": (64, 64, 3), "keypoint_head.0": (64, 64, 1),
                "keypoint_head.1": (64, 64, 1), "keypoint_head.2": (64, 64, 1)}


def fold_xfeat(sd, eps=1e-5) -> dict:
    """state_dict of XFeatModel (models/XFeat.py:22-94) -> folded tensors.  BasicLayer (XFeat.py:7-19) is
    conv(bias=False) + BatchNorm2d(affine=False) + ReLU: w' = w / sqrt(var + eps), b' = -mean / sqrt(var + eps).
    heatmap_head and fine_matcher are constructed by the reference but never used by forward (XFeat.py:112-140)."""
    t = {}
    for n in XFEAT_BASIC:
        w = _np(sd[n + ".layer.0.weight"])
        mu, var = _np(sd[n + ".layer.1.running_mean"]), _np(sd[n + ".layer.1.running_var"])
        s = 1.0 / np.sqrt(var + eps)
        t[n + ".w"] = (w * s[:, None, None, None]).astype(np.float32)
        t[n + ".b"] = (-mu * s).astype(np.float32)
    t["block_fusion.2.w"] = _np(sd["block_fusion.2.weight"]).astype(np.float32)
    t["block_fusion.2.b"] = _np(sd["block_fusion.2.bias"]).astype(np.float32)
    t["keypoint_head.3.w"] = _np(sd["keypoint_head.3.weight"]).astype(np.float32)
    t["keypoint_head.3.b"] = _np(sd["keypoint_head.3.bias"]).astype(np.float32)
    t["skip1.w"] = _np(sd["skip1.1.weight"]).astype(np.float32).reshape(24)
    t["skip1.b"] = _np(sd["skip1.1.bias"]).astype(np.float32)
    return t


def random_xfeat_state_dict(seed: int) -> dict:
    """Seeded stand-in for the absent xfeat.pt: the state_dict entries XFeatModel.forward reads."""
    rng = np.random.default_rng(seed)
    sd = {}
    for n in XFEAT_BASIC:
        co, ci, k = XFEAT_SHAPES[n]
        bound = np.sqrt(6.0 / (ci * k * k))
        sd[n + ".layer.0.weight"] = rng.uniform(-bound, bound, size=(co, ci, k, k)).astype(np.float32)
        sd[n + ".layer.1.running_mean"] = rng.normal(0.0, 0.1, size=(co,)).astype(np.float32)
        sd[n + ".layer.1.running_var"] = rng.uniform(0.5, 1.5, size=(co,)).astype(np.float32)
    for n, (co, ci) in (("block_fusion.2", (64, 64)), ("keypoint_head.3", (65, 64))):
        bound = np.sqrt(6.0 / ci)
        sd[n + ".weight"] = rng.uniform(-bound, bound, size=(co, ci, 1, 1)).astype(np.float32)
        sd[n + ".bias"] = rng.uniform(-0.05, 0.05, size=(co,)).astype(np.float32)
    sd["skip1.1.weight"] = rng.uniform(-1, 1, size=(24, 1, 1, 1)).astype(np.float32)
    sd["skip1.1.bias"] = rng.uniform(-0.05, 0.05, size=(24,)).astype(np.float32)
    return sd

DISK_BLOCKS = (("down1", "unet.path_down.1.1", 16, 32), ("down2", "unet.path_down.2.1", 32, 64),
               ("down3", "unet.path_down.3.1", 64, 64), ("down4", "unet.path_down.4.1", 64, 64),
               ("up0", "unet.path_up.0.conv", 128, 64), ("up1", "unet.path_up.1.conv", 128, 64),
               ("up2", "unet.path_up.2.conv", 96, 64), ("up3", "unet.path_up.3.conv", 80, 129))


def tensors_disk(sd) -> dict:
    """state_dict of DISK (models/disk.py:293-307; the reference loads checkpoint['extractor'],
    model_interface.py:76-78) -> tensors named as csrc/convnet.hip expects.  Conv = Sequential(norm, PReLU,
    dropout, conv) (disk.py:76-97): index 1 is the PReLU slope, index 3 the convolution."""
    t = {"down0.w": _np(sd["unet.path_down.0.1.3.weight"]).astype(np.float32),
         "down0.b": _np(sd["unet.path_down.0.1.3.bias"]).astype(np.float32)}
    for name, key, cin, cout in DISK_BLOCKS:
        t[name + ".w"] = _np(sd[key + ".3.weight"]).astype(np.float32)
        t[name + ".b"] = _np(sd[key + ".3.bias"]).astype(np.float32)
        slope = _np(sd[key + ".1.weight"]).astype(np.float32).reshape(-1)
        t[name + ".slope"] = np.broadcast_to(slope, (cin,)).copy() if slope.size == 1 else slope
    return t


def random_disk_state_dict(seed: int) -> dict:
    """Seeded stand-in for the absent disk.pth."""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(key, co, ci):
        bound = np.sqrt(6.0 / (ci * 25))
        sd[key + ".3.weight"] = rng.uniform(-bound, bound, size=(co, ci, 5, 5)).astype(np.float32)
        sd[key + ".3.bias"] = rng.uniform(-0.05, 0.05, size=(co,)).astype(np.float32)

    conv("unet.path_down.0.1", 16, 3)
    for name, key, cin, cout in DISK_BLOCKS:
        conv(key, cout, cin)
        sd[key + ".1.weight"] = rng.uniform(0.1, 0.4, size=(cin,)).astype(np.float32)
    return sd


LG_LAYERS = 9


def tensors_lightglue(sd) -> dict:
    """state_dict of LightGlue (models/lightglue.py:392-409) -> fp32 tensors under the reference's own names; old
    checkpoints' self_attn.{i} / cross_attn.{i} keys are renamed as the reference does at load time (427-433)."""
    t = {}
    for k, v in sd.items():
        for i in range(LG_LAYERS):
            if not k.startswith("transformers."):
                k = k.replace("self_attn.%d" % i, "transformers.%d.self_attn" % i)
                k = k.replace("cross_attn.%d" % i, "transformers.%d.cross_attn" % i)
        if k == "confidence_thresholds" or k.endswith("inner_attn") or not hasattr(v, "shape"):
            continue
        t[k] = _np(v).astype(np.float32)
    return t


def random_lightglue_state_dict(seed: int, input_dim: int = 256, variant: str = "plain") -> dict:
    """Seeded stand-in for the absent superpoint_lightglue / disk_lightglue checkpoints.  The transformer is a
    damped random network (its residual branches are scaled down so descriptors stay close to their input) and
    final_proj is near a scaled identity, so that corresponding keypoints of a synthetic pair really match.
      plain  : no layer is confident -> all nine layers run, nothing is pruned
      stop   : token confidence saturates from layer 2 on -> check_if_stop fires (lightglue.py:670-681)
      prune  : token confidence and matchability are bimodal at layers 0-3 -> get_pruning_mask drops points (659-668)"""
    rng = np.random.default_rng(seed)
    sd = {}

    def lin(name, co, ci, scale=1.0, bias=0.02):
        b = np.sqrt(3.0 / ci) * scale
        sd[name + ".weight"] = rng.uniform(-b, b, size=(co, ci)).astype(np.float32)
        sd[name + ".bias"] = rng.uniform(-bias, bias, size=(co,)).astype(np.float32)

    if input_dim != 256:
        lin("input_proj", 256, input_dim, 1.0)
    sd["posenc.Wr.weight"] = rng.normal(0, 1.0, size=(32, 2)).astype(np.float32)
    for i in range(LG_LAYERS):
        for blk, names in (("self_attn", (("Wqkv", 768, 256, 1.0), ("out_proj", 256, 256, 0.3))),
                           ("cross_attn", (("to_qk", 256, 256, 1.0), ("to_v", 256, 256, 1.0), ("to_out", 256, 256, 0.3)))):
            p = "transformers.%d.%s" % (i, blk)
            for n, co, ci, sc in names:
                lin(p + "." + n, co, ci, sc)
            lin(p + ".ffn.0", 512, 512, 1.0)
            sd[p + ".ffn.1.weight"] = rng.uniform(0.8, 1.2, size=(512,)).astype(np.float32)
            sd[p + ".ffn.1.bias"] = rng.uniform(-0.1, 0.1, size=(512,)).astype(np.float32)
            lin(p + ".ffn.3", 256, 512, 0.15)
        p = "log_assignment.%d" % i
        sd[p + ".final_proj.weight"] = (10.0 * np.eye(256) + rng.normal(0, 0.05, size=(256, 256))).astype(np.float32)
        sd[p + ".final_proj.bias"] = rng.uniform(-0.02, 0.02, size=(256,)).astype(np.float32)
        sd[p + ".matchability.weight"] = rng.normal(0, 0.05, size=(1, 256)).astype(np.float32)
        sd[p + ".matchability.bias"] = np.array([3.0], np.float32)
        if i < LG_LAYERS - 1:
            p = "token_confidence.%d.token.0" % i
            sd[p + ".weight"] = rng.normal(0, 0.05, size=(1, 256)).astype(np.float32)
            sd[p + ".bias"] = np.array([-1.0], np.float32)
    if variant == "stop":
        for i in range(2, LG_LAYERS - 1):
            sd["token_confidence.%d.token.0.bias" % i] = np.array([6.0], np.float32)
    elif variant == "prune":
        for i in range(0, 4):
            sd["token_confidence.%d.token.0.weight" % i] = rng.normal(0, 4.0, size=(1, 256)).astype(np.float32)
            sd["token_confidence.%d.token.0.bias" % i] = np.array([0.0], np.float32)
            sd["log_assignment.%d.matchability.weight" % i] = rng.normal(0, 4.0, size=(1, 256)).astype(np.float32)
            sd["log_assignment.%d.matchability.bias" % i] = np.array([0.0], np.float32)
    elif variant != "plain":
        raise ValueError(variant)
    return sd
