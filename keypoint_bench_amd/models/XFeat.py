"""Drop-in for the reference's models/XFeat.py: ``XFeatModel()`` with ``load_state_dict`` / ``eval`` /
``__call__(image) -> (heatmap [B,1,H,W], feats [B,64,H/8,W/8])`` (XFeat.py:112-140), computed by
csrc/convnet.hip through libkpb.so.  The feature map is stored channels-last."""
from .. import weights as _weights
from ._base import HipNet


class XFeatModel(HipNet):
    ARCH = _weights.ARCH_XFEAT

    def load_state_dict(self, state_dict, strict=True):
        self.load_packed(_weights.pack(_weights.fold_xfeat(state_dict), _weights.ARCH_XFEAT))
        return "<All keys matched successfully>"


def xfeat_random(seed=0) -> "XFeatModel":
    """XFeat with seeded random weights (the reference checkpoint xfeat.pt is not in its tree)."""
    net = XFeatModel()
    net.load_state_dict(_weights.random_xfeat_state_dict(seed))
    return net
