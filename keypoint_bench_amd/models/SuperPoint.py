"""Drop-in for the reference's models/SuperPoint.py: ``SuperPointNet()`` with ``load_state_dict`` / ``eval`` /
``__call__(image) -> (heatmap [B,1,H,W], desc [B,256,H/8,W/8])`` (SuperPoint.py:30-71), computed by
csrc/convnet.hip through libkpb.so.  The descriptor map is stored channels-last."""
from .. import weights as _weights
from ._base import HipNet


class SuperPointNet(HipNet):
    ARCH = _weights.ARCH_SUPERPOINT

    def load_state_dict(self, state_dict, strict=True):
        self.load_packed(_weights.pack(_weights.tensors_superpoint(state_dict), _weights.ARCH_SUPERPOINT))
        return "<All keys matched successfully>"


def superpoint_random(seed=0) -> "SuperPointNet":
    """SuperPoint with seeded random weights (the reference checkpoint superpoint_v1.pth is not in its tree)."""
    net = SuperPointNet()
    net.load_packed(_weights.pack(_weights.random_superpoint(seed), _weights.ARCH_SUPERPOINT))
    return net
