"""Drop-in for the reference's models/disk.py: ``DISK()`` with ``load_state_dict`` / ``eval`` /
``__call__(image) -> (score_map [B,1,H,W], desc_map [B,128,H,W])`` (disk.py:309-313), computed by
csrc/convnet.hip through libkpb.so.  The descriptor map is stored channels-last."""
from .. import weights as _weights
from ._base import HipNet


class DISK(HipNet):
    ARCH = _weights.ARCH_DISK

    def __init__(self, desc_dim=128, setup=None, kernel_size=5):
        if desc_dim != 128 or kernel_size != 5 or setup is not None:
            raise NotImplementedError("this build carries kernels for the default DISK (desc_dim=128, 5x5, thin U-Net)")
        super().__init__()

    def load_state_dict(self, state_dict, strict=True):
        if "extractor" in state_dict:      # model_interface.py:78 passes checkpoint['extractor']; accept both
            state_dict = state_dict["extractor"]
        self.load_packed(_weights.pack(_weights.tensors_disk(state_dict), _weights.ARCH_DISK))
        return "<All keys matched successfully>"


def disk_random(seed=0) -> "DISK":
    """DISK with seeded random weights (the reference checkpoint disk.pth is not in its tree)."""
    net = DISK()
    net.load_state_dict(_weights.random_disk_state_dict(seed))
    return net
