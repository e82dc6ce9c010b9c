"""install(): put this package's kernels under an importable reference checkout (its root on sys.path) so that
``python3 main.py -c config/config_MHA.yaml test`` and every task in tasks/ run unchanged.

Every swapped name keeps the reference's ORIGINAL callable (taken from the reference's own module at install time,
never from oracle/) and routes to it whatever lies outside the library's contract:

  * a tensor that is not on a HIP device (``accelerator: cpu`` runs, BASELINE configs[0]);
  * a score map with negative values -- models/Harris.py:13-22 returns cv2.cornerHarris responses, and the
    reference's NMS lets suppressed zeros win against negative neighbours.  The library finds this with a device-side
    flag raised by the NMS kernels themselves (KPB_E_NEGATIVE), not with a host scan of the map;
  * arguments the kernels do not carry (fast_nms max_iter / min_value, a non-euclidean metric, nms_dist > 16, more matches
    than a RANSAC kernel holds, ALIKE channel plans other than -t, DISK variants): KPB_E_UNSUPPORTED / NotImplementedError.

Nothing else falls through: KPB_E_INVALID (a malformed call) and every other library error RAISE -- an in-contract call that
starts failing must not turn into a silent thousandfold slowdown on the reference's CPU code.

So the Harris / repeatability configuration keeps running on the reference's code while ALIKE / SuperPoint / XFeat /
DISK runs go through libkpb.so.  ``uninstall()`` restores every original.
"""
import importlib
import sys
import warnings

import torch

from ._lib import KpbError

KPB_E_INVALID, KPB_E_NEGATIVE, KPB_E_UNSUPPORTED = -1, -4, -7
FALLBACK_CODES = (KPB_E_NEGATIVE, KPB_E_UNSUPPORTED)       # the whitelist: documented out-of-contract reasons only

_state = {"originals": {}, "swapped": [], "skipped": {}, "rebound": []}
_ROOTS = ("utils", "models", "tasks")


def _on_device(*tensors):
    return all(torch.is_tensor(t) and t.is_cuda for t in tensors)


def _note(name, why):
    warnings.warn("keypoint_bench_amd: %s handled by the reference's own code (%s)" % (name, why), RuntimeWarning, stacklevel=3)


def guarded(name, hip_fn, ref_fn, in_contract):
    """hip_fn where the call is inside the library's contract, ref_fn (the reference's original) elsewhere."""

    def call(*a, **k):
        if ref_fn is None:
            return hip_fn(*a, **k)
        why = in_contract(*a, **k)
        if why is not None:
            call.fallbacks += 1
            return ref_fn(*a, **k)
        try:
            return hip_fn(*a, **k)
        except KpbError as e:
            if e.code not in FALLBACK_CODES:
                raise
            why = "negative scores" if e.code == KPB_E_NEGATIVE else str(e)
        except NotImplementedError as e:
            why = str(e)
        if call.fallbacks == 0:
            _note(name, why)
        call.fallbacks += 1
        return ref_fn(*a, **k)

    call.__name__ = getattr(hip_fn, "__name__", name)
    call.__doc__ = hip_fn.__doc__
    call.hip, call.reference, call.fallbacks = hip_fn, ref_fn, 0
    return call


class DualNet:
    """What a swapped model constructor returns: the HIP net for device images, the reference's own nn.Module
    (built lazily with the same constructor arguments and the same state_dict) for host images."""

    def __init__(self, hip, make_ref):
        self.__dict__.update(_hip=hip, _make_ref=make_ref, _ref=None, _sd=None, _sd_args=((), {}))

    def load_state_dict(self, state_dict, *a, **k):
        self.__dict__.update(_sd=state_dict, _sd_args=(a, k), _ref=None)
        return self._hip.load_state_dict(state_dict, *a, **k)

    def _reference(self):
        if self._ref is None:
            ref = self._make_ref()
            if self._sd is not None:
                ref.load_state_dict(self._sd, *self._sd_args[0], **self._sd_args[1])
            self.__dict__["_ref"] = ref.eval()
        return self._ref

    def eval(self):
        self._hip.eval()
        return self

    def to(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def parameters(self):
        return iter(())

    def __call__(self, image, *a, **k):
        if _on_device(image):
            return self._hip(image, *a, **k)
        with torch.no_grad():
            return self._reference()(image, *a, **k)

    forward = __call__

    def __getattr__(self, k):
        return getattr(self._hip, k)

    def __setattr__(self, k, v):
        setattr(self._hip, k, v)


class NetFactory:
    """Stands where the reference's model class stood (``ALNet(params)``, ``DISK()`` ...)."""

    def __init__(self, name, hip_cls, ref_cls):
        self.name, self.hip, self.reference = name, hip_cls, ref_cls
        self.__name__, self.__doc__ = hip_cls.__name__, hip_cls.__doc__

    def __call__(self, *a, **k):
        try:
            hip = self.hip(*a, **k)
        except NotImplementedError as e:
            _note(self.name, str(e))
            return self.reference(*a, **k)
        return DualNet(hip, lambda: self.reference(*a, **k))


class DualMatcher:
    """LightGlue(features=, weight_path=).match(...) (model_interface.py:62-63, 80-81): HIP on device keypoints, the
    reference's class on host keypoints."""

    def __init__(self, hip, make_ref):
        self._hip, self._make_ref, self._ref = hip, make_ref, None

    def match(self, pts0, pts1, desc_map_0, desc_map_1, params=None):
        if _on_device(pts0, pts1):
            return self._hip.match(pts0, pts1, desc_map_0, desc_map_1, params)
        if self._ref is None:
            self._ref = self._make_ref()
        return self._ref.match(pts0, pts1, desc_map_0, desc_map_1, params)

    def __getattr__(self, k):
        return getattr(self.__dict__["_hip"], k)


class MatcherFactory(NetFactory):
    def __call__(self, *a, **k):
        try:
            hip = self.hip(*a, **k)
        except (NotImplementedError, ValueError) as e:
            _note(self.name, str(e))
            return self.reference(*a, **k)
        return DualMatcher(hip, lambda: self.reference(*a, **k))


def _lk_factory(hip_cls, ref_cls):
    class OpticalFlow(hip_cls):
        """utils/matcher.py:7-142 on the device; host images go to the reference's class."""
        hip, reference = hip_cls, ref_cls

        def __init__(self, params=None):
            hip_cls.__init__(self, params)
            self._params, self._ref = params, None

        def __call__(self, img1, img2, pts1, pts2, *a, **k):
            if ref_cls is None or _on_device(img1, img2, pts1):
                return hip_cls.__call__(self, img1, img2, pts1, pts2, *a, **k)
            if self._ref is None:
                self._ref = ref_cls(self._params)
            return self._ref(img1, img2, pts1, pts2, *a, **k)

    return OpticalFlow


# --------------------------------------------------------------------------------------------- contracts
def _c_detection(score_map, params=None):
    return None if _on_device(score_map) else "score map is not on a HIP device"


def _c_fast_nms(image_probs, nms_dist=4, max_iter=-1, min_value=0.0):
    if not _on_device(image_probs):
        return "map is not on a HIP device"
    return None if (max_iter == -1 and min_value == 0.0) else "max_iter / min_value"


def _c_bf(pts0, pts1, desc_map_0, desc_map_1, params=None):
    if not _on_device(pts0, pts1):
        return "keypoints are not on a HIP device"
    return None if (params or {}).get("metric", "euclidean") == "euclidean" else "metric"


def _c_first(t, *a, **k):
    return None if _on_device(t) else "points are not on a HIP device"


def _c_lk(pts0, pts1, img0, img1, params=None):
    return None if _on_device(pts0, img0, img1) else "inputs are not on a HIP device"


def _c_vkp(kps0, kps1, warp01, warp10, th=3):
    return None if _on_device(kps0, kps1) else "keypoints are not on a HIP device"


def _table():
    from .models.ALike import ALNet
    from .models.SuperPoint import SuperPointNet
    from .models.XFeat import XFeatModel
    from .models.disk import DISK
    from .models.lightglue import LightGlue
    from .tasks import repeatability as rp
    from .tasks import FundamentalMatrix as fm
    from .tasks import visual_odometer as vo
    from .utils import extracter as ex, matcher as ma, mvg, projection as pj
    fn, net = "fn", "net"
    return [
        ("utils.extracter", "detection", fn, ex.detection, _c_detection),
        ("utils.extracter", "fast_nms", fn, ex.fast_nms, _c_fast_nms),
        ("utils.matcher", "brute_force_matcher", fn, ma.brute_force_matcher, _c_bf),
        ("utils.matcher", "optical_flow_tensor", fn, ma.optical_flow_tensor, _c_lk),
        ("utils.matcher", "OpticalFlow", "lk", ma.OpticalFlow, None),
        ("utils.projection", "warp_homography", fn, pj.warp_homography, _c_first),
        ("utils.projection", "warp_se3", fn, pj.warp_se3, _c_first),
        ("tasks.repeatability", "val_key_points", fn, rp.val_key_points, _c_vkp),
        ("tasks.FundamentalMatrix", "fundamental_matrix", fn, fm.fundamental_matrix, fm.in_contract),
        ("tasks.FundamentalMatrix", "fundamental_matrix_ransac", fn, fm.fundamental_matrix_ransac, fm.ransac_in_contract),
        ("utils.mvg", "fundamental_estimate", fn, mvg.fundamental_estimate, _c_first),
        ("tasks.visual_odometer", "visual_odometry", fn, vo.visual_odometry, vo.in_contract),
        ("models.ALike", "ALNet", net, ALNet, None),
        ("models.SuperPoint", "SuperPointNet", net, SuperPointNet, None),
        ("models.XFeat", "XFeatModel", net, XFeatModel, None),
        ("models.disk", "DISK", net, DISK, None),
        ("models.lightglue", "LightGlue", "matcher", LightGlue, None),
    ]


def install():
    """Returns the list of swapped ``module.name`` strings ([] when no reference checkout is importable).
    A reference module that exists but fails to import (a missing third-party package) is reported with a warning and
    listed in ``installed()['skipped']`` -- it is not silently ignored."""
    if _state["swapped"]:
        return list(_state["swapped"])
    for modname, attr, kind, hip, contract in _table():
        try:
            mod = importlib.import_module(modname)
        except ModuleNotFoundError as e:
            if e.name in _ROOTS or e.name == modname:          # no reference checkout on sys.path (or this module absent)
                continue
            _state["skipped"][modname] = "%s: %s" % (type(e).__name__, e)
            continue
        except Exception as e:     # the reference module is there but cannot be imported: say so
            _state["skipped"][modname] = "%s: %s" % (type(e).__name__, e)
            continue
        if not hasattr(mod, attr):
            _state["skipped"][modname + "." + attr] = "name not found in the reference module"
            continue
        full = modname + "." + attr
        ref = getattr(mod, attr)
        _state["originals"][full] = ref
        if kind == "fn":
            new = guarded(full, hip, ref, contract)
        elif kind == "lk":
            new = _lk_factory(hip, ref)
        elif kind == "matcher":
            new = MatcherFactory(full, hip, ref)
        else:
            new = NetFactory(full, hip, ref)
        setattr(mod, attr, new)
        _state["swapped"].append(full)
    for m, why in _state["skipped"].items():
        warnings.warn("keypoint_bench_amd.install(): reference module %s not swapped (%s)" % (m, why), RuntimeWarning, stacklevel=2)
    _rebind()
    return list(_state["swapped"])


def _rebind():
    """Names that task / harness modules imported with ``from utils.x import y`` before install() ran."""
    new = {}
    for full in _state["swapped"]:
        modname, attr = full.rsplit(".", 1)
        new[(_state["originals"][full], attr)] = getattr(sys.modules[modname], attr)
    for name, mod in list(sys.modules.items()):
        if mod is None or not (name.startswith("tasks.") or name.startswith("utils.") or name == "models.model_interface"):
            continue
        for (orig, attr), repl in new.items():
            if getattr(mod, attr, None) is orig:
                setattr(mod, attr, repl)
                _state["rebound"].append(name + "." + attr)


def uninstall():
    for full, ref in _state["originals"].items():
        modname, attr = full.rsplit(".", 1)
        if modname in sys.modules:
            setattr(sys.modules[modname], attr, ref)
    for bound in _state["rebound"]:
        modname, attr = bound.rsplit(".", 1)
        for full, ref in _state["originals"].items():
            if full.rsplit(".", 1)[1] == attr and modname in sys.modules:
                setattr(sys.modules[modname], attr, ref)
    _state.update(originals={}, swapped=[], skipped={}, rebound=[])


def installed():
    return {k: (dict(v) if isinstance(v, dict) else list(v)) for k, v in _state.items()}
