"""keypoint_bench_amd.shim.install() against the REAL reference modules (build container only: /root/reference is not on the
GPU box).  cv2 / skimage / torchvision / utils.export are absent from this image; they are supplied blank, exactly as
tests/golden/make_golden*.py do (none is touched on the paths exercised here).

What is checked: every listed name is swapped and re-bound in the task modules that imported it earlier; the
reference's ORIGINAL callables (taken from the checkout, not from oracle/) are kept and reached for inputs outside the
library's contract -- host tensors and negative score maps (BASELINE configs[0]: models/Harris.py returns signed
cv2.cornerHarris responses); uninstall() restores everything."""
import os
import sys
import types
import warnings

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (build container only)")

_STUBBED = ["cv2", "skimage", "skimage.feature", "torchvision", "torchvision.models", "torchvision.models.resnet", "openvino", "tensorrt"]


@pytest.fixture()
def reference_on_path():
    import torch.nn as nn
    saved = {k: sys.modules.get(k) for k in _STUBBED}
    before = set(sys.modules)
    sk, skf = types.ModuleType("skimage"), types.ModuleType("skimage.feature")
    skf.match_descriptors = lambda *a, **k: (_ for _ in ()).throw(AssertionError("not on this path"))
    sk.feature = skf
    tv, tvm, tvr = types.ModuleType("torchvision"), types.ModuleType("torchvision.models"), types.ModuleType("torchvision.models.resnet")
    tvr.conv3x3 = lambda i, o, stride=1, groups=1, dilation=1: nn.Conv2d(i, o, 3, stride, dilation, dilation, groups, False)
    tvr.conv1x1 = lambda i, o, stride=1: nn.Conv2d(i, o, 1, stride, bias=False)
    tv.models, tvm.resnet = tvm, tvr
    sys.modules.update({"cv2": types.ModuleType("cv2"), "skimage": sk, "skimage.feature": skf, "torchvision": tv, "torchvision.models": tvm,
                        "torchvision.models.resnet": tvr, "openvino": types.ModuleType("openvino"), "tensorrt": types.ModuleType("tensorrt")})
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from keypoint_bench_amd import shim
    try:
        yield shim
    finally:
        shim.uninstall()
        sys.path.remove(REF)
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("utils", "models", "tasks"):
                del sys.modules[k]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_install_swaps_rebinds_and_keeps_the_originals(reference_on_path):
    shim = reference_on_path
    import tasks.repeatability as ref_rep            # imported BEFORE install: holds `from utils.extracter import detection`
    import tasks.FundamentalMatrix as ref_fm
    import utils.extracter as ref_ex
    import utils.matcher as ref_ma
    import utils.projection as ref_pj
    orig = dict(detection=ref_ex.detection, fast_nms=ref_ex.fast_nms, bfm=ref_ma.brute_force_matcher, wh=ref_pj.warp_homography,
                vkp=ref_rep.val_key_points, fm=ref_fm.fundamental_matrix)
    assert ref_rep.detection is orig["detection"] and ref_fm.brute_force_matcher is orig["bfm"]

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        swapped = shim.install()
    assert not [x for x in w if "not swapped" in str(x.message)], [str(x.message) for x in w]
    want = {"utils.extracter.detection", "utils.extracter.fast_nms", "utils.matcher.brute_force_matcher", "utils.matcher.optical_flow_tensor",
            "utils.matcher.OpticalFlow", "utils.projection.warp_homography", "utils.projection.warp_se3", "tasks.repeatability.val_key_points",
            "tasks.FundamentalMatrix.fundamental_matrix", "tasks.FundamentalMatrix.fundamental_matrix_ransac", "utils.mvg.fundamental_estimate",
            "tasks.visual_odometer.visual_odometry", "models.ALike.ALNet", "models.SuperPoint.SuperPointNet", "models.XFeat.XFeatModel",
            "models.disk.DISK", "models.lightglue.LightGlue"}
    assert set(swapped) == want
    assert shim.installed()["skipped"] == {}

    # the swapped names keep the reference's own callable, and it IS the checkout's (module + file), not the oracle's
    assert ref_ex.detection is not orig["detection"] and ref_ex.detection.reference is orig["detection"]
    assert orig["detection"].__module__ == "utils.extracter" and orig["detection"].__code__.co_filename.startswith(REF)
    assert ref_ma.brute_force_matcher.reference is orig["bfm"] and ref_pj.warp_homography.reference is orig["wh"]
    assert ref_fm.fundamental_matrix.reference is orig["fm"]
    # names imported earlier by task modules were re-bound
    assert ref_rep.detection is ref_ex.detection
    assert ref_fm.detection is ref_ex.detection and ref_fm.brute_force_matcher is ref_ma.brute_force_matcher
    assert ref_fm.optical_flow_tensor is ref_ma.optical_flow_tensor
    assert ref_rep.val_key_points.reference is orig["vkp"]
    bound = shim.installed()["rebound"]
    assert "tasks.repeatability.detection" in bound and "tasks.FundamentalMatrix.brute_force_matcher" in bound
    import utils.mvg as ref_mvg
    assert "tasks.FundamentalMatrix.fundamental_estimate" in bound and ref_fm.fundamental_estimate is ref_mvg.fundamental_estimate
    # host tensors are outside the contract: the reference's own utils/mvg.py:13-15 answers (fewer than 8 points: no cv2 call)
    few = torch.arange(10, dtype=torch.float32).reshape(5, 2)
    F, a, b = ref_fm.fundamental_estimate(few, few + 1)
    assert F is None and a.shape == (5, 2) and ref_mvg.fundamental_estimate.fallbacks == 1
    # `warp` of the reference dispatches through its module globals: it now reaches the swapped warp_homography
    assert ref_pj.warp.__globals__["warp_homography"] is ref_pj.warp_homography

    assert shim.install() == swapped            # idempotent
    shim.uninstall()
    assert ref_ex.detection is orig["detection"] and ref_rep.detection is orig["detection"]
    assert ref_fm.brute_force_matcher is orig["bfm"] and ref_rep.val_key_points is orig["vkp"]


def test_out_of_contract_inputs_reach_the_reference(reference_on_path):
    """configs[0] plumbing: a host tensor with negative responses (what models/Harris.py hands to detection) runs the
    reference's own extracter.py:193-221, not a kernel and not the oracle."""
    shim = reference_on_path
    import utils.extracter as ref_ex
    import tasks.repeatability as ref_rep
    import utils.projection as ref_pj
    orig_detection = ref_ex.detection
    shim.install()
    rng = np.random.default_rng(5)
    harris = torch.from_numpy((rng.normal(size=(1, 1, 64, 96)) * 1e-3).astype(np.float32))      # signed, like cv2.cornerHarris
    prm = dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=50, min_score=0.0)
    want = orig_detection(harris.clone(), prm)
    assert ref_ex.detection.fallbacks == 0
    got = ref_ex.detection(harris.clone(), prm)
    assert ref_ex.detection.fallbacks == 1
    assert torch.equal(got, want) and got.shape[0] == 50
    seen = ref_ex.fast_nms.fallbacks          # the reference's detection reaches fast_nms through its module globals: counted too
    assert seen == 2
    nms = ref_ex.fast_nms(harris.clone(), 4)
    assert torch.equal(nms, ref_ex.fast_nms.reference(harris.clone(), 4)) and ref_ex.fast_nms.fallbacks == seen + 1

    # the repeatability task end to end on host tensors (detector-only model: desc map None), through the re-bound names
    k0 = want
    H = torch.tensor([[1.0, 0.01, 0.002], [-0.01, 1.0, 0.001], [0.0, 0.0, 1.0]])
    w01 = {"mode": "homo", "width": torch.tensor(96), "height": torch.tensor(64), "homography_matrix": H}
    w10 = {"mode": "homo", "width": torch.tensor(96), "height": torch.tensor(64), "homography_matrix": torch.inverse(H)}
    res = ref_rep.val_key_points(k0, k0.clone(), w01, w10, th=3)
    assert ref_rep.val_key_points.fallbacks == 1 and ref_pj.warp_homography.fallbacks == 2      # the reference's own warp called it twice
    ref = ref_rep.val_key_points.reference(k0, k0.clone(), w01, w10, th=3)
    assert res["num_feat"] == ref["num_feat"] and float(res["repeatability"]) == float(ref["repeatability"])


def test_model_constructors_fall_back(reference_on_path):
    shim = reference_on_path
    import models.ALike as ref_alike
    ref_cls = ref_alike.ALNet
    shim.install()
    assert ref_alike.ALNet.reference is ref_cls
    # a channel plan the library has no kernels for -> the reference's own nn.Module
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        big = ref_alike.ALNet({"c1": 16, "c2": 32, "c3": 64, "c4": 128, "dim": 128})
    assert isinstance(big, ref_cls)
    # ALIKE-t: HIP net for device images, the reference module (same state_dict) for host images
    net = ref_alike.ALNet({"c1": 8, "c2": 16, "c3": 32, "c4": 64, "dim": 64})
    sd = torch.load(os.path.join(REF, "weights", "alike-t.pth"), map_location="cpu")
    net.load_state_dict(sd)
    net.eval()
    img = torch.rand(1, 3, 64, 96)
    score, desc = net(img)
    plain = ref_cls({"c1": 8, "c2": 16, "c3": 32, "c4": 64, "dim": 64})
    plain.load_state_dict(sd)
    with torch.no_grad():
        s2, d2 = plain.eval()(img)
    assert torch.equal(score, s2) and torch.equal(desc, d2)
