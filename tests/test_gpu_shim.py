"""install() on the GPU box.  The reference checkout does not travel, so a STAND-IN checkout is written into a temp
directory by this test (tiny torch functions of the test's own making, in modules named like the reference's): what is
exercised is the dispatch itself -- a signed score map on the device raises the library's device-side KPB_E_NEGATIVE flag
and the call lands in the checkout's original; a non-negative map stays on the HIP path.  (The real reference modules are
put under install() by tests/test_shim_reference.py in the build container.)"""
import sys
import textwrap
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

EXTRACTER = '''
import torch
import torch.nn.functional as F
CALLS = []

def fast_nms(image_probs, nms_dist=4, max_iter=-1, min_value=0.0):
    CALLS.append("fast_nms")
    mx = F.max_pool2d(image_probs, 2 * nms_dist + 1, 1, nms_dist)
    return torch.where(image_probs == mx, image_probs, torch.zeros_like(image_probs))

def detection(score_map, params=None):
    CALLS.append("detection")
    m = fast_nms(score_map, params["nms_dist"])[0, 0]
    idx = torch.nonzero(m > params["threshold"])
    s = m[idx[:, 0], idx[:, 1]]
    H, W = m.shape
    pts = torch.stack([(idx[:, 1] + 0.5) / W, (idx[:, 0] + 0.5) / H, s], 1)
    if pts.shape[0] > params["top_k"]:
        pts = pts[torch.argsort(s, descending=True)[:params["top_k"]]]
    return pts
'''
REPEAT = '''
from utils.extracter import detection
CALLS = []

def val_key_points(kps0, kps1, warp01, warp10, th=3):
    CALLS.append("val_key_points")
    return {"num_feat": min(len(kps0), len(kps1)), "repeatability": 0.0, "mean_error": 0.0, "errors": None}

def repeatability(idx, img_0, score_map_0, img_1, score_map_1, warp01, warp10, params):
    kps0 = detection(score_map_0, params["extractor_params"])
    kps1 = detection(score_map_1, params["extractor_params"])
    return val_key_points(kps0, kps1, warp01, warp10, th=params["repeatability_params"]["th"])
'''


@pytest.fixture()
def standin_checkout(tmp_path):
    for pkg, files in (("utils", {"extracter.py": EXTRACTER}), ("tasks", {"repeatability.py": REPEAT})):
        d = tmp_path / pkg
        d.mkdir()
        (d / "__init__.py").write_text("")
        for name, src in files.items():
            (d / name).write_text(textwrap.dedent(src))
    before = set(sys.modules)
    sys.path.insert(0, str(tmp_path))
    from keypoint_bench_amd import shim
    try:
        yield shim
    finally:
        shim.uninstall()
        sys.path.remove(str(tmp_path))
        for k in set(sys.modules) - before:
            if k.split(".")[0] in ("utils", "tasks", "models"):
                del sys.modules[k]


def test_signed_map_of_a_detector_only_model_reaches_the_original(standin_checkout):
    shim = standin_checkout
    import tasks.repeatability as rep          # imported before install(): its `detection` must be re-bound
    import utils.extracter as ex
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        swapped = shim.install()
    assert set(swapped) == {"utils.extracter.detection", "utils.extracter.fast_nms", "tasks.repeatability.val_key_points"}
    assert rep.detection is ex.detection and "tasks.repeatability.detection" in shim.installed()["rebound"]
    prm = {"extractor_params": dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=100, min_score=0.0), "repeatability_params": {"th": 3}}
    rng = np.random.default_rng(0)
    harris = torch.from_numpy((rng.normal(size=(1, 1, 96, 128)) * 1e-3).astype(np.float32)).to(DEV)    # models/Harris.py:13-22: signed, on x.device
    w = dict(mode="homo", homography_matrix=torch.eye(3, device=DEV), width=128, height=96)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        res = rep.repeatability(0, None, harris, None, harris.clone(), w, w, prm)      # (score_map, None) model: must not raise
    assert any("negative scores" in str(c.message) for c in caught)
    assert ex.CALLS.count("detection") == 2 and ex.detection.fallbacks == 2            # both maps went to the checkout's original
    assert res["num_feat"] > 0
    # keypoints came back on the device from the original -> val_key_points runs on the HIP kernels
    assert rep.CALLS == [] and rep.val_key_points.fallbacks == 0

    # a non-negative map of the same shape never leaves the library
    ex.CALLS.clear()
    good = harris.abs()
    k = ex.detection(good, prm["extractor_params"])
    assert ex.CALLS == [] and ex.detection.fallbacks == 2 and k.shape == (100, 3)
    # host tensors go to the original without touching the GPU path
    k = ex.detection(good.cpu(), prm["extractor_params"])
    assert ex.CALLS.count("detection") == 1 and not k.is_cuda
