"""North-star acceptance on synthetic pairs: the repeatability task (net forward -> detection -> covisibility warp ->
val_key_points) computed end to end on the GPU must agree with the same chain computed by the oracle to +-0.001
(BASELINE.json: "repeatability/MHA within +-0.001 of the reference").  The real datasets are not available offline;
the synthetic views are related by a pure translation, which is a homography."""
import numpy as np
import pytest
import torch

import oracle
from oracle import alike_ref
from keypoint_bench_amd import synthetic, weights

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EP = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)     # config/config_MHA.yaml:68-73


@pytest.mark.parametrize("seed", [40, 41, 42, 43])
def test_repeatability_gpu_vs_oracle(seed):
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.utils.extracter import detection
    from keypoint_bench_amd.tasks.repeatability import val_key_points
    v0, v1 = synthetic.image_pair(seed)                    # view1 = the canvas 3 px right / 2 px down of view0
    H, W = v0.shape[1:]
    h01 = np.array([[1, 0, -3], [0, 1, -2], [0, 0, 1]], np.float32)
    h10 = np.linalg.inv(h01).astype(np.float32)
    # GPU chain
    net = alike_t().eval()
    t = lambda a: torch.from_numpy(a).to(DEV)
    s0, _ = net(t(v0)[None]); s1, _ = net(t(v1)[None])
    k0, k1 = detection(s0, EP), detection(s1, EP)
    w01 = dict(mode="homo", homography_matrix=t(h01), width=W, height=H)
    w10 = dict(mode="homo", homography_matrix=t(h10), width=W, height=H)
    got = val_key_points(k0, k1, w01, w10, th=3)
    # oracle chain
    tw = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    with torch.no_grad():
        o0, _ = alike_ref.alnet_forward(torch.from_numpy(v0)[None], tw)
        o1, _ = alike_ref.alnet_forward(torch.from_numpy(v1)[None], tw)
    e0, _ = oracle.detection(o0[0, 0].numpy(), EP)
    e1, _ = oracle.detection(o1[0, 0].numpy(), EP)
    exp = oracle.val_key_points(e0, e1, dict(homography_matrix=h01, width=W, height=H), dict(homography_matrix=h10, width=W, height=H), th=3)
    assert got["num_feat"] == exp["num_feat"] == 1000
    assert abs(float(got["repeatability"]) - float(exp["repeatability"])) <= 1e-3
    assert abs(float(got["mean_error"]) - float(exp["mean_error"])) <= 1e-3
    assert float(exp["repeatability"]) > 0.5            # the metric is doing something on this pair


def test_runner_repeatability_task_end_to_end():
    """PairRunner with task_type 'repeatability' over a small homography pair dataset: per-pair rows and the on_test_end
    style reduction, against the oracle chain."""
    from keypoint_bench_amd import runner
    seeds = (44, 45, 46)
    h01 = np.array([[1, 0, -3], [0, 1, -2], [0, 0, 1]], np.float32)
    h10 = np.linalg.inv(h01).astype(np.float32)
    ds = []
    for s in seeds:
        v0, v1 = synthetic.image_pair(s)
        ds.append({"image0": torch.from_numpy(v0)[None], "image1": torch.from_numpy(v1)[None], "dataset": ["synthetic"],
                   "warp01_params": dict(mode="homo", homography_matrix=torch.from_numpy(h01).to(DEV), width=640, height=480),
                   "warp10_params": dict(mode="homo", homography_matrix=torch.from_numpy(h10).to(DEV), width=640, height=480)})
    params = {"model_type": "Alike", "task_type": "repeatability", "Alike_params": dict(c1=8, c2=16, c3=32, c4=64, dim=64),
              "extractor_params": EP, "repeatability_params": {"th": 3}}
    agg, rows = runner.PairRunner(params, device=DEV).run(ds)
    assert rows.shape == (3, runner.ROW_WIDTH - 1)
    tw = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    exp = []
    for s in seeds:
        v0, v1 = synthetic.image_pair(s)
        with torch.no_grad():
            o0, _ = alike_ref.alnet_forward(torch.from_numpy(v0)[None], tw)
            o1, _ = alike_ref.alnet_forward(torch.from_numpy(v1)[None], tw)
        e0, _ = oracle.detection(o0[0, 0].numpy(), EP); e1, _ = oracle.detection(o1[0, 0].numpy(), EP)
        r = oracle.val_key_points(e0, e1, dict(homography_matrix=h01, width=640, height=480), dict(homography_matrix=h10, width=640, height=480), th=3)
        exp.append([r["num_feat"], float(r["repeatability"]), float(r["mean_error"])])
    exp = np.asarray(exp)
    np.testing.assert_allclose(rows[:, :3], exp, rtol=0, atol=1e-3)
    assert abs(agg["repeatability"] - exp[:, 1].mean()) <= 1e-3 and abs(agg["rep_mean_err"] - exp[:, 2].mean()) <= 1e-3
