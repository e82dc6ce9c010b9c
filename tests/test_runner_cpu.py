"""Host logic of the runner and the N>1 path on CPU: sharding, the fixed-width row all-gather over gloo
(world_size 2), and the on_test_end reductions."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import ROOT
from keypoint_bench_amd import runner


def test_shard_indices_partition():
    for n in (0, 1, 7, 580, 1500):
        for w in (1, 2, 4, 8):
            got = sorted(i for r in range(w) for i in runner.shard_indices(n, r, w))
            assert got == list(range(n))
            assert max((len(runner.shard_indices(n, r, w)) for r in range(w)), default=0) <= runner.rows_per_rank(n, w)


def test_pose_auc_matches_reference_formula():
    # the reference formula (tasks/AUC.py:86-98) evaluated by hand on a tiny case
    errs = [1.0, 3.0, 12.0, 30.0]
    aucs = runner.pose_auc(errs, [5, 10, 20])
    # recall steps at 1, 3, 12, 30 -> areas
    a5 = (0.25 * (1 - 0) / 2 + (0.25 + 0.5) / 2 * (3 - 1) + 0.5 * (5 - 3)) / 5
    a10 = (0.125 + 0.75 + 0.5 * 7) / 10
    a20 = (0.125 + 0.75 + (0.5 + 0.75) / 2 * 9 + 0.75 * 8) / 20
    np.testing.assert_allclose(aucs, [a5, a10, a20], rtol=1e-12)


def test_aggregate_mirrors_on_test_end():
    rows = np.array([[1, 0, 1], [0, 0, 1], [1, 1, 1], [0, 0, 0]], np.float32)
    out = runner.aggregate("MHA", rows, {"MHA_params": {"th": [3, 5, 7]}})
    assert out["MHA"] == [0.5, 0.25, 0.75]
    rep = np.array([[100, 0.5, 1.0], [200, 0.25, np.nan]], np.float32)
    out = runner.aggregate("repeatability", rep)
    assert out["num_feat"] == 150 and out["repeatability"] == 0.375 and out["rep_mean_err"] == 1.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_items", [1, 9, 10])
def test_two_rank_gather_over_gloo(n_items):
    import json
    import subprocess
    port = _free_port()
    worker = os.path.join(ROOT, "tests", "_gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(n_items)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-2000:]
        line = [l for l in so.splitlines() if l.startswith("RESULT ")][0]
        outs.append(json.loads(line[7:]))
    want = [[float(i), float(2 * i + 1), float((i * 7) % 5)] for i in range(n_items)]
    for o in outs:
        assert o["rows"] == want, "rank %d" % o["rank"]
        assert abs(o["agg"]["mean_matches"] - np.mean([w[2] for w in want])) < 1e-6


def test_crop32_and_config_loading(tmp_path):
    assert runner.crop32(torch.zeros(1, 3, 500, 660)).shape == (1, 3, 480, 640)
    assert runner.crop32(torch.zeros(3, 64, 96)).shape == (3, 64, 96)
    cfg = tmp_path / "c.yaml"
    cfg.write_text("test:\n  trainer: {devices: [1]}\n  data:\n    params: {batch_size: 1}\n  model:\n    params:\n"
                   "      model_type: Alike\n      task_type: MHA\n      Alike_params: {c1: 8, c2: 16, c3: 32, c4: 64, dim: 64}\n"
                   "      extractor_params: {nms_dist: 6, min_score: 0.0, top_k: 1000, threshold: 0, border_dist: 8}\n")
    p = runner.load_config(str(cfg))
    assert p["model_type"] == "Alike" and p["extractor_params"]["top_k"] == 1000 and p["data_params"]["batch_size"] == 1
