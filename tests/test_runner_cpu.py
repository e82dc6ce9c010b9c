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

sys.path.insert(0, os.path.join(ROOT, "tests"))
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


def _reference_chain(rows):
    """model_interface.py:116-117, 296-297 with tasks/visual_odometer.py:84-91 written out."""
    r_est, t_est = [np.eye(3)], [np.zeros([3, 1])]
    for row in np.asarray(rows, np.float64):
        R, t, scale = row[:9].reshape(3, 3), row[9:12].reshape(3, 1), row[12]
        R_est, T_est = r_est[-1], t_est[-1]
        if scale >= 0.001:
            R_est = r_est[-1].dot(R)
            T_est = t_est[-1] + float(scale) * r_est[-1].dot(t)
        r_est.append(R_est)
        t_est.append(T_est)
    return np.stack(r_est), np.stack(t_est)


def test_visual_odometer_rows_compose_to_the_reference_chain():
    from vo_rows import vo_rows
    from keypoint_bench_amd.tasks.visual_odometer import compose, step_length
    rows = np.asarray(vo_rows(11), np.float32)
    agg = runner.aggregate("visual_odometer", rows)
    r, t = _reference_chain(rows)
    assert np.array_equal(agg["r_est"], r) and np.array_equal(agg["t_est"], t) and r.shape == (12, 3, 3) and t.shape == (12, 3, 1)
    assert np.array_equal(r[1], r[0]) and np.array_equal(t[1], t[0])               # frame 0 stands still
    r2, t2 = compose(rows)
    assert np.array_equal(r2, r) and np.array_equal(t2, t)

    class Pose:                     # pypose LieTensors answer .tensor() (datasets/euroc.py:188-189)
        def __init__(self, v): self.v = torch.tensor(v, dtype=torch.float32)
        def tensor(self): return self.v
    b = {"ground_truth": Pose([1.0, 2.0, 2.0, 0, 0, 0, 1]), "last_ground_truth": Pose([0.0, 0.0, 0.0, 0, 0, 0, 1])}
    assert step_length(b) == 3.0
    assert step_length({"ground_truth": np.array([0.0, 0.0, 1.0]), "last_ground_truth": torch.zeros(3)}) == 1.0


def test_two_rank_sequence_chunks_over_gloo():
    """A sequence task on two ranks: contiguous chunks, one all-gather of 13-wide rows, the chain composed on each rank."""
    import json
    import subprocess
    from vo_rows import vo_rows
    n = 9
    port = _free_port()
    worker = os.path.join(ROOT, "tests", "_gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(n), "vo"], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    r, t = _reference_chain(np.asarray(vo_rows(n), np.float32))
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-2000:]
        o = json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][0][7:])
        assert o["rows"] == vo_rows(n)
        np.testing.assert_array_equal(np.asarray(o["t_end"]), t[-1].ravel())
        np.testing.assert_array_equal(np.asarray(o["r_end"]), r[-1].ravel())


def test_crop32_and_config_loading(tmp_path):
    assert runner.crop32(torch.zeros(1, 3, 500, 660)).shape == (1, 3, 480, 640)
    assert runner.crop32(torch.zeros(3, 64, 96)).shape == (3, 64, 96)
    cfg = tmp_path / "c.yaml"
    cfg.write_text("test:\n  trainer: {devices: [1]}\n  data:\n    params: {batch_size: 1}\n  model:\n    params:\n"
                   "      model_type: Alike\n      task_type: MHA\n      Alike_params: {c1: 8, c2: 16, c3: 32, c4: 64, dim: 64}\n"
                   "      extractor_params: {nms_dist: 6, min_score: 0.0, top_k: 1000, threshold: 0, border_dist: 8}\n")
    p = runner.load_config(str(cfg))
    assert p["model_type"] == "Alike" and p["extractor_params"]["top_k"] == 1000 and p["data_params"]["batch_size"] == 1


def _bench_rank(rank, world, port, q):
    """One rank of bench.py's timed region and row exchange over gloo, with a stub step."""
    import importlib.util
    import time as _t
    import torch.distributed as dist
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    calls = []

    def step():                                   # rank 1 is the slow one: the reported time must be ITS time
        calls.append(1)
        _t.sleep(0.002 * (1 + 2 * rank))

    elapsed, tail, own = bench.timed_steps(step, 100, 3, dev, True, dist)
    B = 5
    rows = torch.tensor([[100.0 * rank + i, 1.0 + rank, float(i)] for i in range(B)])
    allrows = bench.exchange_rows(rows, world, True, dist)
    per_rank = bench.per_rank_figures(1e3 * own / 100, 7.0 + rank, world, True, dist, dev)      # each rank's own step time and "dominant kernel" figure
    q.put((rank, len(calls), elapsed, tail, allrows.numpy().tolist(), dist.get_world_size(), own, per_rank))
    dist.destroy_process_group()


def test_bench_rank_logic_over_gloo():
    """bench.py's contract under two ranks: W untimed + exactly K timed steps per rank, MAX-over-ranks time (and last-half time),
    one all-gather with rank r's rows at [r B, (r + 1) B) -- the 8-GPU run is the driver's, this is what can be checked here."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_bench_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, e0, t0, rows0, w0, own0, pr0), (r1, n1, e1, t1, rows1, w1, own1, pr1) = res
    assert n0 == n1 == 103 and w0 == w1 == 2
    assert e0 == e1 and t0 == t1                           # both ranks report the MAX
    assert e0 >= 100 * 0.006 and e0 < 100 * 0.006 * 3      # the slow rank's 6 ms steps, not the fast rank's 2 ms
    assert 0.4 * e0 < t0 < 0.6 * e0                        # the last 50 of 100 steps
    assert rows0 == rows1 and [r[0] for r in rows0] == [0, 1, 2, 3, 4, 100, 101, 102, 103, 104]
    # r06: the MAX hides which rank is slow -- every rank's own figures are gathered and identical on both ranks
    assert own0 < 0.6 * e0 and 0.9 * e0 < own1 <= e1       # rank 0 finished its 2 ms steps long before the barrier released it
    assert pr0 == pr1 and [p["rank"] for p in pr0] == [0, 1]
    assert abs(pr0[0]["ms_per_step"] - 10 * own0) < 0.01 and abs(pr0[1]["ms_per_step"] - 10 * own1) < 0.01      # 1e3 * own / 100 steps
    assert pr0[1]["ms_per_step"] > 2 * pr0[0]["ms_per_step"] and [p["dominant_kernel_ms"] for p in pr0] == [7.0, 8.0]


def test_generator_threads_share_the_cores():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cores = len(os.sched_getaffinity(0))
    assert bench.generator_threads(1) == max(1, min(cores, 16))
    assert bench.generator_threads(8) == max(1, min(cores // 8, 16))
    assert bench.generator_threads(8) * 8 <= max(cores, 8)
