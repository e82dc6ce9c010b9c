"""PairRunner on the device: the batched path (PairPipeline under the runner) must give the rows of the single-pair
drop-in path bit for bit; dataset items are numpy dicts as datasets/hpatches.py:74-83 returns them; sequence datasets
follow model_interface.py:217-228 and their contiguous chunks with one-frame overlap reproduce the single-rank rows."""
import numpy as np
import pytest
import torch

from keypoint_bench_amd import runner, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EP = dict(nms_dist=4, threshold=0.0, border_dist=8, top_k=300, min_score=0.0)
BF = dict(metric="euclidean", max_distance=5, cross_check=True)
H01 = np.array([[1, 0, -3], [0, 1, -2], [0, 0, 1]], np.float32)


def params(task):
    return {"model_type": "Alike", "task_type": task, "Alike_params": dict(c1=8, c2=16, c3=32, c4=64, dim=64), "extractor_params": EP,
            "matcher_params": {"type": "brute_force", "brute_force_params": BF}, "repeatability_params": {"th": 3},
            "FundamentalMatrix_params": {"th": 3.0}, "MHA_params": {"th": [3, 5, 7]}}


def pair_dataset(n, shapes=((96, 128),)):
    ds = []
    for i in range(n):
        h, w = shapes[i % len(shapes)]
        v0, v1 = synthetic.image_pair(100 + i, h, w)        # numpy [3,H,W] float32: what the reference datasets hand out
        hm = H01 + (0.001 * (i % 3)) * np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 0]], np.float32)
        ds.append({"image0": v0, "image1": v1, "dataset": "HPatches",
                   "warp01_params": dict(mode="homo", homography_matrix=hm, width=np.int64(w), height=np.int64(h), resize=np.int64(w)),
                   "warp10_params": dict(mode="homo", homography_matrix=np.linalg.inv(hm).astype(np.float32), width=w, height=h)})
    return ds


@pytest.mark.parametrize("task", ["match_stats", "repeatability"])
@pytest.mark.parametrize("dense", [False, True])
def test_batched_rows_equal_single_pair_rows(task, dense):
    ds = pair_dataset(11, shapes=((96, 128), (96, 128), (96, 128), (64, 96), (96, 128), (100, 130)))     # ragged: 100x130 crops to 96x128
    single = runner.PairRunner(params(task), device=DEV, batch=1, dense_descriptors=dense)
    _, rows1 = single.run(ds)
    assert single.batched_pairs == 0
    batched = runner.PairRunner(params(task), device=DEV, batch=4, dense_descriptors=dense)
    agg, rowsb = batched.run(ds)
    assert batched.batched_pairs == 11
    assert np.array_equal(rows1.view(np.uint64), rowsb.view(np.uint64)), (rows1, rowsb)
    assert rowsb[:, 0].min() > 50                                         # the rows carry real keypoint counts
    assert np.isfinite(agg["mean_matches" if task == "match_stats" else "repeatability"])


def test_numpy_items_and_user_task_fn_take_the_single_pair_path():
    ds = pair_dataset(3)
    seen = []

    def task(idx, img0, s0, d0, img1, s1, d1, w01, w10, prm):
        seen.append((idx, tuple(s0.shape), w01["mode"]))
        return [float(s0.shape[2]), float(s0.shape[3])]

    r = runner.PairRunner(params("match_stats"), task_fn=task, device=DEV, batch=8)
    _, rows = r.run(ds, task_type="match_stats")
    assert [s[0] for s in seen] == [0, 1, 2] and seen[0][1] == (1, 1, 96, 128) and r.batched_pairs == 0
    assert rows[:, :2].tolist() == [[96.0, 128.0]] * 3


def sequence_dataset(n, h=96, w=128):
    rng = np.random.default_rng(9)
    canvas, _ = synthetic.image_pair(300, h + 40, w + 40)
    ds = []
    for i in range(n):
        F = rng.normal(size=(3, 3)).astype(np.float32)
        ds.append({"image0": canvas[:, i:i + h, 2 * i:2 * i + w].copy(), "fundamental": torch.from_numpy(F), "dataset": "TartanAir"})
    return ds


@pytest.mark.parametrize("dense", [False, True])
def test_sequence_rows_batched_equal_single_and_chunks_reproduce_them(dense):
    ds = sequence_dataset(9)
    prm = params("FundamentalMatrix")
    single = runner.PairRunner(prm, device=DEV, batch=1, dense_descriptors=dense)
    agg1, rows1 = single.run(ds)
    assert single.batched_pairs == 0 and rows1.shape[0] == 9
    # frame 0 pairs with itself (last_batch starts as the batch, model_interface.py:218-219): every keypoint matches itself
    assert agg1["fundamental_num"] >= 0
    batched = runner.PairRunner(prm, device=DEV, batch=4, dense_descriptors=dense)
    _, rowsb = batched.run(ds)
    assert batched.batched_pairs == 9
    assert np.array_equal(rows1.view(np.uint64), rowsb.view(np.uint64)), (rows1, rowsb)
    # contiguous chunks with one-frame overlap: what ranks 0..2 of a 3-rank run would each compute
    for world in (2, 3):
        got = np.zeros_like(rows1[:, :3])
        for rank in range(world):
            idx = runner.shard_chunk(len(ds), rank, world)
            for mode_batch in (1, 4):
                r = runner.PairRunner(prm, device=DEV, batch=mode_batch, dense_descriptors=dense)
                rows = np.asarray(r._run_sequence(ds, idx), np.float64)           # rows are float64 end to end (runner.pack_rows)
                got[idx] = rows
                assert np.array_equal(rows.view(np.uint64), np.ascontiguousarray(rows1[idx, :3]).view(np.uint64)), (world, rank, mode_batch)
        assert np.array_equal(got, rows1[:, :3])


# ------------------------------------------------------------------------------------------------ visual odometer
def vo_dataset(n, h=96, w=128):
    """Frames sliding over one canvas (a camera translating sideways over a fronto-parallel plane), with the fields
    datasets/euroc.py:184-196 hands out; poses as plain 7-vectors (pypose is absent here)."""
    canvas, _ = synthetic.image_pair(321, h + 40, w + 60)
    ds = []
    for i in range(n):
        pos = [0.05 * i, 0.02 * i, 0.0, 0, 0, 0, 1]
        last = [0.05 * max(i - 1, 0), 0.02 * max(i - 1, 0), 0.0, 0, 0, 0, 1]
        ds.append({"image0": canvas[:, 2 * i:2 * i + h, 5 * i:5 * i + w].copy(), "dataset": "Euroc", "fx": 120.0, "fy": 120.0, "cx": 63.5, "cy": 47.5,
                   "ground_truth": np.asarray(pos, np.float32), "last_ground_truth": np.asarray(last, np.float32)})
    return ds


def test_visual_odometer_rows_batched_equal_single_and_chunks_reproduce_them():
    from oracle import geometry_ref as g
    import oracle
    ds = vo_dataset(7)
    prm = params("visual_odometer")
    single = runner.PairRunner(prm, device=DEV, batch=1)
    agg1, rows1 = single.run(ds)
    assert single.batched_pairs == 0 and rows1.shape[0] == 7
    batched = runner.PairRunner(prm, device=DEV, batch=4)
    aggb, rowsb = batched.run(ds)
    assert batched.batched_pairs == 7
    assert np.array_equal(rows1.view(np.uint64), rowsb.view(np.uint64)), (rows1, rowsb)
    assert np.array_equal(agg1["t_est"], aggb["t_est"]) and agg1["r_est"].shape == (8, 3, 3) and agg1["t_est"].shape == (8, 3, 1)
    # rotations are rotations; the camera slides sideways: the composed path has moved, mostly in the image plane
    for R in agg1["r_est"]:
        assert abs(np.linalg.det(R) - 1) < 1e-5 and np.abs(R @ R.T - np.eye(3)).max() < 1e-5
    assert np.array_equal(agg1["t_est"][1], np.zeros((3, 1)))          # frame 0 pairs with itself: zero step length, no update
    end = agg1["t_est"][-1].ravel()
    assert np.linalg.norm(end) > 0.1 and abs(end[2]) < 0.5 * np.linalg.norm(end[:2]) + 0.05, end
    # what ranks of a 2- and 3-rank run would compute on their chunks (one-frame overlap), batched and not
    for world in (2, 3):
        for rank in range(world):
            idx = runner.shard_chunk(len(ds), rank, world)
            for mode_batch in (1, 4):
                r = runner.PairRunner(prm, device=DEV, batch=mode_batch)
                rows = np.asarray(r._run_sequence(ds, idx, "visual_odometer"), np.float64)
                assert np.array_equal(rows.view(np.uint64), np.ascontiguousarray(rows1[idx, :13]).view(np.uint64)), (world, rank, mode_batch)
    # frame pair (2, 3) against the same steps taken with the oracle's pieces
    net = runner.build_model(prm)
    t = lambda a: torch.from_numpy(a)[None].to(DEV)
    s0, d0 = net(t(ds[2]["image0"]))
    s1, d1 = net(t(ds[3]["image0"]))
    k0, _ = oracle.detection(s0[0, 0].cpu().numpy(), EP)
    k1, _ = oracle.detection(s1[0, 0].cpu().numpy(), EP)
    m0, m1 = oracle.brute_force_matcher(k0, k1, d0[0].cpu().numpy(), d1[0].cpu().numpy(), BF)
    px = np.array([127, 95], np.float32)
    c, f = np.array([63.5, 47.5]), 120.0
    x0, x1 = ((m0[:, :2] * px).astype(np.float64) - c) / f, ((m1[:, :2] * px).astype(np.float64) - c) / f
    E, mask, info = g.find_essential_ransac(x0, x1, seed=0, threshold=1.0 / f, prob=0.999)
    n, R, tt, _ = g.recover_pose(E, x0, x1, np.ones(len(x0), np.uint8), dist=50.0)
    # Same sampler and solver on both sides; since r03 the keep / niters rule is OpenCV's sequential one, so a hypothesis whose
    # inlier count differs by one between the device's and numpy's float64 summation order (a Sampson error at the threshold) can
    # change which model is kept.  The two poses are then neighbouring minimal models of the same motion: compared as poses.
    Rg, tg = rows1[3, :9].reshape(3, 3), rows1[3, 9:12]
    if np.abs(Rg - R).max() > 2e-5:
        ang = np.degrees(np.arccos(np.clip((np.trace(Rg.T @ R) - 1) / 2, -1, 1)))
        dirs = np.degrees(np.arccos(np.clip(abs(float(tg @ tt)) / (np.linalg.norm(tg) * np.linalg.norm(tt)), -1, 1)))
        assert ang < 0.5 and dirs < 5.0, (ang, dirs)
    else:
        np.testing.assert_allclose(tg, tt, atol=2e-5)
    assert abs(rows1[3, 12] - np.hypot(0.05, 0.02)) < 1e-6


def test_host_items_are_staged_through_pinned_buffers_and_decoded_uint8_images_equal_their_fp32_form():
    """numpy items take the HostStager path (pinned ring, one PCIe copy per batch, copy under compute); a
    decoded uint8 [H,W,3] image gives the rows of its ToTensor form (megadepth.py:312-313) handed over as fp32 CHW."""
    ds = pair_dataset(10, shapes=((96, 128), (96, 128), (64, 96)))
    u8, f32, dev = [], [], []
    for it in ds:
        a = np.ascontiguousarray((it["image0"].transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8))
        b = np.ascontiguousarray((it["image1"].transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8))
        u8.append(dict(it, image0=a, image1=b))
        fa = np.ascontiguousarray((a.astype(np.float32) / np.float32(255)).transpose(2, 0, 1))
        fb = np.ascontiguousarray((b.astype(np.float32) / np.float32(255)).transpose(2, 0, 1))
        f32.append(dict(it, image0=fa, image1=fb))
        dev.append(dict(it, image0=torch.from_numpy(fa).to(DEV), image1=torch.from_numpy(fb).to(DEV)))
    rows = {}
    for name, d in (("u8", u8), ("f32", f32), ("dev", dev)):
        r = runner.PairRunner(params("repeatability"), device=DEV, batch=4)
        _, rows[name] = r.run(d)
        assert r.batched_pairs == 10
        assert (r.staged_batches == 0) == (name == "dev"), (name, r.staged_batches)
    assert np.array_equal(rows["f32"].view(np.uint64), rows["dev"].view(np.uint64))
    assert np.array_equal(rows["u8"].view(np.uint64), rows["f32"].view(np.uint64))
    single = runner.PairRunner(params("repeatability"), device=DEV, batch=1)
    _, rows1 = single.run(u8)                     # the single-pair path takes decoded images too
    assert np.array_equal(rows1.view(np.uint64), rows["u8"].view(np.uint64))


def test_a_failing_dataset_item_surfaces_from_the_staging_thread():
    """An exception raised while the producer thread reads the dataset reaches the caller (no hang, no lost slot), and the
    runner works again afterwards."""
    good = pair_dataset(6)

    class Broken(list):
        def __getitem__(self, i):
            if i == 4:
                raise OSError("unreadable image")
            return list.__getitem__(self, i)

    r = runner.PairRunner(params("match_stats"), device=DEV, batch=2)
    with pytest.raises(OSError, match="unreadable image"):
        r.run(Broken(good))
    _, rows = r.run(good)
    assert rows.shape[0] == 6 and r.staged_batches > 0


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fail_at", [0, 2])
def test_a_failing_batched_task_raises_instead_of_hanging(monkeypatch, fail_at):
    """A batched task that throws (fundamental_ransac_batch, relative_motion_batch and the kernels do, on purpose) on a
    staged batch -- the last one included, when no further slot is outstanding -- reaches the caller; the staging ring's slot
    accounting survives it and the same runner works afterwards."""
    good = pair_dataset(6)
    calls = {"n": 0}
    fn, match, covis = runner.BATCHED_TASKS["match_stats"]

    def flaky(pipe, items, params, indices=None):
        calls["n"] += 1
        if calls["n"] - 1 == fail_at:
            raise RuntimeError("task failed on batch %d" % fail_at)
        return fn(pipe, items, params, indices)

    monkeypatch.setitem(runner.BATCHED_TASKS, "match_stats", (flaky, match, covis))
    r = runner.PairRunner(params("match_stats"), device=DEV, batch=2)        # three staged batches: fail_at 2 = the last
    with pytest.raises(RuntimeError, match="task failed"):
        r.run(good)
    st = r._stager
    assert st.free.qsize() == st.slots and st.order.qsize() == 0            # every slot is back
    _, rows = r.run(good)
    assert rows.shape[0] == 6 and np.isfinite(rows[:, :3]).all()


def test_encoded_image_files_go_through_the_decode_pool_and_give_the_rows_of_their_decoded_arrays(tmp_path):
    """SURVEY 8(f)2, decode stage: PNG / JPEG files decoded by PIL on the prefetch pool (datasets.ImagePairFiles, as
    megadepth.py:149-152), staged as uint8, transformed on the device -- the rows of handing the decoded arrays over directly."""
    import io
    from PIL import Image
    from keypoint_bench_amd import datasets
    ds = pair_dataset(7, shapes=((96, 128), (96, 128), (64, 96)))
    recs, arrays = [], []
    for j, it in enumerate(ds):
        views = []
        for v, name in ((it["image0"], "a"), (it["image1"], "b")):
            u8 = np.ascontiguousarray((v.transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8))
            fmt = "PNG" if j % 2 == 0 else "JPEG"
            path = tmp_path / ("%d%s.%s" % (j, name, fmt.lower()))
            Image.fromarray(u8).save(path, format=fmt, quality=92)
            views.append((str(path), datasets.decode_rgb(str(path))))
        recs.append(dict(it, image0=views[0][0], image1=views[1][0] if j != 3 else open(views[1][0], "rb").read()))   # one item as bytes
        arrays.append(dict(it, image0=views[0][1], image1=views[1][1]))
    r_files = runner.PairRunner(params("repeatability"), device=DEV, batch=4)
    r_files.decode_workers = 3
    _, rows_files = r_files.run(datasets.ImagePairFiles(recs))
    r_arr = runner.PairRunner(params("repeatability"), device=DEV, batch=4)
    _, rows_arr = r_arr.run(arrays)
    assert r_files.batched_pairs == 7 and r_files.staged_batches > 0
    assert np.array_equal(rows_files.view(np.uint64), rows_arr.view(np.uint64))
    assert rows_files[:, 0].min() > 20


@pytest.mark.parametrize("task", ["repeatability", "MHA"])
def test_hpatches_ppm_files_are_read_straight_into_the_staging_ring_and_give_the_rows_of_their_arrays(tmp_path, task):
    """HPatches is `.ppm` (datasets/hpatches.py:36, cv2.imread + BGR2RGB at 47-56): a P6 raster is the decoded image, so
    datasets.ImagePairFiles hands such files over as RawImages and HostStager.fill reads each into its pinned row.  Rows must be
    those of the decoded arrays bit for bit -- equal shapes, a shape whose x32 crop cuts rows and columns (row-wise reads), a gray
    .pgm view, a pair of unequal views (single-pair path, read by as_image) and a .ppm handed over as bytes."""
    from PIL import Image
    from keypoint_bench_amd import datasets
    ds = pair_dataset(9, shapes=((96, 128), (96, 128), (96, 128), (107, 141), (96, 128)))
    recs, arrays = [], []
    for j, it in enumerate(ds):
        views = []
        for v, name in ((it["image0"], "a"), (it["image1"], "b")):
            u8 = np.ascontiguousarray((v.transpose(1, 2, 0) * 255.0 + 0.5).astype(np.uint8))
            if j == 2:                                   # a gray pair: P5 files, three equal channels after decoding
                u8 = np.repeat(u8[..., 1:2], 3, axis=2)
                path = tmp_path / ("%d%s.pgm" % (j, name))
                Image.fromarray(u8[..., 0]).save(path)
            else:
                path = tmp_path / ("%d%s.ppm" % (j, name))
                Image.fromarray(u8).save(path)
            assert np.array_equal(datasets.decode_rgb(str(path)), u8)
            views.append((str(path), u8))
        recs.append(dict(it, image0=views[0][0], image1=views[1][0] if j != 1 else open(views[1][0], "rb").read()))
        arrays.append(dict(it, image0=views[0][1], image1=views[1][1]))
    files = datasets.ImagePairFiles(recs)
    assert isinstance(files[0]["image0"], datasets.RawImage) and isinstance(files[1]["image1"], np.ndarray)
    r_files = runner.PairRunner(params(task), device=DEV, batch=4)
    r_files.decode_workers = 3
    _, rows_files = r_files.run(files)
    r_arr = runner.PairRunner(params(task), device=DEV, batch=4)
    _, rows_arr = r_arr.run(arrays)
    assert r_files.batched_pairs == r_arr.batched_pairs > 0 and r_files.staged_batches > 0
    assert np.array_equal(rows_files.view(np.uint64), rows_arr.view(np.uint64))
    single = runner.PairRunner(params(task), device=DEV, batch=1)        # every pair through test_step: as_image reads the RawImages
    _, rows_single = single.run(files)
    assert np.array_equal(rows_single.view(np.uint64), rows_arr.view(np.uint64))
