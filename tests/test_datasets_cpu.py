"""SURVEY 8(f)2, decode stage (keypoint_bench_amd/datasets.py): PIL decoding as datasets/megadepth.py:149-152 does it, and the
order-preserving prefetcher that runs dataset[i] ahead of the staging thread."""
import io
import os
import threading
import time

import numpy as np
import pytest

from keypoint_bench_amd import datasets


def _img(seed, h=37, w=53):
    return np.random.default_rng(seed).integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def _encode(arr, fmt, mode=None, **kw):
    from PIL import Image
    im = Image.fromarray(arr)
    if mode:
        im = im.convert(mode)
    buf = io.BytesIO()
    im.save(buf, format=fmt, **kw)
    return buf.getvalue()


def test_png_is_decoded_exactly_from_path_bytes_and_file(tmp_path):
    a = _img(1)
    data = _encode(a, "PNG")
    p = tmp_path / "a.png"
    p.write_bytes(data)
    for src in (str(p), data, io.BytesIO(data)):
        out = datasets.decode_rgb(src)
        assert out.dtype == np.uint8 and out.shape == a.shape and np.array_equal(out, a)


def test_jpeg_decodes_to_what_pil_gives_and_close_to_the_source():
    from PIL import Image
    a = np.clip(np.add.outer(np.arange(40), np.arange(56))[..., None] * np.array([2, 1, 3]) % 256, 0, 255).astype(np.uint8)
    data = _encode(a, "JPEG", quality=95)
    out = datasets.decode_rgb(data)
    assert np.array_equal(out, np.array(Image.open(io.BytesIO(data)).convert("RGB")))
    assert out.shape == a.shape and np.abs(out.astype(int) - a.astype(int)).mean() < 12


def test_binary_ppm_is_its_own_raster_from_path_bytes_and_lazily(tmp_path):
    """HPatches' format (datasets/hpatches.py:36: `.ppm`, read with cv2.imread + BGR2RGB at 47-56): the P6 raster is the decoded image.
    The fast path must give exactly what PIL gives, for every header spelling netpbm allows, and hand a file over unread when asked."""
    from PIL import Image
    a = _img(11, 45, 70)
    p = tmp_path / "a.ppm"
    Image.fromarray(a).save(p)
    want = np.array(Image.open(p).convert("RGB"))
    assert np.array_equal(want, a)
    for src in (str(p), p, p.read_bytes(), io.BytesIO(p.read_bytes())):
        out = datasets.decode_rgb(src)
        assert isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and np.array_equal(out, a)
    raw = datasets.decode_rgb(str(p), lazy=True)
    assert isinstance(raw, datasets.RawImage) and raw.shape == a.shape and raw.dtype == np.uint8 and raw.ndim == 3
    assert np.array_equal(np.asarray(raw), a)
    dst = np.zeros((3,) + a.shape, np.uint8)            # a row of a staging buffer
    raw.read_into(dst[1])
    assert np.array_equal(dst[1], a) and not dst[0].any() and not dst[2].any()
    c = raw.crop(32, 64)                                # the x32 crop cuts rows AND columns: row-wise reads
    assert c.shape == (32, 64, 3) and np.array_equal(np.asarray(c), a[:32, :64])
    c = raw.crop(32, 70)                                # rows only: one read
    assert np.array_equal(np.asarray(c), a[:32])
    with pytest.raises(ValueError):
        raw.read_into(np.zeros((45, 70, 4), np.uint8))
    with pytest.raises(ValueError):
        raw.crop(46, 70)
    for hdr in (b"P6 70 45 255\n", b"P6\n# made by hand\n70 45\n#x\n255\n", b"P6\r\n70\t45\r\n255 "):
        data = hdr + a.tobytes()
        assert np.array_equal(datasets.decode_rgb(data), a)
        q = tmp_path / "h.ppm"
        q.write_bytes(data)
        assert np.array_equal(np.asarray(datasets.decode_rgb(str(q), lazy=True)), a)


def test_gray_pgm_becomes_three_equal_channels_and_odd_pnm_goes_to_pil(tmp_path):
    from PIL import Image
    g = _img(12, 33, 40)[..., 0].copy()
    p = tmp_path / "g.pgm"
    Image.fromarray(g).save(p)
    want = np.array(Image.open(p).convert("RGB"))
    assert np.array_equal(datasets.decode_rgb(str(p)), want) and np.array_equal(datasets.decode_rgb(p.read_bytes()), want)
    lazy = datasets.decode_rgb(str(p), lazy=True)
    assert isinstance(lazy, datasets.RawImage) and np.array_equal(np.asarray(lazy.crop(32, 32)), want[:32, :32])
    # not the fast path's business: ASCII PNM, 16-bit samples, a maxval the decoders rescale, a truncated raster -> PIL decides
    assert datasets.parse_pnm_header(b"P3\n2 2\n255\n" + b"0 " * 12) is None
    assert datasets.parse_pnm_header(b"P6\n2 2\n65535\n" + bytes(24)) is None
    assert datasets.parse_pnm_header(b"P6\n2 2\n100\n" + bytes(12)) is None
    assert datasets.parse_pnm_header(b"P6\n2 2\n255") is None and datasets.parse_pnm_header(b"P6\n2 2\n255x" + bytes(12)) is None
    a = _img(13, 6, 5)
    asc = b"P3\n5 6\n255\n" + b" ".join(str(int(v)).encode() for v in a.reshape(-1)) + b"\n"
    assert np.array_equal(datasets.decode_rgb(asc), a)
    q = tmp_path / "t.ppm"
    q.write_bytes(b"P6\n5 6\n255\n" + a.tobytes()[:-7])
    try:                                                # a raster that ends early is never handed over as a RawImage: PIL raises or pads
        out = datasets.decode_rgb(str(q), lazy=True)
    except Exception:
        out = None
    assert not isinstance(out, datasets.RawImage)


def test_pair_dataset_hands_ppm_files_over_unread_unless_told_otherwise(tmp_path):
    from PIL import Image
    a, b = _img(14, 64, 96), _img(15, 64, 96)
    Image.fromarray(a).save(tmp_path / "a.ppm")
    Image.fromarray(b).save(tmp_path / "b.png")
    recs = [{"image0": "a.ppm", "image1": "b.png", "tag": 5}]
    it = datasets.ImagePairFiles(recs, root=str(tmp_path))[0]
    assert isinstance(it["image0"], datasets.RawImage) and isinstance(it["image1"], np.ndarray) and it["tag"] == 5
    assert np.array_equal(np.asarray(it["image0"]), a) and np.array_equal(it["image1"], b)
    it = datasets.ImagePairFiles(recs, root=str(tmp_path), lazy_raw=False)[0]
    assert isinstance(it["image0"], np.ndarray) and np.array_equal(it["image0"], a)


@pytest.mark.parametrize("mode", ["L", "RGBA", "P"])
def test_other_modes_are_converted_to_rgb_like_the_reference(mode):
    """megadepth.py:150-151: `if image.mode != 'RGB': image = image.convert('RGB')`."""
    from PIL import Image
    a = _img(2)
    data = _encode(a, "PNG", mode=mode)
    want = np.array(Image.open(io.BytesIO(data)).convert("RGB"))
    out = datasets.decode_rgb(data)
    assert out.shape == a.shape and np.array_equal(out, want)


def test_pair_dataset_decodes_both_views_and_passes_everything_else_through(tmp_path):
    a, b = _img(3), _img(4, 41, 47)
    (tmp_path / "x").mkdir()
    (tmp_path / "x" / "0.png").write_bytes(_encode(a, "PNG"))
    recs = [{"image0": "x/0.png", "image1": _encode(b, "PNG"), "dataset": "megaDepth", "warp01_params": {"mode": "se3", "k": 7}}]
    ds = datasets.ImagePairFiles(recs, root=str(tmp_path))
    assert len(ds) == 1
    it = ds[0]
    assert np.array_equal(it["image0"], a) and np.array_equal(it["image1"], b)
    assert it["dataset"] == "megaDepth" and it["warp01_params"] == {"mode": "se3", "k": 7}
    assert isinstance(recs[0]["image0"], str)          # the records themselves are not touched


def test_prefetcher_keeps_order_overlaps_fetches_and_bounds_its_look_ahead():
    live, peak, lock = [0], [0], threading.Lock()

    class Slow:
        def __getitem__(self, i):
            with lock:
                live[0] += 1
                peak[0] = max(peak[0], live[0])
            time.sleep(0.02 * ((i * 7) % 3))        # later items may finish first
            with lock:
                live[0] -= 1
            return {"i": i}

    idx = [5, 2, 9, 0, 3, 3, 8, 1, 7, 4, 6, 10, 11]
    with datasets.Prefetcher(Slow(), idx, workers=4, depth=6) as pf:
        got = [(i, it["i"]) for i, it in pf]
    assert got == [(i, i) for i in idx]
    assert 2 <= peak[0] <= 4


def test_prefetcher_raises_at_the_failing_items_turn_and_not_before():
    class Broken:
        def __getitem__(self, i):
            if i == 3:
                raise OSError("cannot identify image file")
            return i

    seen = []
    with datasets.Prefetcher(Broken(), range(6), workers=3, depth=4) as pf:
        with pytest.raises(OSError, match="cannot identify"):
            for i, it in pf:
                seen.append(i)
    assert seen == [0, 1, 2]


def test_prefetcher_with_one_worker_and_an_empty_index_list():
    with datasets.Prefetcher(list(range(4)), [], workers=1) as pf:
        assert list(pf) == []
    with datasets.Prefetcher([10, 11, 12], [2, 0], workers=1, depth=1) as pf:
        assert list(pf) == [(2, 12), (0, 10)]


def test_a_dataset_that_does_not_declare_thread_safety_is_read_by_one_thread_in_order():
    """shared file handles, stateful readers and the global np.random of datasets/megadepth.py:195 must not
    be entered from 16 threads at once; the reference's DataLoader workers each read their items one after another."""
    live, peak, order, tids, lock = [0], [0], [], set(), threading.Lock()

    class Stateful:
        def __getitem__(self, i):
            with lock:
                live[0] += 1
                peak[0] = max(peak[0], live[0])
                order.append(i)
                tids.add(threading.get_ident())
            time.sleep(0.005)
            with lock:
                live[0] -= 1
            return i

    with datasets.Prefetcher(Stateful(), range(12)) as pf:              # workers=None: the dataset decides
        assert pf.workers == 1
        assert [i for i, _ in pf] == list(range(12))
    assert peak[0] == 1 and order == list(range(12)) and len(tids) == 1 and threading.get_ident() not in tids

    class Safe(Stateful):
        thread_safe = True

    with datasets.Prefetcher(Safe(), range(4)) as pf:
        assert pf.workers >= 1 and pf.workers == max(1, min(16, len(os.sched_getaffinity(0))))
    assert datasets.ImagePairFiles.thread_safe is True


def test_zero_workers_means_inline_on_the_callers_thread():
    tids, calls = set(), []

    class D:
        def __getitem__(self, i):
            tids.add(threading.get_ident())
            calls.append(i)
            return i * i

    pf = datasets.Prefetcher(D(), [3, 1, 2], workers=0)
    assert pf.pool is None
    it = iter(pf)
    assert calls == []                              # nothing is fetched ahead
    assert next(it) == (3, 9) and calls == [3]
    assert list(it) == [(1, 1), (2, 4)]
    assert tids == {threading.get_ident()}
    pf.close()
