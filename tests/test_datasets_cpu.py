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
