"""SURVEY 8(f)2, decode stage: the image-reading half of the reference's pair datasets, as a dataset of DECODED uint8 [H, W, 3]
items that PairRunner stages to the device (runner.HostStager) where csrc/preprocess.hip finishes the datasets' transform.

    datasets/megadepth.py:149-168   Image.open(path); convert('RGB') unless it already is; np.array(image)
    datasets/hpatches.py:47-56      cv2.imread(path, IMREAD_COLOR) [BGR] then cvtColor(BGR2RGB): the same RGB bytes PIL hands over

Decoding is PIL's (Pillow is in the image; cv2 is not): `ImagePairFiles[i]` decodes the two images of pair i, and
`Prefetcher` runs those `__getitem__` calls on a thread pool a bounded number of items ahead of the consumer -- PIL releases the
GIL inside its decoders -- and hands the items back IN ORDER.  The reference gets the same effect from DataLoader workers
(config/config_MHA.yaml: num_workers); its resize (cv2.resize, hpatches.py:66-67) is the device transform's job here."""
import io
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def decode_rgb(src):
    """path / bytes / file object -> uint8 [H, W, 3] RGB (megadepth.py:149-152)."""
    from PIL import Image
    if isinstance(src, (bytes, bytearray, memoryview)):
        src = io.BytesIO(bytes(src))
    with Image.open(src) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        return np.array(im)


class ImagePairFiles:
    """Pairs of image files (or encoded byte strings).  `records`: a sequence of dicts with 'image0' and 'image1' = path or
    bytes, plus whatever else an item carries (warp01_params / warp10_params / dataset ...), passed through untouched.
    Items come back with the two images decoded to uint8 [H, W, 3]."""

    def __init__(self, records, root=None):
        self.records, self.root = list(records), root

    def __len__(self):
        return len(self.records)

    def _src(self, v):
        if isinstance(v, str) and self.root is not None and not os.path.isabs(v):
            return os.path.join(self.root, v)
        return v

    def __getitem__(self, i):
        rec = self.records[i]
        item = dict(rec)
        item["image0"] = decode_rgb(self._src(rec["image0"]))
        if "image1" in rec:
            item["image1"] = decode_rgb(self._src(rec["image1"]))
        return item


class Prefetcher:
    """Iterates (index, dataset[index]) over `indices` in order, with up to `depth` items being fetched concurrently on
    `workers` threads.  An exception raised by dataset[index] is re-raised at that index's turn; close() (or exhausting /
    abandoning the iterator inside a `with`) stops the pool."""

    def __init__(self, dataset, indices, workers=None, depth=None):
        self.dataset, self.indices = dataset, list(indices)
        self.workers = workers or max(1, min(16, len(os.sched_getaffinity(0))))
        self.depth = depth or 2 * self.workers
        self.pool = ThreadPoolExecutor(max_workers=self.workers)

    def __iter__(self):
        pending = []
        it = iter(self.indices)
        try:
            for i in it:
                pending.append((i, self.pool.submit(self.dataset.__getitem__, i)))
                if len(pending) >= self.depth:
                    j, f = pending.pop(0)
                    yield j, f.result()
            while pending:
                j, f = pending.pop(0)
                yield j, f.result()
        finally:
            for _, f in pending:
                f.cancel()

    def close(self):
        self.pool.shutdown(wait=True, cancel_futures=True)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
