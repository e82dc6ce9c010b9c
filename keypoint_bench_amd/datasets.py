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

    thread_safe = True          # __getitem__ touches no shared state: Prefetcher may call it from several threads at once

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
    abandoning the iterator inside a `with`) stops the pool.

    How many threads call `dataset.__getitem__` at once is the DATASET's decision, not the runner's: only a dataset that
    declares `thread_safe = True` (ImagePairFiles does) gets the default pool of up to 16; any other dataset -- shared h5py or
    video handles, stateful sequence readers, the global np.random of datasets/megadepth.py:195 -- is read by ONE thread, in
    order, as a DataLoader worker would read its shard.  workers=0 means inline: no thread at all, dataset[i] runs on the
    caller's thread when the iterator reaches i.  An explicit workers > 1 is the caller's promise that the dataset allows it."""

    def __init__(self, dataset, indices, workers=None, depth=None):
        self.dataset, self.indices = dataset, list(indices)
        if workers is None:
            workers = max(1, min(16, len(os.sched_getaffinity(0)))) if getattr(dataset, "thread_safe", False) else 1
        self.workers = int(workers)
        self.depth = depth if depth is not None else 2 * max(1, self.workers)
        self.pool = ThreadPoolExecutor(max_workers=self.workers) if self.workers > 0 else None

    def __iter__(self):
        if self.pool is None or self.depth <= 0:
            for i in self.indices:
                yield i, self.dataset[i]
            return
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
        if self.pool is not None:
            self.pool.shutdown(wait=True, cancel_futures=True)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
