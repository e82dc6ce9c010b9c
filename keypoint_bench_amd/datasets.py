"""SURVEY 8(f)2, decode stage: the image-reading half of the reference's pair datasets, as a dataset of DECODED uint8 [H, W, 3]
items that PairRunner stages to the device (runner.HostStager) where csrc/preprocess.hip finishes the datasets' transform.

    datasets/megadepth.py:149-168   Image.open(path); convert('RGB') unless it already is; np.array(image)
    datasets/hpatches.py:47-56      cv2.imread(path, IMREAD_COLOR) [BGR] then cvtColor(BGR2RGB): the same RGB bytes PIL hands over

Decoding is PIL's (Pillow is in the image; cv2 is not) except for HPatches' own format, binary PNM (`.ppm`, hpatches.py:36), whose raster
IS the decoded image: `RawImage` locates it and the staging thread reads it straight into its pinned slot.  `ImagePairFiles[i]` decodes the two images of pair i, and
`Prefetcher` runs those `__getitem__` calls on a thread pool a bounded number of items ahead of the consumer -- PIL releases the
GIL inside its decoders -- and hands the items back IN ORDER.  The reference gets the same effect from DataLoader workers
(config/config_MHA.yaml: num_workers); its resize (cv2.resize, hpatches.py:66-67) is the device transform's job here."""
import io
import os
import queue
import re
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


_PNM_WS = b" \t\n\r\x0b\x0c"
_PNM_PLAIN = re.compile(rb"P[56][ \t\n\r\x0b\x0c]+(\d+)[ \t\n\r\x0b\x0c]+(\d+)[ \t\n\r\x0b\x0c]+(\d+)[ \t\n\r\x0b\x0c]")


def parse_pnm_header(head):
    """(channels, width, height, data offset) of a binary PNM whose samples are the bytes a decoder would hand over -- `P6` (RGB) or
    `P5` (gray) with maxval 255 -- else None (ASCII forms, 16-bit samples, a maxval the decoders rescale, a truncated header: PIL's
    business).  Netpbm header: magic, width, height, maxval as decimal tokens separated by whitespace, `#` comments running to the end of
    their line allowed between them, then ONE whitespace byte, then the raster."""
    if len(head) < 11 or head[:2] not in (b"P6", b"P5"):
        return None
    m = _PNM_PLAIN.match(head)          # the header every writer in practice produces (no comments): one regular expression instead of the byte loop below
    if m:
        w, h, maxval = int(m.group(1)), int(m.group(2)), int(m.group(3))
        return ((3 if head[1] == 0x36 else 1), w, h, m.end()) if (maxval == 255 and w > 0 and h > 0) else None
    i, n, vals = 2, len(head), []
    while len(vals) < 3:
        while i < n and (head[i] in _PNM_WS or head[i] == 0x23):
            if head[i] == 0x23:         # '#': a comment, to the end of its line
                while i < n and head[i] not in b"\n\r":
                    i += 1
            else:
                i += 1
        j = i
        while j < n and 0x30 <= head[j] <= 0x39:
            j += 1
        if j == i or j >= n:            # no digits, or the token may continue beyond what was read
            return None
        vals.append(int(head[i:j]))
        i = j
    if head[i] not in _PNM_WS:
        return None
    w, h, maxval = vals
    if maxval != 255 or w <= 0 or h <= 0:
        return None
    return (3 if head[:2] == b"P6" else 1), w, h, i + 1


class RawImage:
    """A binary PNM file whose raster IS the decoded image: located, not yet read.  HPatches -- BASELINE configs[0] and [1] -- is `.ppm`
    (datasets/hpatches.py:36, read with cv2.imread + BGR2RGB at 47-56: the P6 raster's own RGB bytes), so its decode stage is a file
    read, and `read_into` makes it ONE copy: from the page cache straight into the row of the pinned staging buffer the image is bound
    for (runner.HostStager.fill) -- no PIL, no intermediate array.  Quacks like the uint8 [H, W, 3] array it stands for (shape, dtype,
    ndim, np.asarray), so every consumer of decoded items takes it; `crop` is the runner's x32 crop (top-left, like crop32)."""

    dtype = np.dtype(np.uint8)
    ndim = 3

    def __init__(self, path, offset, height, width, channels, crop=None):
        self.path, self.offset, self.H, self.W, self.channels = path, int(offset), int(height), int(width), int(channels)
        self.h, self.w = crop if crop is not None else (self.H, self.W)

    @property
    def shape(self):
        return (self.h, self.w, 3)

    def crop(self, h, w):
        if not (0 < h <= self.h and 0 < w <= self.w):
            raise ValueError("crop (%d, %d) outside the %d x %d image" % (h, w, self.h, self.w))
        return RawImage(self.path, self.offset, self.H, self.W, self.channels, (int(h), int(w)))

    @staticmethod
    def _pread_full(fd, mv, offset):
        got = 0
        while got < len(mv):
            k = os.preadv(fd, [mv[got:]], offset + got)
            if k <= 0:
                raise EOFError("PNM raster ends %d bytes early" % (len(mv) - got))
            got += k

    def read_into(self, dst):
        """dst: C-contiguous uint8 [h, w, 3] (any writable buffer of that shape, pinned or not) <- the image's top-left h x w pixels."""
        if dst.shape != self.shape or dst.dtype != np.uint8 or not dst.flags.c_contiguous:
            raise ValueError("read_into wants a contiguous uint8 %s destination, got %s %s" % (self.shape, dst.dtype, tuple(dst.shape)))
        c, W = self.channels, self.W
        fd = os.open(self.path, os.O_RDONLY)
        try:
            if c == 3 and self.w == W:                  # rows are whole: one read of h * W * 3 bytes
                n = self.h * W * 3
                if os.preadv(fd, [dst], self.offset) != n:      # (short reads are rare: the careful loop takes over)
                    self._pread_full(fd, memoryview(dst).cast("B"), self.offset)
            elif c == 3:                                # the x32 crop cut columns: one read per row, each into its place
                mv = memoryview(dst).cast("B")
                rb = self.w * 3
                for r in range(self.h):
                    self._pread_full(fd, mv[r * rb:(r + 1) * rb], self.offset + r * W * 3)
            else:                                       # P5: gray -> the three equal channels IMREAD_COLOR / convert('RGB') give
                g = np.empty((self.h, W), np.uint8)
                self._pread_full(fd, memoryview(g).cast("B"), self.offset)
                dst[...] = g[:, :self.w, None]
        finally:
            os.close(fd)
        return dst

    def __array__(self, dtype=None, copy=None):
        a = self.read_into(np.empty(self.shape, np.uint8))
        return a if dtype is None else a.astype(dtype, copy=False)


def _pnm_from_bytes(data):
    hd = parse_pnm_header(bytes(data[:512]))
    if hd is None:
        return None
    c, w, h, off = hd
    if len(data) - off < c * w * h:
        return None                     # truncated: let PIL raise its own error
    a = np.frombuffer(data, np.uint8, c * w * h, off).reshape(h, w, c)
    return np.ascontiguousarray(a) if c == 3 else np.repeat(a, 3, axis=2)


def decode_rgb(src, lazy=False):
    """path / bytes / file object -> uint8 [H, W, 3] RGB (megadepth.py:149-152; hpatches.py:47-56 gives the same bytes).

    Binary PNM (P6 / P5, maxval 255 -- HPatches' format) never reaches PIL: the raster is the image.  lazy=True returns a `RawImage`
    for such a FILE (read later, straight into its staging slot); everything else is decoded by PIL as before."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        a = _pnm_from_bytes(src)
        if a is not None:
            return a
        src = io.BytesIO(bytes(src))
    elif isinstance(src, (str, os.PathLike)):
        fd = os.open(src, os.O_RDONLY)          # (four system calls and no file object: this runs once per image on the runner's prefetch threads)
        try:
            head = os.read(fd, 512)
            hd = parse_pnm_header(head) if head[:1] == b"P" else None
            size = os.fstat(fd).st_size if hd is not None else 0
        finally:
            os.close(fd)
        if hd is not None:
            c, w, h, off = hd
            if size - off >= c * w * h:
                raw = RawImage(os.fspath(src), off, h, w, c)
                return raw if lazy else np.asarray(raw)
    from PIL import Image
    with Image.open(src) as im:
        if im.mode != "RGB":
            im = im.convert("RGB")
        return np.array(im)


class ImagePairFiles:
    """Pairs of image files (or encoded byte strings).  `records`: a sequence of dicts with 'image0' and 'image1' = path or
    bytes, plus whatever else an item carries (warp01_params / warp10_params / dataset ...), passed through untouched.
    Items come back with the two images decoded to uint8 [H, W, 3]."""

    thread_safe = True          # __getitem__ touches no shared state: Prefetcher may call it from several threads at once

    def __init__(self, records, root=None, lazy_raw=True):
        """lazy_raw: binary PNM files (HPatches' .ppm) come back as `RawImage`s -- located, not read -- which the runner's staging
        thread reads straight into its pinned buffer; False materialises them here like every other format."""
        self.records, self.root, self.lazy_raw = list(records), root, bool(lazy_raw)

    def __len__(self):
        return len(self.records)

    def _src(self, v):
        if isinstance(v, str) and self.root is not None and not os.path.isabs(v):
            return os.path.join(self.root, v)
        return v

    def __getitem__(self, i):
        rec = self.records[i]
        item = dict(rec)
        item["image0"] = decode_rgb(self._src(rec["image0"]), lazy=self.lazy_raw)
        if "image1" in rec:
            item["image1"] = decode_rgb(self._src(rec["image1"]), lazy=self.lazy_raw)
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
