"""Batched pair pipeline: the per-pair body of MInterface.test_step (models/model_interface.py:205-212
-> tasks/*: detection x2, brute_force_matcher) for B independent pairs per launch wave.

One call enqueues, with no host synchronisation in between,
    net forward on 2B images -> detection on 2B score maps -> descriptors at the keypoints ->
    brute-force mutual match of B pairs -> gather of the matched keypoint rows,
then a single sync + NMS convergence check.  Every stage is the same C-ABI entry point the
single-pair drop-ins use; counts stay on the device between stages.
"""
import ctypes

import torch

from ._lib import Context, DetectParams, MatchParams, ptr


PLACE_MIN_BYTES = 4 << 30       # descriptor maps below this are not worth placing (their forward is not bound by the map's stores)
PLACE_CANDIDATES = 4            # allocations compared (the first included), as far as free memory allows
PLACE_HEADROOM = 12 << 30       # bytes that must stay free while the candidates exist (workspaces of the first forward, the caller's tensors)


class PairPipeline:
    def __init__(self, net, extractor_params, brute_force_params, batch, H, W, device="cuda:0", lightglue=None, match=True, place_map=None):
        """match=False stops after detection (the repeatability task never matches: tasks/repeatability.py:95-122).
        place_map: choose WHERE the dense descriptor map lives by measurement at the first run (`_place_map`); None = on unless
        KPB_PLACE_MAP=0 (a host-side knob of this class: the library allocates nothing of the caller's)."""
        import os
        self.place_map = (os.environ.get("KPB_PLACE_MAP", "1") != "0") if place_map is None else bool(place_map)
        self.placement = None       # the record of that choice (bench.py prints it in config.placement)
        self.net, self.B, self.H, self.W = net, int(batch), int(H), int(W)
        self.match = bool(match)
        self.device = torch.device(device)
        self.ctx = Context.get(self.device)
        ep, bf = extractor_params, brute_force_params
        if bf.get("metric", "euclidean") != "euclidean":
            raise NotImplementedError("only metric='euclidean'")
        self.top_k = min(int(ep["top_k"]), H * W)
        self.dprm = DetectParams(int(ep["nms_dist"]), float(ep["threshold"]), int(ep["border_dist"]), self.top_k,
                                 float(ep["min_score"]))
        self.mprm = MatchParams(float(bf["max_distance"]), 1 if bf["cross_check"] else 0)
        net._ensure(self.device)
        self.dense = getattr(net, "dense_descriptors", True)
        self.div = getattr(net, "desc_div", 1)
        B2, K, C = 2 * self.B, self.top_k, (net.param["dim"] if hasattr(net, "param") else net.dim)
        dev = self.device
        f32, i32 = torch.float32, torch.int32
        self.score = torch.empty((B2, 1, H, W), dtype=f32, device=dev)
        self.Hd, self.Wd = H // self.div, W // self.div
        self.desc = torch.empty((B2, self.Hd, self.Wd, C), dtype=f32, device=dev) if self.dense else None
        self.kps = torch.empty((B2, K, 3), dtype=f32, device=dev)
        self.idx = torch.empty((B2, K), dtype=i32, device=dev)
        self.n = torch.empty((B2,), dtype=i32, device=dev)
        self.sdesc = torch.empty((B2, K, C), dtype=f32, device=dev)
        self.pairs = torch.empty((self.B, K, 2), dtype=i32, device=dev)
        self.dist = torch.empty((self.B, K), dtype=torch.float64, device=dev)
        self.k = torch.empty((self.B,), dtype=i32, device=dev)
        self.m0 = torch.empty((self.B, K, 3), dtype=f32, device=dev)
        self.m1 = torch.empty((self.B, K, 3), dtype=f32, device=dev)
        self.C = C
        self.reruns = 0
        self._covis, self.cov, self.m0c, self.m1c = None, None, None, None
        self.lg = lightglue          # a keypoint_bench_amd.models.lightglue.LightGlue: replaces the brute-force matcher
        if self.lg is not None:
            if self.desc is None:
                raise ValueError("the LightGlue matcher samples the dense descriptor map")
            self.lg._ensure(self.device)
            self.lg_scores = torch.empty((self.B, K), dtype=f32, device=dev)
            self.lg_stop = torch.empty((self.B,), dtype=i32, device=dev)
        net._ensure(self.device)

    def _place_map(self, images):
        """Where the descriptor map's pages are decides how fast the dense head stores it -- a property of the ALLOCATION, found in r06
        (profiles/r06_head_modes.txt): on one box, one library, the head takes 9.12 / 9.40 / 9.65 ms per 512 images depending on which
        40 GB allocation it writes, the same figure every time for the same allocation, whatever the process does otherwise (RCCL
        initialised or not, under rocprofv3 or not, any offset inside the allocation, any virtual alignment), while plain streaming
        fills and reads of those buffers run at equal rates -- it is the head's pattern (768 workgroups, each storing 16 KB row pieces
        160 KB apart) against the physical placement the driver chose.  Successive processes alternate between regions, which is
        what r05 read as "two time modes".  The pipeline owns this buffer, so it chooses: up to PLACE_CANDIDATES allocations side by
        side (as far as free memory allows), the network's own forward timed into each (what the map is for; two forwards after one
        warm-up, events on the launch stream), the fastest kept, the others returned to the driver.  Costs about 0.3 s once per
        pipeline; results are untouched (the same kernels write the same values wherever the buffer is)."""
        self.place_map = False
        if self.desc is None or self.desc.numel() * 4 < PLACE_MIN_BYTES:
            return
        ctx, L, net = self.ctx, self.ctx.lib, self.net
        nbytes = self.desc.numel() * 4
        B2, H, W = 2 * self.B, self.H, self.W

        def forward_ms(buf):
            best = None
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(self.device))
                ctx.check(L.kpb_net_forward(net._handle, ptr(images), B2, H, W, ptr(self.score), ptr(buf)))
                e1.record(torch.cuda.current_stream(self.device))
                e1.synchronize()
                if rep:
                    t = e0.elapsed_time(e1)
                    best = t if best is None else min(best, t)
            return best

        cands, times, c = [self.desc], [], None
        times.append(forward_ms(self.desc))          # the first forward also allocates the network's workspaces: measure free memory after it
        while len(cands) < PLACE_CANDIDATES and torch.cuda.mem_get_info(self.device)[0] >= nbytes + PLACE_HEADROOM:
            try:
                c = torch.empty_like(self.desc)
            except torch.OutOfMemoryError:
                break
            cands.append(c)
            times.append(forward_ms(c))
        best = min(range(len(cands)), key=lambda i: times[i])
        self.desc = cands[best]
        self.placement = {"candidates": len(cands), "forward_ms": [round(t, 3) for t in times], "chosen": best,
                          "map_bytes": nbytes, "ptr_mod_2MiB": self.desc.data_ptr() % (2 << 20)}
        del cands, c
        torch.cuda.synchronize(self.device)
        torch.cuda.empty_cache()                     # the losers go back to the driver, not into torch's cache

    def enqueue(self, images, covis=None):
        """images [2B, 3, H, W]: rows 0..B-1 are image0 of each pair, rows B..2B-1 image1.
        covis = (hmat [2B, 9], wh [2B, 2] int32): the MHA flow (tasks/MHA.py:30-39) -- keypoints are filtered by the
        covisibility warp (rows 0..B-1 with warp01, rows B..2B-1 with warp10) BEFORE sampling and matching, and the
        matcher sees the surviving (x, y) rows only."""
        self.ctx = Context.get(self.device)      # follows torch's current stream
        ctx, L, net = self.ctx, self.ctx.lib, self.net
        B, B2, K, C, H, W = self.B, 2 * self.B, self.top_k, self.C, self.H, self.W
        assert images.shape == (B2, 3, H, W) and images.is_contiguous() and images.dtype == torch.float32
        self._covis = covis
        if self.place_map:
            self._place_map(images)
        ctx.check(L.kpb_net_forward(net._handle, ptr(images), B2, H, W, ptr(self.score), ptr(self.desc)))
        net._forward_count += 1
        ctx.check(L.kpb_detect(ctx.handle, ptr(self.score), B2, H, W, ctypes.byref(self.dprm), ptr(self.kps),
                               ptr(self.idx), ptr(self.n), 0))
        self._enqueue_match()

    def _enqueue_covis(self):
        ctx, L = self.ctx, self.ctx.lib
        B2, K = 2 * self.B, self.top_k
        if self.cov is None:
            dev, f32, i32 = self.device, torch.float32, torch.int32
            self.cov = dict(k0=torch.empty((B2, K, 2), dtype=f32, device=dev), k01=torch.empty((B2, K, 2), dtype=f32, device=dev),
                            ids=torch.empty((B2, K), dtype=i32, device=dev), n=torch.empty((B2,), dtype=i32, device=dev))
        hm, wh = self._covis
        c = self.cov
        ctx.check(L.kpb_warp_homography(ctx.handle, ptr(self.kps), B2, K, 3, ptr(self.n), ptr(hm), ptr(wh), ptr(c["k0"]), ptr(c["k01"]),
                                        ptr(c["ids"]), ptr(c["n"])))

    def _enqueue_match(self):
        if self._covis is not None:
            self._enqueue_covis()
        if not self.match:
            return
        ctx, L, net = self.ctx, self.ctx.lib, self.net
        B, B2, K, C, H, W = self.B, 2 * self.B, self.top_k, self.C, self.H, self.W
        pts, cols, n = (self.kps, 3, self.n) if self._covis is None else (self.cov["k0"], 2, self.cov["n"])
        if cols == 2 and self.m0c is None:
            self.m0c = torch.empty((B, K, 2), dtype=torch.float32, device=self.device)
            self.m1c = torch.empty((B, K, 2), dtype=torch.float32, device=self.device)
        m0, m1 = (self.m0, self.m1) if cols == 3 else (self.m0c, self.m1c)
        if self.lg is not None:     # FundamentalMatrix.py:132-133 / visual_odometer.py:60-61: matcher.match(kps0, kps1, desc0, desc1, {'w','h'})
            from ._lib import LgParams
            lg, Hd, Wd = self.lg, self.Hd, self.Wd
            if cols != 3:
                raise NotImplementedError("LightGlue consumes (x, y, score) rows (lightglue.py:451-452)")
            prm = LgParams(float(lg.conf["depth_confidence"]), float(lg.conf["width_confidence"]), float(lg.conf["filter_threshold"]),
                           lg.prune_min_kpts)
            ctx.check(L.kpb_lg_match(lg._handle, ptr(self.kps[:B]), ptr(self.kps[B:]), ptr(self.n[:B]), ptr(self.n[B:]), B, K,
                                     ptr(self.desc[:B]), ptr(self.desc[B:]), C, Hd, Wd, Hd * Wd * C, 1, Wd * C, C, W, H, ctypes.byref(prm),
                                     ptr(self.pairs), ptr(self.lg_scores), ptr(self.k), ptr(self.lg_stop)))
        else:
            if self.desc is not None:   # utils/matcher.py:221-226 on the dense map (channels-last strides)
                Hd, Wd = self.Hd, self.Wd
                ctx.check(L.kpb_sample(ctx.handle, ptr(self.desc), B2, C, Hd, Wd, Hd * Wd * C, 1, Wd * C, C, ptr(pts), cols, K,
                                       ptr(n), ptr(self.sdesc)))
            else:
                ctx.check(L.kpb_net_desc_at(net._handle, ptr(pts), cols, K, ptr(n), ptr(self.sdesc)))
            ctx.check(L.kpb_match(ctx.handle, ptr(self.sdesc[:B]), ptr(self.sdesc[B:]), B, C, K, K, ptr(n[:B]), ptr(n[B:]),
                                  ctypes.byref(self.mprm), ptr(self.pairs), ptr(self.dist), ptr(self.k)))
        ctx.check(L.kpb_gather_rows(ctx.handle, ptr(pts[:B]), B, K, cols, ptr(self.pairs), K, 2, 0, ptr(self.k), ptr(m0)))
        ctx.check(L.kpb_gather_rows(ctx.handle, ptr(pts[B:]), B, K, cols, ptr(self.pairs), K, 2, 1, ptr(self.k), ptr(m1)))

    def finish(self):
        """Sync; re-runs NMS sweeps + everything downstream for the (rare) batch that had not converged."""
        ctx = self.ctx
        rc = ctx.lib.kpb_detect_check(ctx.handle)
        if rc == 1:     # keypoints were rewritten after extra sweeps: redo the stages that consumed them
            self.reruns += 1
            self._enqueue_match()
            ctx.sync()
        elif rc != 0:
            ctx.check(rc)

    def run(self, images, covis=None):
        self.enqueue(images, covis)
        self.finish()
        return self

    def matched(self):
        """(m0, m1) [B, K, cols] of the last run: the matched rows of image 0 / image 1 (cols = 2 under covis)."""
        return (self.m0, self.m1) if self._covis is None else (self.m0c, self.m1c)

    def pair(self, b):
        """Host-side view of pair b's results (numpy): kps0, kps1, matched rows."""
        n0, n1, k = int(self.n[b]), int(self.n[self.B + b]), int(self.k[b])
        return dict(kps0=self.kps[b, :n0].cpu().numpy(), kps1=self.kps[self.B + b, :n1].cpu().numpy(),
                    pairs=self.pairs[b, :k].cpu().numpy(), dist=self.dist[b, :k].cpu().numpy(),
                    m0=self.m0[b, :k].cpu().numpy(), m1=self.m1[b, :k].cpu().numpy())


class SequencePipeline:
    """Frames of one sequence dataset (model_interface.py:217-228): result i pairs frame i-1 with frame i, and the first
    frame with itself (`last_batch` starts as the batch).  The reference runs the net on the previous frame again for
    every step; here every frame goes through net -> detection -> descriptors-at-keypoints ONCE, and pair i matches
    slot i against slot i+1 of the same buffers (kpb_match / kpb_gather_rows take the two sides as separate base
    pointers, so the shift is a pointer offset).  Slot 0 carries the last frame of the previous chunk."""

    def __init__(self, net, extractor_params, brute_force_params, frames, H, W, device="cuda:0"):
        self.net, self.F, self.H, self.W = net, int(frames), int(H), int(W)
        self.device = torch.device(device)
        self.ctx = Context.get(self.device)
        ep, bf = extractor_params, brute_force_params
        if bf.get("metric", "euclidean") != "euclidean":
            raise NotImplementedError("only metric='euclidean'")
        self.top_k = min(int(ep["top_k"]), H * W)
        self.dprm = DetectParams(int(ep["nms_dist"]), float(ep["threshold"]), int(ep["border_dist"]), self.top_k, float(ep["min_score"]))
        self.mprm = MatchParams(float(bf["max_distance"]), 1 if bf["cross_check"] else 0)
        net._ensure(self.device)
        self.dense = getattr(net, "dense_descriptors", True)
        self.div = getattr(net, "desc_div", 1)
        F, K, C = self.F, self.top_k, (net.param["dim"] if hasattr(net, "param") else net.dim)
        dev, f32, i32 = self.device, torch.float32, torch.int32
        self.C, self.Hd, self.Wd = C, H // self.div, W // self.div
        self.score = torch.empty((F, 1, H, W), dtype=f32, device=dev)
        self.desc = torch.empty((F, self.Hd, self.Wd, C), dtype=f32, device=dev) if self.dense else None
        self.kps = torch.zeros((F + 1, K, 3), dtype=f32, device=dev)
        self.idx = torch.empty((F + 1, K), dtype=i32, device=dev)
        self.n = torch.zeros((F + 1,), dtype=i32, device=dev)
        self.sdesc = torch.zeros((F + 1, K, C), dtype=f32, device=dev)
        self.pairs = torch.empty((F, K, 2), dtype=i32, device=dev)
        self.dist = torch.empty((F, K), dtype=torch.float64, device=dev)
        self.k = torch.zeros((F,), dtype=i32, device=dev)
        self.m0 = torch.empty((F, K, 3), dtype=f32, device=dev)
        self.m1 = torch.empty((F, K, 3), dtype=f32, device=dev)
        self._last = None       # slot that holds the newest frame of the previous chunk

    def run(self, images, first):
        """images [f, 3, H, W], f <= F consecutive frames; first = the chunk starts at frame 0 of the whole sequence
        (otherwise the previous call -- or `prime` -- supplied the frame before images[0]).  Leaves pair j =
        (frame before images[j], images[j]) in pairs/dist/k/m0/m1[j]."""
        self.ctx = Context.get(self.device)      # follows torch's current stream
        ctx, L, net = self.ctx, self.ctx.lib, self.net
        K, C, H, W = self.top_k, self.C, self.H, self.W
        f = images.shape[0]
        assert 0 < f <= self.F and images.shape[1:] == (3, H, W) and images.is_contiguous() and images.dtype == torch.float32
        if not first:
            if self._last is None:
                raise RuntimeError("SequencePipeline.run(first=False) needs the previous frame: call prime() or run() first")
            if self._last != 0:
                for t in (self.kps, self.n, self.sdesc):
                    t[0].copy_(t[self._last])
        ctx.check(L.kpb_net_forward(net._handle, ptr(images), f, H, W, ptr(self.score), ptr(self.desc)))
        net._forward_count += 1
        ctx.check(L.kpb_detect(ctx.handle, ptr(self.score), f, H, W, ctypes.byref(self.dprm), ptr(self.kps[1:]), ptr(self.idx[1:]),
                               ptr(self.n[1:]), 1))
        if self.desc is not None:
            Hd, Wd = self.Hd, self.Wd
            ctx.check(L.kpb_sample(ctx.handle, ptr(self.desc), f, C, Hd, Wd, Hd * Wd * C, 1, Wd * C, C, ptr(self.kps[1:]), 3, K,
                                   ptr(self.n[1:]), ptr(self.sdesc[1:])))
        else:
            ctx.check(L.kpb_net_desc_at(net._handle, ptr(self.kps[1:]), 3, K, ptr(self.n[1:]), ptr(self.sdesc[1:])))
        if first:
            for t in (self.kps, self.n, self.sdesc):
                t[0].copy_(t[1])
        ctx.check(L.kpb_match(ctx.handle, ptr(self.sdesc), ptr(self.sdesc[1:]), f, C, K, K, ptr(self.n), ptr(self.n[1:]),
                              ctypes.byref(self.mprm), ptr(self.pairs), ptr(self.dist), ptr(self.k)))
        ctx.check(L.kpb_gather_rows(ctx.handle, ptr(self.kps), f, K, 3, ptr(self.pairs), K, 2, 0, ptr(self.k), ptr(self.m0)))
        ctx.check(L.kpb_gather_rows(ctx.handle, ptr(self.kps[1:]), f, K, 3, ptr(self.pairs), K, 2, 1, ptr(self.k), ptr(self.m1)))
        self._last = f
        return self

    def prime(self, image):
        """The one-frame overlap of a chunk that does not start the sequence: runs the frame before the
        chunk through net / detection / sampling so that it can serve as `previous`; emits nothing."""
        self.run(image.reshape(1, 3, self.H, self.W), first=True)
        self._last = 1
        return self
