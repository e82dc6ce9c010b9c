#!/usr/bin/env python3
"""bench.py -- image-pairs/sec of the extract + NMS + brute-force-match hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs-per-step B] [--sparse]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): ALIKE-t extract + NMS(nms_dist 6, border 8, top_k 1000) + brute-force
mutual match (euclidean, max_distance 5, cross_check) on synthetic 640x480 pairs,
inputs resident in HBM before the timed region.  One step = one pass of the whole path over one batch of
B pairs per GPU; pairs shard across ranks with no data-path collective (weak scaling); the only exchange
is the end-of-run RCCL all-gather of per-pair metric rows, outside the timed region.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the kernel that takes the most time in the step, timed live with HIP events on the
                  launch stream (kpb_prof_*), against its algorithmic FLOPs/bytes
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm) on a bounded sample of the
                  same pairs, one single-threaded worker per host core, rank 0 at N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 480, 640
EXTRACTOR = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)   # config/config_MHA.yaml:68-73
BRUTE_FORCE = dict(metric="euclidean", max_distance=5, cross_check=True)                  # config/config_MHA.yaml:82-85
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 matrix (= vector) peak; the path computes in fp32
PEAK_F64_TFLOPS = 78.6
PEAK_F16_TFLOPS = 16 * 157.3   # MI355X_MICROARCH.md, matrix cores: the F16/BF16 forms run at 16x the fp32 MFMA rate (~2.5 PF dense)
# Kernels that take their fp32 products as three f16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate): their matrix roof,
# counted in the fp32 FLOPs of the algorithm, is a third of the f16 peak.
PEAK_SPLIT_TFLOPS = PEAK_F16_TFLOPS / 3.0


def strict_fp32():
    """KPB_FP32_MATRIX=1: the strict-fp32 kernels (fp32 MFMA / fp32 vector ALUs) instead of the split-f16 matrix forms."""
    return os.environ.get("KPB_FP32_MATRIX", "0") not in ("", "0")


def mfma_peak(kernel):
    """Matrix roof (TFLOP/s of algorithmic fp32 FLOPs) of `kernel` in this build."""
    if kernel.startswith("match_approx"):
        return PEAK_SPLIT_TFLOPS
    split = ("alike_head_dense", "alike_block1", "alike_block2", "conv3x3_b3c1", "conv3x3_b3c2", "conv3x3_b4c1", "conv3x3_b4c2")
    if kernel in split or kernel.startswith(("sp_conv", "xf_", "disk_", "lg_")):
        return PEAK_F32_TFLOPS if strict_fp32() else PEAK_SPLIT_TFLOPS
    return PEAK_F32_TFLOPS


# ---------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(i):
    import numpy as np
    import torch
    torch.set_num_threads(1)
    import oracle
    from oracle import alike_ref
    from keypoint_bench_amd import synthetic, weights
    t = {k: torch.from_numpy(v) for k, v in weights.load_alike_t().items()}
    v0, v1 = synthetic.image_pair(i, H, W)
    t0 = time.perf_counter()
    with torch.no_grad():
        s0, d0 = alike_ref.alnet_forward(torch.from_numpy(v0)[None], t)
        s1, d1 = alike_ref.alnet_forward(torch.from_numpy(v1)[None], t)
    k0, _ = oracle.detection(s0[0, 0].numpy(), EXTRACTOR)
    k1, _ = oracle.detection(s1[0, 0].numpy(), EXTRACTOR)
    f0 = oracle.sample(d0[0].numpy(), k0)
    f1 = oracle.sample(d1[0].numpy(), k1)
    pairs, _ = oracle.match(f0, f1, BRUTE_FORCE["max_distance"], BRUTE_FORCE["cross_check"])
    return time.perf_counter() - t0, len(pairs)


def cpu_baseline(pairs_per_worker=6):
    """Oracle pipeline on host cores: one single-threaded worker per core, each doing whole pairs."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))   # the 1-GPU box's CPU share
    n = cores * pairs_per_worker
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, range(n), chunksize=pairs_per_worker)
    wall = time.perf_counter() - t0
    busy = sum(r[0] for r in res)
    # throughput from the workers' own timers (excludes interpreter start-up and input synthesis)
    value = n / (busy / cores)
    return dict(value=round(value, 4), unit="pairs/s", cores=cores, kind="port",
                sample="%d synthetic 640x480 pairs, oracle/ (torch-fp32 ALIKE-t restatement + C NMS/top-k/match), "
                       "%d single-thread workers, %.1f s wall, %.2f s/pair/core" % (n, cores, wall, busy / n))


def fp32_companion(args):
    """The same command with the strict-fp32 kernels (KPB_FP32_MATRIX=1: fp32 MFMA / fp32 vector ALUs, no half-precision
    operand anywhere), run in a CHILD process before this one touches the GPU: what the split-f16 matrix arithmetic buys."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(max(3, args.steps // 2)), "--warmup", str(args.warmup), "--no-cpu-baseline",
           "--no-variants", "--model", args.model, "--matcher", args.matcher, "--lg-attention", args.lg_attention]
    if args.pairs_per_step:
        cmd += ["--pairs-per-step", str(args.pairs_per_step)]
    if args.sparse:
        cmd.append("--sparse")
    if args.distinct:
        cmd += ["--distinct", str(args.distinct)]
    try:
        p = subprocess.run(cmd, env=dict(os.environ, KPB_FP32_MATRIX="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        j = json.loads(line)
        return {"arithmetic": "strict fp32: fp32 MFMA (v_mfma_f32_32x32x2_f32) and fp32 vector ALUs, no half-precision operands",
                "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                "dominant_kernel": (j.get("roofline") or {}).get("kernel"), "dominant_kernel_ms": (j.get("roofline") or {}).get("avg_ms")}
    except Exception as e:      # the companion is a reported extra: its failure must not cost the main figure
        return {"error": "%s: %s" % (type(e).__name__, e)}



# ---------------------------------------------------------------------------------------- what ran, and the rank logic
def build_identity():
    """Which library produced this line: the .so's content hash and mtime, the commit it was built from (recorded by build.py; the GPU
    box has no .git) and, where git is at hand, the tree's own HEAD."""
    import hashlib
    import subprocess
    so = os.environ.get("KPB_LIB_PATH") or os.path.join(ROOT, "keypoint_bench_amd", "libkpb.so")      # the library _lib.load() takes
    out = {"lib_sha256": None, "lib_mtime": None, "built_from": None, "tree_head": None}
    if os.environ.get("KPB_LIB_PATH"):
        out["lib_path_override"] = so        # a measurement build, not the tree's library: built_from below does not describe it
    try:
        out["lib_sha256"] = hashlib.sha256(open(so, "rb").read()).hexdigest()[:12]
        out["lib_mtime"] = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime(os.path.getmtime(so)))
    except OSError:
        pass
    try:
        out["built_from"] = json.load(open(os.path.join(ROOT, "keypoint_bench_amd", "_obj", "build_info.json")))
    except (OSError, ValueError):
        pass
    try:        # do the sources in this tree hash to what the library says it was built from?  (build.source_hash: csrc/ + kpb.h + build.py + isa_fixup.py)
        from keypoint_bench_amd import build as kbuild
        out["src_sha256_tree"] = kbuild.source_hash()
        out["lib_matches_sources"] = bool(out["built_from"]) and out["built_from"].get("src_sha256") == out["src_sha256_tree"]
    except Exception:
        pass
    try:
        r = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10)
        out["tree_head"] = r.stdout.strip() or None
    except Exception:
        pass
    return out


def timed_steps(run, steps, warmup, dev, use_dist, dist=None):
    """The contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by synchronize + barrier on both sides,
    MAX over ranks.  Returns (seconds, seconds of the last half or None, THIS rank's own seconds -- its steps up to its own synchronisation,
    before the barrier: the MAX hides which rank is slow, `per_rank_figures` gathers these); the second figure comes from two events on the launch stream
    (no extra synchronisation inside the region) and is what the chip SUSTAINS -- a 20-step run right after an idle period reads ~3 %
    faster than the state an uninterrupted stream settles into.  `run()` performs one step; on a CPU device (the
    gloo test) the events are perf_counter stamps."""
    import torch
    cuda = dev.type == "cuda"

    def barrier():
        if cuda:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            if cuda:
                torch.cuda.synchronize(dev)

    for _ in range(warmup):
        run()
    barrier()
    half = steps // 2 if steps >= 100 else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if (cuda and half) else None
    t_half = None
    t0 = time.perf_counter()
    for i in range(steps):
        if half is not None and i == half:
            if ev:
                ev[0].record(torch.cuda.current_stream(dev))
            else:
                t_half = time.perf_counter()
        run()
    if ev:
        ev[1].record(torch.cuda.current_stream(dev))
    t_own = time.perf_counter()         # this rank's own end (the CPU stand-in of the second event)
    if cuda:
        torch.cuda.synchronize(dev)
    own = time.perf_counter() - t0      # this rank alone: its K steps to its own synchronisation, no barrier
    barrier()
    elapsed = time.perf_counter() - t0
    tail = None
    if half is not None:
        tail = ev[0].elapsed_time(ev[1]) * 1e-3 if ev else (t_own - t_half)
    if use_dist:
        t = torch.tensor([elapsed, tail if tail is not None else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, tail = float(t[0].item()), (float(t[1].item()) if tail is not None else None)
    return elapsed, tail, own


def per_rank_figures(own_ms_per_step, kernel_ms, world, use_dist, dist=None, dev=None):
    """[{rank, ms_per_step, dominant_kernel_ms}] of every rank, on every rank: one all-gather of two floats per rank (outside the timed region).
    `value` is computed from the MAX over ranks; this says which rank that was and whether its dominant kernel or something else was slow."""
    import torch
    if not use_dist:
        return [{"rank": 0, "ms_per_step": round(own_ms_per_step, 3), "dominant_kernel_ms": round(kernel_ms, 4) if kernel_ms else None}]
    mine = torch.tensor([own_ms_per_step, kernel_ms or 0.0], dtype=torch.float64, device=dev)
    allv = torch.empty((world * 2,), dtype=torch.float64, device=dev)      # flat: rank r's two figures at [2 r, 2 r + 2) (the concatenating form gloo takes too)
    dist.all_gather_into_tensor(allv, mine)
    allv = allv.view(world, 2)
    return [{"rank": r, "ms_per_step": round(float(allv[r, 0]), 3), "dominant_kernel_ms": round(float(allv[r, 1]), 4) if float(allv[r, 1]) else None}
            for r in range(world)]


def exchange_rows(rows, world, use_dist, dist=None):
    """fixed-width per-pair rows, ONE all-gather at the end of the run (rank r's rows land at [r * B, (r + 1) * B))."""
    import torch
    if not use_dist:
        return rows
    allrows = torch.empty((world * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
    dist.all_gather_into_tensor(allrows, rows.contiguous())
    return allrows


def generator_threads(world):
    """Host threads one rank may use to synthesise its pairs: the visible cores shared by the ranks of the node (8 ranks x 16 threads on
    one node's cores would oversubscribe them)."""
    return max(1, min(len(os.sched_getaffinity(0)) // max(world, 1), 16))

# whole-path algorithmic work per pair: conv / GEMM FLOPs (2 x MAC) and compulsory HBM bytes
NET_GFLOP_PER_IMAGE = {"alike": 3.8885, "alike_sparse": 1.3720 + 0.0082, "superpoint": 52.10, "xfeat": 2.54, "disk": 197.8}
DESC_CHANNELS = {"alike": 64, "superpoint": 256, "xfeat": 64, "disk": 128}


def step_roofline(model, matcher, sparse, pairs_per_s_per_gpu):
    """Whole-step roofline: the path's algorithmic FLOPs and compulsory bytes per pair times the measured rate of
    ONE GPU, against the matrix and HBM peaks.  The dense ALIKE mode adds the 2 x 2 x 78.6 MB of descriptor maps the
    reference's layering asks for (written, then sampled), as 8d says to."""
    C = DESC_CHANNELS[model]
    key = "alike_sparse" if (model == "alike" and sparse) else model
    flops = 2 * NET_GFLOP_PER_IMAGE[key] * 1e9 + (2 * 1000 * 1000 * C if matcher == "brute_force" else 0)
    nbytes = 2 * (3.686e6 + 1.229e6 + 1.229e6 + 0.012e6) + 2 * 1000 * 4 * C * 4 + 2 * 1000 * C * 4 + 0.016e6
    dense_extra = 2 * 2 * 78.6e6 if (model == "alike" and not sparse) else 0.0
    peak = PEAK_F32_TFLOPS if strict_fp32() else PEAK_SPLIT_TFLOPS
    tf, gbs = pairs_per_s_per_gpu * flops / 1e12, pairs_per_s_per_gpu * (nbytes + dense_extra) / 1e9
    return {"flops_per_pair": flops, "compulsory_bytes_per_pair": nbytes, "dense_map_bytes_per_pair": dense_extra,
            "achieved_tflops": round(tf, 2), "mfma_peak_tflops": round(peak, 1), "frac_mfma": round(tf / peak, 4),
            "achieved_gbs": round(gbs, 1), "hbm_peak_gbs": PEAK_HBM_GBS, "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
            "note": "LightGlue's FLOPs are not in SURVEY 8d and are not counted" if matcher == "lightglue" else None}


# ---------------------------------------------------------------------------------------- roofline table
def kernel_costs(B2, B, K, C, dense, sweeps):
    """Algorithmic (compulsory) FLOPs and HBM bytes PER LAUNCH of each kernel."""
    P = H * W
    c = {}
    c["alike_block1"] = (2 * P * (27 * 8 + 72 * 8) * B2, P * (12 + 32 + 8) * B2)      # image in, x1 out, and its 2 x 2 max-pool (P / 4 x 32 B) block 2 reads: DESIGN.md section 5
    c["conv3x3_b2c1"] = (2 * (P // 4) * 9 * 8 * 16 * B2, (P * 32 + (P // 4) * 64) * B2)
    c["conv3x3_b2c2"] = (2 * (P // 4) * (9 * 16 * 16 + 8 * 16) * B2, ((P // 4) * 64 + P * 32 + (P // 4) * 64) * B2)
    c["conv3x3_b3c1"] = (2 * (P // 64) * 9 * 16 * 32 * B2, ((P // 4) * 64 + (P // 64) * 128) * B2)
    c["conv3x3_b3c2"] = (2 * (P // 64) * (9 * 32 * 32 + 16 * 32) * B2, ((P // 64) * 128 * 2 + (P // 4) * 64) * B2)
    c["conv3x3_b4c1"] = (2 * (P // 1024) * 9 * 32 * 64 * B2, ((P // 64) * 128 + (P // 1024) * 256) * B2)
    c["conv3x3_b4c2"] = (2 * (P // 1024) * (9 * 64 * 64 + 32 * 64) * B2, ((P // 1024) * 256 * 2 + (P // 64) * 128) * B2)
    c["conv1x1_agg2"] = (2 * (P // 4) * 16 * 16 * B2, (P // 4) * 128 * B2)
    # block 2 fused (b2c1 + b2c2 + identity branch + agg2): reads the pooled block-1 map (32 B per pixel of H/2 x W/2), writes a2
    # (64 B), the score share (4 B) and the 4 x 4 max-pool of x2 block 3 reads (64 B per pixel of H/8 x W/8); x2 stays in LDS
    c["alike_block2"] = (c["conv3x3_b2c1"][0] + c["conv3x3_b2c2"][0] + c["conv1x1_agg2"][0], ((P // 4) * (32 + 64 + 4) + (P // 64) * 64) * B2)
    c["conv1x1_agg3"] = (2 * (P // 64) * 32 * 16 * B2, (P // 64) * 192 * B2)
    c["conv1x1_agg4"] = (2 * (P // 1024) * 64 * 16 * B2, (P // 1024) * 320 * B2)
    feat = 2 * 8 * 16 + 3 * 16 * 8 + 2 * 64            # agg1 + three 4-tap lerps + score dot (score-only kernel)
    # head 64 -> 65 = 2 555.9 MFLOP/img, plus agg1 (8 -> 16 at full resolution, 78.6 MFLOP/img) which this kernel fuses
    c["alike_head_dense"] = ((2 * 64 * 65 + 2 * 8 * 16) * P * B2, P * (32 + 4 + 256) * B2 + (P // 4 + P // 64 + P // 1024) * 64 * B2)
    c["alike_head_score"] = (feat * P * B2, P * (32 + 4) * B2 + (P // 4 + P // 64 + P // 1024) * 64 * B2)
    c["nms_sweep"] = (0, 2 * P * 4 * B2 / max(sweeps, 1))       # map read once + written once, spread over the sweeps
    c["select_topk"] = (0, (P * 4 + K * 16) * B2)
    c["sample_bilinear"] = (7 * K * C * B2, (4 * K * C * 4 + K * C * 4) * B2)
    c["alike_desc_at"] = ((2 * 64 * 64 + 4 * 64 * 8) * K * B2, (4 * 4 * 64 * 4 + K * 0 + 256) * K * B2)
    c["match_tile"] = (3 * K * K * C * B, 2 * K * C * 4 * B)     # float64 sub/mul/add per element
    c["match_approx_min"] = (2 * K * K * C * B, 2 * K * C * 4 * B)     # the prefilter's approximate distance matrix (2 MAC flops per element), twice
    c["match_approx_cand"] = c["match_approx_min"]
    c["match_finalize"] = (0, (2 * K * 16 * 12 + K * 20) * B)
    c["gather_rows"] = (0, K * 32 * B)
    return c


def other_net_costs(B2):
    """(FLOPs, bytes) per launch of the dominant XFeat / DISK convolutions (2*MAC on the reference's channel counts;
    NHWC fp32 in + out)."""
    P = H * W
    c = {}
    def conv(name, px, cin, cout, k):
        c[name] = (2 * px * k * k * cin * cout * B2, px * (cin + cout) * 4 * B2)
    conv("xf_block1.3", P // 16, 8, 24, 3)
    # the fused block-1 kernels (r04): block1.0 (1 -> 4 at full resolution) + block1.1 (4 -> 8, stride 2) read the grey image and write the
    # 8-channel half-resolution map; block1.2 (8 -> 8) + block1.3 (8 -> 24, stride 2) + the 4 x 4-averaged skip term read that map and the
    # grey image and write the 24-channel quarter-resolution map (XFeat.py:44-59, 124-126)
    c["xf_block1.01"] = (2 * (P * 9 * 1 * 4 + (P // 4) * 9 * 4 * 8) * B2, (P * 4 + (P // 4) * 32) * B2)
    c["xf_block1.23"] = (2 * ((P // 4) * 9 * 8 * 8 + (P // 16) * 9 * 8 * 24 + (P // 16) * 24) * B2, ((P // 4) * 32 + P * 4 + (P // 16) * 96) * B2)
    conv("xf_block2.0", P // 16, 24, 24, 3); conv("xf_block2.1", P // 16, 24, 24, 3)
    conv("xf_block3.1", P // 64, 64, 64, 3); conv("xf_block_fusion.0", P // 64, 64, 64, 3); conv("xf_block_fusion.1", P // 64, 64, 64, 3)
    conv("disk_up3", P, 80, 129, 5); conv("disk_up2", P // 4, 96, 64, 5); conv("disk_up1", P // 16, 128, 64, 5)
    return c


def pmc_traffic(kernel, pairs, dense):
    """HBM bytes per launch of `kernel` from the committed PMC passes (scripts/prof_pmc.sh: FETCH_SIZE and WRITE_SIZE
    in separate rocprofv3 --pmc runs of this same command; FETCH_SIZE doubled per the gfx950 correction).  Counters
    cannot be collected from inside a timed run, so this is the figure of the profile named in profiles/README.md."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_b%d_%s.json" % (pairs, "dense" if dense else "sparse"))
    try:
        return json.load(open(path))[kernel]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def superpoint_costs(B2):
    """(FLOPs, bytes) per launch of the SuperPoint conv kernels (2*MAC; NHWC fp32 in + out, weights negligible)."""
    P = H * W
    c = {}
    def conv(name, px, cin, cout, k, in_px=None, out_px=None):
        c[name] = (2 * px * k * k * cin * cout * B2, ((in_px or px) * cin + (out_px or px) * cout) * 4 * B2)
    conv("sp_conv1a", P, 1, 64, 3)
    conv("sp_conv1b", P, 64, 64, 3, out_px=P // 4)
    conv("sp_conv2a", P // 4, 64, 64, 3)
    conv("sp_conv2b", P // 4, 64, 64, 3, out_px=P // 16)
    conv("sp_conv3a", P // 16, 64, 128, 3)
    conv("sp_conv3b", P // 16, 128, 128, 3, out_px=P // 64)
    conv("sp_conv4a", P // 64, 128, 128, 3)
    conv("sp_conv4b", P // 64, 128, 128, 3)
    conv("sp_convPa", P // 64, 128, 256, 3)
    conv("sp_convPb", P // 64, 256, 65, 1)
    conv("sp_convDa", P // 64, 128, 256, 3)
    conv("sp_convDb", P // 64, 256, 256, 1)
    return c


def workload_label(model, matcher):
    """Names the workload and the BASELINE.json config it belongs to (configs[1] is the headline; the others are run with
    --model / --matcher and seeded stand-in weights)."""
    net = {"alike": "ALIKE-t", "superpoint": "SuperPoint", "xfeat": "XFeat", "disk": "DISK"}[model]
    cfg = {("alike", "brute_force"): "BASELINE configs[1]", ("superpoint", "brute_force"): "BASELINE configs[2] extract+match stage",
           ("xfeat", "brute_force"): "BASELINE configs[3] extract+match stage", ("disk", "lightglue"): "BASELINE configs[4] extract+match stage"}
    tag = cfg.get((model, matcher), "not a BASELINE config")
    m = ("brute-force mutual match (euclidean fp64, max_distance=5, cross_check)" if matcher == "brute_force"
         else "LightGlue attention matcher (split-f16 MFMA with fp32 accumulation, 9 layers, early stop + pruning)")
    return "%s extract + NMS(nms_dist=6, border=8, top_k=1000) + %s, 640x480 pairs [%s]" % (net, m, tag)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (SURVEY 8d: at least 200 timed batches)")
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps before them (SURVEY 8d: 20)")
    ap.add_argument("--pairs-per-step", type=int, default=None, help="pairs per GPU per step (default: 256 ALIKE and XFeat [SURVEY 8d], 16 SuperPoint and DISK)")
    ap.add_argument("--sparse", action="store_true", help="keypoint-only descriptors (no dense 78.6 MB/img map)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the keypoint-only and strict-fp32 companion figures")
    ap.add_argument("--spawn", action="store_true", help="take the multi-GPU launch path (a child torchrun, one rank per GPU over RCCL) even at --gpus 1")
    ap.add_argument("--distinct", type=int, default=None, help="distinct synthetic pairs generated (default: one per pair of the batch; fewer are cycled)")
    ap.add_argument("--matcher", default="brute_force", choices=["brute_force", "lightglue"],
                    help="lightglue = BASELINE configs[4] (with --model disk or superpoint), seeded stand-in weights")
    ap.add_argument("--lg-attention", default="fp32", choices=["fp32", "f16"],
                    help="LightGlue attention arithmetic: fp32 = the reference's CPU branch as split-f16 MFMA triples (default, what the parity fixtures pin); "
                         "f16 = what the reference runs on a GPU (q.half(), k.half(), v.half() through SDPA, lightglue.py:129-134), reported separately")
    ap.add_argument("--model", default="alike", choices=["alike", "superpoint", "xfeat", "disk"],
                    help="alike = BASELINE configs[1] (the headline); superpoint = configs[2] with seeded random weights")
    args = ap.parse_args()

    if (args.gpus > 1 or args.spawn) and "RANK" not in os.environ:
        # a bare `python bench.py --gpus N` (or --spawn at N = 1): start the N ranks as a CHILD torchrun (nothing in this process has touched the
        # GPU yet -- a process that has must never be replaced by another program on this pool) and relay its JSON line
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--spawn"]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
        if lines:
            print(lines[-1])
        raise SystemExit(proc.returncode if proc.returncode else (0 if lines else 1))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print("warning: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()          # before anything touches the GPU (spawned workers, CPU only)
    fp32 = None
    if rank == 0 and world == 1 and "RANK" not in os.environ and not args.no_variants and not strict_fp32():
        fp32 = fp32_companion(args)   # a child process with KPB_FP32_MATRIX=1 (the library reads the knob once per process)

    import numpy as np
    import torch
    import torch.distributed as dist
    from keypoint_bench_amd import synthetic
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.pipeline import PairPipeline

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X GPU; keypoint_bench_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ        # under torchrun the RCCL path is exercised even at N=1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # RCCL prints a banner (host name, library path) on stdout when its communicator comes up; stdout must carry the
        # one JSON line only, so the file descriptor points at stderr until the first collective has run
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    B = args.pairs_per_step or {"alike": 256, "xfeat": 256, "superpoint": 16, "disk": 16}[args.model]
    if args.model == "superpoint":
        from keypoint_bench_amd.models.SuperPoint import superpoint_random
        net = superpoint_random(7).eval()
    elif args.model == "xfeat":
        from keypoint_bench_amd.models.XFeat import xfeat_random
        net = xfeat_random(9).eval()
    elif args.model == "disk":
        from keypoint_bench_amd.models.disk import disk_random
        net = disk_random(5).eval()
    else:
        net = alike_t(dense_descriptors=not args.sparse).eval()
    lg = None
    if args.matcher == "lightglue":
        from keypoint_bench_amd import weights as kw
        from keypoint_bench_amd.models.lightglue import LightGlue
        dim, scale = {"disk": (128, 1), "superpoint": (256, 8)}[args.model]
        lg = LightGlue(features=None, desc_scale=scale, attention=args.lg_attention)
        lg.load_state_dict(kw.random_lightglue_state_dict(31, dim, "plain"))
    pipe = PairPipeline(net, EXTRACTOR, BRUTE_FORCE, B, H, W, device=dev, lightglue=lg)
    # synthetic pairs, different per rank, resident in HBM
    nd = min(args.distinct or B, B)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(generator_threads(world)) as ex:     # numpy releases the GIL: ~0.1 s per pair per thread; cores // ranks
        v0s, v1s = zip(*ex.map(lambda i: synthetic.image_pair(rank * 1000 + i, H, W), range(nd)))
    sel = [i % nd for i in range(B)]
    images = torch.from_numpy(np.stack([v0s[i] for i in sel] + [v1s[i] for i in sel])).to(dev).contiguous()

    elapsed, tail, own = timed_steps(lambda: pipe.run(images), args.steps, args.warmup, dev, use_dist, dist)
    value = world * B * args.steps / elapsed
    value_sustained = (world * B * (args.steps - args.steps // 2) / tail) if tail else None

    # end-of-run exchange: fixed-width per-pair rows [n0, n1, matches], one RCCL all-gather
    rows = torch.stack([pipe.n[:B].float(), pipe.n[B:].float(), pipe.k.float()], dim=1).contiguous()
    allrows = exchange_rows(rows, world, use_dist, dist).cpu().numpy()
    world_seen = dist.get_world_size() if use_dist else 1

    # the same pairs with keypoint-only descriptors (ALNet(dense_descriptors=False): same keypoints and matches, the
    # 78.6 MB/image descriptor map is never written; SURVEY 8d asks to say which was run): timed like the main loop
    variant = None
    if args.model == "alike" and not args.sparse and args.matcher == "brute_force" and not args.no_variants:
        pipe2 = PairPipeline(alike_t(dense_descriptors=False).eval(), EXTRACTOR, BRUTE_FORCE, B, H, W, device=dev)
        e2, _, _ = timed_steps(lambda: pipe2.run(images), args.steps, args.warmup, dev, use_dist, dist)
        same = bool(torch.equal(pipe2.k, pipe.k) and torch.equal(pipe2.pairs[:, :16], pipe.pairs[:, :16]))
        variant = {"descriptors": "keypoint-only", "value": round(world * B * args.steps / e2, 2), "unit": "pairs/s",
                   "ms_per_step": round(1e3 * e2 / args.steps, 3), "same_matches_as_dense": same}
        del pipe2

    # roofline leg: per-kernel durations from HIP events on the launch stream, same workload
    roof = None
    ctx = pipe.ctx
    # r06: two untimed steps first, then ten profiled ones.  The first step after the host-side pause that precedes this leg (row exchange, the
    # variant's set-up) runs its big kernels 5-7 % slower (kernel trace: the head 9.89 ms, then 9.21, 9.24) -- with r05's three profiled steps and no
    # warm-up that one launch put `avg_ms` 2.4 % above the kernel-trace average of the same process (profiles/r06_head_modes.txt)
    for _ in range(2):
        pipe.run(images)
    ctx.prof_enable(True)
    prof_steps = 10
    for _ in range(prof_steps):
        pipe.run(images)
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    # every rank's own step time and the average of ITS dominant kernel: gathered, reported under quality.per_rank
    dom = max(prof, key=lambda k: prof[k][1]) if prof else None
    per_rank = per_rank_figures(1e3 * own / args.steps, (prof[dom][1] / prof[dom][0]) if dom else None, world, use_dist, dist, dev)
    if rank == 0 and prof:
        sweeps = prof.get("nms_sweep", (1, 0))[0] / prof_steps
        costs = kernel_costs(2 * B, B, EXTRACTOR["top_k"], net.dim, not args.sparse, sweeps)
        costs.update(superpoint_costs(2 * B))
        costs.update(other_net_costs(2 * B))
        name = max(prof, key=lambda k: prof[k][1])
        calls, total_ms = prof[name]
        avg_ms = total_ms / calls
        flops, nbytes = costs.get(name, (0, 0))
        tf, gbs = flops / avg_ms / 1e9, nbytes / avg_ms / 1e6
        if name == "match_tile":
            bound, achieved, peak, unit = "mfma", tf, PEAK_F64_TFLOPS, "TFLOP/s"     # fp64 vector peak (no MFMA used)
        elif flops and (flops / max(nbytes, 1)) > (mfma_peak(name) * 1e3 / PEAK_HBM_GBS):
            bound, achieved, peak, unit = "mfma", tf, round(mfma_peak(name), 1), "TFLOP/s"
        else:
            bound, achieved, peak, unit = "hbm", gbs, PEAK_HBM_GBS, "GB/s"
        tot = sum(v[1] for v in prof.values())
        traffic = pmc_traffic(name, B, not args.sparse) if args.model == "alike" else None
        roof = dict(bound=bound, achieved=round(achieved, 3), peak=peak, unit=unit, frac=round(achieved / peak, 4),
                    traffic=traffic, other_roof=dict(bound="hbm" if bound == "mfma" else "mfma", achieved=round(gbs if bound == "mfma" else tf, 3),
                                                     peak=PEAK_HBM_GBS if bound == "mfma" else round(mfma_peak(name), 1), unit="GB/s" if bound == "mfma" else "TFLOP/s",
                                                     frac=round((gbs / PEAK_HBM_GBS) if bound == "mfma" else (tf / mfma_peak(name)), 4)),
                    kernel=name, avg_ms=round(avg_ms, 4), launches_per_step=calls / prof_steps,
                    share_of_step=round(total_ms / tot, 3),
                    kernels_ms_per_step={k: round(v[1] / prof_steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])})

    if rank == 0:
        out = {
            "metric": "image-pairs/sec (extract+NMS+%s, 640x480, top_k=1000)" % ("BF-match" if args.matcher == "brute_force" else "LightGlue-match"),
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # rate of the LAST HALF of the timed steps (events on the launch stream; steps >= 100): the state an uninterrupted launch stream settles into
            "value_sustained": round(value_sustained, 2) if value_sustained else None,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_label(args.model, args.matcher),
                       "matcher": args.matcher, "lightglue_attention": (args.lg_attention if args.matcher == "lightglue" else None), "pairs_per_step_per_gpu": B, "descriptors": "keypoint-only" if args.sparse else "dense-map",
                       "weights": "alike-t (reference checkpoint, BN folded)" if args.model == "alike" else args.model + ", seeded random (checkpoint absent from the reference tree)", "parallelism": "pairs sharded, dp%d" % world,
                       "nms_reruns": pipe.reruns, "build": build_identity(), "world_size_seen": world_seen,
                       # where the descriptor map lives was chosen by measurement (PairPipeline._place_map): the forward's time into each candidate allocation
                       "placement": pipe.placement,
                       "arithmetic": ("strict fp32 (KPB_FP32_MATRIX=1): fp32 MFMA / fp32 vector ALUs, fp64 match" if strict_fp32() else
                                      "fp32 results; matrix products as split-f16 MFMA triples with fp32 accumulation (2^-22 per product), operands "
                                      "scaled per tile to the f16 window (no fixed range), fp64 match")},
            "quality": {"mean_kps": round(float(allrows[:, :2].mean()), 1), "mean_matches": round(float(allrows[:, 2].mean()), 1),
                        "pairs_gathered": int(allrows.shape[0]),
                        "pairs_gathered_per_rank": [int((allrows[r * B:(r + 1) * B, 0] > 0).sum()) for r in range(world)],
                        "per_rank": per_rank},
            "roofline": roof, "roofline_step": step_roofline(args.model, args.matcher, args.sparse, value / world),
            "cpu_baseline": cpu, "variant": variant, "variant_fp32": fp32,
        }
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
