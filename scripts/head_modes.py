#!/usr/bin/env python3
"""head_modes.py -- why does `alike_head_f16p` have two time modes on ONE library?

r05 read 9.30-9.44 ms per launch in plain `python bench.py` runs and 9.84-9.95 ms under `--spawn` (torchrun + RCCL) and under
rocprofv3, every other kernel within 2 %.  One process per ARM, all arms interleaved on one box by scripts/head_modes.sh; every arm
builds bench.py's own pipeline (256 pairs, dense descriptor maps), warms it, times N steps and reads the per-kernel HIP-event
averages, and logs what could differ between the launch paths:

  * device addresses of the buffers the head touches (images, score, the 40 GB map: torch; the activations: KPB_LOG_ALLOC lines
    on stderr), modulo 2 MiB / 64 KiB / 4 KiB,
  * allocation order against RCCL's own buffers (arms nccl_before / nccl_after / gloo),
  * HSA_* / NCCL_* / HIP_* / ROC* / TORCH_* / OMP_* of the process,
  * sclk / mclk / fclk / socclk / power from sysfs at 10 Hz while the timed steps run.

Arms:
  plain          no torch.distributed at all (what `python bench.py` runs)
  nccl_before    init_process_group("nccl", world 1) BEFORE any buffer is allocated (what bench.py does under torchrun)
  nccl_after     the same call AFTER the pipeline has allocated and warmed up
  gloo           init_process_group("gloo"): torch.distributed without RCCL
  env_dist       like nccl_before, rank / world from the environment (started by torch.distributed.run: the --spawn path itself)
  dummy:<bytes>  plain, with one torch allocation of <bytes> made (and kept) first: shifts every later address
  map_first      plain, the 40 GB map allocated before the context, the network and every other buffer
  offsets        plain, then the SAME process re-times the step with the map at byte offsets 0 ... 2 MiB + 4 KiB inside one
                 larger allocation: placement of the map alone, nothing else changed
  candidates:<n> plain, then n MORE 40 GB maps are allocated side by side and the step is re-timed on each, round robin, three
                 times over -- does the time mode belong to the ALLOCATION (where its pages are) or to the moment? -- together with
                 each buffer's plain streaming rates (fill, read): is a slow buffer slow for any access or for the head's pattern?
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H, W = 480, 640
EXTRACTOR = dict(nms_dist=6, threshold=0.0, border_dist=8, top_k=1000, min_score=0.0)
BRUTE_FORCE = dict(metric="euclidean", max_distance=5, cross_check=True)


class SysfsSampler(threading.Thread):
    """Current sclk / mclk / fclk / socclk (the starred line of pp_dpm_*) and hwmon power of card 0..n, 10 Hz."""

    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []
        self.cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))

    @staticmethod
    def _star(path):
        try:
            for line in open(path).read().splitlines():
                if line.rstrip().endswith("*"):
                    return int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    def sample(self):
        out = []
        for p in self.cards:
            d = os.path.dirname(p)
            row = {k: self._star(os.path.join(d, "pp_dpm_" + k)) for k in ("sclk", "mclk", "fclk", "socclk")}
            pw = None
            for f in glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input")):
                try:
                    pw = int(open(f).read()) / 1e6
                    break
                except (OSError, ValueError):
                    pass
            row["power_w"] = pw
            out.append(row)
        return out

    def run(self):
        while not self.stop:
            self.rows.append(self.sample())
            time.sleep(0.1)

    def summary(self):
        """Per card: min / median / max of every quantity over the samples; only cards whose power moved are interesting."""
        res = []
        if not self.rows:
            return res
        for c in range(len(self.cards)):
            s = {}
            for k in ("sclk", "mclk", "fclk", "socclk", "power_w"):
                v = sorted(r[c][k] for r in self.rows if r[c][k] is not None)
                if v:
                    s[k] = [v[0], v[len(v) // 2], v[-1]]
            res.append(s)
        return res


def mods(p):
    return {"ptr": hex(p), "mod_2MiB": p % (2 << 20), "mod_64KiB": p % (64 << 10), "mod_4KiB": p % 4096, "mod_1GiB_MiB": (p % (1 << 30)) >> 20}


def measure(pipe, images, steps, prof_steps, torch):
    """(ms per step over `steps` steps between two synchronisations, per-kernel average ms from HIP events on the launch stream)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.run(images)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    ctx = pipe.ctx
    ctx.prof_enable(True)
    for _ in range(prof_steps):
        pipe.run(images)
    prof = ctx.prof_report()
    ctx.prof_enable(False)
    ker = {k: round(v[1] / v[0], 4) for k, v in prof.items()}
    return round(ms, 4), ker


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arm", default="plain")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--tag", default="")
    ap.add_argument("--place", action="store_true", help="let PairPipeline place the descriptor map by measurement (its default); without it the arms see the driver's own placement")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "head_modes.jsonl"))
    args = ap.parse_args()
    arm = args.arm
    os.environ["KPB_LOG_ALLOC"] = "1"

    import numpy as np
    import torch
    import torch.distributed as dist
    from keypoint_bench_amd import synthetic
    from keypoint_bench_amd.models.ALike import alike_t
    from keypoint_bench_amd.pipeline import PairPipeline

    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    rec = {"arm": arm, "tag": args.tag, "pid": os.getpid(), "under_rocprof": any("rocprof" in (os.environ.get(k) or "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")),
           "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("HSA_", "NCCL_", "RCCL_", "HIP_", "ROC", "TORCH_", "OMP_", "GPU_", "AMD_", "LD_PRELOAD", "RANK", "WORLD_SIZE", "LOCAL_RANK"))}}

    def init(backend):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29731")
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)             # RCCL's banner goes to stderr (as in bench.py)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    free0 = torch.cuda.mem_get_info(dev)[0]
    if arm in ("nccl_before", "env_dist"):
        init("nccl")
    elif arm == "gloo":
        init("gloo")
    rec["free_after_init_MiB"] = (free0 - torch.cuda.mem_get_info(dev)[0]) >> 20
    keep = None
    if arm.startswith("dummy:"):
        keep = torch.empty(int(arm.split(":")[1]), dtype=torch.uint8, device=dev)
        rec["dummy"] = mods(keep.data_ptr())
    B = args.pairs
    premap = None
    if arm == "map_first":
        premap = torch.empty((2 * B, H, W, 64), dtype=torch.float32, device=dev)
    net = alike_t(dense_descriptors=True).eval()
    pipe = PairPipeline(net, EXTRACTOR, BRUTE_FORCE, B, H, W, device=dev, place_map=args.place)
    if premap is not None:
        pipe.desc = premap
    nd = 32
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(16) as ex:
        v0s, v1s = zip(*ex.map(lambda i: synthetic.image_pair(i, H, W), range(nd)))
    sel = [i % nd for i in range(B)]
    images = torch.from_numpy(np.stack([v0s[i] for i in sel] + [v1s[i] for i in sel])).to(dev).contiguous()
    for _ in range(args.warmup):
        pipe.run(images)
    torch.cuda.synchronize()
    rec["ptrs"] = {"images": mods(images.data_ptr()), "score": mods(pipe.score.data_ptr()), "desc": mods(pipe.desc.data_ptr())}
    rec["placement"] = pipe.placement
    if arm == "nccl_after":
        init("nccl")
    smp = SysfsSampler()
    rec["sysfs_cards"] = len(smp.cards)
    rec["sysfs_before"] = smp.sample()
    smp.start()
    ms, ker = measure(pipe, images, args.steps, 5, torch)
    smp.stop = True
    smp.join()
    rec["sysfs_during"] = smp.summary()
    rec["ms_per_step"], rec["kernels_ms"] = ms, {k: ker[k] for k in ("alike_head_dense", "alike_block1", "alike_block2", "nms_sweep") if k in ker}
    rec["sum_kernels_ms"] = round(sum(ker.values()), 4)
    if arm == "offsets":
        n = 2 * B * H * W * 64
        slack = (4 << 20) // 4
        big = torch.empty(n + slack, dtype=torch.float32, device=dev)
        sweep = []
        for off in (0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (2 << 20) - 4096, 2 << 20, (2 << 20) + 4096, 0):
            pipe.desc = big[off // 4: off // 4 + n].view(2 * B, H, W, 64)
            for _ in range(5):
                pipe.run(images)
            m, k = measure(pipe, images, 30, 5, torch)
            sweep.append({"offset": off, "desc": mods(pipe.desc.data_ptr()), "ms_per_step": m, "head_ms": k.get("alike_head_dense"), "block1_ms": k.get("alike_block1")})
            print("offset %8d  step %.3f  head %.3f" % (off, m, k.get("alike_head_dense", 0)), file=sys.stderr, flush=True)
        rec["offset_sweep"] = sweep
    if arm.startswith("candidates:"):
        ncand = int(arm.split(":")[1])
        cands = [("first", pipe.desc)] + [("cand%d" % i, torch.empty((2 * B, H, W, 64), dtype=torch.float32, device=dev)) for i in range(ncand)]
        rec["free_after_candidates_GiB"] = round(torch.cuda.mem_get_info(dev)[0] / 2 ** 30, 1)
        rows = []
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rnd in range(3):
            for name, buf in cands:
                pipe.desc = buf
                for _ in range(4):
                    pipe.run(images)
                m, k = measure(pipe, images, 16, 4, torch)
                row = {"round": rnd, "buffer": name, "desc": mods(buf.data_ptr()), "ms_per_step": m, "head_ms": k.get("alike_head_dense"),
                       "block1_ms": k.get("alike_block1"), "nms_ms": k.get("nms_sweep"), "sample_ms": k.get("sample_bilinear")}
                if rnd == 2:          # plain streaming rates of the same bytes (after the head runs: they overwrite the map)
                    flat = buf.view(-1)
                    torch.cuda.synchronize()
                    ev[0].record(); flat.zero_(); ev[1].record(); torch.cuda.synchronize()
                    row["fill_GBs"] = round(flat.numel() * 4 / ev[0].elapsed_time(ev[1]) / 1e6, 1)
                    ev[0].record(); sm = flat.view(torch.int32).max(); ev[1].record(); torch.cuda.synchronize()
                    row["read_GBs"] = round(flat.numel() * 4 / ev[0].elapsed_time(ev[1]) / 1e6, 1)
                    del sm
                rows.append(row)
                print("round %d %-6s %s step %.3f head %.3f %s" % (rnd, name, hex(buf.data_ptr()), m, k.get("alike_head_dense", 0),
                                                                   ("fill %.0f read %.0f GB/s" % (row["fill_GBs"], row["read_GBs"])) if rnd == 2 else ""), file=sys.stderr, flush=True)
        rec["candidates"] = rows
    if dist.is_initialized():
        dist.destroy_process_group()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("%-14s %-8s step %.3f ms  head %.3f  block1 %.3f  block2 %.3f  nms %.3f  %s" % (
        arm, args.tag, ms, ker.get("alike_head_dense", 0), ker.get("alike_block1", 0), ker.get("alike_block2", 0), ker.get("nms_sweep", 0),
        ("placed: forward ms %s -> #%d" % (pipe.placement["forward_ms"], pipe.placement["chosen"])) if pipe.placement else "driver's placement"), flush=True)


if __name__ == "__main__":
    main()
