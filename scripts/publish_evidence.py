#!/usr/bin/env python3
"""Copy the summaries of scripts/collect_evidence.sh from gpurun_out/ into profiles/ (tracked).  Usage: publish_evidence.py r01"""
import glob, json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
b = json.load(open(os.path.join(src, R, "bench_default.json")))
B = b["config"]["pairs_per_step_per_gpu"]
shutil.copy(max(glob.glob(os.path.join(src, R, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime), os.path.join(dst, "%s_bench_default_kernel_stats.csv" % R))
shutil.copy(os.path.join(src, R, "bench_default.json"), os.path.join(dst, "%s_bench_default.json" % R))
shutil.copy(os.path.join(src, R, "bench_sparse.json"), os.path.join(dst, "%s_bench_sparse.json" % R))
for mode in ("dense", "sparse"):
    shutil.copy(os.path.join(src, "pmc_%s_%s" % (R, mode), "summary.txt"), os.path.join(dst, "%s_pmc_per_kernel_b%d_%s.csv" % (R, B, mode)))
    shutil.copy(os.path.join(src, "pmc_%s_%s" % (R, mode), "traffic.json"), os.path.join(dst, "pmc_traffic_b%d_%s.json" % (B, mode)))
for extra in ("bench_spawn_w1.json", "single_pair_latency.txt", "bench_profiled.json"):
    f = os.path.join(src, R, extra)
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (R, extra)))
for f in glob.glob(os.path.join(src, R, "bench_*_*.json")):
    name = os.path.basename(f)
    if name.startswith(("bench_superpoint", "bench_xfeat", "bench_disk")) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (R, name)))
for net in ("superpoint", "disk_lightglue", "xfeat"):
    d = os.path.join(src, "pmc_%s_%s" % (R, net))
    if os.path.exists(os.path.join(d, "summary.txt")):
        shutil.copy(os.path.join(d, "summary.txt"), os.path.join(dst, "%s_pmc_per_kernel_%s.csv" % (R, net)))
        if os.path.exists(os.path.join(d, "traffic.json")):
            shutil.copy(os.path.join(d, "traffic.json"), os.path.join(dst, "%s_pmc_traffic_%s.json" % (R, net)))
for f in (glob.glob(os.path.join(src, R, "runner_rate*.json")) + glob.glob(os.path.join(src, R, "parity_sweep*.json")) +
          glob.glob(os.path.join(src, R, "bench_500_steps.json")) + glob.glob(os.path.join(src, R, "rates_build.txt")) +
          glob.glob(os.path.join(src, R, "soak_*.txt")) + glob.glob(os.path.join(src, R, "soak_*.json"))):      # scripts/soak.sh
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, "%s_%s" % (R, os.path.basename(f))))
# which library every bench line of the round came from (config.build of bench.py's JSON line): one build, or the README must say so
libs = {}
for f in sorted(glob.glob(os.path.join(dst, "%s_bench_*.json" % R))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        libs[os.path.basename(f)] = (j.get("config", {}).get("build") or {}).get("lib_sha256")
    except Exception:
        libs[os.path.basename(f)] = None
json.dump(libs, open(os.path.join(dst, "%s_builds.json" % R), "w"), indent=1)
print("libraries:", sorted(set(v for v in libs.values() if v)))
print("published", R, "value", b["value"], "roofline", {k: b["roofline"][k] for k in ("kernel", "frac", "avg_ms", "traffic")})
