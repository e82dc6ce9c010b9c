#!/usr/bin/env python3
"""Fills the @@NAME@@ fields of a DESIGN.md template from the round's published evidence: scripts/fill_design.py TEMPLATE ROUND > DESIGN.md

The prose of DESIGN.md is written by hand; the figures that belong to the evidence build come from profiles/<ROUND>_*.json so that the document
and the records cannot disagree (figures of three builds side by side)."""
import json
import re
import sys


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    tmpl, rnd = sys.argv[1], sys.argv[2]
    P = "profiles/%s_" % rnd
    d = line(P + "bench_default.json")
    k = d["roofline"]["kernels_ms_per_step"]
    f = {}
    f["VALUE"] = "{:,.0f}".format(d["value"]).replace(",", " ")
    f["MS"] = "%.2f" % d["ms_per_step"]
    f["SUST"] = "{:,.0f}".format(d["value_sustained"]).replace(",", " ")
    f["SPARSE"] = "{:,.0f}".format(d["variant"]["value"]).replace(",", " ")
    f["FP32"] = "{:,.0f}".format(d["variant_fp32"]["value"]).replace(",", " ")
    f["CPU"] = "%.1f" % d["cpu_baseline"]["value"]
    f["HEAD"] = "%.2f" % k["alike_head_dense"]
    f["HEADFRAC"] = "%.2f" % d["roofline"]["frac"]
    f["HEADGBS"] = "{:,.0f}".format(d["roofline"]["achieved"]).replace(",", " ")
    f["B1"] = "%.2f" % k["alike_block1"]
    f["B2"] = "%.2f" % k["alike_block2"]
    f["NMS"] = "%.2f" % k["nms_sweep"]
    f["TAIL"] = "%.2f" % k["nms_tail"]
    f["SELTOPK"] = "%.2f" % k["select_topk"]
    f["SAMPLE"] = "%.2f" % k["sample_bilinear"]
    f["MATCH"] = "%.2f" % sum(v for n, v in k.items() if n.startswith("match_"))
    for a, b in (("B3C1", "conv3x3_b3c1"), ("B3C2", "conv3x3_b3c2"), ("B4C1", "conv3x3_b4c1"), ("B4C2", "conv3x3_b4c2")):
        f[a] = "%.2f" % k[b]
    f["V2000"] = "{:,.0f}".format(line(P + "soak_bench_2000_steps.json")["value"]).replace(",", " ")
    f["STEPMFMA"] = "%.2f" % d["roofline_step"]["frac_mfma"]
    f["STEPHBM"] = "%.2f" % d["roofline_step"]["frac_hbm"]
    f["V500"] = "{:,.0f}".format(line(P + "bench_500_steps.json")["value"]).replace(",", " ")
    f["SPAWN"] = "{:,.0f}".format(line(P + "bench_spawn_w1.json")["value"]).replace(",", " ")
    sp = line(P + "bench_superpoint_brute_force.json")
    f["SP"] = "{:,.0f}".format(sp["value"]).replace(",", " ")
    f["CONV1B"] = "%.2f" % sp["roofline"]["kernels_ms_per_step"]["sp_conv1b"]
    f["CONV1BFRAC"] = "%.2f" % sp["roofline"]["frac"]
    dk = line(P + "bench_disk_brute_force.json")
    f["DISK"] = "{:,.0f}".format(dk["value"]).replace(",", " ")
    f["UP3"] = "%.1f" % dk["roofline"]["kernels_ms_per_step"]["disk_up3"]
    f["UP3FRAC"] = "%.2f" % dk["roofline"]["frac"]
    f["XF"] = "{:,.0f}".format(line(P + "bench_xfeat_brute_force.json")["value"]).replace(",", " ")
    f["SPLG"] = "{:,.0f}".format(line(P + "bench_superpoint_lightglue.json")["value"]).replace(",", " ")
    f["SPLGF16"] = "{:,.0f}".format(line(P + "bench_superpoint_lightglue_f16attn.json")["value"]).replace(",", " ")
    f["DISKLG"] = "{:,.0f}".format(line(P + "bench_disk_lightglue.json")["value"]).replace(",", " ")
    f["DISKLGF16"] = "{:,.0f}".format(line(P + "bench_disk_lightglue_f16attn.json")["value"]).replace(",", " ")
    for key, name in (("LGMS", "bench_superpoint_lightglue.json"), ("LGMSF16", "bench_superpoint_lightglue_f16attn.json")):
        kk = line(P + name)["roofline"]["kernels_ms_per_step"]
        f[key] = "%.1f" % sum(v for n, v in kk.items() if n.startswith("lg_"))
    m = re.search(r"^dense ([0-9.]+) ms/pair", open(P + "single_pair_latency.txt").read(), re.M)
    f["SINGLE"] = m.group(1) if m else "?"
    num = lambda v: "{:,.0f}".format(v["pairs_per_s"] if isinstance(v, dict) else v).replace(",", " ")
    rr = json.load(open(P + "runner_rate.json"))
    f.update(RRMS=num(rr["match_stats"]), RRREP=num(rr["repeatability"]), RRMHA=num(rr["MHA"]), RRAUC=num(rr["AUC"]))
    sq, sx = json.load(open(P + "runner_rate_seq.json")), json.load(open(P + "runner_rate_seq_xfeat.json"))
    f.update(SEQFM=num(sq["FundamentalMatrix"]), SEQFMXF=num(sx["FundamentalMatrix"]), SEQVO=num(sq["visual_odometer"]), SEQVOXF=num(sx["visual_odometer"]))
    f.update(HOSTF32=num(json.load(open(P + "runner_rate_host.json"))["match_stats"]), HOSTU8=num(json.load(open(P + "runner_rate_u8.json"))["match_stats"]))
    png, jpg = json.load(open(P + "runner_rate_files_png.json")), json.load(open(P + "runner_rate_files_jpeg.json"))
    f.update(PNG=num(png["match_stats"]), JPEG=num(jpg["match_stats"]), PNGPOOL=num(png["decode_pool_alone_pairs_per_s"]), JPEGPOOL=num(jpg["decode_pool_alone_pairs_per_s"]))
    try:        # r06: HPatches' own format through the file path
        ppm = json.load(open(P + "runner_rate_files_ppm.json"))
        f.update(PPM=num(ppm["match_stats"]), PPMPOOL=num(ppm["decode_pool_alone_pairs_per_s"]))
    except OSError:
        pass
    try:        # the head's kernel-trace average of the profiled run of the same command (must agree with the line's HIP-event average)
        import csv
        for row in csv.DictReader(open(P + "bench_default_kernel_stats.csv")):
            if "alike_head_f16p" in row["Name"]:
                f["HEADTRACE"] = "%.2f" % (float(row["AverageNs"]) / 1e6)
                f["HEADCALLS"] = row["Calls"]
    except OSError:
        pass
    text = open(tmpl).read()
    missing = set(re.findall(r"@@([A-Z0-9]+)@@", text)) - set(f)
    if missing:
        raise SystemExit("no value for " + ", ".join(sorted(missing)))
    sys.stdout.write(re.sub(r"@@([A-Z0-9]+)@@", lambda mm: f[mm.group(1)], text))


if __name__ == "__main__":
    main()
