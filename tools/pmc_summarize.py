#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/pmc_run.sh into profiles/<tag>_hbm_traffic.json and
profiles/<tag>_kernel_stats.csv.

    python tools/pmc_summarize.py <tag> [config]      (reads gpurun_out/prof_<tag>/{trace,fetch,write})

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes): MI355X_MICROARCH.md, HBM / rocprofv3 section -- on
gfx950 FETCH_SIZE reports half of wide coalesced reads; the counters are collected in separate --pmc passes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402  (CONFIGS, algorithmic figures, kernel-source hash)
tag = sys.argv[1]
config = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
cfg = bench.CONFIGS[config]
base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
alg = bench.algorithmic(cfg, cfg["B"], cfg["T"])
if cfg["precision"] == "fp32":
    ALG = {"lstm2_fwd48_kernel": alg["fwd_bytes"], "lstm2_bwd48_kernel": alg["bwd_bytes"],
           "lstm2_fwd48x4_kernel": alg["fwd_bytes"], "lstm2_bwd48x4_kernel": alg["bwd_bytes"]}
else:
    ALG = dict(zip(bench.scan_kernel_names(cfg), (alg["fwd_bytes"], alg["bwd_bytes"])))


def counters(sub, name):
    files = glob.glob(os.path.join(base, sub, "**", "*counter_collection.csv"), recursive=True)
    files = sorted(files, key=os.path.getmtime)[-1:]          # the newest pass only (gpurun merges every run's files into gpurun_out/)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
            acc[k].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(kernel_name):
    return kernel_name.replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]


def rocprof_averages():
    """Average launch duration per kernel (us) from the kernel-trace pass of the PRODUCT library, instantiations of one template pooled
    by launch count -- what bench.py quotes beside its live HIP-event figure."""
    stats = sorted(glob.glob(os.path.join(base, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    if stats:
        for r in csv.DictReader(open(stats[-1])):
            tot[short(r["Name"])] += float(r["TotalDurationNs"])
            n[short(r["Name"])] += int(r["Calls"])
    return {k: tot[k] / n[k] / 1e3 for k in tot if n[k]}


fetch, write = counters("fetch", "FETCH_SIZE"), counters("write", "WRITE_SIZE")
avg_us = rocprof_averages()
out = {"workload": f"bench.py --config {config} (B={cfg['B']}, T={cfg['T']}) on one MI355X", "config": config, "B": cfg["B"], "T": cfg["T"],
       "source_sha256": bench.kernel_source_hash(cfg["precision"]),
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_run.sh); FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); units KB; mean over the launches of the pass",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if k.startswith("__amd") or "at::" in k:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    e = {"FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024)}
    if k in avg_us:
        e["rocprof_avg_us"] = round(avg_us[k], 2)
    if k in ALG:
        e["algorithmic_bytes_per_launch"] = ALG[k]
        e["traffic_over_algorithmic"] = round(e["hbm_bytes_per_launch_corrected"] / ALG[k], 2)
    out["kernels"][k] = e
dst = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
stats = sorted(glob.glob(os.path.join(base, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    print("wrote", os.path.join("profiles", f"{tag}_kernel_stats.csv"))
for k, e in out["kernels"].items():
    print(f"  {k:32s} {e['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB/launch")
