"""Summarise tools/build_pmc.sh: per build kernel, mean duration and per-dispatch means of every counter collected
(FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, as tools/save_profiles.py does --
an upper bound for kernels whose reads are narrow gathers)."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_chunk_masks", "k_insert", "k_elevate", "k_final_vid", "k_neighbors", "k_radix_scatter", "k_table_insert")


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


out = collections.defaultdict(dict)
f = max(glob.glob(os.path.join(ROOT, "gpurun_out/bpmc_stats/*/*kernel_stats.csv")), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    k = short(r["Name"])
    if k.split("<")[0] in KERNELS:
        out[k]["avg_us"] = round(float(r["AverageNs"]) / 1e3, 1)
        out[k]["calls"] = int(r["Calls"])
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out/bpmc_p*"))):
    fs = glob.glob(os.path.join(d, "*/*counter_collection.csv"))
    if not fs:
        continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        if k.split("<")[0] in KERNELS:
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for (k, c), (s, n) in acc.items():
        out[k][c] = round(s / n, 1)
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_read_MB"] = round(v["FETCH_SIZE"] * 2 * 1024 / 1e6, 1)
        v["hbm_write_MB"] = round(v["WRITE_SIZE"] * 1024 / 1e6, 1)
        if "avg_us" in v:
            v["hbm_TBps"] = round((v["hbm_read_MB"] + v["hbm_write_MB"]) / v["avg_us"], 3)      # MB / us = TB/s
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        w = v["SQ_WAVE_CYCLES"]
        v["wave_state"] = {"waiting_any": round(v.get("SQ_WAIT_ANY", 0) / w, 3), "issue_stalled": round(v.get("SQ_WAIT_INST_ANY", 0) / w, 3),
                           "issuing": round(v.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)}
json.dump(out, open(os.path.join(ROOT, "gpurun_out/build_pmc.json"), "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("avg_us", 0)):
    print(k, v.get("avg_us"), "us", {x: v[x] for x in ("hbm_read_MB", "hbm_write_MB", "hbm_TBps", "wave_state") if x in v})
