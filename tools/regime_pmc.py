"""Condense tools/regime_pmc.sh's rocprofv3 outputs (gpurun_out/rpmc_*) into gpurun_out/regime_pmc.json: per regime and kernel
the average duration, HBM traffic (FETCH_SIZE x2 as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE), L2 hit rate and
the SQ counters, per dispatch."""
import collections
import csv
import glob
import json
import sys

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
KERNELS = ("k_splat_tiled", "k_splat_reduce_long", "k_splat_reduce", "k_blur2", "k_slice_tiled")


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def kname(n):
    s = short(n)
    for k in KERNELS:
        if s.startswith(k):
            return s.split("<")[0] if k != "k_splat_tiled" else s
    return None


out = {"workload": wl, "note": "rocprofv3 per-dispatch means; FETCH_SIZE in KB doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE in KB; "
       "SQ_* counters count quad-cycles summed over waves (WAVE_CYCLES, WAIT_*, ACTIVE_INST_*); separate --pmc passes", "regimes": {}}
for tag in ("default", "tsu_1_1", "tsu_08_03"):
    reg = collections.defaultdict(dict)
    for f in glob.glob(f"gpurun_out/rpmc_{tag}_stats/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            k = kname(r["Name"])
            if k:
                reg[k]["avg_us"] = round(float(r["AverageNs"]) / 1e3, 2)
                reg[k]["calls"] = int(r["Calls"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/rpmc_{tag}_p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        d = reg[k]
        if "FETCH_SIZE" in m:
            d["hbm_read_MB"] = round(m["FETCH_SIZE"] * 2048 / 1e6, 1)
        if "WRITE_SIZE" in m:
            d["hbm_write_MB"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 1)
        if "TCC_HIT_sum" in m:
            d["L2_hit_rate"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0)), 3)
        for c, v in sorted(m.items()):
            if c.startswith("SQ_"):
                d[c] = int(v)
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            w = m["SQ_WAVE_CYCLES"]
            d["frac_wave_parked(WAIT_ANY)"] = round(m.get("SQ_WAIT_ANY", 0) / w, 3)
            d["frac_issue_stall(WAIT_INST_ANY)"] = round(m.get("SQ_WAIT_INST_ANY", 0) / w, 3)
            d["frac_issuing(ACTIVE_INST_ANY)"] = round(m.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
        if "avg_us" in d and "hbm_read_MB" in d:
            d["hbm_GBps"] = round((d["hbm_read_MB"] + d.get("hbm_write_MB", 0)) / d["avg_us"] * 1e3 / 1e3, 1)
    out["regimes"][tag] = reg
json.dump(out, open("gpurun_out/regime_pmc.json", "w"), indent=1)
for tag, reg in out["regimes"].items():
    for k, d in sorted(reg.items()):
        print(tag, k, {x: d[x] for x in ("avg_us", "hbm_read_MB", "hbm_write_MB", "L2_hit_rate", "hbm_GBps", "frac_wave_parked(WAIT_ANY)", "frac_issuing(ACTIVE_INST_ANY)") if x in d})
