"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory: python tools/kstats.py DIR [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:52]
    print(f"{n:54s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e3:10.1f} us")
print(f"all kernels: {tot / 1e3:.1f} us")
