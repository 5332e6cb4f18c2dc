"""Timeline of the last warm reference-table build under `rocprofv3 --kernel-trace -- python3 tools/reftable_time.py c3`:
python tools/build_timeline.py <kernel_trace.csv>  prints start / duration / hardware queue of every launch from the last
k_elevate on, and the idle gaps of the caller's queue (profiles/r04*_build_timeline.txt)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = []
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
ks.sort()
last = max(i for i, k in enumerate(ks) if k[2].startswith("k_elevate"))
t0 = ks[last][0]
queues = {}
print("# start_us duration_us hardware_queue kernel")
prev_end = {}
gaps = []
for s, e, n, q in ks[last:]:
    qi = queues.setdefault(q, len(queues) + 1)
    if qi == 1 and qi in prev_end and s - prev_end[qi] > 8000:
        gaps.append(((prev_end[qi] - t0) / 1e3, (s - prev_end[qi]) / 1e3, n))
    prev_end[qi] = max(prev_end.get(qi, 0), e)
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{qi} {n[:40]}")
end = max(e for s, e, n, q in ks[last:])
small = [(e - s) / 1e3 for s, e, n, q in ks[last:] if queues[q] == 1 and e - s < 8000]
print(f"# whole build on the device: {(end - t0) / 1e3:.1f} us; launches under 8 us on the caller's queue: {len(small)}, {sum(small):.0f} us")
print("# idle gaps > 8 us on the caller's queue (end of previous launch, gap, next kernel):")
for g in gaps:
    print(f"#   at {g[0]:8.1f} us: {g[1]:6.1f} us before {g[2][:40]}")
