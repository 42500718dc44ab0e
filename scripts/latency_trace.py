"""Where a single-pair tick goes: reads a rocprofv3 kernel trace of scripts/latency_r02.py and prints, per pyramid level of the
launch (by grid size), the k_tick duration, the gap to k_finalize, its duration, and the gap to the next k_tick (host turn-around).
usage: latency_trace.py kernel_trace.csv"""
import csv, sys, collections, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_tick" in r["Kernel_Name"] or "k_finalize" in r["Kernel_Name"]]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "T" if "k_tick" in r["Kernel_Name"] else "F", int(r["Grid_Size_X"])) for r in rows)
ks = ks[len(ks) // 2:]  # the timed half
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(len(ks) - 2):
    a, b, c = ks[i], ks[i + 1], ks[i + 2]
    if a[2] == "T" and b[2] == "F" and c[2] == "T":
        g = a[3]
        acc[g]["tick"].append(a[1] - a[0]); acc[g]["gap_tf"].append(b[0] - a[1]); acc[g]["fin"].append(b[1] - b[0]); acc[g]["turnaround"].append(c[0] - b[1])
print("grid_x  n   k_tick  gap  k_finalize  to-next-k_tick  sum (us, medians)")
tot = 0; n = 0
for g in sorted(acc):
    m = {k: statistics.median(v) / 1e3 for k, v in acc[g].items()}
    s = sum(m.values())
    print(f"{g:7d} {len(acc[g]['tick']):5d} {m['tick']:7.2f} {m['gap_tf']:5.2f} {m['fin']:7.2f} {m['turnaround']:9.2f} {s:8.2f}")
