"""Summarise a rocprofv3 kernel_trace.csv: per-kernel durations by grid size and the gaps between consecutive kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r.get("Grid_Size_Y", 1)))
    by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(by):
    v = by[k]
    print(f"{k[0]:40s} blocks_x={k[1]:6d} y={k[2]:3d} n={len(v):5d} avg={sum(v)/len(v):8.2f}us min={min(v):8.2f} max={max(v):8.2f}")
gaps = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    na = a["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
    nb = b["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
    gaps[(na[:12], nb[:12])].append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
for k in sorted(gaps):
    v = sorted(gaps[k])
    print(f"gap {k[0]:12s}->{k[1]:12s} n={len(v):5d} median={v[len(v)//2]:8.2f}us min={v[0]:8.2f} p90={v[int(len(v)*0.9)]:8.2f}")
