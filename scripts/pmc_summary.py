"""Summarise rocprofv3 --pmc counter_collection.csv: per (kernel, grid) averages of every counter."""
import csv, sys, collections, glob
files = sys.argv[1:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("::")[-1]
        gx = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 256)) or 256)
        gy = int(r.get("Grid_Size_Y", 1) or 1)
        key = (name, gx // max(wg, 1), gy)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    if not key[0].startswith(("k_tick", "k_fin")):
        continue
    c = acc[key]
    n = max(len(v) for v in c.values())
    print(f"{key[0]} blocks_x={key[1]} y={key[2]} dispatches={n}")
    for cn in sorted(c):
        v = c[cn]
        print(f"    {cn:28s} avg={sum(v)/len(v):16.1f}")
