"""How many k_tick kernels run at the same time, and how busy each hardware queue is, from a rocprofv3 kernel trace of the
bench (second half of the trace = the timed region).  usage: concurrency_from_trace.py bench_kernel_trace.csv"""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "T" if "k_tick" in r["Kernel_Name"] else "F", r["Queue_Id"], r["Stream_Id"])
            for r in rows if "k_tick" in r["Kernel_Name"] or "k_finalize" in r["Kernel_Name"])
t0 = ks[len(ks) // 2][0]
ks = [k for k in ks if k[0] >= t0]
span = ks[-1][1] - ks[0][0]
ev = []
for s, e, tag, q, st in ks:
    ev.append((s, 1, tag)); ev.append((e, -1, tag))
ev.sort()
cur = {"T": 0, "F": 0}; last = ks[0][0]; hist = collections.Counter(); none = 0; only_f = 0
for t, d, tag in ev:
    hist[cur["T"]] += t - last
    if cur["T"] == 0 and cur["F"] == 0: none += t - last
    if cur["T"] == 0 and cur["F"] > 0: only_f += t - last
    cur[tag] += d; last = t
tot = sum(hist.values())
print(f"{len(ks)} kernels over {span / 1e6:.1f} ms")
print("k_tick kernels running at once -> share of the time:", ", ".join(f"{k}: {hist[k] / tot:.3f}" for k in sorted(hist)))
print(f"mean {sum(k * v for k, v in hist.items()) / tot:.2f}; nothing running {none / tot:.3f}; only k_finalize running {only_f / tot:.3f}")
byq = collections.defaultdict(list)
for k in ks: byq[k[3]].append(k)
for q, l in sorted(byq.items()):
    busy = sum(e - s for s, e, *_ in l)
    print(f"hardware queue {q}: {len(set(k[4] for k in l))} streams, {len(l)} kernels, busy {busy / span:.3f} of the time")
