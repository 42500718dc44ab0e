"""Turns the rocprofv3 outputs of a bench.py run into the summaries committed under profiles/.

  python scripts/profile_summary.py agreement <kernel_trace.csv> <bench.json of the same command, un-profiled> <skip_tail> > out.txt
      average k_tick duration over the launches of the timed region (rocprofv3 kernel trace) next to the figure bench.py
      measured live with dispatch time stamps (hipExtLaunchKernel start / stop events)
  python scripts/profile_summary.py traffic <pmc FETCH_SIZE csv> <pmc WRITE_SIZE csv> <bench.json of the pmc run> > out.json
      HBM-side traffic per k_tick launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
"""
import csv
import json
import sys


def k_tick_rows(path):
    # the batch form only ("k_tick<...>"): single match() calls (k_tick_small: the latency probe, the timed-region check) launch
    # behind the small argument block and are not what the timed region runs
    rows = [r for r in csv.DictReader(open(path)) if "k_tick<" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def main():
    mode = sys.argv[1]
    if mode == "agreement":
        rows = k_tick_rows(sys.argv[2])
        bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        skip_tail = int(sys.argv[4])  # launches after the timed region (isolated-kernel micro-benchmark: 1 warm-up + reps)
        n = bench["roofline"]["launches"]
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        timed = d[len(d) - skip_tail - n: len(d) - skip_tail]
        print(f"k_tick dispatches in the trace: {len(d)}, average over all of them {sum(d) / len(d):.2f} us")
        print(f"launches of the timed region ({n}, the ones bench.py times): rocprofv3 average {sum(timed) / len(timed):.2f} us")
        print(f"bench.py, same command without the profiler, dispatch-stamped HIP events: {bench['roofline']['avg_launch_us']:.2f} us")
        print(f"ratio {bench['roofline']['avg_launch_us'] / (sum(timed) / len(timed)):.3f}")
    elif mode == "traffic":
        out = {}
        bench = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
        # Like for like: bench.py's algorithmic bytes per launch are those of its single-stream timing pass, the LAST
        # roofline.launches batch-form dispatches of the run (--no-extras: nothing of that form follows); the counters are
        # averaged over exactly those dispatches.  The average over every dispatch of the run (streaming region, warm-up,
        # priming included, whose launches carry other mixes of levels) is kept beside it.
        n_timed = int(bench["roofline"]["launches"])
        for name, path in (("FETCH_SIZE", sys.argv[2]), ("WRITE_SIZE", sys.argv[3])):
            rows = [r for r in csv.DictReader(open(path))
                    if "k_tick<" in r["Kernel_Name"] and r["Counter_Name"] == name]  # (the batch form: see k_tick_rows)
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            v = [float(r["Counter_Value"]) for r in rows]
            out[name + "_kb_avg_per_launch"] = sum(v[-n_timed:]) / n_timed
            out[name + "_kb_avg_over_every_dispatch_of_the_run"] = sum(v) / len(v)
            out[name + "_dispatches"] = len(v)
        out["launches_averaged"] = n_timed
        # FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read on gfx950: doubled (upper bound for this kernel,
        # whose reads are a mix of 16-byte gathers and 4-byte streaming loads); WRITE_SIZE is exact for streaming stores
        out["traffic_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_kb_avg_per_launch"] + out["WRITE_SIZE_kb_avg_per_launch"]) * 1024.0
        out["traffic_bytes_per_launch_uncorrected"] = (out["FETCH_SIZE_kb_avg_per_launch"] + out["WRITE_SIZE_kb_avg_per_launch"]) * 1024.0
        out["command"] = ("python bench.py --steps 2 --warmup 1 --prime 1 --no-extras --no-cpu-baseline under rocprofv3 --pmc "
                          "FETCH_SIZE and --pmc WRITE_SIZE (one pass each)")
        out["bench_value_under_pmc"] = bench["value"]
        out["alg_bytes_per_launch"] = bench["roofline"]["alg_bytes_per_launch"]
        out["note"] = ("average over the k_tick dispatches of the run's single-stream timing pass, the launches alg_bytes_per_launch "
                       "describes (default workload: every pair of a step a different "
                       "(keyframe, frame) combination, 108 pyramids = 2.1 GB, far beyond the 256 MiB Infinity Cache).  FETCH_SIZE "
                       "counts 64 B per 128-B request of a wide coalesced read on gfx950, so the corrected figure (2 x FETCH + "
                       "WRITE) is an upper bound for this kernel's mix of 16-byte gathers and 4-byte streaming loads and the "
                       "uncorrected one a lower bound; alg_bytes_per_launch is the algorithmic figure (56 B per selected pixel) of "
                       "the single-stream timing pass of the same run.  The algorithmic figure lies between the two bounds: no wasted "
                       "re-reads; part of the residual spill / re-read (16 of the 56 B) stays in L2 / Infinity Cache.")
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
