"""Turns the rocprofv3 outputs of a bench.py run into the summaries committed under profiles/.

  python scripts/profile_summary.py agreement <kernel_trace.csv> <bench.json of the same command, un-profiled> > out.txt
      average k_tick duration over the launches of bench.py's timing pass (rocprofv3 kernel trace) next to the figure bench.py
      measured live with dispatch time stamps (hipExtLaunchKernel start / stop events)
  python scripts/profile_summary.py traffic <pmc FETCH_SIZE csv> <pmc WRITE_SIZE csv> <bench.json of the FETCH pass> [<bench.json of the WRITE pass>] > out.json
      HBM-side traffic per k_tick launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950

Both select the batch-form k_tick dispatches BETWEEN the two k_marker dispatches that bracket bench.py's timing pass
(dvo_slam_amd/pmc.py) -- not "the last N of the run" (21 launches of the isolated level-0 micro-benchmark used
to be averaged into the traffic figure that way).
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import pmc  # noqa: E402


def main():
    mode = sys.argv[1]
    if mode == "agreement":
        rows = list(csv.DictReader(open(sys.argv[2])))
        bench = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
        n = bench["roofline"]["launches"]
        timed = pmc.between_markers(rows, "Start_Timestamp")
        assert timed is not None and len(timed) == n, (None if timed is None else len(timed), n)
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in timed]
        every = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_tick<" in r["Kernel_Name"]]
        print(f"k_tick dispatches in the trace: {len(every)}, average over all of them {sum(every) / len(every):.2f} us")
        print(f"launches of bench.py's timing pass ({n}, between the two k_marker dispatches): rocprofv3 average {sum(d) / len(d):.2f} us")
        print(f"bench.py, same command without the profiler, dispatch-stamped HIP events: {bench['roofline']['avg_launch_us']:.2f} us")
        print(f"ratio {bench['roofline']['avg_launch_us'] / (sum(d) / len(d)):.3f}")
    elif mode == "traffic":
        bench = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
        bench_w = json.loads(open(sys.argv[5]).read().strip().splitlines()[-1]) if len(sys.argv) > 5 else None
        out = pmc.traffic_summary(sys.argv[2], sys.argv[3], bench, bench_line_write=bench_w,
                                  command="python bench.py --steps 2 --warmup 1 --prime 1 --no-extras --no-cpu-baseline --no-live-counters "
                                          "under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (one pass each)")
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
