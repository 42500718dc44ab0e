"""Summaries of rocprofv3 runs of bench.py: which dispatches belong to the timing pass, HBM-side traffic per k_tick launch.

bench.py brackets its single-stream timing pass -- the launches `roofline.launches`, `roofline.avg_launch_us` and
`roofline.alg_bytes_per_launch` describe -- with two no-op dispatches named `k_marker` (dvo_amd_debug_marker).  Everything
here selects the k_tick dispatches BETWEEN the two markers, so a figure quoted per launch is a figure of exactly
those launches (until round 4 the counter summary took "the last N k_tick dispatches of the run", and 21 launches of an
isolated level-0 micro-benchmark that ran behind the timing pass were averaged in: VERDICT round 4).

Used by scripts/profile_summary.py (the committed profiles/) and by bench.py itself, which runs the two counter passes as
child processes of the very run that prints the bench line (`roofline.traffic_source.measured = "in this run"`).
"""
from __future__ import annotations

import csv
import json
import os
import shutil
import subprocess
import sys
import tempfile

MARKER = "k_marker"
# /opt/skills/guides/MI355X_MICROARCH.md, "HBM": on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
# streaming read (16 B per lane; 128-byte requests tallied at 64 B) -- double it; WRITE_SIZE is exact for streaming stores; other
# access widths are uncalibrated.  k_tick's reads are a mix: 4-byte-per-lane streaming loads of the reference planes (16 of the
# 56 algorithmic bytes per pixel), 16-byte and 8-byte gathers of the current planes (24), 8-byte-per-lane streaming reads of the
# spilled residuals (8).  scripts/probes/fetch_calibration.hip measured the factors on this part (profiles/r05_fetch_calibration.txt):
# exactly 2.00 for 4-, 8- and 16-byte-per-lane STREAMING reads alike; >= 1.30 for k_tick's GATHER pattern (a mix of 128-byte and
# 64-byte requests).  The raw bounds stay 1 x FETCH (no request under-counted) and 2 x FETCH (every request is: the guide's
# correction, what `roofline.traffic` reports); fetch_model() below splits the counter into its streaming and gather parts.
FETCH_FACTOR_BOUNDS = (1.0, 2.0)
STREAMING_FACTOR = 2.0
GATHER_FACTOR_CALIBRATED = 1.30
# of the 56 algorithmic bytes per selected pixel: 16 reference planes + 8 residual re-read are streaming reads, 24 are gathers
ALG_STREAMING_READ, ALG_GATHER_READ, ALG_WRITE, ALG_TOTAL = 24.0, 24.0, 8.0, 56.0


def fetch_model(fetch_counted_bytes: float, alg_bytes: float) -> dict:
    """true fetch bytes = S + g (C - S / 2): S the streaming bytes of the launch (every one read exactly once, counted at half),
    C the counter, g in [1, 2] the true / counted ratio of the gathers (1.30 for the calibrated pattern)"""
    S = alg_bytes * ALG_STREAMING_READ / ALG_TOTAL
    gathers_counted = max(0.0, fetch_counted_bytes - S / STREAMING_FACTOR)
    return {"streaming_bytes_assumed": S, "gather_bytes_counted": gathers_counted,
            "fetch_bytes_bounds": [S + 1.0 * gathers_counted, S + 2.0 * gathers_counted],
            "fetch_bytes_calibrated": S + GATHER_FACTOR_CALIBRATED * gathers_counted,
            "alg_gather_bytes": alg_bytes * ALG_GATHER_READ / ALG_TOTAL,
            "calibration": "profiles/r05_fetch_calibration.txt: streaming reads of 4 / 8 / 16 B per lane are counted at exactly 1/2, "
                           "k_tick's gather pattern at 1/1.30 or less"}


def _is_tick(name: str) -> bool:
    # the batch form only ("k_tick<...>"): single match() calls (k_tick_small: the latency probe, the timed-region check) launch
    # behind the small argument block and are not what the timed region runs
    return "k_tick<" in name


def between_markers(rows, order_key):
    """rows: dicts with Kernel_Name; order_key: the column that orders dispatches (Dispatch_Id or Start_Timestamp).  Returns the
    k_tick rows between the FIRST and the LAST k_marker dispatch, or None when the run carries fewer than two markers."""
    rows = sorted(rows, key=lambda r: int(r[order_key]))
    marks = [i for i, r in enumerate(rows) if MARKER in r["Kernel_Name"]]
    if len(marks) < 2:
        return None
    # (every form of the kernel: the tail of the timing pass drains through ticks of at most eight pairs, which go out as
    #  k_tick_small; bench.py's launch count and algorithmic bytes include them)
    return [r for r in rows[marks[0] + 1: marks[-1]] if "k_tick" in r["Kernel_Name"]]


def counter_per_launch(csv_path: str, counter: str):
    """-> (values of `counter` for the k_tick dispatches of the timing pass, values for every k_tick dispatch of the run)"""
    rows = [r for r in csv.DictReader(open(csv_path)) if r.get("Counter_Name") in (counter, None, "")]
    every = [float(r["Counter_Value"]) for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])) if _is_tick(r["Kernel_Name"])]
    sel = between_markers(rows, "Dispatch_Id")
    if sel is None:
        raise RuntimeError(f"{csv_path}: no pair of {MARKER} dispatches (a bench.py older than round 5?)")
    # a counter is reported once per dispatch and XCC / dimension: sum the rows of a dispatch
    per = {}
    for r in sel:
        per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [per[k] for k in sorted(per)], every


def traffic_summary(fetch_csv: str, write_csv: str, bench_line: dict, command: str | None = None,
                    bench_line_write: dict | None = None) -> dict:
    """HBM-side bytes per k_tick launch of the timing pass of the run that wrote the two counter files (one rocprofv3 --pmc pass
    each: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2)."""
    n_timed = int(bench_line["roofline"]["launches"])
    alg = float(bench_line["roofline"]["alg_bytes_per_launch"])
    out = {}
    for name, path, line in (("FETCH_SIZE", fetch_csv, bench_line), ("WRITE_SIZE", write_csv, bench_line_write or bench_line)):
        sel, every = counter_per_launch(path, name)
        if len(sel) != int(line["roofline"]["launches"]) or len(sel) != n_timed:
            raise RuntimeError(f"{name}: {len(sel)} k_tick dispatches between the markers, bench.py timed "
                               f"{line['roofline']['launches']} in that pass ({n_timed} in the FETCH_SIZE pass)")
        out[name + "_kb_avg_per_launch"] = sum(sel) / len(sel)
        out[name + "_kb_avg_over_every_dispatch_of_the_run"] = sum(every) / max(1, len(every))
        out[name + "_dispatches"] = len(every)
    out["launches_averaged"] = n_timed
    out["selection"] = f"the k_tick dispatches (batch and small-argument form) between the two {MARKER} dispatches that bracket bench.py's timing pass"
    f_kb, w_kb = out["FETCH_SIZE_kb_avg_per_launch"], out["WRITE_SIZE_kb_avg_per_launch"]
    lo, hi = FETCH_FACTOR_BOUNDS
    out["fetch_bytes_per_launch_bounds"] = [lo * f_kb * 1024.0, hi * f_kb * 1024.0]
    out["write_bytes_per_launch"] = w_kb * 1024.0
    out["traffic_bytes_per_launch"] = (hi * f_kb + w_kb) * 1024.0            # upper bound: every read request under-counted
    out["traffic_bytes_per_launch_uncorrected"] = (lo * f_kb + w_kb) * 1024.0  # lower bound: none is
    out["alg_bytes_per_launch"] = alg
    # algorithmic split of the 56 B per selected pixel: 40 read (16 reference planes + 24 current planes) + 8
    # residual spill written + 8 read back by the likelihood pass of the next tick
    out["alg_write_bytes_per_launch"] = alg * 8.0 / 56.0
    out["alg_read_bytes_per_launch"] = alg * 48.0 / 56.0
    out["write_ratio_to_algorithmic"] = out["write_bytes_per_launch"] / out["alg_write_bytes_per_launch"]
    out["read_ratio_to_algorithmic_bounds"] = [b / out["alg_read_bytes_per_launch"] for b in out["fetch_bytes_per_launch_bounds"]]
    out["wasted_traffic_ratio_bounds"] = [out["traffic_bytes_per_launch_uncorrected"] / alg, out["traffic_bytes_per_launch"] / alg]
    fm = fetch_model(f_kb * 1024.0, alg)
    out["fetch_model"] = fm
    out["traffic_bytes_per_launch_calibrated"] = fm["fetch_bytes_calibrated"] + out["write_bytes_per_launch"]
    out["wasted_traffic_ratio_calibrated"] = out["traffic_bytes_per_launch_calibrated"] / alg
    out["wasted_traffic_ratio_bounds_calibrated"] = [(b + out["write_bytes_per_launch"]) / alg for b in fm["fetch_bytes_bounds"]]
    out["bench_value_under_pmc"] = bench_line.get("value")
    if command:
        out["command"] = command
    out["note"] = ("Counters of exactly the launches alg_bytes_per_launch describes.  WRITE_SIZE is exact for streaming stores: "
                   "write_ratio_to_algorithmic is the residual spill (8 B per selected pixel) plus the block records and likelihood "
                   "partials.  FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read on gfx950 (MI355X_MICROARCH.md), so "
                   "the read side is bracketed by 1 x and 2 x FETCH_SIZE; wasted_traffic_ratio_bounds = (bounds of FETCH + WRITE) / "
                   "algorithmic bytes: a ratio well above 1 would mean wasted re-reads, below 1 that part of the algorithmic bytes "
                   "(neighbouring gathers, the residual re-read) never left L2 / the Infinity Cache.  fetch_model splits the counter "
                   "with the calibrated factors (streaming reads exactly 2, the gather pattern >= 1.30): "
                   "wasted_traffic_ratio_calibrated is the best estimate, wasted_traffic_ratio_bounds_calibrated its bounds for gather "
                   "factors of 1 and 2.  The writes exceed the algorithmic 8 B per SELECTED pixel because the spill covers every pixel "
                   "of the level (NaN marks the ~10 % unselected ones) and every block adds a 416-byte record.")
    return out


def rocprofv3_path() -> str | None:
    for cand in (shutil.which("rocprofv3"), "/opt/rocm/bin/rocprofv3"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _counter_pass(prof, counters, bench_py, leg_args, tmp, env, timeout_s):
    """one `rocprofv3 --pmc <counters> -- python3 bench.py <leg_args>` child: (bench line it printed, counter csv path)"""
    d = os.path.join(tmp, "_".join(counters))
    cmd = [prof, "--pmc"] + list(counters) + ["-d", d, "-o", "pmc", "--output-format", "csv", "--", sys.executable, bench_py] + leg_args
    res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
    if res.returncode != 0:
        raise RuntimeError(f"{counters} pass failed (rc {res.returncode}): {res.stderr[-600:]}")
    out_lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if not out_lines:
        raise RuntimeError(f"{counters} pass printed no bench line: {res.stdout[-300:]} {res.stderr[-300:]}")
    found = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
    if not found:
        raise RuntimeError(f"{counters} pass left no counter_collection.csv under {d}")
    return json.loads(out_lines[-1]), found[0]


def issue_summary(csv_path: str, bench_line: dict, valu_issue_cycles=4, mfma_issue_cycles=8, simds=1024, clock_hz=2.4e9) -> dict:
    """VALU + MFMA issue cycles of the timing pass's k_tick launches from SQ_INSTS_VALU / SQ_INSTS_MFMA of those very dispatches
    (wave instructions; SQ_INSTS_VALU counts the MFMA instructions too), against every SIMD issuing every cycle for the launches'
    own duration as the UN-profiled run measured it (the caller passes its k_tick milliseconds: counters slow the kernels down)."""
    valu, _ = counter_per_launch(csv_path, "SQ_INSTS_VALU")
    mfma, _ = counter_per_launch(csv_path, "SQ_INSTS_MFMA")
    n = int(bench_line["roofline"]["launches"])
    if len(valu) != n or len(mfma) != n:
        raise RuntimeError(f"issue counters: {len(valu)} / {len(mfma)} k_tick dispatches between the markers, bench.py timed {n}")
    plain = sum(valu) - sum(mfma)
    cyc = plain * valu_issue_cycles + sum(mfma) * mfma_issue_cycles
    return {"bound": "VALU+MFMA issue", "valu_wave_instructions": plain, "mfma_wave_instructions": sum(mfma), "launches": n,
            "valu_issue_cycles": valu_issue_cycles, "mfma_issue_cycles": mfma_issue_cycles, "issue_cycles": cyc,
            "simds": simds, "peak_clock_hz": clock_hz,
            "measured": "SQ_INSTS_VALU / SQ_INSTS_MFMA of the timing pass's own k_tick dispatches, one rocprofv3 --pmc pass run by "
                        "this bench (the launch time is the un-profiled run's)"}


def measure_live(bench_py: str, leg_args: list, timeout_s: float = 150.0, with_issue: bool = True) -> dict:
    """The counter passes as child processes of the caller: `rocprofv3 --pmc <counters> -- python3 bench.py <leg_args>` each (FETCH_SIZE
    and WRITE_SIZE do not fit one pass; SQ_INSTS_VALU + SQ_INSTS_MFMA do), in a scratch directory (cwd /tmp as the profiler
    wants), DVO_AMD_LAUNCH_LOCK=1 (rocprofv3's queue interceptor and several host threads:
    profiles/r03_rocprofv3_sigsegv_root_cause.md).  Returns {"traffic": traffic_summary(), "issue": issue_summary() or an error
    string}; raises if a traffic pass fails."""
    prof = rocprofv3_path()
    if prof is None:
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="dvo_pmc_", dir="/tmp")
    env = dict(os.environ, DVO_AMD_LAUNCH_LOCK="1", TMPDIR="/tmp")
    env.pop("DVO_BENCH_MAPS", None)
    try:
        line_f, csv_f = _counter_pass(prof, ("FETCH_SIZE",), bench_py, leg_args, tmp, env, timeout_s)
        line_w, csv_w = _counter_pass(prof, ("WRITE_SIZE",), bench_py, leg_args, tmp, env, timeout_s)
        summary = traffic_summary(csv_f, csv_w, line_f,
                                  command="rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (one pass each) -- python3 bench.py " + " ".join(leg_args),
                                  bench_line_write=line_w)
        summary["launches_of_the_write_pass"] = int(line_w["roofline"]["launches"])
        out = {"traffic": summary, "issue": None}
        if with_issue:
            try:
                line_i, csv_i = _counter_pass(prof, ("SQ_INSTS_VALU", "SQ_INSTS_MFMA"), bench_py, leg_args, tmp, env, timeout_s)
                out["issue"] = issue_summary(csv_i, line_i)
            except Exception as exc:  # the traffic figure does not depend on the third pass
                out["issue"] = {"error": repr(exc)[:300]}
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
