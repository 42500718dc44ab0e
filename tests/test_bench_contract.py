"""bench.py's one-line contract with the driver, checked on the GPU with a small workload: the fields the driver parses, the
`roofline` and `cpu_baseline` objects, and the internal consistency of the line (value from its own ms_per_step, frac from
achieved / peak, traffic only for the workload the counters ran)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_carries_what_the_driver_and_the_judge_read():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "144",
                          "--no-extras", "--cpu-seconds", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    for key, want in (("unit", "frame-pairs/s"), ("n_gpus", 1), ("steps", 3), ("warmup", 1), ("higher_is_better", True),
                      ("scaling", "weak"), ("vs_baseline", None), ("dtype", "f32")):
        assert line[key] == want, key
    assert line["metric"].startswith("frame-pairs/s (640x480, 4-level GN align)")
    assert "synthetic" in line["data"] and "workload" in line["config"] and "model" not in line["config"]
    # value = pairs of the timed region / its time
    assert line["value"] == pytest.approx(144 * 3 / (line["ms_per_step"] * 3e-3), rel=1e-6)
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9) and 0.02 < r["frac"] < 0.8
    assert r["launches"] > 0 and r["avg_launch_us"] > 0
    assert r["achieved"] == pytest.approx(r["alg_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9, rel=1e-6)
    assert r["traffic"] is None and r["traffic_source"] is None  # 144 pairs per step is not the workload the counters ran
    assert 0.0 <= r["speculation_waste"]["discarded_fraction_of_submitted_bytes"] < 0.2
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "frame-pairs/s" and c["value"] > 0 and c["sample"]
    assert line["value"] > 50 * c["value"]  # a GPU that is not two orders above one CPU core is not running the HIP path


@pytest.mark.gpu
def test_counter_passes_select_the_timing_pass_and_measure_its_traffic():
    """roofline.traffic is a property of the run that prints it (round 5): bench.py brackets its timing pass with two k_marker
    dispatches and starts the rocprofv3 counter passes itself (dvo_slam_amd/pmc.py).  Here the same machinery on a small workload:
    the three child passes run, the summaries find exactly the k_tick dispatches bench.py timed (every form of the kernel between the
    markers), the writes are what the algorithm writes plus the unselected pixels' markers and the block records, and the calibrated
    traffic is of the size of the algorithmic bytes -- no wasted re-reads."""
    import shutil

    sys.path.insert(0, ROOT)
    from dvo_slam_amd import pmc

    if pmc.rocprofv3_path() is None or not shutil.which("python3"):
        pytest.skip("rocprofv3 is not installed on this box")
    leg = ["--counter-leg", "--steps", "1", "--warmup", "0", "--prime", "1", "--batch", "288", "--threads", "2", "--in-flight", "72"]
    live = pmc.measure_live(os.path.join(ROOT, "bench.py"), leg, timeout_s=300.0)
    t = live["traffic"]
    assert t["launches_averaged"] > 10 and t["launches_averaged"] == t["launches_of_the_write_pass"]
    assert 1.0 <= t["write_ratio_to_algorithmic"] <= 1.6, t["write_ratio_to_algorithmic"]
    lo, hi = t["wasted_traffic_ratio_bounds"]
    assert 0.3 < lo <= t["wasted_traffic_ratio_calibrated"] <= hi < 2.0, (lo, t["wasted_traffic_ratio_calibrated"], hi)
    assert t["traffic_bytes_per_launch"] == pytest.approx((2 * t["FETCH_SIZE_kb_avg_per_launch"] + t["WRITE_SIZE_kb_avg_per_launch"]) * 1024)
    i = live["issue"]
    assert "error" not in i, i
    assert i["mfma_wave_instructions"] > 0 and i["valu_wave_instructions"] > 5 * i["mfma_wave_instructions"]
