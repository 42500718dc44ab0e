"""Dynamic instruction counts of k_tick per 64-pixel wave step, for bench.py's roofline.issue (profiles/r02_issue.json).

  step 1 (under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA):  python3 scripts/issue_counts.py run OUT.json
          one batch of 72 pairs in timing mode (every launch logged with its residual / likelihood wave steps); nothing else
          launches k_tick in that process
  step 2:  python3 scripts/issue_counts.py fit OUT.json <pmc counter_collection.csv> <isolated-kernel counter csv> > r02_issue.json
          the isolated residual-pass launches give VALU and MFMA per residual step; the batch gives the likelihood step
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(out_path):
    from dvo_slam_amd import capi, synth

    W, H = 640, 480
    K = synth.intrinsics_for(W, H)
    ref = capi.RgbdImagePyramid(*synth.render(W, H, None, frame_id=0), K, 4)
    curs = [capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * (0.6 + 0.1 * i)), frame_id=1 + 2 * i), K, 4)
            for i in range(8)]
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    trk.kernel_timing(True, reset=True)
    trk.match_batch([ref] * 72, [curs[i % 8] for i in range(72)], stats=False, in_flight=36)
    ms, n = trk.kernel_timing(False)
    log = trk.tick_log()
    json.dump({"launches": int(n), "res_steps": float(log[:, 6].sum()), "ll_steps": float(log[:, 7].sum()), "ms": ms},
              open(out_path, "w"))


def counters(path, name_filter="k_tick"):
    tot, n = {}, {}
    for r in csv.DictReader(open(path)):
        if name_filter not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
    return tot, n


def fit(run_json, batch_csv, isolated_csv, isolated_steps_per_launch):
    b = json.load(open(run_json))
    iso, iso_n = counters(isolated_csv)
    launches = iso_n["SQ_INSTS_VALU"]
    tot, _ = counters(batch_csv)
    # The isolated pass walks the compacted selection (since the end of round 5): its steps per launch are no longer pixels / 64.
    # The batch logs its steps launch by launch, so the matrix instructions per step are known from it (36: nine tiles x four
    # groups; 16 for the 16x16x4 form) -- and the isolated pass's step count follows from its own matrix-instruction count.
    mfma_per = float(round(tot["SQ_INSTS_MFMA"] / max(b["res_steps"], 1.0)))
    isolated_steps_per_launch = iso["SQ_INSTS_MFMA"] / launches / mfma_per
    valu_per = (iso["SQ_INSTS_VALU"] - iso["SQ_INSTS_MFMA"]) / launches / isolated_steps_per_launch  # SQ_INSTS_VALU counts MFMA too
    valu_batch = tot["SQ_INSTS_VALU"] - tot["SQ_INSTS_MFMA"]
    ll_per = (valu_batch - valu_per * b["res_steps"]) / max(b["ll_steps"], 1.0)
    print(json.dumps({
        "valu_per_res_step": valu_per, "mfma_per_res_step": mfma_per, "valu_per_ll_step": ll_per,
        # 36 per step: the nine-tile form, v_mfma_f32_4x4x1_16b_f32, two passes = 8 issue cycles (8.9 measured back to back,
        # scripts/probes/mfma_4x4_blocks.hip); 16 per step: v_mfma_f32_16x16x4_f32, 32 cycles
        "mfma_issue_cycles": 8 if mfma_per > 24 else 32,
        "mfma_check_batch": tot["SQ_INSTS_MFMA"] / max(b["res_steps"], 1.0),
        "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA: isolated level-0 residual pass (36 pairs per launch, prologue / "
                  "epilogue amortised over the steps) and one 72-pair batch in timing mode (scripts/issue_counts.py)",
        "isolated_steps_per_launch": isolated_steps_per_launch, "batch": b}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        fit(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]) if len(sys.argv) > 5 else 36 * 307200 / 64)
