"""CPU restatement of the per-frame dual match of dvo_slam::LocalTracker::update -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.  PARITY UNPINNED.
Follows local_tracker.cpp:157-191 (the two match() calls and their initial transforms) and the quantities the accept
callbacks of KeyframeTracker read from the two results (keyframe_tracker.cpp:105-190).
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc


def track_frame(cfg, keyframe, last_frame, frame, last_keyframe_pose=None):
    P = np.eye(4) if last_keyframe_pose is None else np.asarray(last_keyframe_pose, dtype=np.float64)
    init_kf = np.eye(4)  # last_keyframe_pose_.inverse(Eigen::Isometry), local_tracker.cpp:173
    init_kf[:3, :3] = P[:3, :3].T
    init_kf[:3, 3] = -P[:3, :3].T @ P[:3, 3]
    r_keyframe = orc.match(cfg, keyframe, frame, init_kf)   # :179
    r_odometry = orc.match(cfg, last_frame, frame, np.eye(4))  # :172,180
    return r_keyframe, r_odometry, criteria(r_keyframe, r_odometry)


def criteria(r_keyframe, r_odometry):
    """what the accept callbacks read from the two results.  A result: dict with T, information, loglik, is_nan and either
    `levels` (match()'s) or `constraint_ratio` (a continuation's summary, tests/fork_criterion.py)"""
    with np.errstate(all="ignore"):
        if "constraint_ratio" in r_keyframe:
            ratio = r_keyframe["constraint_ratio"]
        else:
            L = r_keyframe["levels"][-1]
            ratio = float(np.float64(L["iterations"][-1]["valid_constraints"]) / np.float64(L["valid_pixels"]))  # :165
        crit = dict(
            odometry_is_nan=bool(r_odometry.get("is_nan", False)), keyframe_is_nan=bool(r_keyframe.get("is_nan", False)),
            odometry_translation_norm=float(np.linalg.norm(r_odometry["T"][:3, 3])),      # keyframe_tracker.cpp:135
            keyframe_translation_norm=float(np.linalg.norm(r_keyframe["T"][:3, 3])),      # :135,160
            keyframe_constraint_ratio=ratio,
            odometry_neg_loglik=-r_odometry["loglik"], keyframe_neg_loglik=-r_keyframe["loglik"],  # :108-110
        )
        for name, r in (("odometry", r_odometry), ("keyframe", r_keyframe)):  # :172-184
            ev = np.sort(np.linalg.eigvalsh(r["information"])) if np.isfinite(r["information"]).all() else np.full(6, np.nan)
            crit[name + "_condition_number"] = float(abs(ev[5] / ev[0]))
    return crit
