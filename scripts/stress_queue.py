"""Soak of the queue API (dvo_amd_match_submit / _wait / _poll): several host threads, one tracker each, random batch sizes
queued behind each other (up to three submissions outstanding per tracker), completed by wait() or by poll() loops in random
order, pyramids created per round and dropped by the caller right after submitting (the queue holds its own references),
statistics on and off.  Every result must be finite and BIT-IDENTICAL to the single-pair result of the same pair (a pair's
result is a function of its inputs alone since round 4; until then the bar was 3e-4 and the worst seen 5.76e-5).
usage: stress_queue.py [seconds] [threads]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from dvo_slam_amd import capi, synth

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
W, H = 640, 480
K = synth.intrinsics_for(W, H)
SCALES = (0.0, 0.6, -0.7, 0.9, -1.1, 0.3)
CFG = dict(FirstLevel=3, LastLevel=0)


def make_frames():
    return [capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * s), frame_id=i), K, 4)
            for i, s in enumerate(SCALES)]


frames0 = make_frames()
single = capi.DenseTracker(capi.Config(**CFG))
truth = {(a, b): single.match(frames0[a], frames0[b]).Transformation for a in range(len(SCALES)) for b in range(len(SCALES))}
errors = []
stats = {"submissions": 0, "pairs": 0, "worst": 0.0, "polls": 0}
lock = threading.Lock()
t_end = time.time() + BUDGET


def check(idx, sub, raw_results):
    worst = 0.0
    for (a, b), o in zip(idx, raw_results):
        T = np.array(o.transformation[:]).reshape(4, 4).T
        if o.is_nan or not np.isfinite(T).all():
            errors.append(f"NaN result for pair {(a, b)} of a submission of {len(idx)}")
            return
        if not np.array_equal(truth[(a, b)], T):
            worst = max(worst, synth.pose_error(truth[(a, b)], T), 1e-300)
    if worst > 0.0:
        errors.append(f"pose differs by {worst:.2e} from the single match() in a submission of {len(idx)}")
        return
    with lock:
        stats["submissions"] += 1
        stats["pairs"] += len(idx)
        stats["worst"] = max(stats["worst"], worst)


def worker(t, seed):
    r = np.random.default_rng(seed)
    trk = capi.DenseTracker(capi.Config(**CFG))
    in_flight = int(r.choice([8, 36, 96]))
    outstanding = []
    frames = frames0
    while time.time() < t_end and not errors:
        if r.random() < 0.1:  # fresh pyramids: the previous ones are dropped while submissions that use them may still be queued
            frames = make_frames()
        n = int(r.integers(1, 150))
        idx = [(int(r.integers(0, len(frames))), int(r.integers(0, len(frames)))) for _ in range(n)]
        sub = trk.submit([frames[a] for a, _ in idx], [frames[b] for _, b in idx], stats=bool(r.random() < 0.5),
                         in_flight=in_flight)
        outstanding.append((idx, sub))
        while len(outstanding) >= 3 or (outstanding and r.random() < 0.3):
            k = int(r.integers(0, len(outstanding)))  # not necessarily the oldest
            idx_k, sub_k = outstanding.pop(k)
            if r.random() < 0.5:
                n_polls = 0
                while not trk.poll(sub_k):
                    n_polls += 1
                with lock:
                    stats["polls"] += n_polls
                check(idx_k, sub_k, sub_k.results(raw=True))
            else:
                check(idx_k, sub_k, trk.wait(sub_k, raw=True))
    for idx_k, sub_k in outstanding:
        check(idx_k, sub_k, trk.wait(sub_k, raw=True))
    trk.wait()


n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ths = [threading.Thread(target=worker, args=(t, 77 + t)) for t in range(n_threads)]
t0 = time.time()
for x in ths:
    x.start()
while any(x.is_alive() for x in ths):  # a progress line every half minute (a silent run is taken to be hung)
    ths[0].join(timeout=30.0)
    with lock:
        print(f"[{time.time() - t0:6.1f} s] submissions {stats['submissions']}, pairs {stats['pairs']}, polls {stats['polls']}, "
              f"worst deviation from the single-pair result {stats['worst']:.2e}", flush=True)
for x in ths:
    x.join()
print("threads", n_threads, "errors:", errors)
sys.exit(1 if errors else 0)
