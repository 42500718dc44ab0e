"""Stress run of the batched / grouped / threaded paths: random batch sizes, residencies and thread counts for a fixed
wall-clock budget; every result must be finite and equal (to 1e-5) to the single-pair result of the same pair."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from dvo_slam_amd import capi, synth

BUDGET = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
W, H = 640, 480
K = synth.intrinsics_for(W, H)
rng = np.random.default_rng(7)
frames = [capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * s), frame_id=i), K, 4)
          for i, s in enumerate((0.0, 0.6, -0.7, 0.9, -1.1, 0.3))]
single = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
truth = {}
for a in range(len(frames)):
    for b in range(len(frames)):
        truth[(a, b)] = single.match(frames[a], frames[b]).Transformation
trackers = [capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0)) for _ in range(8)]
errors = []
stats = {"calls": 0, "pairs": 0, "worst": 0.0}
lock = threading.Lock()
t_end = time.time() + BUDGET


def worker(t, seed):
    r = np.random.default_rng(seed)
    while time.time() < t_end and not errors:
        n = int(r.integers(1, 200))
        in_flight = int(r.choice([0, 1, 2, 7, 36, 37, 72, 100, 144, 300]))
        idx = [(int(r.integers(0, len(frames))), int(r.integers(0, len(frames)))) for _ in range(n)]
        out = trackers[t].match_batch([frames[a] for a, _ in idx], [frames[b] for _, b in idx], stats=False,
                                      in_flight=in_flight, raw=True)
        worst = 0.0
        for (a, b), o in zip(idx, out):
            T = np.array(o.transformation[:]).reshape(4, 4).T
            if o.is_nan or not np.isfinite(T).all():
                errors.append(f"NaN result for pair {(a, b)} n={n} in_flight={in_flight}")
                return
            worst = max(worst, synth.pose_error(truth[(a, b)], T))
        if worst > 3e-4:
            errors.append(f"pose differs by {worst:.2e} n={n} in_flight={in_flight}")
            return
        with lock:
            stats["calls"] += 1
            stats["pairs"] += n
            stats["worst"] = max(stats["worst"], worst)


n_threads = 0
round_ = 0
t0 = time.time()
while time.time() < t_end and not errors:
    n_threads = int(rng.choice([1, 2, 4, 8]))
    ths = [threading.Thread(target=worker, args=(t, 1000 * round_ + t)) for t in range(n_threads)]
    t_round_end = time.time() + 5.0
    for x in ths:
        x.start()
    while any(x.is_alive() for x in ths):  # a progress line every half minute (a silent run is taken to be hung)
        ths[0].join(timeout=30.0)
        with lock:
            print(f"[{time.time() - t0:6.1f} s] calls {stats['calls']}, pairs {stats['pairs']}, worst {stats['worst']:.2e}", flush=True)
    for x in ths:
        x.join()
    round_ += 1
    print(f"[{time.time() - t0:6.1f} s] calls {stats['calls']}, pairs {stats['pairs']}, worst deviation from single-pair result "
          f"{stats['worst']:.2e}", flush=True)
    break  # workers run until the budget ends; one round with the drawn thread count
print("threads", n_threads, "errors:", errors)
sys.exit(1 if errors else 0)
