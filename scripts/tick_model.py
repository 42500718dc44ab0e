"""Per-launch durations of k_tick over a bench-like batch against the launch's composition (timing mode, one stream)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from dvo_slam_amd import capi, synth

W, H, B = 640, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 256
IN_FLIGHT = int(sys.argv[2]) if len(sys.argv) > 2 else 27
K = synth.intrinsics_for(W, H)
ref = capi.RgbdImagePyramid(*synth.render(W, H, None, frame_id=0), K, 4)
curs = [capi.RgbdImagePyramid(*synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * (0.6 + 0.1 * i) * (1 if i % 2 == 0 else -1)),
                                            frame_id=1 + 2 * i), K, 4) for i in range(8)]
trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
refs, cb = [ref] * B, [curs[i % 8] for i in range(B)]
trk.match_batch(refs, cb, stats=False, in_flight=IN_FLIGHT)
trk.kernel_timing(True, reset=True)
out = trk.match_batch(refs, cb, stats=False, in_flight=IN_FLIGHT)
ms, n = trk.kernel_timing(False)
log = trk.tick_log()
print(f"launches {n}, total {ms:.2f} ms, avg {ms / n * 1e3:.1f} us; alg GB {sum(o.alg_bytes for o in out) / 1e9:.2f}")
t, items, rb, lb, gx, px = log.T[:6]
A = np.stack([np.ones_like(t), px * 56 / 1e6, lb, items], 1)
coef, *_ = np.linalg.lstsq(A, t * 1e3, rcond=None)
print("fit us = %.2f + %.3f * MB_alg + %.4f * ll_blocks + %.3f * items" % tuple(coef))
print("residual rms %.2f us" % np.sqrt(np.mean((A @ coef - t * 1e3) ** 2)))
order = np.argsort(px)
for q in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
    i = order[min(int(q * (len(order) - 1)), len(order) - 1)]
    print(f"q{q:.2f}: {t[i] * 1e3:7.1f} us items {int(items[i]):3d} res_blocks {int(rb[i]):6d} ll_blocks {int(lb[i]):6d} grid.x {int(gx[i]):5d} "
          f"px {int(px[i]):9d} -> {px[i] * 56 / t[i] / 1e9:7.2f} TB/s")
print("time share by px-quartile:", [round(float(t[order[int(a * len(order)):int(b * len(order))]].sum() / t.sum()), 3)
                                    for a, b in ((0, .25), (.25, .5), (.5, .75), (.75, 1))])
# by level (lock-step batches are homogeneous): group launches by their selected pixels per residual item
per_item = np.where(items > 0, px / np.maximum(items, 1), 0)
for lo, hi, name in ((0, 1, "LL only"), (1, 6000, "L3"), (6000, 25000, "L2"), (25000, 100000, "L1"), (100000, 1e9, "L0")):
    m = (per_item >= lo) & (per_item < hi)
    if m.any():
        print(f"{name:8s}: {int(m.sum()):4d} launches, avg {t[m].mean() * 1e3:7.1f} us, time share {t[m].sum() / t.sum():.3f}, "
              f"bytes share {px[m].sum() / max(px.sum(), 1):.3f}, {px[m].sum() * 56 / t[m].sum() / 1e9:.2f} TB/s, "
              f"avg items {items[m].mean():.1f}, avg ll_blocks/item {lb[m].sum() / items[m].sum():.0f}")
