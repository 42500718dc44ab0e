"""Where does a block's life go, and how full are the block slots?  Needs a library built with the block trace:
    scripts/variant.sh trace -DDVO_TRACE_BLOCKS      ->  DVO_AMD_LIB=dvo_slam_amd/libdvo_amd_var_trace.so python scripts/block_trace.py
Runs (a) the streaming batch of bench.py (6 threads x 96 resident pairs, every step queued behind the previous one) and (b) the
residual pass alone (level 0, 36 pairs per launch), reads the per-block trace {start, after the first step, end} and prints: the
average number of resident k_tick blocks against the 1024 block slots of the GPU, and per kind of block the median duration, time to
the end of the first step (prologue + one step) and time per further step."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dvo_slam_amd import capi, synth

L = capi.lib()
L.dvo_amd_debug_block_trace.restype = C.c_longlong
L.dvo_amd_debug_block_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_longlong]
CAP = 1 << 22


def read_trace(trk):
    buf = np.zeros((CAP, 4), np.uint64)
    n = L.dvo_amd_debug_block_trace(trk._h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), CAP)
    if n < 0:
        raise SystemExit("this library was not built with -DDVO_TRACE_BLOCKS (scripts/variant.sh trace -DDVO_TRACE_BLOCKS)")
    return buf[: min(n, CAP)], n


def analyse(tr, label, wall_s=None):
    t0, tf, te = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2].astype(np.int64)
    info = (tr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    hw = (tr[:, 3] >> np.uint64(32)).astype(np.int64)
    steps = 1 << (info & 15)
    is_ll = (info >> 4) & 1
    width = ((info >> 8) & 255) * 8
    xcc = (hw >> 28) & 15
    span = (te.max() - t0.min()) * 10e-9  # 100 MHz clock
    dur = (te - t0) * 10e-9
    print(f"== {label}: {len(tr)} blocks over {span * 1e3:.2f} ms of GPU time" + (f" (host wall {wall_s * 1e3:.2f} ms)" if wall_s else ""))
    print(f"   resident k_tick blocks on average: {dur.sum() / span:7.1f} of 1024 slots ({dur.sum() / span / 1024:.2f})")
    # occupancy over time: sweep line at 1 us resolution
    grid = np.zeros(int(span * 1e6) + 2)
    a, b = ((t0 - t0.min()) // 100).astype(int), ((te - t0.min()) // 100).astype(int)
    np.add.at(grid, a, 1)
    np.add.at(grid, b + 1, -1)
    occ = np.cumsum(grid)[:-1]
    qs = np.percentile(occ, [5, 25, 50, 75, 95])
    print(f"   resident blocks per microsecond: p5 {qs[0]:.0f}  p25 {qs[1]:.0f}  median {qs[2]:.0f}  p75 {qs[3]:.0f}  p95 {qs[4]:.0f};  "
          f"time with < 256 resident: {np.mean(occ < 256):.2f}, < 768: {np.mean(occ < 768):.2f}, >= 960: {np.mean(occ >= 960):.2f}")
    print(f"   blocks per XCD: {[int((xcc == x).sum()) for x in range(8)]}")
    for ll in (0, 1):
        for w in sorted(set(width)):
            for s in sorted(set(steps)):
                m = (is_ll == ll) & (width == w) & (steps == s)
                if m.sum() < 20:
                    continue
                d = dur[m] * 1e6
                line = f"   {'likelihood' if ll else 'residual  '} width {w:4d} steps {s:2d}: {int(m.sum()):7d} blocks, {d.sum() / dur.sum() * 100:5.1f} % of block time, " \
                       f"duration median {np.median(d):6.2f} us (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})"
                if not ll and s > 1:
                    first = (tf[m] - t0[m]) * 10e-3
                    rest = (te[m] - tf[m]) * 10e-3 / (s - 1)
                    ok = tf[m] > 0
                    line += f"; to the end of the first step {np.median(first[ok]):5.2f} us, then {np.median(rest[ok]):5.2f} us per step (incl. the epilogue's share)"
                print(line)


def main():
    W, H = 640, 480
    K = synth.intrinsics_for(W, H)
    n_refs, n_curs, B, T, RES = 12, 96, 1152, 6, 96

    def cur_pose(i):
        return synth.se3_exp(synth.XI_GT_PAIR * (0.5 + 0.9 * ((i * 7) % 13) / 13.0) * (1 if i % 2 == 0 else -1)
                             + synth.XI_GT_PAIR[::-1] * 0.03 * ((i * 5) % 11 - 5))
    refs = [capi.RgbdImagePyramid(*synth.render(W, H, None if i == 0 else synth.se3_exp(synth.XI_GT_PAIR * 0.05 * i), frame_id=1000 + i), K, 4)
            for i in range(n_refs)]
    curs = [capi.RgbdImagePyramid(*synth.render(W, H, cur_pose(i), frame_id=1 + 2 * i), K, 4) for i in range(n_curs)]
    idx = [(i % n_refs, (i // n_refs) % n_curs) for i in range(B)]
    cfg = capi.Config(FirstLevel=3, LastLevel=0)
    trackers = [capi.DenseTracker(cfg) for _ in range(T)]
    shares = [list(range(t, B, T)) for t in range(T)]

    def run(n_steps):
        def worker(t):
            r, c = [refs[idx[i][0]] for i in shares[t]], [curs[idx[i][1]] for i in shares[t]]
            prev = None
            for _ in range(n_steps):
                sub = trackers[t].submit(r, c, stats=False, in_flight=RES)
                if prev is not None:
                    trackers[t].wait(prev, raw=True)
                prev = sub
            trackers[t].wait(prev, raw=True)
        th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        return time.perf_counter() - t0

    run(2)
    read_trace(trackers[0])  # reset
    steps = 3
    wall = run(steps)
    tr, n = read_trace(trackers[0])
    print(f"streaming batch: {B * steps / wall:.0f} pairs/s with the trace on ({n} blocks recorded, capacity {CAP})")
    analyse(tr, "streaming batch, 6 threads x 96 resident", wall)
    # the residual pass alone
    (Ir, Zr), (Ic, Zc), Tgt = synth.make_pair(W, H)
    pr, pc = capi.RgbdImagePyramid(Ir, Zr, K, 4), capi.RgbdImagePyramid(Ic, Zc, K, 4)
    for level, items in ((0, 36), (0, 4), (1, 16)):
        trackers[0].bench_residual_pass(pr, pc, level, Tgt, items, 0, reps=3)
        read_trace(trackers[0])
        trackers[0].bench_residual_pass(pr, pc, level, Tgt, items, 0, reps=1)  # (warm-up launch + 1)
        tr, n = read_trace(trackers[0])
        half = tr[np.argsort(tr[:, 0])][len(tr) // 2:]  # the second launch
        analyse(half, f"residual pass alone: level {level}, {items} pairs in one launch")


main()
