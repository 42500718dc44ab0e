"""Where does a block's life go, and how full are the block slots?  Needs a library built with the block trace:
    scripts/variant.sh trace -DDVO_TRACE_BLOCKS      ->  DVO_AMD_LIB=dvo_slam_amd/libdvo_amd_var_trace.so python scripts/block_trace.py
Runs (a) the streaming batch of bench.py (6 threads x TRACE_IN_FLIGHT (default 62) resident pairs, every step queued behind the previous one) and (b) the
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
# step codes of a work item (csrc/dvo_types.h: steps_of_code): 0..6 = 1 << code, then the lengths that are no power of two
STEPS_OF_CODE = np.array([1, 2, 4, 8, 16, 32, 64, 10, 20, 12, 14, 6, 18, 24, 40, 30], np.int64)


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
    steps = STEPS_OF_CODE[info & 15]
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
    print(f"   blocks per XCD: {[int((xcc == x).sum()) for x in range(8)]}; block time per XCD (slots occupied of 128): "
          f"{[round(float(dur[xcc == x].sum() / span), 1) for x in range(8)]}")
    # hardware queues (HW_ID: PIPE_ID bits 7:6, QUEUE_ID bits 26:24): when does each one have blocks on the GPU, and how full are
    # the slots while k of them do?
    hwq = ((hw >> 6) & 3) * 8 + ((hw >> 24) & 7)
    busy = np.zeros((0, len(occ)))
    names = []
    for q in sorted(set(hwq)):
        m = hwq == q
        if m.sum() < 100:
            continue
        g = np.zeros(len(occ) + 1)
        np.add.at(g, a[m], 1)
        np.add.at(g, b[m] + 1, -1)
        busy = np.vstack([busy, (np.cumsum(g)[:-1] > 0)[None, :]])
        names.append(f"pipe {q // 8} queue {q % 8}: {int(m.sum())} blocks, on the GPU {busy[-1].mean():.2f} of the time")
    print("   hardware queues: " + "; ".join(names))
    # the stretches in which a hardware queue has NO block on the GPU: kernel-to-kernel transitions (its k_finalize runs in them) if
    # they are all alike, starvation (no launch waiting in the queue) if some are long
    for qi, row in enumerate(busy):
        edges = np.flatnonzero(np.diff(np.concatenate([[1], row.astype(int), [1]])))
        gaps = (edges[1::2] - edges[0::2]).astype(float)  # in microseconds (the grid's resolution)
        if len(gaps):
            print(f"      {names[qi].split(':')[0]}: {len(gaps)} stretches without a block, median {np.median(gaps):.0f} us, p10 "
                  f"{np.percentile(gaps, 10):.0f}, p90 {np.percentile(gaps, 90):.0f}, longest {gaps.max():.0f}; stretches > 40 us hold "
                  f"{gaps[gaps > 40].sum() / max(gaps.sum(), 1):.2f} of the queue's empty time")
    # Is the dispatcher work-conserving?  Blocks "pending" at time t: blocks of a launch (a queue's stretch with blocks on the GPU)
    # that start later than t.  If slots are empty while blocks are pending, something other than the supply of blocks limits
    # the residency.
    pending = np.zeros(len(occ) + 1)
    qids = [q for q in sorted(set(hwq)) if (hwq == q).sum() >= 100]
    for qi, q in enumerate(qids):
        m = np.flatnonzero(hwq == q)
        row = busy[qi].astype(int)
        launch_id = np.cumsum(np.diff(np.concatenate([[0], row])) == 1)  # per microsecond: index of the queue's current launch
        lid = launch_id[np.minimum(a[m], len(occ) - 1)]
        first_us = np.full(lid.max() + 1, len(occ), dtype=np.int64)
        np.minimum.at(first_us, lid, a[m])
        # a block is pending from its launch's first microsecond until it starts
        np.add.at(pending, first_us[lid], 1)
        np.add.at(pending, a[m], -1)
    pend = np.cumsum(pending)[:-1]
    for lo_, hi_ in ((0, 0), (1, 63), (64, 255), (256, 1023), (1024, 1 << 30)):
        sel = (pend >= lo_) & (pend <= hi_)
        if sel.any():
            print(f"      blocks pending in [{lo_}, {min(hi_, 99999)}]: {sel.mean():.2f} of the time, {occ[sel].mean():.0f} slots occupied")
    n_busy = busy.sum(axis=0).astype(int)
    print("   slots occupied while k hardware queues have blocks resident: " +
          "; ".join(f"k={k}: {np.mean(n_busy == k):.2f} of the time, {occ[n_busy == k].mean():.0f} blocks" for k in range(len(names) + 1) if (n_busy == k).any()))
    for ll in (0, 1):
        for w in sorted(set(width)):
            for s in sorted(set(steps)):
                m = (is_ll == ll) & (width == w) & (steps == s)
                if m.sum() < 20:
                    continue
                d = dur[m] * 1e6
                line = f"   {'likelihood' if ll else 'residual  '} width {w:4d} steps {s:2d}: {int(m.sum()):7d} blocks, {dur[m].sum() / dur.sum() * 100:5.1f} % of block time, " \
                       f"duration median {np.median(d):6.2f} us (p10 {np.percentile(d, 10):6.2f}, p90 {np.percentile(d, 90):6.2f})"
                if not ll and s > 1:
                    first = (tf[m] - t0[m]) * 10e-3
                    rest = (te[m] - tf[m]) * 10e-3 / (s - 1)
                    ok = tf[m] > 0
                    line += f"; to the end of the first step {np.median(first[ok]):5.2f} us, then {np.median(rest[ok]):5.2f} us per step (incl. the epilogue's share)"
                print(line)


def lone_launches(tr, event_us):
    tr = tr[np.argsort(tr[:, 0])]
    t0, tf, te = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2].astype(np.int64)
    info = (tr[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    steps, is_ll = STEPS_OF_CODE[info & 15], (info >> 4) & 1
    end_so_far = np.maximum.accumulate(te)
    cuts = np.flatnonzero(t0[1:] > end_so_far[:-1]) + 1  # a block that starts behind every earlier block's end opens a launch
    bounds = np.concatenate([[0], cuts, [len(tr)]])
    spans, occ_sum, n_big, last_start, tail_kinds = [], np.zeros(80), 0, [], {}
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b - a < 900:
            continue
        n_big += 1
        z = t0[a:b].min()
        span = (te[a:b].max() - z) / 100.0
        spans.append(span)
        last_start.append((t0[a:b].max() - z) / 100.0)
        grid = np.zeros(81)
        np.add.at(grid, np.minimum((t0[a:b] - z) // 100, 79).astype(int), 1)
        np.add.at(grid, np.minimum((te[a:b] - z) // 100 + 1, 80).astype(int), -1)
        occ_sum += np.cumsum(grid)[:80]
        late = te[a:b] > te[a:b].max() - 300  # blocks still running 3 us before the launch's end
        for s_, l_ in zip(steps[a:b][late], is_ll[a:b][late]):
            key = ("likelihood" if l_ else "residual") + f" {s_} steps"
            tail_kinds[key] = tail_kinds.get(key, 0) + 1
    if not n_big:
        print("== lone launches: none of >= 900 blocks found")
        return
    spans = np.array(spans)
    print(f"== lone launches of a batch (one stream, per-launch timing on): {n_big} launches of >= 900 blocks; HIP-event duration of "
          f"the average launch (all sizes) {event_us:.1f} us")
    print(f"   first block's start -> last block's end: median {np.median(spans):.1f} us (p10 {np.percentile(spans, 10):.1f}, p90 "
          f"{np.percentile(spans, 90):.1f}); the last block STARTS at {np.median(last_start):.1f} us")
    occ = occ_sum / n_big
    print("   resident blocks by microsecond since the first block's start:")
    print("   " + " ".join(f"{int(o):4d}" for o in occ[:48]))
    tot = sum(tail_kinds.values())
    print("   blocks still running 3 us before their launch's end: " + ", ".join(f"{k}: {v / tot:.2f}" for k, v in sorted(tail_kinds.items())))


def main():
    W, H = 640, 480
    K = synth.intrinsics_for(W, H)
    n_refs, n_curs, B, T, RES = 12, 96, 1152, int(os.environ.get("TRACE_THREADS", 6)), int(os.environ.get("TRACE_IN_FLIGHT", 62))

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
    analyse(tr, f"streaming batch, {T} threads x {RES} resident", wall)
    # (c) the launches of a batch one at a time: one thread's share with the per-launch timing on (every launch alone on the GPU,
    # the configuration roofline.frac is quoted on): what does a lone launch look like from the first block's start to the last's end?
    t = trackers[0]
    r, c = [refs[idx[i][0]] for i in shares[0]], [curs[idx[i][1]] for i in shares[0]]
    t.match_batch(r, c, stats=False, in_flight=RES)
    read_trace(t)
    t.kernel_timing(True, reset=True)
    t.match_batch(r, c, stats=False, in_flight=RES)
    k_ms, k_n = t.kernel_timing(False)
    tr, n = read_trace(t)
    lone_launches(tr, k_ms * 1e3 / max(k_n, 1))
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


if __name__ == "__main__":
    main()
