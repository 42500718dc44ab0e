#!/usr/bin/env python3
"""bench.py -- frame-pairs/s of the dense RGB-D alignment hot path on MI355X (BASELINE.json metric).

A "step" = one pass over B independent 640x480 frame pairs (4-level coarse-to-fine Gauss-Newton,
FirstLevel 3 -> LastLevel 0, reference defaults otherwise) aligned on ONE GPU through the C ABI: T host threads, each with
its own tracker, work through an equal share of the batch with a fixed number of pairs resident at a time.  A thread hands
its tracker the next step's share (dvo_amd_match_submit) before it collects the current one (dvo_amd_match_wait), so a
tracker does not run dry between steps -- the shape of the loop-closure validator fed proposal list after proposal list
(keyframe_graph.cpp:434-498, 576-593); --drain-between-steps gives the old one-call-per-step behaviour (dvo_amd_match_many).  Pyramids of all frames are built beforehand and stay resident in HBM, as
LocalTracker::update pre-builds them (dvo_slam/src/local_tracker.cpp:163-169), so the timed region is
DenseTracker::match only.  With N GPUs every rank aligns its own B pairs (independent units, no data-path
collective): weak scaling, value = N * B * K / max-over-ranks(time).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_POINT = 56.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node; N > 1 without a torchrun environment makes this process spawn the N ranks "
                         "itself (python -m torch.distributed.run ... bench.py <same arguments>)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1152, help="frame pairs aligned per step per GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--distinct", type=int, default=96, help="distinct current frames")
    ap.add_argument("--distinct-refs", type=int, default=12,
                    help="distinct keyframes.  Pair i of a step = (keyframe i %% R, frame (i // R) %% C): with the defaults "
                         "(96 x 12 = 1152) every pair of a step is a different combination and the 108 pyramids (2.1 GB) "
                         "are far beyond the 256 MiB Infinity Cache")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a one-GPU box: every rank uses GPU 0 and the ranks synchronise over gloo "
                         "(exercises the N > 1 code path; the figure is meaningless)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="no GPU work at all: ranks rendezvous over gloo, 'align' their share with a stand-in sleep and print "
                         "the line (tests/test_distributed.py checks the spawn / aggregate / one-line plumbing on the CPU)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational side measurements (cache-resident variant, STREAM copy, frame ingest, "
                         "loop-closure validator, all-core CPU baseline); rank 0 at N=1 only")
    ap.add_argument("--counter-leg", action="store_true",
                    help="the short run a rocprofv3 counter pass profiles (bench.py starts two of them itself, see "
                         "--no-live-counters): like --no-extras --no-cpu-baseline, and without the single-pair latency probe and the "
                         "timed-region check")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not run the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) as child processes of this run; "
                         "roofline.traffic then comes from the committed profile of the same command (traffic_source says which)")
    ap.add_argument("--tile-shard", action="store_true",
                    help="BASELINE config 4 instead of the default: ONE pair at a time, every level tile-sharded over the "
                         "N GPUs with a per-iteration exchange of the band records (strong scaling, expected to be "
                         "slower than 1 GPU at 640x480: a tick is ~26 us, so is a small-message all-gather)")
    ap.add_argument("--prime", type=int, default=2,
                    help="untimed priming steps run as part of set-up before the W warm-up steps: the first calls grow "
                         "scratch buffers, streams and the HIP runtime's internal pools (a one-off ~40 ms stall)")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="pairs resident per tracker at a time (0 = the whole share in lock step); default 96, 72 from 1280x960 on")
    ap.add_argument("--threads", type=int, default=None,
                    help="host threads per GPU, each with its own tracker (HIP stream) and an equal share of the batch; default 6, "
                         "8 from 1280x960 on (measured: a 288-pair step of 1280x960 runs 14.4k pairs/s with 8 x 72 and 10.8k with 6 x 96)")
    ap.add_argument("--drain-between-steps", action="store_true",
                    help="one dvo_amd_match_many call per step and thread (the tracker drains to empty at the end of every step) "
                         "instead of submitting the next step's share before waiting for the current one")
    ap.add_argument("--no-stats", action="store_true",
                    help="drop the per-iteration statistics (Result.Statistics) in the timed region; by default they are "
                         "delivered, as the reference's callers read them (keyframe_tracker.cpp:167, "
                         "constraint_proposal_voter.cpp:128)")
    args = ap.parse_args()
    if args.counter_leg:
        args.no_extras = args.no_cpu_baseline = True
    if args.no_extras:
        args.no_live_counters = True  # (the counter passes are a side measurement of the full line)
    if args.threads is None:
        args.threads = 6
    if args.in_flight is None:
        # one group of up to 62 pairs per tracker: every tick one launch as large as the argument block allows
        # (kMaxItemsPerLaunch).  Until the end of round 5: 6 x 124 = two groups per tracker (8 x 72 for 1280x960); with 16-step
        # wave segments one group is +1.0 ... +1.9 % in interleaved runs on three boxes (1280x960, 288 pairs per step: +5.5 %) and
        # needs half the resident scratch (profiles/r05_residency_ab.txt)
        args.in_flight = 62
    return args


def maybe_spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks as fresh child processes (one per
    GPU) BEFORE this process touches the GPU, forward their output (rank 0 prints the one JSON line) and exit with their
    code.  The reference deals its proposals over TBB workers the same way (keyframe_graph.cpp:576-593)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def plumbing_only(args):
    """The N-rank plumbing without a GPU: rendezvous (gloo), barrier, stand-in work, MAX / SUM aggregation, one line."""
    from dvo_slam_amd import sharding

    rank, _, world = sharding.rank_info()
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (1 + rank))  # the straggler sets the time
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed, pairs = sharding.aggregate(elapsed, args.batch * args.steps, dist, None if dist is None else "cpu")
    if rank == 0:
        print(json.dumps({"metric": "frame-pairs/s (640x480, 4-level GN align)", "value": pairs / elapsed,
                          "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none (plumbing rehearsal: no alignment ran)",
                          "config": {"workload": "plumbing only", "pairs_total": pairs}}), file=OUT, flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_with_deadline(fn, seconds, what):
    """fn() on its own daemon thread -> (result, hung).  An exception becomes {"error": ...}; a call still running at the
    deadline becomes ({"error": ...}, True) and keeps its thread: the caller must then leave through os._exit once its output
    is written (a rank stuck inside a collective cannot be cancelled)."""
    import threading

    box = {}

    def run():
        try:
            box["out"] = fn()
        except BaseException as exc:  # a side measurement never fails the bench line
            box["out"] = {"error": repr(exc)}

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": f"the {what} did not finish within {seconds:g} s (a rank failed inside a collective?); dropped"}, True
    return box.get("out"), False


def pair_index(i, n_refs, n_curs):
    """pair i of a step -> (keyframe, frame): all n_refs * n_curs combinations before any repeats"""
    return i % n_refs, (i // n_refs) % n_curs


def workload_cur_pose(synth, i):
    """camera pose of current frame i of the synthetic workload (hashed: every frame a different motion of 0.5 .. 1.4 x the headline
    pair's, alternating sign, plus a small second component)"""
    return synth.se3_exp(synth.XI_GT_PAIR * (0.5 + 0.9 * ((i * 7) % 13) / 13.0) * (1 if i % 2 == 0 else -1)
                         + synth.XI_GT_PAIR[::-1] * 0.03 * ((i * 5) % 11 - 5))


def workload_ref_frame(synth, W, H, rank, k):
    """keyframe k of the synthetic workload (k = 0: the scene's reference view)"""
    if k == 0:
        return synth.render(W, H, None, frame_id=2 * rank)
    return synth.render(W, H, synth.se3_exp(synth.XI_GT_PAIR * 0.05 * k), frame_id=1000 + 2 * rank + k)


def workload_cur_frame(synth, W, H, rank, i):
    return synth.render(W, H, workload_cur_pose(synth, i), frame_id=2 * rank + 1 + 2 * i)


OUT = sys.stdout


def private_stdout():
    """The one JSON line gets a private copy of the process's stdout; file descriptor 1 itself is pointed at stderr for everything
    else, so that whatever a library prints there (RCCL's version block on some boxes, seen in front of the line of the
    --tile-shard leg in round 5) cannot end up in front of it."""
    global OUT
    sys.stdout.flush()
    OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)


def main():
    args = parse()
    maybe_spawn_ranks(args)
    private_stdout()
    if args.plumbing_only:
        return plumbing_only(args)
    from dvo_slam_amd import sharding

    rank, local_rank, world = sharding.rank_info()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist  # noqa: F811

        if args.rehearse_on_one_gpu:
            local_rank = 0
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from dvo_slam_amd import capi, synth

    if capi.lib().dvo_amd_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    device = local_rank
    numa = pin_to_gpu_numa_node(device)
    thread_plan = host_thread_plan(args, world, numa)
    W, H = args.width, args.height
    levels = 5 if W >= 1280 else 4
    first_level = levels - 1
    K = synth.intrinsics_for(W, H)
    n_curs, n_refs = max(1, args.distinct), max(1, args.distinct_refs)

    # ---- synthetic frames (seeded, rank-dependent): `distinct` current frames and `distinct-refs` keyframes at hashed poses
    # (tests/test_bench_workload.py checks a sample of exactly these pairs against the oracle)
    def cur_pose(i):
        return workload_cur_pose(synth, i)

    t0 = time.perf_counter()
    ref_frame = workload_ref_frame(synth, W, H, rank, 0)
    cur_frames = [workload_cur_frame(synth, W, H, rank, i) for i in range(n_curs)]
    t_render = time.perf_counter() - t0

    # ---- pyramids (prep): built once, resident in HBM
    t0 = time.perf_counter()
    ref = capi.RgbdImagePyramid(ref_frame[0], ref_frame[1], K, levels, device=device)
    curs = [capi.RgbdImagePyramid(f[0], f[1], K, levels, device=device) for f in cur_frames]
    prep_first_ms = (time.perf_counter() - t0) * 1e3 / (1 + len(curs))  # includes the one-off slab allocations
    # steady state (slabs come from the pool): build some of the same frames again and drop them
    t0 = time.perf_counter()
    for f in cur_frames[:8]:
        capi.RgbdImagePyramid(f[0], f[1], K, levels, device=device)
    prep_ms = (time.perf_counter() - t0) * 1e3 / len(cur_frames[:8])
    if args.tile_shard:
        return tile_shard_bench(args, capi, synth, sharding, dist, rank, world, device, ref_frame, cur_frames, K, levels,
                                first_level)

    import threading

    T = max(1, min(args.threads, args.batch))
    cfg = capi.Config(FirstLevel=first_level, LastLevel=0)
    # (diagnostic, DVO_BENCH_FOREIGN_STREAMS=n: n streams of "the application" -- idle contexts -- created in front of every
    #  tracker: how much of the throughput rests on the runtime's stream -> hardware queue assignment,
    #  profiles/r05_stream_queue_assignment_ab.txt)
    foreign, trackers = [], []
    for _ in range(T):
        foreign += [capi.DenseTracker(cfg, device=device) for _ in range(int(os.environ.get("DVO_BENCH_FOREIGN_STREAMS", "0")))]
        trackers.append(capi.DenseTracker(cfg, device=device))
    if os.environ.get("DVO_BENCH_TRACKER_PERM"):  # (diagnostic) host thread t drives the tracker created perm[t]-th
        perm = [int(x) for x in os.environ["DVO_BENCH_TRACKER_PERM"].split(",")]
        if sorted(perm) == list(range(T)):
            trackers = [trackers[i] for i in perm]
    trk = trackers[0]
    B = args.batch
    ref_pyrs = [ref]
    ref_frames_extra = []
    for i in range(1, n_refs):
        fr = workload_ref_frame(synth, W, H, rank, i)
        ref_frames_extra.append(fr)
        ref_pyrs.append(capi.RgbdImagePyramid(fr[0], fr[1], K, levels, device=device))
    n_pyramids = len(ref_pyrs) + len(curs)
    pyramid_bytes = n_pyramids * 76.0 * sum((W >> l) * (H >> l) for l in range(levels))
    idx = [pair_index(i, n_refs, n_curs) for i in range(B)]
    refs = [ref_pyrs[r] for r, _ in idx]
    curb = [curs[c] for _, c in idx]
    distinct_pairs = len(set(idx))
    shares = sharding.split_for_threads(list(range(B)), T)
    with_stats = not args.no_stats
    # result structs (+ per-iteration statistics arrays) are allocated once per host thread (two sets: the step being
    # collected and the step already queued behind it) and refilled every step
    result_bufs = [[trackers[t].alloc_results(len(shares[t])) for _ in range(2)] if with_stats else None for t in range(T)]
    streaming = not args.drain_between_steps

    def run_steps(n_steps, collect, r_list=None, c_list=None, stats=with_stats, keep=None):
        """n_steps batches of B pairs on this GPU; with T > 1 every thread drives its own tracker"""
        r_list, c_list = r_list or refs, c_list or curb

        def worker(t):
            ix = shares[t]
            r, c = [r_list[i] for i in ix], [c_list[i] for i in ix]

            def tally(out):
                collect.append((sum(o.alg_bytes for o in out), sum(o.n_residual_passes for o in out),
                                sum(o.is_nan for o in out), sum(o.n_iterations for o in out),
                                sum(o.alg_bytes_discarded for o in out)))
            prev = None
            if keep is not None:
                keep[t] = None
            for s_ in range(n_steps):
                ts = time.perf_counter()
                bufs = result_bufs[t][s_ % 2] if stats else None
                if streaming:
                    # the next step's share is queued behind the current one before the current one is collected
                    sub = trackers[t].submit(r, c, stats=False, in_flight=args.in_flight, results=bufs)
                    if prev is not None:
                        tally(trackers[t].wait(prev, raw=True))
                    prev = sub
                else:
                    out_ = trackers[t].match_batch(r, c, stats=False, in_flight=args.in_flight, raw=True, results=bufs)
                    tally(out_)
                    if keep is not None:
                        keep[t] = (ix, out_)
                if os.environ.get("DVO_BENCH_DEBUG"):
                    print(f"thread {t} step {s_}: {(time.perf_counter() - ts) * 1e3:.2f} ms", file=sys.stderr, flush=True)
            if prev is not None:
                last = trackers[t].wait(prev, raw=True)
                tally(last)
                if keep is not None:
                    keep[t] = (ix, last)  # the raw result structs of this thread's share of the LAST step
        if T == 1:
            worker(0)
        else:
            th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            for x in th:
                x.start()
            for x in th:
                x.join()

    def sync_all():
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    run_steps(args.prime, [])
    run_steps(args.warmup, [])
    if os.environ.get("DVO_BENCH_MAPS"):
        # every library of the run is mapped by now: raw-address stacks (glog under rocprofv3) can be attributed afterwards
        with open(os.environ["DVO_BENCH_MAPS"], "w") as fh:
            fh.write(open("/proc/self/maps").read())
    # single-pair latency (informational)
    # One match() per frame is the reference's default deployment (dvo_ros/src/camera_dense_tracking.cpp:269): a tracker configured
    # for it (dvo_amd_config::segment_geometry = DVO_AMD_GEOMETRY_LATENCY, round 5) next to the batch's own configuration.
    n_lat = 10

    def latency_of(tracker):
        rounds = []
        for _ in range(0 if args.counter_leg else 3):  # the median of three rounds of ten: one round is at the mercy of whatever the box does in those 7 ms
            t0 = time.perf_counter()
            for i in range(n_lat):
                tracker.match(ref, curs[i % len(curs)])
            rounds.append((time.perf_counter() - t0) * 1e3 / n_lat)
        return sorted(rounds)[1] if rounds else None
    trk_latency = capi.DenseTracker(capi.Config(FirstLevel=first_level, LastLevel=0, SegmentGeometry=capi.GEOMETRY_LATENCY), device=device)
    if not args.counter_leg:
        trk_latency.match(ref, curs[0])  # (its scratch and streams exist before the clock starts)
    single_ms = latency_of(trk_latency)
    single_ms_batch_geometry = latency_of(trk)
    del trk_latency

    # With one host thread the HIP events around every k_tick launch are taken inside the timed region.  With several
    # threads kernels of different streams overlap on the GPU, so the per-launch durations are measured in a second,
    # single-stream pass of the same batch right after the timed region.
    # The launches the per-launch roofline figures describe are bracketed by two no-op dispatches named k_marker, so that a
    # profiler's summary can pick exactly them (dvo_slam_amd/pmc.py).
    if T == 1:
        trk.marker(1)
        trk.kernel_timing(True, reset=True)
    sync_all()
    t0 = time.perf_counter()
    col = []
    kept = [None] * T
    run_steps(args.steps, col, keep=kept)
    sync_all()
    elapsed = time.perf_counter() - t0
    if T == 1:
        trk.marker(2)
    region_check = None if args.counter_leg else check_timed_region(capi, synth, cfg, device, kept, refs, curb)
    alg_bytes = sum(c[0] for c in col)
    discarded_bytes = sum(c[4] for c in col)
    passes = sum(c[1] for c in col)
    iterations_delivered = sum(c[3] for c in col)
    if sum(c[2] for c in col):
        raise SystemExit("bench.py: a pair of the timed region came back NaN")
    if T == 1:
        k_ms, k_launches = trk.kernel_timing(False)
        log = trk.tick_log()
    else:
        # one host thread's share of the batch, same residency as in the timed region, on one stream
        trk.marker(1)
        trk.kernel_timing(True, reset=True)
        idx0 = shares[0]
        out = trk.match_batch([refs[i] for i in idx0], [curb[i] for i in idx0], stats=False, in_flight=args.in_flight)
        k_ms, k_launches = trk.kernel_timing(False)
        trk.marker(2)
        log = trk.tick_log()
        alg_bytes_k = sum(o.alg_bytes for o in out)
        discarded_k = sum(o.alg_bytes_discarded for o in out)
    elapsed_local = elapsed
    # MAX over ranks of the elapsed time, SUM over ranks of the pairs aligned
    elapsed, pairs = sharding.aggregate(elapsed, B * args.steps, dist,
                                        None if dist is None else ("cpu" if args.rehearse_on_one_gpu else "cuda"))
    value = pairs / elapsed

    tile_shard = None
    side_hung = False
    if world > 1 and (not args.rehearse_on_one_gpu or os.environ.get("DVO_AMD_EXCHANGE") == "peer"):
        # (a one-GPU rehearsal can only exercise the peer exchange: RCCL refuses two ranks on one device)
        # BASELINE config 4 from the driver's own multi-GPU command: every rank takes part (the exchange is collective).
        # The headline figure above is complete at this point.  The side measurement runs under a deadline on its own thread:
        # a collective that one rank never enters cannot be cancelled, so a rank whose measurement is still running at the
        # deadline reports that, prints the line (rank 0) and leaves with os._exit -- the bench line is never lost to it.
        del trackers[1:]

        def side():
            import torch

            if torch.cuda.is_available():
                torch.cuda.set_device(device)
            out = tile_shard_measure(args, capi, synth, dist, rank, world, device, K, levels, first_level,
                                     steps=max(2, args.steps // 4), warmup=1)
            if (W, H) == (640, 480):
                # SURVEY.md 8e expects sharding within a pair to pay only from 1280x960 on: measure that next to it
                big = argparse.Namespace(**vars(args))
                big.width, big.height = 1280, 960
                try:
                    out["at_1280x960"] = tile_shard_measure(big, capi, synth, dist, rank, world, device,
                                                            synth.intrinsics_for(1280, 960), 5, 4, steps=2, warmup=1)
                except Exception as exc:  # pragma: no cover - keeps the 640x480 figure
                    out["at_1280x960"] = {"error": repr(exc)}
            return out

        tile_shard, side_hung = run_with_deadline(side, float(os.environ.get("DVO_BENCH_SIDE_DEADLINE_S", "120")),
                                                  "tile-shard side measurement on rank %d" % rank)

    if rank == 0:
        if T == 1:
            alg_bytes_k, discarded_k = alg_bytes, discarded_bytes
        # USEFUL algorithmic bytes: speculative residual passes whose iteration was rolled back are not counted
        useful_k = alg_bytes_k - discarded_k
        achieved = useful_k / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        useful = alg_bytes - discarded_bytes
        line = {
            "metric": "frame-pairs/s (640x480, 4-level GN align)" if W == 640 else f"frame-pairs/s ({W}x{H}, {levels}-level GN align)",
            "value": value,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic {W}x{H} RGB-D pairs (analytic room corner, seed 20131103), {levels}-level "
                            f"coarse-to-fine Gauss-Newton (FirstLevel {first_level} -> LastLevel 0, MaxIter 100, "
                            f"Precision 5e-7, Mu 0), {B} independent pairs per step per GPU = {distinct_pairs} distinct "
                            f"(keyframe, frame) combinations of {len(ref_pyrs)} keyframes x {len(curs)} frames "
                            f"({n_pyramids} pyramids, {pyramid_bytes / 2**20:.0f} MiB resident: "
                            f"{'beyond' if pyramid_bytes > 256 * 2**20 else 'inside'} the 256 MiB Infinity Cache), worked "
                            f"through by {T} host threads (one tracker / HIP stream each) with at most "
                            f"{args.in_flight or 'all'} pairs resident per tracker"
                            f"{' and the next step queued behind the current one' if streaming else ''}, pyramids pre-built and resident in HBM, "
                            f"per-iteration statistics {'delivered' if with_stats else 'dropped'}",
                "pairs_per_step_per_gpu": B,
                "distinct_pairs_per_step": distinct_pairs,
                "pyramids_resident": n_pyramids,
                "pyramid_working_set_mib": pyramid_bytes / 2**20,
                "host_threads_per_gpu": T,
                "host_threads_pinned_to_numa_node": numa,
                "host_threads": thread_plan,
                "pairs_in_flight_per_tracker": args.in_flight,
                "residency": "kept across steps (next step's share submitted before the current one is collected)" if streaming
                             else "drained at the end of every step",
                "iteration_statistics": "delivered" if with_stats else "dropped",
                "sharding": "independent pairs per rank, no collective on the data path",
            },
            "build_id": capi.build_id(),
            "iterations_per_pair": iterations_delivered / max(1, B * args.steps),
            "timed_region_check": region_check,
            "max_deviation_from_single_match": region_check["max_deviation_from_single_match"] if region_check else None,
            "single_pair_latency_ms": single_ms,
            "single_pair_latency": {
                "ms": single_ms, "configuration": "segment_geometry = DVO_AMD_GEOMETRY_LATENCY (levels 3..0 in 1/2/2/4 steps per wave)",
                "ms_with_the_batch_configuration": single_ms_batch_geometry,
                "batch_configuration": "segment_geometry = DVO_AMD_GEOMETRY_THROUGHPUT (4/4/10/10: row-aligned segments on the fine levels), what the timed region runs",
                "what": "one dvo_amd_match() at a time through the Python binding, median of three rounds of ten pairs; the geometry "
                        "is a field of the tracker's configuration and part of what a result is a function of"},
            "prep_ms_per_frame": prep_ms,
            "prep_ms_per_frame_first_use": prep_first_ms,
            # build the new frame's pyramid from host planes + one match()
            "end_to_end_ms_per_frame_one_stream": None if single_ms is None else prep_ms + single_ms,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "achieved_is": "USEFUL algorithmic bytes (56 B per selected reference pixel per residual pass, SURVEY.md 8d, without "
                               "the speculative passes that were rolled back: `speculation_waste`) / k_tick launch duration; an "
                               "upper-level view, not the bytes HBM actually moved: see `traffic` and `issue`",
                "speculation_waste": {"discarded_fraction_of_submitted_bytes": (discarded_k / alg_bytes_k) if alg_bytes_k else None,
                                      "achieved_counting_discarded_passes": alg_bytes_k / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0},
                "traffic": None,          # filled in below: roofline_traffic()
                "traffic_source": None,
                "kernel": "k_tick (fused warp+residual+weights+normal equations, with the log-likelihood items of the tick)",
                "launches": int(k_launches),
                "avg_launch_us": (k_ms * 1e3 / k_launches) if k_launches else None,
                "alg_bytes_per_launch": (useful_k / k_launches) if k_launches else None,
                "measured": "HIP events stamped by the dispatch itself (hipExtLaunchKernel start / stop) on the launching "
                            "stream, every launch alone on the GPU, " + ("inside the timed region" if T == 1 else
                            "in a single-stream pass over one host thread's share of the batch, same pairs in flight, right after "
                            "the timed region (which ran %d streams at once: its launches overlap, so the per-launch figure "
                            "here is not the concurrent rate)" % T),
                "residual_passes": int(passes),
                "concurrent": {
                    "achieved": useful / elapsed_local / 1e9, "unit": "GB/s", "frac": useful / elapsed_local / 1e9 / HBM_PEAK_GBS,
                    "discarded_fraction_of_submitted_bytes": (discarded_bytes / alg_bytes) if alg_bytes else None,
                    "what": "all launches of the timed region of this GPU together: their USEFUL algorithmic bytes / the wall time of "
                            "the region (%d host threads, k_finalize and the host turn-around included)" % T},
                "issue": issue_roofline(log, k_ms),
            },
        }
        line["roofline"]["traffic"], line["roofline"]["traffic_source"] = roofline_traffic(args, world, line)
        try:
            if args.no_extras:  # (a counter / trace pass must not find batch-form k_tick launches behind the timing pass)
                raise RuntimeError("skipped under --no-extras")
            # (the level's own geometry -- ten steps of 64 pixels per wave, one image row, for a 640x480 level 0 since the end of round 5
            # -- and, beside it, 8 steps)
            ms_i, ab_i, nl_i = trk.bench_residual_pass(ref, curs[0], 0, cur_pose(0), 36, 0, reps=20)
            ms_8, ab_8, nl_8 = trk.bench_residual_pass(ref, curs[0], 0, cur_pose(0), 36, 2, reps=20)
            line["roofline_isolated_kernel"] = {
                "what": "the residual pass alone: level 0, 36 pairs in one launch (one launch's worth of resident pairs), the wave "
                        "segments match() gives this level (segment_geometry = DVO_AMD_GEOMETRY_THROUGHPUT)",
                "achieved": ab_i / ms_i / 1e6, "unit": "GB/s", "frac": ab_i / ms_i / 1e6 / HBM_PEAK_GBS,
                "launch_us": ms_i * 1e3 / nl_i, "alg_bytes_per_launch": ab_i / nl_i,
                "with_8_steps_per_wave": {"frac": ab_8 / ms_8 / 1e6 / HBM_PEAK_GBS, "launch_us": ms_8 * 1e3 / nl_8}}
            # the timed region against what the kernel body reaches on its best case in this very run (the level-0 pass alone on
            # the GPU sits on the kernel's practical issue ceiling): how much of that the whole job keeps
            line["roofline"]["concurrent"]["fraction_of_the_isolated_kernel"] = (
                line["roofline"]["concurrent"]["frac"] / line["roofline_isolated_kernel"]["frac"])
        except Exception as exc:  # pragma: no cover
            line["roofline_isolated_kernel"] = {"error": str(exc)}
        if tile_shard is not None:
            line["tile_shard"] = tile_shard
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, ref_frame, cur_frames[:8], K, levels, first_level)
        if world == 1 and not args.no_extras:
            def cache_resident():
                # round 1's workload for comparison: 8 distinct frames against ONE keyframe (9 pyramids, 180 MB: inside the
                # Infinity Cache, each distinct pair ~4x in every launch, identical reads that L2 dedups)
                r_l, c_l = [ref] * B, [curs[i % 8] for i in range(B)]
                run_steps(1, [], r_l, c_l)
                t0 = time.perf_counter()
                n = max(2, args.steps // 2)
                run_steps(n, [], r_l, c_l)
                dt = time.perf_counter() - t0
                return {"value": B * n / dt, "unit": "frame-pairs/s", "steps": n,
                        "what": "the same batch size with 8 distinct frames against 1 keyframe (9 pyramids inside the "
                                "Infinity Cache, pairs repeated 144x per step): round 1's default workload"}
            def sensor_noise():
                # The regime the reference actually runs in (benchmark_slam.cpp:56-80): the SAME views and the same pair list as
                # the timed region, delivered as a sensor delivers them -- 8-bit grey, uint16 depth at 1/5000 m, depth noise of
                # the sigma the reference models itself (dense_tracking_impl.cpp:122-128) -- and ingested on the device
                # (dvo_amd_pyramid_create_raw).  Not the headline: BASELINE.json's metric is quoted on the noise-free pair.
                t_ingest = [0.0]

                def raw_pyr(frame, fid):
                    g, z = synth.sensor_from_analytic(frame[0], frame[1], frame_id=fid)
                    t0 = time.perf_counter()
                    p = capi.RgbdImagePyramid.from_raw(g, z, K, levels, device=device)
                    t_ingest[0] += time.perf_counter() - t0
                    return p
                s_refs = [raw_pyr(ref_frame, 5000)] + [raw_pyr(ref_frames_extra[i - 1], 5000 + i) for i in range(1, n_refs)]
                s_curs = [raw_pyr(f, 6000 + i) for i, f in enumerate(cur_frames)]
                t_build = t_ingest[0]
                r_l, c_l = [s_refs[r] for r, _ in idx], [s_curs[c] for _, c in idx]
                run_steps(1, [], r_l, c_l)
                n = max(2, args.steps // 2)
                tally = []
                t0 = time.perf_counter()
                run_steps(n, tally, r_l, c_l)
                dt = time.perf_counter() - t0
                if sum(c[2] for c in tally):
                    raise RuntimeError("a pair of the sensor-noise workload came back NaN")
                return {"value": B * n / dt, "unit": "frame-pairs/s", "steps": n,
                        "iterations_per_pair": sum(c[3] for c in tally) / (B * n),
                        "residual_passes_per_pair": sum(c[1] for c in tally) / (B * n),
                        "useful_algorithmic_mb_per_pair": sum(c[0] - c[4] for c in tally) / (B * n) / 1e6,
                        "concurrent_frac_of_hbm_peak": sum(c[0] - c[4] for c in tally) / dt / 1e9 / HBM_PEAK_GBS,
                        "frames_ingested_from_raw_ms_each": t_build * 1e3 / (len(s_refs) + len(s_curs)),
                        "what": "the timed region's pair list on sensor-realistic input: the same views as 8-bit grey + uint16 "
                                "depth at 1/5000 m with hashed Gaussian depth noise sigma_z(z) = 0.0012 + 0.0019 (z - 0.4)^2 and "
                                "1.5 grey levels of intensity noise, ingested on the device; same trackers, same residency"}
            def host_rcpps():
                # the opt-in reciprocal mode that reproduces the host's _mm_rcp_ps bit for bit (dense_tracking_impl.cpp:192,700)
                # from a table: the timed region's workload once more with it switched on, and what it costs
                for t_ in trackers:
                    t_.set_reciprocal_mode("host_sse")
                try:
                    run_steps(1, [])
                    n = max(2, args.steps // 2)
                    tally = []
                    t0 = time.perf_counter()
                    run_steps(n, tally)
                    dt = time.perf_counter() - t0
                    mb_pair = sum(c[0] - c[4] for c in tally) / (B * n) / 1e6
                    conc = sum(c[0] - c[4] for c in tally) / dt / 1e9 / HBM_PEAK_GBS
                    return {"value": B * n / dt, "unit": "frame-pairs/s", "steps": n,
                            "relative_to_the_default_mode": (B * n / dt) / value,
                            "iterations_per_pair": sum(c[3] for c in tally) / (B * n),
                            "useful_algorithmic_mb_per_pair": mb_pair,
                            "concurrent_frac_of_hbm_peak": conc,
                            # the same work per second?  rcpps arithmetic converges differently (more iterations, more of them on
                            # the fine levels): pairs/s compares two different amounts of work, bytes/s does not
                            "relative_to_the_default_mode_at_equal_work": conc / (useful / elapsed_local / 1e9 / HBM_PEAK_GBS),
                            "form": trackers[0].reciprocal_form()[0],
                            "table_mantissa_bits": trackers[0].reciprocal_mode()[1],
                            "what": "dvo_amd_set_reciprocal_mode(DVO_AMD_RCP_HOST_SSE): 1 / z of the projection and the reciprocal of "
                                    "the t-distribution weights are this host's _mm_rcp_ps (the device's own reciprocal plus 4-bit "
                                    "corrections in LDS, or a device-resident table: `form`); every weight of a pass is computeWeightsSse's, "
                                    "its exact-division tail included (k_q7_tail: one more small dispatch per tick)"}
                finally:
                    for t_ in trackers:
                        t_.set_reciprocal_mode("exact")
            for name, fn in (("sensor_noise_workload", sensor_noise),
                             ("host_rcpps_mode", host_rcpps),
                             ("cache_resident_workload", cache_resident),
                             ("no_stats_variant", lambda: stats_variant(run_steps, B, args, with_stats)),
                             ("stream_copy", lambda: stream_copy(device)),
                             ("ingest", lambda: ingest_timing(capi, synth, cur_frames, K, levels, device)),
                             ("loop_closure_validator", lambda: validator_timing(capi, synth, W, H, device)),
                             ("dual_match_front_end", lambda: dual_match_timing(capi, ref, curs, levels, first_level, device)),
                             ("cpu_baseline_many_threads", lambda: None if args.no_cpu_baseline else
                              cpu_baseline_threads(args, ref_frame, cur_frames[:8], K, levels, first_level))):
                try:
                    line[name] = fn()
                except Exception as exc:  # pragma: no cover - side measurements never fail the bench line
                    line[name] = {"error": repr(exc)}
        print(json.dumps(line), file=OUT, flush=True)
    if side_hung:  # a thread of this process is stuck inside a collective: no orderly teardown is possible
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


def check_timed_region(capi, synth, cfg, device, kept, refs, curb, n_samples=48):
    """Results of the LAST step of the timed region (as its host threads collected them) against a single match() of the same
    pair on a fresh tracker: a pair's result is a function of its inputs alone (tests/test_determinism.py), so every sampled
    pair must come back bit for bit -- whatever it shared its ticks with in the timed region.  A deviation fails the bench."""
    import numpy as np

    single = capi.DenseTracker(cfg, device=device)
    per_thread = max(1, -(-n_samples // max(1, len(kept))))
    n = identical = 0
    worst = 0.0
    for entry in kept:
        if entry is None:
            continue
        ix, raw = entry
        for k in np.linspace(0, len(ix) - 1, num=min(per_thread, len(ix)), dtype=int):
            i = ix[int(k)]
            want = single.match(refs[i], curb[i])
            got_T = np.array(raw[int(k)].transformation[:]).reshape(4, 4).T
            got_I = np.array(raw[int(k)].information[:]).reshape(6, 6).T
            same = np.array_equal(got_T, want.Transformation) and np.array_equal(got_I, want.Information) \
                and raw[int(k)].n_iterations == sum(len(L["Iterations"]) for L in want.Levels)
            n += 1
            identical += 1 if same else 0
            if not np.array_equal(got_T, want.Transformation):  # (identical matrices are at distance 0, not at the rounding of log())
                worst = max(worst, synth.pose_error(want.Transformation, got_T))
    out = {"sampled_pairs": n, "bit_identical_to_single_match": identical, "max_deviation_from_single_match": worst,
           "what": "pairs of the last step of the timed region, re-aligned one at a time by dvo_amd_match() on a fresh tracker: "
                   "transformation, information and iteration count compared bit for bit; deviation = |log(T_single^-1 T_batch)|"}
    if n == 0 or identical != n:
        raise SystemExit("bench.py: a result of the timed region differs from the same pair's single match(): " + json.dumps(out))
    return out


def stats_variant(run_steps, B, args, with_stats):
    """the timed workload once more with the per-iteration statistics toggled: what delivering them costs"""
    n = max(2, args.steps // 2)
    run_steps(1, [], stats=not with_stats)
    t0 = time.perf_counter()
    run_steps(n, [], stats=not with_stats)
    dt = time.perf_counter() - t0
    return {"value": B * n / dt, "unit": "frame-pairs/s", "steps": n,
            "iteration_statistics": "dropped" if with_stats else "delivered"}


SIMDS = 256 * 4
PEAK_CLOCK_HZ = 2.4e9
VALU_ISSUE_CYCLES = 4   # one fp32 VALU wave instruction holds its SIMD for 4 cycles (measured: profiles/r04_issue.json)
MFMA_ISSUE_CYCLES = 8   # v_mfma_f32_4x4x1_16b_f32, the nine-tile Gram form (8.9 measured back to back, scripts/probes/mfma_4x4_blocks.hip)
ISSUE_FILES = ("r05_issue.json", "r04_issue.json")  # the newest committed PMC pass of the kernel that exists


def issue_roofline(log, k_ms):
    """The bound k_tick actually sits on: VALU + MFMA issue cycles.  Instructions per 64-pixel wave step come from the
    committed PMC pass of the kernel (profiles/r04_issue.json: SQ_INSTS_VALU / SQ_INSTS_MFMA per step of the residual and of
    the likelihood pass); the steps are this run's (tick log); peak = every SIMD issuing every cycle at the 2.4 GHz peak clock."""
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        name = next(f for f in ISSUE_FILES if os.path.exists(os.path.join(here, "profiles", f)))
        c = json.load(open(os.path.join(here, "profiles", name)))
        res_steps, ll_steps = float(log[:, 6].sum()), float(log[:, 7].sum())
        mfma_cyc = c.get("mfma_issue_cycles", MFMA_ISSUE_CYCLES)
        cyc = (res_steps * (c["valu_per_res_step"] * VALU_ISSUE_CYCLES + c["mfma_per_res_step"] * mfma_cyc)
               + ll_steps * c["valu_per_ll_step"] * VALU_ISSUE_CYCLES)
        peak = SIMDS * PEAK_CLOCK_HZ * k_ms * 1e-3
        return {"bound": "VALU+MFMA issue", "valu_per_res_step": c["valu_per_res_step"], "mfma_per_res_step": c["mfma_per_res_step"],
                "valu_per_ll_step": c["valu_per_ll_step"], "valu_issue_cycles": VALU_ISSUE_CYCLES, "mfma_issue_cycles": mfma_cyc,
                "res_steps": res_steps, "ll_steps": ll_steps, "issue_cycles": cyc, "peak_cycles": peak, "frac": cyc / peak,
                "source": c.get("source"), "file": "profiles/" + name,
                "measured": "instruction counts per step: not in this run (a committed SQ_INSTS_VALU / SQ_INSTS_MFMA pass of this "
                            "kernel); steps and launch time: this run"}
    except Exception as exc:
        return {"error": repr(exc)}


def tile_shard_measure(args, capi, synth, dist, rank, world, device, K, levels, first_level, steps, warmup):
    """BASELINE config 4: ONE pair at a time, every level tile-sharded over the ranks with a per-tick exchange of the 784-byte
    band records.  Both exchanges are measured when they come up: the RCCL all-gather (`rccl`, the path whose collective is
    the library's own) and the one-hop peer exchange (`peer`: mapped fine-grained buffers written by the tail of k_finalize --
    never run across two GPUs before the first multi-GPU run of this bench, so its poses are cross-checked against the RCCL
    path's here and the verdict is part of the output).  A peer exchange that does not come up on every rank (allocation,
    hipIpc attach, or a first tick that times out) falls back to RCCL automatically and says why.  DVO_AMD_EXCHANGE = peer |
    rccl restricts the measurement to one path."""
    import numpy as np

    from dvo_slam_amd import sharding

    _all_ranks = sharding.all_ranks
    # every rank must hold the SAME frames: rank-independent ids
    ref_frame = synth.render(args.width, args.height, None, frame_id=0)
    cur_frames = [synth.render(args.width, args.height, synth.se3_exp(synth.XI_GT_PAIR * (0.6 + 0.1 * i)), frame_id=1 + 2 * i)
                  for i in range(4)]
    ref = capi.RgbdImagePyramid(ref_frame[0], ref_frame[1], K, levels, device=device)
    curs = [capi.RgbdImagePyramid(f[0], f[1], K, levels, device=device) for f in cur_frames]
    # one pair at a time is a latency workload: the latency-first wave segments (a field of the configuration, include/dvo_amd.h)
    cfg = capi.Config(FirstLevel=first_level, LastLevel=0, SegmentGeometry=capi.GEOMETRY_LATENCY)
    want = os.environ.get("DVO_AMD_EXCHANGE", "auto")
    pairs_per_step = 8

    def barrier():
        if dist is not None:
            import torch

            dist.barrier()
            if dist.get_backend() == "nccl":
                torch.cuda.synchronize()

    def timed(trk):
        poses = []
        for _ in range(warmup + 1):
            for i in range(pairs_per_step):
                poses.append(trk.match_sharded(ref, curs[i % len(curs)]).Transformation)
        barrier()
        t0 = time.perf_counter()
        ticks = 0
        for _ in range(steps):
            for i in range(pairs_per_step):
                ticks += trk.match_sharded(ref, curs[i % len(curs)]).n_ticks
        barrier()
        elapsed = max(_all_ranks(dist, time.perf_counter() - t0))
        n_pairs = pairs_per_step * steps  # the SAME pairs on every rank: total work is fixed
        return {"pairs_per_s": n_pairs / elapsed, "us_per_tick": elapsed * 1e6 / max(ticks, 1), "ms_per_step": elapsed * 1e3 / steps,
                "ticks": ticks}, poses[:pairs_per_step]

    out = {"ranks": world, "pairs_per_step": pairs_per_step, "steps": steps,
           "what": f"ONE synthetic {args.width}x{args.height} pair at a time, every pyramid level tile-sharded over {world} GPU(s) "
                   "(bands of scan-order blocks; segment_geometry = DVO_AMD_GEOMETRY_LATENCY), per-tick exchange of the 784-byte band records "
                   "(BASELINE config 4); strong scaling, "
                   "expected to be slower than one GPU at this size (a tick is ~24 us, so is a small-message exchange)"}
    poses = {}
    if want in ("auto", "rccl"):
        trk = capi.DenseTracker(cfg, device=device)
        ids = [capi.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(ids, src=0)
        trk.comm_create(ids[0], world, rank)
        out["rccl"], poses["rccl"] = timed(trk)
        del trk
    if want in ("auto", "peer"):
        trk = capi.DenseTracker(cfg, device=device)
        # every stage runs on every rank and counts only if it succeeded everywhere: the ranks agree on a fallback
        handle, ok, why = sharding.collective_stage(dist, lambda: trk.exchange_create(world, rank), capi.DvoAmdError)
        if ok:
            handles = _all_ranks(dist, handle)
            _, ok, why = sharding.collective_stage(dist, lambda: trk.exchange_attach(handles), capi.DvoAmdError)
        if ok:  # the first tick: a peer whose writes never become visible times out here
            _, ok, why = sharding.collective_stage(dist, lambda: trk.match_sharded(ref, curs[0]), capi.DvoAmdError)
        if ok:
            out["peer"], poses["peer"] = timed(trk)
            out["peer"]["verified_over_xgmi_before_this_run"] = False
            if "rccl" in poses:
                diff = max(float(np.abs(a - b).max()) for a, b in zip(poses["peer"], poses["rccl"]))
                out["peer"]["max_abs_pose_difference_to_rccl_path"] = diff  # same records, same fold: expected exactly 0
                out["peer"]["agrees_with_rccl_path"] = bool(diff == 0.0)
        else:
            out["peer"] = {"unavailable": why, "fallback": "rccl"}
            if want == "peer":
                raise SystemExit(f"bench.py: DVO_AMD_EXCHANGE=peer but the peer exchange did not come up ({why})")
        del trk
    # the figure quoted for config 4: the collective path unless only the peer path was asked for (or only it came up)
    best = "rccl" if "rccl" in out and "pairs_per_s" in out.get("rccl", {}) else "peer"
    out["quoted"] = best
    out["pairs_per_s"] = out[best]["pairs_per_s"]
    out["us_per_tick"] = out[best]["us_per_tick"]
    out["ms_per_step"] = out[best]["ms_per_step"]
    return out


def tile_shard_bench(args, capi, synth, sharding, dist, rank, world, device, ref_frame, cur_frames, K, levels, first_level):
    """--tile-shard: config 4 as the bench line itself (strong scaling of one pair)."""
    m = tile_shard_measure(args, capi, synth, dist, rank, world, device, K, levels, first_level, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "frame-pairs/s (640x480, 4-level GN align)", "value": m["pairs_per_s"], "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": m["what"], "exchange": m["quoted"],
                       "sharding": "tile-shard with per-iteration exchange of the band records (BASELINE config 4)"},
            "us_per_tick": m["us_per_tick"], "tile_shard": m,
        }), file=OUT, flush=True)
    if dist is not None:
        dist.destroy_process_group()


def host_thread_plan(args, world, numa):
    """Every host thread of a rank spins on pinned memory while it waits for a tick (wait_tick): ranks x threads spinners per
    node.  Records what this rank runs with and caps the threads so that the ranks that share this rank's cores (all `world`
    ranks spread evenly over the NUMA nodes, when the rank could be pinned to its GPU's node; else over the whole affinity mask)
    leave one core in four free: 8 ranks x 6 threads are 48 spinners -- fine on the 2 x 64-core hosts of the MI355X nodes seen
    so far, not on a 32-core host."""
    try:
        cpus = len(os.sched_getaffinity(0))
        n_nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()]) or 1
    except Exception:
        cpus, n_nodes = os.cpu_count() or 1, 1
    sharing = -(-world // n_nodes) if numa is not None else world  # ranks that spin on the cores this rank may use
    cap = max(1, (3 * cpus // 4) // max(1, sharing))
    asked = args.threads
    if args.threads > cap:
        args.threads = cap
    return {"threads_per_rank": args.threads, "threads_asked_for": asked, "ranks": world, "spin_polling_threads_in_the_job": args.threads * world,
            "cpus_in_this_ranks_affinity_mask": cpus, "ranks_sharing_those_cpus": sharing, "cap_applied": asked > cap,
            "numa_node_of_the_gpu": numa,
            "rule": "threads per rank <= 3/4 of the cores in the rank's affinity mask / ranks sharing them"}


def pin_to_gpu_numa_node(device):
    """Keep this rank's host threads (they poll pinned memory the GPU writes) on the CPU socket the GPU hangs off.  Best
    effort: any failure, or an affinity mask that does not reach that node, leaves the process as it is."""
    try:
        import torch

        p = torch.cuda.get_device_properties(device)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0) & cpus
        if len(allowed) >= 8:
            os.sched_setaffinity(0, allowed)
            return node
    except Exception:
        pass
    return None


TRAFFIC_FILE = "r05_traffic.json"


def profiled_workload(args):
    """The counter passes ran the default workload (the driver's command): their figure says nothing about another one."""
    return (args.width, args.height, args.batch, args.distinct, args.distinct_refs, args.threads, args.in_flight) == \
           (640, 480, 1152, 96, 12, 6, 62) and not args.no_stats and not args.drain_between_steps


def roofline_traffic(args, world, line):
    """HBM-side bytes per k_tick launch of the timing pass: (traffic, traffic_source).

    A property of THIS run when it can be: on rank 0 of a one-GPU run bench.py starts the rocprofv3 counter passes itself
    (--pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each: they do not fit one; a third pass with SQ_INSTS_VALU + SQ_INSTS_MFMA
    replaces roofline.issue's per-step model by the instruction counts of the launches themselves) as child processes running the same workload
    in its short form (--counter-leg), and summarises the k_tick dispatches between the two k_marker dispatches that bracket
    the timing pass (dvo_slam_amd/pmc.py; corrected as MI355X_MICROARCH.md prescribes for gfx950).  `traffic` is the upper
    bound (2 x FETCH_SIZE + WRITE_SIZE); the bounds, the write ratio and the wasted-traffic ratio are in traffic_source.
    Otherwise (--no-live-counters, several GPUs, no rocprofv3, a failing pass): the committed profile of the driver's command,
    for that workload only; null for any other."""
    here = os.path.dirname(os.path.abspath(__file__))
    why_not = None
    if world == 1 and not args.no_live_counters and os.environ.get("DVO_BENCH_LIVE_COUNTERS", "1") != "0":
        from dvo_slam_amd import pmc

        leg = ["--counter-leg", "--steps", "1", "--warmup", "0", "--prime", "1", "--batch", str(args.batch), "--width", str(args.width),
               "--height", str(args.height), "--distinct", str(args.distinct), "--distinct-refs", str(args.distinct_refs),
               "--threads", str(args.threads), "--in-flight", str(args.in_flight)]
        leg += ["--no-stats"] if args.no_stats else []
        leg += ["--drain-between-steps"] if args.drain_between_steps else []
        try:
            t0 = time.perf_counter()
            live = pmc.measure_live(os.path.abspath(__file__), leg)
            d = live["traffic"]
            d["measured"] = "in this run"
            d["seconds_spent_on_the_counter_passes"] = time.perf_counter() - t0
            d["alg_bytes_per_launch_of_this_runs_timing_pass"] = line["roofline"]["alg_bytes_per_launch"]
            iss = live.get("issue")
            if iss and "issue_cycles" in iss and line["roofline"].get("avg_launch_us"):
                # the issue roofline from the instruction counts of these very launches (the per-step model from a committed
                # profile stays beside it as `issue_model`)
                k_s = line["roofline"]["avg_launch_us"] * 1e-6 * line["roofline"]["launches"]
                iss["peak_cycles"] = iss["simds"] * iss["peak_clock_hz"] * k_s
                iss["frac"] = iss["issue_cycles"] / iss["peak_cycles"]
                line["roofline"]["issue_model"] = line["roofline"].get("issue")
                line["roofline"]["issue"] = iss
            elif iss:
                line["roofline"]["issue_live_pass"] = iss
            return d["traffic_bytes_per_launch"], d
        except Exception as exc:  # the bench line never depends on the profiler
            why_not = repr(exc)[:400]
    if not profiled_workload(args):
        return None, ({"live_counter_passes_failed": why_not} if why_not else None)
    try:
        d = json.load(open(os.path.join(here, "profiles", TRAFFIC_FILE)))
        src = {"measured": "not in this run: the committed counter passes of the same command", "file": "profiles/" + TRAFFIC_FILE,
               "command": d.get("command"), "alg_bytes_per_launch_of_that_run": d.get("alg_bytes_per_launch"),
               "pairs_per_s_under_pmc": d.get("bench_value_under_pmc"),
               "wasted_traffic_ratio_bounds": d.get("wasted_traffic_ratio_bounds"),
               "write_ratio_to_algorithmic": d.get("write_ratio_to_algorithmic")}
        if why_not:
            src["live_counter_passes_failed"] = why_not
        return d["traffic_bytes_per_launch"], src
    except Exception:
        return None, ({"live_counter_passes_failed": why_not} if why_not else None)


def stream_copy(device):
    """The 'achievable' HBM figure SURVEY.md 8d asks for: a plain device-to-device copy of 1 GiB on the same GPU
    (read + write bytes / time), next to the nominal 8 TB/s the roofline uses."""
    import torch

    n = 1 << 28  # floats: 1 GiB
    a = torch.empty(n, dtype=torch.float32, device=f"cuda:{device}").normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    return {"GBps": 2 * 4 * n / ms / 1e6, "what": "torch device-to-device copy of 1 GiB, read+write bytes / time",
            "frac_of_nominal_peak": 2 * 4 * n / ms / 1e6 / HBM_PEAK_GBS}


def ingest_timing(capi, synth, frames, K, levels, device):
    """Frame ingest: a full pyramid straight from a raw uint8 BGR + uint16 depth frame."""
    import torch

    raws = [synth.to_raw(I, Z) for I, Z in frames[:4]]
    h, w = raws[0][1].shape
    for bgr, z in raws:  # warm the slab pool
        capi.RgbdImagePyramid.from_raw(bgr, z, K, levels, device=device)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        for bgr, z in raws:
            capi.RgbdImagePyramid.from_raw(bgr, z, K, levels, device=device)
    host_ms = (time.perf_counter() - t0) * 1e3 / (reps * len(raws))
    dev = [(torch.from_numpy(bgr).to(f"cuda:{device}"), torch.from_numpy(z.view("int16")).to(f"cuda:{device}")) for bgr, z in raws]
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        for bgr, z in dev:
            capi.RgbdImagePyramid.from_raw_device(bgr.data_ptr(), 3, z.data_ptr(), w, h, K, levels, device=device)
    dev_ms = (time.perf_counter() - t0) * 1e3 / (reps * len(raws))
    return {"ms_per_frame_from_host_raw": host_ms, "ms_per_frame_from_device_raw": dev_ms,
            "what": f"{w}x{h} uint8 BGR + uint16 depth -> {levels}-level pyramid (gray conversion, depth scaling, "
                    "derivatives, gather layout) through the Python binding; host: pageable memory, 5 B/px over PCIe"}


def validator_timing(capi, synth, W, H, device):
    """BASELINE config 5 in the reference's real shape: one keyframe against 32 candidates, two
    proposals each, stage 1 = level 3 only on proposal + inverse, stage 2 = levels 3..1 on the survivors, keep-best."""
    from dvo_slam_amd import constraints as Cn

    K = synth.intrinsics_for(W, H)
    key, cands = synth.loop_closure_scenario(W, H, 32, decoys=False)
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=1), device=device)

    def mk(e):
        p = capi.RgbdImagePyramid(e["frame"][0], e["frame"][1], K, 4, device=device)
        return Cn.Keyframe(e["id"], p, e["pose"], Cn.LogLikelihoodTrackingResultEvaluation(trk.match(p, p)))

    kkey, kc = mk(key), [mk(c) for c in cands]
    val = Cn.createConstraintProposalValidator(min_constraint_ratio=0.2, ratio_coarse=-1e300, ratio_fine=-1e300,
                                               device=device, max_in_flight=72)
    val.validate(Cn.proposalsForCandidates(kkey, kc))
    reps = 5
    native = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        out = val.validate(Cn.proposalsForCandidates(kkey, kc))
        native += val.native_ms
    ms = (time.perf_counter() - t0) * 1e3 / reps
    native /= reps
    n_align = 2 * 64 + 64
    # the metric form of config 5: the same 64 (keyframe, candidate, initial transform) pairs as 64 full
    # alignments over levels 3..0 in one batch, the keyframe's point selection cached once
    props = Cn.proposalsForCandidates(kkey, kc)
    full = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0, UseInitialEstimate=True), device=device)
    refs, curs_ = [pr.Reference.image for pr in props], [pr.Current.image for pr in props]
    inits = [pr.InitialTransformation for pr in props]
    full.match_batch(refs, curs_, T_inits=inits, stats=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        full.match_batch(refs, curs_, T_inits=inits, stats=False)
    ms_full = (time.perf_counter() - t0) * 1e3 / reps
    return {"ms_per_validate": ms, "ms_per_validate_library_call": native, "proposals": 64, "alignments": n_align,
            "alignments_per_s": n_align / ms * 1e3, "alignments_per_s_library_call": n_align / native * 1e3,
            "constraints_kept": len(out),
            "full_alignment_of_the_64_pairs": {"ms": ms_full, "pairs_per_s": 64 / ms_full * 1e3,
                                               "what": "64 match() over levels 3..0 from the proposals' initial transforms, "
                                                       "one batch on one tracker (one host thread, two launches per tick)"},
            "what": "dvo_amd_validate_proposals: 64 proposals (32 candidates x {identity, relative pose}), stage 1 = 128 "
                    "level-3 alignments (proposals + cross-validation inverses), stage 2 = 64 alignments over levels 3..1, "
                    "evaluation thresholds open so that every proposal reaches stage 2"}


def dual_match_timing(capi, keyframe, frames, levels, first_level, device):
    """the two alignments LocalTracker::update runs per frame (keyframe -> frame, last frame -> frame) as
    one two-pair batch (dvo_amd_track_frame) against two single match() calls one after the other."""
    trk = capi.DenseTracker(capi.Config(FirstLevel=first_level, LastLevel=1, UseInitialEstimate=True), device=device)
    eye = np.eye(4)
    n = 20
    trk.track_frame(keyframe, frames[0], frames[1], eye)
    t0 = time.perf_counter()
    for i in range(n):
        trk.track_frame(keyframe, frames[i % 4], frames[(i + 1) % 4], eye)
    fused = (time.perf_counter() - t0) * 1e3 / n
    t0 = time.perf_counter()
    for i in range(n):
        trk.match(keyframe, frames[(i + 1) % 4], eye)
        trk.match(frames[i % 4], frames[(i + 1) % 4], eye)
    serial = (time.perf_counter() - t0) * 1e3 / n
    return {"ms_per_frame_two_pair_batch": fused, "ms_per_frame_two_single_matches": serial,
            "what": "levels 3..1 (the reference's default LastLevel), keyframe -> frame and last frame -> frame"}


def cpu_baseline_threads(args, ref_frame, cur_frames, K, levels, first_level):
    """one oracle tracker per host thread over independent pairs (the shape of tbb::parallel_reduce over
    proposals, keyframe_graph.cpp:587-590), the threads driven from C (orc_bench_threads).  Threads = the CPUs of this process's
    affinity mask, at most 64: a bounded side measurement on a host that other jobs share -- named for what it is, not "all
    cores" (round 4 ran the same 64 threads from Python and scaled 10.6x: most of a call was the binding's result marshalling
    under the interpreter lock, not the alignment)."""
    from oracle import oracle as orc

    orc.select_build("native")  # -O3 -march=native, the reference's flags (dvo_core/CMakeLists.txt:38-40)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    n_threads = max(1, min(avail, 64))
    pr = orc.Pyramid(ref_frame[0], ref_frame[1], K, levels)
    pcs = [orc.Pyramid(f[0], f[1], K, levels) for f in cur_frames]
    cfg = orc.default_config(first_level=first_level, last_level=0, rcp_mode=orc.RCP_SSE)
    budget = min(args.cpu_seconds, 8.0)
    n, dt = orc.bench_threads(cfg, pr, pcs, n_threads, budget)
    del pr, pcs
    orc.select_build("parity")
    quota = None  # the CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota): what the threads really share
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) if q > 0 else None
        except Exception:
            pass
    return {"value": n / dt, "unit": "frame-pairs/s", "cores": n_threads, "kind": "port", "cpus_in_the_affinity_mask": avail,
            "host_hardware_threads": os.cpu_count(), "container_cpu_quota_cores": quota,
            "sample": f"{n} match() calls in {dt:.1f} s on {n_threads} threads (C threads, one oracle tracker each, shared pyramids); "
                      f"the cap of 64 threads is this bench's, not the host's"}


def cpu_baseline(args, ref_frame, cur_frames, K, levels, first_level):
    """The oracle (CPU restatement of the reference's SSE path, rcp_mode = SSE like the reference) timed single-threaded
    on this host over a bounded sample of the same pairs; match() only, pyramids pre-built."""
    from oracle import oracle as orc

    orc.select_build("native")  # -O3 -march=native, the reference's flags (dvo_core/CMakeLists.txt:38-40)
    pr = orc.Pyramid(ref_frame[0], ref_frame[1], K, levels)
    pcs = [orc.Pyramid(f[0], f[1], K, levels) for f in cur_frames]
    cfg = orc.default_config(first_level=first_level, last_level=0, rcp_mode=orc.RCP_SSE)
    orc.match(cfg, pr, pcs[0])
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        orc.match(cfg, pr, pcs[n % len(pcs)])
        n += 1
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n_pairs = len(pcs)
    del pr, pcs
    orc.select_build("parity")
    return {"value": n / dt, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} match() calls over {n_pairs} of the synthetic pairs in {dt:.1f} s, single thread, "
                      f"oracle/dvo_oracle.c (restated reference SSE path with _mm_rcp_ps) built with the reference's flags "
                      f"-O3 -march=native -msse3 on this host, pyramids pre-built",
            "host_cpu": model, "host_cores": os.cpu_count()}


if __name__ == "__main__":
    main()
