"""A pair's result is a function of its inputs alone.

The reference runs independent match() calls under tbb::parallel_reduce (dvo_slam/src/keyframe_graph.cpp:587-590) and
tbb::parallel_invoke (dvo_slam/src/local_tracker.cpp:184): whatever runs beside a pair cannot change its result.  Here the
geometry of a residual pass -- where the fp32 sums are cut -- is a function of the pyramid level alone (level_steps,
csrc/dvo_tracker.cpp), and everything summed across blocks follows one tree per level (csrc/dvo_types.h, "the summation tree of a level"), so

    match()  ==  the same pair in match_batch at any residency  ==  through the submit queue  ==  validated by any number of
    validator workers  ==  match_banded / match_sharded at 1, 2, 4, 8, 16 bands

BIT FOR BIT: transformation, information, likelihood and every per-iteration statistic (np.array_equal, no tolerance).  Band
counts that do not divide 16 (3, 5, ...) cut a level off the chunk boundaries of that tree: same iteration path, equal to
~1e-15 (asserted at 1e-12).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from dvo_slam_amd import capi as c

    if c.lib().dvo_amd_device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return c


@pytest.fixture(scope="module")
def frames(capi, synth):
    """six 640x480 views along a trajectory (every ordered pair of them is a different alignment problem)"""
    K = synth.intrinsics_for(640, 480)
    poses = synth.stream_poses(6, synth.XI_STEP_STREAM * 1.5)
    return [capi.RgbdImagePyramid(*synth.render(640, 480, poses[t], frame_id=t), K, 4) for t in range(6)]


def assert_same_result(a, b, what=""):
    """bit equality of everything a Result carries (NaN == NaN: a NaN result must be the same NaN result)"""
    assert np.array_equal(a.Transformation, b.Transformation, equal_nan=True), what
    assert np.array_equal(a.Information, b.Information, equal_nan=True), what
    assert a.LogLikelihood == b.LogLikelihood or (a.LogLikelihood != a.LogLikelihood and b.LogLikelihood != b.LogLikelihood), what
    assert a.isNaN() == b.isNaN() and len(a.Levels) == len(b.Levels), what
    for la, lb in zip(a.Levels, b.Levels):
        assert (la["Id"], la["ValidPixels"], la["TerminationCriterion"], len(la["Iterations"])) == \
               (lb["Id"], lb["ValidPixels"], lb["TerminationCriterion"], len(lb["Iterations"])), what
        for ia, ib in zip(la["Iterations"], lb["Iterations"]):
            assert ia["ValidConstraints"] == ib["ValidConstraints"], what
            assert ia["TDistributionLogLikelihood"] == ib["TDistributionLogLikelihood"], what
            for key in ("TDistributionPrecision", "EstimateIncrement", "EstimateInformation", "estimate", "initial"):
                assert np.array_equal(ia[key], ib[key]), (what, key)


def _pairs(frames, n):
    idx = [(i, j) for i in range(len(frames)) for j in range(len(frames)) if i != j]
    return [idx[k % len(idx)] for k in range(n)]


@pytest.fixture(scope="module")
def singles(capi, frames):
    """every ordered pair through a single match() on its own tracker"""
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    return {ij: trk.match(frames[ij[0]], frames[ij[1]]) for ij in _pairs(frames, 30)}


@pytest.mark.parametrize("residency", [2, 8, 9, 36, 62, 96, 124])
def test_batch_at_any_residency_equals_single_match(capi, frames, singles, residency):
    """2 and 8 resident pairs run behind the small argument blocks and the 512-thread reducer, 9 and more behind the full-size
    launch and the 256-thread reducer; 62 fill one launch (items sorted by block life, blocks rotated over the XCDs per item),
    96 and 124 make two groups on their own streams: the same bits every time"""
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    pairs = _pairs(frames, 130)
    out = trk.match_batch([frames[i] for i, _ in pairs], [frames[j] for _, j in pairs], in_flight=residency)
    for ij, r in zip(pairs, out):
        assert_same_result(singles[ij], r, f"pair {ij} at residency {residency}")


def test_lock_step_batch_and_initial_estimates(capi, synth, frames, singles):
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    pairs = _pairs(frames, 30)
    out = trk.match_batch([frames[i] for i, _ in pairs], [frames[j] for _, j in pairs])  # in_flight = 0: all 30 resident
    for ij, r in zip(pairs, out):
        assert_same_result(singles[ij], r, f"pair {ij} in a 30-pair lock-step batch")
    # the reference's default levels with an initial estimate and a prior: again single == batched
    cfg = capi.Config(FirstLevel=3, LastLevel=1, UseInitialEstimate=True, Mu=0.05)
    a, b = capi.DenseTracker(cfg), capi.DenseTracker(cfg)
    T0 = [synth.se3_exp(np.array([0.003, -0.002, 0.001, 0.001, -0.001, 0.0005]) * (k % 5)) for k in range(20)]
    pairs = _pairs(frames, 20)
    batch = b.match_batch([frames[i] for i, _ in pairs], [frames[j] for _, j in pairs], T_inits=T0, in_flight=12)
    for (i, j), t0, r in zip(pairs, T0, batch):
        assert_same_result(a.match(frames[i], frames[j], t0), r, f"pair {(i, j)} with an initial estimate")


def test_submit_queue_equals_single_match(capi, frames, singles):
    """three overlapping submissions, waited for out of order and polled: what a pair shares its ticks with is decided by
    timing; its result is not"""
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    plans = [_pairs(frames, 50), _pairs(frames, 7)[::-1], _pairs(frames, 90)[5:]]
    subs = [trk.submit([frames[i] for i, _ in p], [frames[j] for _, j in p], in_flight=40) for p in plans]
    spins = 0
    while not trk.poll(subs[2]):
        spins += 1
        assert spins < 10_000_000
    for p, res in ((plans[1], trk.wait(subs[1])), (plans[0], trk.wait(subs[0])), (plans[2], subs[2].results())):
        for ij, r in zip(p, res):
            assert_same_result(singles[ij], r, f"pair {ij} through the queue")


@pytest.mark.parametrize("n_bands", [1, 2, 4, 8, 16])
def test_bands_that_divide_16_equal_the_unsharded_match_bit_for_bit(capi, frames, singles, n_bands):
    """a band is a run of whole chunks of the level's summation tree; its reducer returns its subtree and the host folds the
    bands with the rest of the same tree (csrc/dvo_types.h): the record of the level is the unsharded one, bit for bit"""
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    for ij in _pairs(frames, 6):
        assert_same_result(singles[ij], trk.match_banded(frames[ij[0]], frames[ij[1]], n_bands), f"pair {ij}, {n_bands} bands")


@pytest.mark.parametrize("n_bands", [3, 5, 12])
def test_other_band_counts_agree_to_rounding(capi, synth, frames, singles, n_bands):
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    for ij in _pairs(frames, 3):
        a, b = singles[ij], trk.match_banded(frames[ij[0]], frames[ij[1]], n_bands)
        assert [[it["ValidConstraints"] for it in L["Iterations"]] for L in a.Levels] == \
               [[it["ValidConstraints"] for it in L["Iterations"]] for L in b.Levels]
        assert [L["TerminationCriterion"] for L in a.Levels] == [L["TerminationCriterion"] for L in b.Levels]
        assert synth.pose_error(a.Transformation, b.Transformation) <= 1e-12


def test_1280x960_five_levels_single_equals_batch_equals_bands(capi, synth):
    """config 3's size: level 0 has 19 200 wave steps (16-step segments, 300 blocks), five levels"""
    K = synth.intrinsics_for(1280, 960)
    poses = synth.stream_poses(3, synth.XI_STEP_STREAM * 1.5)
    pyr = [capi.RgbdImagePyramid(*synth.render(1280, 960, poses[t], frame_id=t), K, 5) for t in range(3)]
    trk = capi.DenseTracker(capi.Config(FirstLevel=4, LastLevel=0))
    pairs = [(0, 1), (1, 2), (2, 0), (0, 2)]
    single = [trk.match(pyr[i], pyr[j]) for i, j in pairs]
    batch = trk.match_batch([pyr[i] for i, _ in pairs] * 5, [pyr[j] for _, j in pairs] * 5, in_flight=12)
    for k, r in enumerate(batch):
        assert_same_result(single[k % 4], r, f"1280x960 pair {pairs[k % 4]} in a batch")
    for n_bands in (2, 8):
        assert_same_result(single[0], trk.match_banded(pyr[0], pyr[1], n_bands), f"1280x960, {n_bands} bands")


def test_odd_sizes_single_equals_batch(capi, synth):
    """level sizes that are not multiples of a block (ragged last blocks, chunks of unequal length, empty chunks on the
    coarsest levels)"""
    for (w, h, levels) in ((336, 250, 3), (64, 48, 2), (1008, 500, 3)):
        K = synth.intrinsics_for(w, h)
        poses = synth.stream_poses(3, synth.XI_STEP_STREAM * 1.5)
        pyr = [capi.RgbdImagePyramid(*synth.render(w, h, poses[t], frame_id=t), K, levels) for t in range(3)]
        trk = capi.DenseTracker(capi.Config(FirstLevel=levels - 1, LastLevel=0))
        single = [trk.match(pyr[0], pyr[1]), trk.match(pyr[1], pyr[2])]
        batch = trk.match_batch([pyr[0], pyr[1]] * 6, [pyr[1], pyr[2]] * 6, in_flight=10)
        for k, r in enumerate(batch):
            assert_same_result(single[k % 2], r, f"{w}x{h} in a batch")
        for n_bands in (2, 4, 16):
            assert_same_result(single[0], trk.match_banded(pyr[0], pyr[1], n_bands), f"{w}x{h}, {n_bands} bands")


def test_stage_probe_runs_the_geometry_match_runs(capi, synth, frames, singles):
    """dvo_amd_debug_iteration (the teacher-forced stage probe of the parity tests) at the pose and precision of a match()'s own
    iteration returns that iteration's valid count, precision and likelihood bit for bit: the probe measures the very
    arithmetic match() ran, not a sibling of it"""
    trk = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    r = singles[(0, 1)]
    checked = 0
    for L in r.Levels:
        prev_P = None
        for it in L["Iterations"]:
            probe = trk.iteration_probe(frames[0], frames[1], L["Id"], it["estimate"], precision_in=prev_P)
            assert probe["n"] == it["ValidConstraints"]
            if it["ValidConstraints"] >= 6:
                assert np.array_equal(probe["precision"].astype(np.float64), it["TDistributionPrecision"])
                assert -probe["ll"] == it["TDistributionLogLikelihood"]
                checked += 1
            prev_P = it["TDistributionPrecision"]
    assert checked >= 10


def test_the_latency_geometry_is_a_configuration_like_any_other(capi, synth, frames, singles):
    """dvo_amd_config::segment_geometry = DVO_AMD_GEOMETRY_LATENCY (round 5; 640x480 levels 3..0 in 1 / 2 / 2 / 4 steps per wave:
    the shortest single match()).  It is part of what a result is a function of, like every other field of the configuration:
    under it match() == batch at any residency == the queue == every band count, bit for bit -- and it is NOT the default
    geometry's result bit for bit (another summation order: agreement to summation noise, the same path or a fork)."""
    cfg = capi.Config(FirstLevel=3, LastLevel=0, SegmentGeometry=capi.GEOMETRY_LATENCY)
    single = capi.DenseTracker(cfg)
    pairs = _pairs(frames, 30)
    want = {ij: single.match(frames[ij[0]], frames[ij[1]]) for ij in pairs}
    trk = capi.DenseTracker(cfg)
    many = _pairs(frames, 130)
    for residency in (8, 62, 124):
        out = trk.match_batch([frames[i] for i, _ in many], [frames[j] for _, j in many], in_flight=residency)
        for ij, r in zip(many, out):
            assert_same_result(want[ij], r, f"latency geometry: pair {ij} at residency {residency}")
    subs = [trk.submit([frames[i] for i, _ in p], [frames[j] for _, j in p], in_flight=40) for p in (pairs, pairs[::-1][:7])]
    for p, s in zip((pairs, pairs[::-1][:7]), subs):
        for ij, r in zip(p, trk.wait(s)):
            assert_same_result(want[ij], r, f"latency geometry: pair {ij} through the queue")
    for n_bands in (1, 2, 8, 16):
        for ij in pairs[:4]:
            assert_same_result(want[ij], trk.match_banded(frames[ij[0]], frames[ij[1]], n_bands), f"latency geometry, {n_bands} bands")
    # another configuration, another summation order: close, not identical
    differs = 0
    for ij in pairs:
        d = synth.pose_error(singles[ij].Transformation, want[ij].Transformation)
        assert d < 3e-4, (ij, d)
        differs += not np.array_equal(singles[ij].Transformation, want[ij].Transformation)
    assert differs > 0
    # an unknown geometry is refused like any insane configuration
    with pytest.raises(capi.DvoAmdError):
        capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0, SegmentGeometry=7))


def test_the_segment_tables_are_the_documented_ones(capi, synth, frames):
    """dvo_amd.h documents what dvo_amd_config::segment_geometry selects; dvo_amd_debug_level_geometry reads it back.  Throughput:
    640x480 levels 3..0 in 4 / 4 / 10 / 10 steps of 64 pixels per wave -- on a level of 64 000 pixels or more a wave segment is a
    whole number of image rows (ten steps = one 640-pixel row = two 320-pixel rows, twenty for 1280), 16 steps where a row is no
    whole number of steps; latency: 1 / 2 / 2 / 4 (1280x960 levels 4..0: 1 / 2 / 2 / 4 / 4).  A function of the level's size and
    the configuration alone: two trackers of one configuration agree, whatever they ran before."""
    thr = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    lat = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0, SegmentGeometry=capi.GEOMETRY_LATENCY))
    vga = frames[0]
    assert [thr.level_geometry(vga, l)[0] for l in (3, 2, 1, 0)] == [4, 4, 10, 10]
    assert [lat.level_geometry(vga, l)[0] for l in (3, 2, 1, 0)] == [1, 2, 2, 4]
    for trk in (thr, lat):
        for l in range(4):
            steps, blocks, points = trk.level_geometry(vga, l)
            n = (640 >> l) * (480 >> l)
            assert 0 < points <= n and points % 2 == 0  # the selected pixels, without an odd trailing one (Q3)
            count, mask = vga.select(l)  # PointSelection::select through the boundary: the same pixels, counted before Q3
            assert count == int(mask.sum()) and points == count - (count & 1)
            assert (blocks - 1) * steps * 256 < points <= blocks * steps * 256  # the blocks cover the points, none is empty
    big = capi.RgbdImagePyramid(*synth.render(1280, 960, frame_id=40), synth.intrinsics_for(1280, 960), 5)
    assert [thr.level_geometry(big, l)[0] for l in (4, 3, 2, 1, 0)] == [4, 4, 10, 10, 20]
    assert [lat.level_geometry(big, l)[0] for l in (4, 3, 2, 1, 0)] == [1, 2, 2, 4, 4]
    # a row that is no whole number of steps keeps the table's 16 (600 = 9.375 steps; level 1: 300 x 224 pixels)
    odd = capi.RgbdImagePyramid(*synth.render(600, 448, frame_id=41), synth.intrinsics_for(600, 448), 2)
    assert [thr.level_geometry(odd, l)[0] for l in (1, 0)] == [16, 16]
    thr.match(frames[1], frames[2])
    other = capi.DenseTracker(capi.Config(FirstLevel=3, LastLevel=0))
    assert [thr.level_geometry(vga, l) for l in range(4)] == [other.level_geometry(vga, l) for l in range(4)]
    with pytest.raises(capi.DvoAmdError):
        thr.level_geometry(vga, 4)


def test_trackers_created_back_to_back_are_spread_over_the_hardware_queues(capi, capsys):
    """The runtime maps streams onto its four hardware queues and a hardware queue runs one kernel at a time: how the trackers of a
    GPU are spread over them decides up to a third of a batch's throughput (profiles/r05_stream_queue_assignment_ab.txt: 55-56 k
    pairs/s for six trackers two to a queue with neighbours together, 47 k on two queues, 39 k on one).  The library takes the stream
    the runtime deals it (a context that chose among probed candidates made the common case worse); this is the diagnostic that shows
    the outcome -- the hardware queue of a tracker's stream, asked of the GPU -- on six trackers created back to back, as
    INTEGRATION.md advises: no queue may carry more than two of them."""
    from collections import Counter

    cfg = capi.Config(FirstLevel=3, LastLevel=0)
    trackers = [capi.DenseTracker(cfg) for _ in range(6)]
    queues = [t.hw_queue() for t in trackers]
    with capsys.disabled():
        print(f"\n[hardware queues] six trackers created back to back: pipe << 3 | queue = {queues}")
    assert all(0 <= q < 64 for q in queues)
    assert max(Counter(queues).values()) <= 2
