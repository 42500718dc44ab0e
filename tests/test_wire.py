"""The host side of the record hand-off (csrc/dvo_tracker.cpp: take_wire): a tick's record arrives as 16-byte pieces made of two
8-byte halves {payload word, tick number}; a piece counts when BOTH its tags are the tick waited for.  Host logic only, no GPU."""
import numpy as np
import pytest


def _aligned(n_words):
    raw = np.zeros(n_words + 4, np.uint32)
    off = (-raw.ctypes.data // 4) % 4
    a = raw[off:off + n_words]
    assert a.ctypes.data % 16 == 0
    return a


@pytest.fixture(scope="module")
def capi():
    from dvo_slam_amd import capi as c
    c.lib()
    return c


def _wire_for(payload, pieces, tick):
    w = _aligned(pieces * 4).reshape(pieces, 4)
    for i in range(pieces):
        w[i, 0] = payload[2 * i]
        w[i, 1] = tick
        if 2 * i + 1 < len(payload):
            w[i, 2] = payload[2 * i + 1]
        w[i, 3] = tick
    return w


def test_layout_covers_the_record(capi):
    pieces, words = capi.wire_layout()
    assert words == 196 and pieces == (words + 1) // 2  # FinOut is 784 bytes (include/dvo_amd.h: the record of a tick)


def test_complete_record_is_reassembled(capi):
    pieces, words = capi.wire_layout()
    rng = np.random.default_rng(3)
    payload = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
    rec = np.zeros(words, np.uint32)
    assert capi.take_wire(_wire_for(payload, pieces, 7), 7, 0, rec) == pieces
    want = payload.copy()
    want[3] = 7  # the record's own sequence field is stamped with the tick once it is complete
    assert np.array_equal(rec, want)


def test_pieces_of_an_older_tick_are_not_taken_and_the_scan_resumes(capi):
    pieces, words = capi.wire_layout()
    rng = np.random.default_rng(4)
    old = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
    new = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
    wire = _wire_for(old, pieces, 11)
    rec = np.full(words, 0xDEADBEEF, np.uint32)
    assert capi.take_wire(wire, 12, 0, rec) == 0 and np.all(rec == 0xDEADBEEF)  # nothing of tick 12 has arrived
    # the pieces arrive in no particular order: the last ones first, then a block in the middle, then the rest
    fresh = _wire_for(new, pieces, 12)
    arrival = list(range(pieces - 5, pieces)) + list(range(20, 40)) + list(range(0, 20)) + list(range(40, pieces - 5))
    have = 0
    for k, i in enumerate(arrival):
        wire[i] = fresh[i]
        nxt = capi.take_wire(wire, 12, have, rec)
        arrived = set(arrival[:k + 1])
        first_missing = next((p for p in range(pieces) if p not in arrived), pieces)
        assert nxt == first_missing and nxt >= have
        have = nxt
    assert have == pieces
    want = new.copy()
    want[3] = 12
    assert np.array_equal(rec, want)


def test_a_piece_with_the_right_payload_but_the_wrong_tag_stops_the_scan(capi):
    pieces, words = capi.wire_layout()
    payload = np.arange(words, dtype=np.uint32)
    wire = _wire_for(payload, pieces, 5)
    wire[17, 3] = 4
    rec = np.zeros(words, np.uint32)
    assert capi.take_wire(wire, 5, 0, rec) == 17
    assert np.array_equal(rec[:34], payload[:34]) and not rec[34:].any()  # 17 pieces x 2 words taken, nothing beyond
    assert rec[3] == 3  # not stamped: the record is incomplete


@pytest.mark.parametrize("late_half", [0, 1])
def test_a_torn_piece_is_not_accepted_until_both_halves_carry_the_tick(capi, late_half):
    """Should a 16-byte store ever reach the host as two 8-byte halves at different times, the half that has landed carries the
    new tick and the other still the old one: the piece is refused -- never new tag with old payload -- and taken on a later poll
    once the second half is there (the hand-off must not rest on 16-byte single-copy atomicity)."""
    pieces, words = capi.wire_layout()
    rng = np.random.default_rng(9)
    old = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
    new = rng.integers(0, 2**32, size=words, dtype=np.uint64).astype(np.uint32)
    wire = _wire_for(old, pieces, 20)
    fresh = _wire_for(new, pieces, 21)
    torn = 33
    for i in range(pieces):
        if i != torn:
            wire[i] = fresh[i]
    early = 1 - late_half
    wire[torn, 2 * early:2 * early + 2] = fresh[torn, 2 * early:2 * early + 2]  # one half of the piece has landed
    rec = np.full(words, 0xABABABAB, np.uint32)
    assert capi.take_wire(wire, 21, 0, rec) == torn
    assert np.all(rec[2 * torn:] == 0xABABABAB)  # neither word of the torn piece (nor anything behind it) was taken
    wire[torn, 2 * late_half:2 * late_half + 2] = fresh[torn, 2 * late_half:2 * late_half + 2]
    assert capi.take_wire(wire, 21, torn, rec) == pieces
    want = new.copy()
    want[3] = 21
    assert np.array_equal(rec, want)


def test_a_fresh_zeroed_buffer_is_never_a_record(capi):
    """Sequence numbers skip 0 (csrc/dvo_types.h next_seq), the value a wire buffer is created with: an untouched buffer
    cannot be mistaken for tick 0's record at the 2^32 wrap.  The tick counter itself is exercised by the library's users;
    here: the receiving side refuses an all-zero buffer for every tick it can be asked for."""
    pieces, words = capi.wire_layout()
    wire = _aligned(pieces * 4).reshape(pieces, 4)
    rec = np.zeros(words, np.uint32)
    for tick in (1, 2, 0xFFFFFFFF):
        assert capi.take_wire(wire, tick, 0, rec) == 0


def test_misaligned_or_missing_buffers_are_refused(capi):
    pieces, words = capi.wire_layout()
    rec = np.zeros(words, np.uint32)
    raw = np.zeros(pieces * 4 + 8, np.uint32)
    off = (-raw.ctypes.data // 4) % 4
    mis = raw[off + 1:off + 1 + pieces * 4]
    with pytest.raises(capi.DvoAmdError):
        capi.take_wire(mis, 1, 0, rec)
    with pytest.raises(capi.DvoAmdError):
        capi.take_wire(_aligned(pieces * 4), 1, pieces + 1, rec)
