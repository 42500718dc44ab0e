"""Pair-level sharding for multi-GPU runs: one process per GPU, independent frame pairs per rank, no data-path collective.

A DenseTracker::match() call is an independent unit (the reference runs them as TBB tasks: local_tracker.cpp:184,
keyframe_graph.cpp:576-593), so N GPUs simply take N disjoint subsets of the pairs.  torch.distributed (backend "nccl" =
RCCL on ROCm, "gloo" in the CPU tests) is used only for the barrier around the timed region, the MAX over ranks of the
elapsed time and the SUM of the pairs processed.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple


def rank_info() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin share of `n_items` global items for `rank`: every item belongs to exactly one rank and the shares
    differ by at most one item (a loop-closure proposal list is dealt over the GPUs like this)."""
    if world < 1 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard request")
    return list(range(rank, n_items, world))


def shard_weak(per_rank: int, rank: int, world: int) -> List[int]:
    """Weak scaling: every rank gets `per_rank` items of its own; global ids are contiguous per rank."""
    if world < 1 or not (0 <= rank < world) or per_rank < 0:
        raise ValueError("bad shard request")
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def aggregate(elapsed_s: float, n_pairs_local: int, dist=None, device=None) -> Tuple[float, int]:
    """(max over ranks of elapsed, total pairs over ranks).  `dist` is torch.distributed (initialised) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), int(n_pairs_local)
    import torch

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([n_pairs_local], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def all_ranks(dist, value) -> list:
    """Every rank's `value` in rank order (collective); a one-entry list without a process group."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [value]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, value)
    return out


def collective_stage(dist, fn, exc_type=Exception):
    """Run fn() on every rank; the stage counts only if it succeeded on ALL ranks, so that every rank takes the same decision
    (bench.py: the peer exchange of a tile-sharded pair falls back to RCCL when its set-up or its first tick fails anywhere).
    Returns (fn's value or None, ok, reason): `reason` names the lowest rank that failed and is the same string on every rank."""
    err = None
    val = None
    try:
        val = fn()
    except exc_type as exc:
        err = str(exc) or type(exc).__name__
    errs = all_ranks(dist, err)
    bad = [(r, e) for r, e in enumerate(errs) if e]
    if bad:
        return None, False, f"rank {bad[0][0]}: {bad[0][1]}"
    return val, True, None


def split_for_threads(items: Sequence, n_threads: int) -> List[list]:
    """Deal a rank's items over its host threads (one tracker / HIP stream each)."""
    n_threads = max(1, min(n_threads, max(1, len(items))))
    return [list(items[t::n_threads]) for t in range(n_threads)]


# ---------------------------------------------------------------------------------------------------------------------
# One-hop exchange of band records (tile-sharded pairs): the slot / tag protocol of exchange_records (csrc/dvo_kernels.hip),
# restated on the host so that its ordering rules can be exercised without GPUs (tests/test_distributed.py).
# ---------------------------------------------------------------------------------------------------------------------
def next_seq(seq: int) -> int:
    """Tick numbers of the exchange (csrc/dvo_types.h next_seq): 32 bits, 0 is never used (it is what a fresh buffer holds) and
    the wrap lands on 2, so that the parity -- the generation of a slot -- keeps alternating."""
    seq = (seq + 1) & 0xFFFFFFFF
    return 2 if seq == 0 else seq


def exchange_slot(seq: int, n_ranks: int, rank: int) -> int:
    """Slot of `rank`'s record of tick `seq` in every rank's exchange buffer: two generations of n_ranks slots."""
    return (seq & 1) * n_ranks + rank


class ExchangeBuffer:
    """One rank's exchange buffer over any writable (2 * n_ranks, pieces, 4) array (shared memory in the tests, fine-grained
    device memory mapped by the peers on the GPUs).  A record travels as pieces of two halves {payload word, tick number},
    each piece written in one go: a piece is valid when both its tags are the tick, so there is no "ready" word and no
    ordering between the pieces (csrc/dvo_types.h: FinWire)."""

    PAYLOAD = 2

    def __init__(self, array, n_ranks: int):
        self.a = array
        self.n = n_ranks
        assert array.shape[0] == 2 * n_ranks and array.shape[2] == 4

    @classmethod
    def pieces_for(cls, words: int) -> int:
        return (words + cls.PAYLOAD - 1) // cls.PAYLOAD

    def publish(self, seq: int, rank: int, payload, order=None):
        """`order`: the order the pieces are written in (any order is fine; the tests write them backwards)"""
        row = self.a[exchange_slot(seq, self.n, rank)]
        n = self.pieces_for(len(payload))
        for i in (range(n) if order is None else order):
            chunk = list(payload[self.PAYLOAD * i:self.PAYLOAD * (i + 1)]) + [0]
            row[i] = [chunk[0], seq, chunk[1], seq]  # one store per piece

    def ready(self, seq: int, words: int) -> bool:
        n = self.pieces_for(words)
        return all(all(int(t) == seq for t in self.a[exchange_slot(seq, self.n, r), :n, 1::2].reshape(-1)) for r in range(self.n))

    def collect(self, seq: int, words: int):
        """the n records of tick `seq` in rank order (call once ready(seq, words))"""
        n = self.pieces_for(words)
        return [self.a[exchange_slot(seq, self.n, r), :n, 0::2].reshape(-1)[:words].copy() for r in range(self.n)]
