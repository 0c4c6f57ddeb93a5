"""Per-utterance data parallelism: one process per GPU, full weight replica on each, no collective on the data
path.  Utterances are independent in the reference's generate loop (qwen_tts/src/nn/generation.rs:815-1194 has no
cross-row coupling except loop termination), so the only communication is the final gather of results."""
from __future__ import annotations

from typing import Callable, Sequence


def assign_utterances(costs: Sequence[float], world_size: int) -> list:
    """Longest-processing-time-first assignment of utterances to ranks.  `costs[i]` is any monotone proxy of the
    work of utterance i (text length, requested frames).  Deterministic: ties go to the lower rank.
    Returns one ascending index list per rank."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    count = [0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], count[k], k))
        out[r].append(i)
        load[r] += float(costs[i])
        count[r] += 1
    return [sorted(x) for x in out]


def run_sharded(items: Sequence, costs: Sequence[float], fn: Callable[[list], list], rank: int, world_size: int):
    """Runs fn on this rank's share of `items`; returns (indices, results) for the share."""
    mine = assign_utterances(costs, world_size)[rank]
    res = fn([items[i] for i in mine]) if mine else []
    if len(res) != len(mine):
        raise RuntimeError("shard function returned a different number of results than utterances")
    return mine, res


def gather_results(indices: list, results: list, n_total: int, dist=None, dst: int = 0):
    """Result gather to rank `dst` (the only cross-rank traffic of the path).  Returns the results in utterance
    order on `dst`, None elsewhere.  Without an initialised process group it is the identity."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        out = [None] * n_total
        for i, r in zip(indices, results):
            out[i] = r
        return out
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object((indices, results), bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = [None] * n_total
    seen = 0
    for idx, res in bucket:
        for i, r in zip(idx, res):
            if out[i] is not None:
                raise RuntimeError(f"utterance {i} was produced by two ranks")
            out[i] = r
            seen += 1
    if seen != n_total:
        raise RuntimeError(f"gathered {seen} of {n_total} utterances")
    return out
