"""How a batch is dealt to the GPUs of one node.  QPs are independent (the reference runs one controller
instance per QP, controller.hpp:50-51), so a rank owns a contiguous block of the batch index and nothing is exchanged on the
solve path.  Homogeneous batch: equal counts.  Ragged (mixed-robot) batch: equal predicted COST, cost of one QP ~ n^3
(factorisation + equality phase dominate: Structure.flops_estimate), so that the ranks finish together.
Pure host logic; used by bench.py and the world_size-2 tests."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def contiguous_shards(total: int, world: int) -> List[Tuple[int, int]]:
    """rank p owns [p total / world, (p + 1) total / world)"""
    return [(total * p // world, total * (p + 1) // world) for p in range(world)]


def ragged_shards(groups: Sequence[Tuple[int, int]], world: int) -> List[List[Tuple[int, int, int]]]:
    """groups: (n, count) per structure, in batch order.  Returns, per rank, the pieces (group index, begin, end) it owns:
    contiguous in the concatenated batch index, cut where the cumulative n^3 cost crosses p / world of the total.
    Every QP lands on exactly one rank; a rank's cost differs from the mean by less than one QP of the largest structure."""
    costs = [float(n) ** 3 for n, _ in groups]
    total = sum(c * cnt for c, (_, cnt) in zip(costs, groups))
    out: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
    if total <= 0.0:
        return out
    acc = 0.0
    rank = 0
    for gi, ((_, cnt), c) in enumerate(zip(groups, costs)):
        begin = 0
        while begin < cnt:
            # QPs of this group that still fit under this rank's boundary (the last rank takes whatever is left)
            if rank == world - 1:
                take = cnt - begin
            else:
                bound = total * (rank + 1) / world
                take = int((bound - acc) // c)
                if take <= 0:
                    # the boundary falls inside one QP: it goes to whichever side it overlaps more
                    if bound - acc >= 0.5 * c:
                        take = 1
                    else:
                        rank += 1
                        continue
                take = min(take, cnt - begin)
            out[rank].append((gi, begin, begin + take))
            acc += take * c
            begin += take
            if rank < world - 1 and acc >= total * (rank + 1) / world - 1e-9 * total:
                rank += 1
    return out


def shard_costs(groups: Sequence[Tuple[int, int]], shards: List[List[Tuple[int, int, int]]]) -> List[float]:
    return [sum((float(groups[g][0]) ** 3) * (e - b) for g, b, e in pieces) for pieces in shards]
