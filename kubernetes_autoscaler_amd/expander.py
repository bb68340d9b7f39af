"""Mirror of the reference's expander filters (CA/expander/*) on top of option_kernel.

  expander.Option                    CA/expander/expander.go:46-53
  leastnodes / leastwaste / mostpods CA/expander/{leastnodes,waste,mostpods}/*.go  (Filter.BestOptions)
  chainStrategy.BestOption           CA/expander/factory/chain.go:36-45
The reduction itself runs on the device (`casim_best_option`); the random fallback of the reference
(CA/expander/random/random.go:49-56) is replaced by "lowest group index" so that results are
reproducible; `n_best` and `best_set` are returned so that a caller can apply its own random pick."""
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import _abi

LEAST_NODES = "least-nodes"
LEAST_WASTE = "least-waste"
MOST_PODS = "most-pods"
RANDOM = "random"

_KIND = {LEAST_NODES: _abi.EXPANDER_LEAST_NODES, LEAST_WASTE: _abi.EXPANDER_LEAST_WASTE, MOST_PODS: _abi.EXPANDER_MOST_PODS}


@dataclass
class Option:
    """expander.Option (NodeGroup, NodeCount, Pods)."""
    node_group: object
    node_count: int
    pods: List[object] = field(default_factory=list)
    debug: str = ""


def kinds_of(names: Sequence[str]) -> List[int]:
    """--expander=a,b,c (CA/expander/factory/expander_factory.go:55-100); `random` ends the chain."""
    out = []
    for n in names:
        if n == RANDOM:
            break
        if n not in _KIND:
            raise ValueError(f"expander {n!r} is not backed by the device reduce (supported: {sorted(_KIND)} + random)")
        out.append(_KIND[n])
    return out


class ChainStrategy:
    """chainStrategy: apply the filters in order, stop when one option is left, else fall back."""

    def __init__(self, names: Sequence[str] = (LEAST_WASTE,)):
        self.names = list(names)
        self.kinds = kinds_of(names)

    def best_option_index(self, problem, group_id_base: int = 0, valid=None):
        """Returns (best local group index or -1, number of equally good survivors, survivor mask).  `valid` ([NG] 0/1):
        options the caller already dropped (empty / partial under all-or-nothing, orchestrator.go:1057-1063) never compete."""
        if valid is None:
            best, n_best, best_set, _key = problem.best_option(self.kinds, group_id_base)
            return best, n_best, best_set
        out = problem.best_option_sims(self.kinds, per_sim=False, valid=valid, group_id_base=group_id_base)
        return int(out["best"][0]), int(out["n_best"][0]), out["best_set"]
