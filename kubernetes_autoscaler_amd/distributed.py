"""Multi-GPU: node groups are independent Estimate() calls, so they shard across the GPUs of one node
with NO data-path collective; the only exchange step is the expander's "pick the best option"
(CA/core/scaleup/orchestrator/orchestrator.go:1079, CA/expander/factory/chain.go:36-45), which becomes
one RCCL collective over xGMI on a 10-int64 key block per rank.

One process per GPU, launched with torch.distributed.run; backend "nccl" (= RCCL on ROCm) on the
GPU box, "gloo" in the CPU tests.  The payload is <= 80 B per rank: latency-bound, so a single-shot
all-gather (or one all-reduce(min) for the integer metrics) is used, never a ring of many steps."""
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist

NONE_KEY = 0x7FFFFFFFFFFFFFFF
KEY_WORDS = 10   # layout written by option_kernel: [0] packed, [1..8] per-filter metrics, [9] global group id


def shard_bounds(n_groups: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition of the node groups (C3: 64 groups / 8 GPUs = 8 each)."""
    base, extra = divmod(n_groups, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_groups(groups: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_bounds(len(groups), rank, world)
    return list(groups[lo:hi])


def _for_backend(t: torch.Tensor) -> torch.Tensor:
    """gloo (CPU tests / the bench's self-test mode) reduces host tensors; RCCL reduces in place on the device."""
    if dist.is_initialized() and dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def reduce_best_min(key_block: torch.Tensor) -> int:
    """Single-filter chains on an integer metric (least-nodes / most-pods): ONE all-reduce(min) on the
    packed key (metric << 20 | global group id) is exact and already breaks ties towards the lowest
    group id.  Returns the global group id or -1."""
    k = _for_backend(key_block[0:1].clone())
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(k, op=dist.ReduceOp.MIN)
    v = int(k.item())
    return -1 if v == NONE_KEY else v & 0xFFFFF


def reduce_best_gather(key_block: torch.Tensor, n_kinds: int) -> int:
    """General chains (any mix incl. least-waste's float64 metric): all-gather the per-rank winners'
    key blocks and take the lexicographic minimum of (m_1, .., m_k, group id) — exactly what the
    filter chain computes over the union of all options."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world > 1:
        kb = _for_backend(key_block.contiguous())
        out = torch.empty(world * KEY_WORDS, dtype=torch.int64, device=kb.device)
        dist.all_gather_into_tensor(out, kb)
        blocks = out.view(world, KEY_WORDS)
    else:
        blocks = key_block.view(1, KEY_WORDS)
    rows = blocks.cpu().tolist()
    best = None
    for r in rows:
        if r[9] == NONE_KEY:
            continue
        t = tuple(r[1:1 + n_kinds]) + (r[9],)
        if best is None or t < best:
            best = t
    return -1 if best is None else int(best[-1])


def global_best_option(problem, kinds: Sequence[int], group_id_base: int, key_block: torch.Tensor, mode: str = "auto") -> int:
    """Device-side local reduce (option_kernel) + one RCCL collective.  `key_block` is a 10-int64
    tensor on the problem's device; the kernel writes into it directly (no host round trip)."""
    from . import _abi
    problem.best_option_device(kinds, group_id_base, key_block.data_ptr())
    integer_chain = len(kinds) == 1 and kinds[0] in (_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_MOST_PODS)
    if mode == "min" or (mode == "auto" and integer_chain):
        return reduce_best_min(key_block)
    return reduce_best_gather(key_block, len(kinds))
