"""world_size-2 gloo tests of the multi-GPU path: node-group sharding and the expander reduce
(all_reduce(MIN) on the packed key / all_gather of key blocks).  CPU only."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kubernetes_autoscaler_amd import distributed as D


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert D.shard_groups(list(range(64)), 3, 8) == list(range(24, 32))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, blocks, n_kinds, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kb = torch.tensor(blocks[rank], dtype=torch.int64)
    out[rank] = (D.reduce_best_min(kb), D.reduce_best_gather(kb, n_kinds))
    dist.barrier()
    dist.destroy_process_group()


def _block(metrics, gid):
    none = D.NONE_KEY
    if gid is None:
        return [none] * 10
    b = [none] * 10
    b[0] = (metrics[0] << 20) | gid
    for i, m in enumerate(metrics):
        b[1 + i] = m
    b[9] = gid
    return b


@pytest.mark.parametrize("blocks,n_kinds,want_min,want_gather", [
    ([_block([5], 3), _block([4], 70)], 1, 70, 70),                 # smaller node count wins
    ([_block([5], 3), _block([5], 70)], 1, 3, 3),                   # tie -> lowest group id
    ([_block([5], 3), _block(None, None)], 1, 3, 3),                # one rank has no option
    ([_block(None, None), _block(None, None)], 1, -1, -1),          # nobody has one
    ([_block([7, 2], 1), _block([7, 1], 9)], 2, 1, 9),              # second filter decides (gather only)
])
def test_two_rank_reduce(blocks, n_kinds, want_min, want_gather):
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, blocks, n_kinds, out), nprocs=2, join=True)
    for r in range(2):
        got_min, got_gather = out[r]
        assert got_gather == want_gather
        if n_kinds == 1:
            assert got_min == want_min
