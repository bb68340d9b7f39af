"""enter -> return of a streamed batch (casim_estimate_batch_query with casim_options.n_streams: upload, kernels, expander reduce and fetch of every part
by its own worker) against ONE unstreamed problem over the same batch and against the oracle — in a process of its own so that the switches read once
per process can be set from outside: CASIM_UPLOAD_FIFO=1 (the parts take the link in turn), CASIM_JOINED_FETCH=1 (round 5's fetch order),
CASIM_POOL_THREADS=0 (no host pool: the parts run one after the other on the calling thread).  Tables pageable and page-locked, requests int64 and
narrowed by the caller, every list and winners only.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _abi  # noqa: E402
from kubernetes_autoscaler_amd.engine import BatchCall  # noqa: E402
from harness import assert_matches_oracle, encode_batch, run_gpu_tables  # noqa: E402
from test_gpu_round2 import _oracle_of, _scenario  # noqa: E402

os.environ.setdefault("CASIM_TEST_UPLOAD_CHUNK", "4096")     # small tables in many pieces
os.environ.setdefault("CASIM_TEST_DIRECT_MIN", "2048")       # page-locked columns of a few KB go out from where they lie
ctx = kaa.Context(0)
scs = [_scenario(1300 + k) for k in range(13)]
enc, ts, bases = encode_batch(scs)
kinds = [_abi.EXPANDER_LEAST_NODES]
whole, wexp = run_gpu_tables(ts, ctx, kinds=kinds)
assert_matches_oracle(whole, _oracle_of(scs, bases), "whole batch")
nnz = int(whole.offsets[-1])
cells = []
for name, tables in (("pageable", ts), ("pinned", ts.pinned())):
    for narrow in (False, True):
        for k in (2, 4):
            for winners in (False, True):
                call = BatchCall(ctx, *tables.structs(narrow_requests=narrow), kinds=kinds, n_streams=k, winners_only=winners)
                for rep in range(3):
                    res, exp = call.call()
                    for f in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum"):
                        assert list(getattr(res, f)) == list(getattr(whole, f)), (name, narrow, k, winners, f)
                    assert list(exp["best"]) == list(wexp["best"]) and list(exp["packed"]) == list(wexp["packed"])
                    if winners:
                        at = 0
                        for s, b in enumerate(exp["best"]):
                            if b < 0:
                                continue
                            lo, hi = int(whole.offsets[b]), int(whole.offsets[b + 1])
                            assert list(res.order[at:at + hi - lo]) == list(whole.order[lo:hi]) and list(res.placed[at:at + hi - lo]) == list(whole.placed[lo:hi]), (name, k, s)
                            at += hi - lo
                    else:
                        assert list(res.offsets) == list(whole.offsets) and list(res.order[:nnz]) == list(whole.order[:nnz]) and list(res.placed[:nnz]) == list(whole.placed[:nnz])
                cells.append(f"{name}/{'req32' if narrow else 'int64'}/K{k}/{'winners' if winners else 'lists'}")
enc.close()
ctx.close()
print(json.dumps({"cells": len(cells), "simulations": ts.n_sims, "groups": ts.n_groups,
                  "switches": {k: os.environ.get(k) for k in ("CASIM_UPLOAD_FIFO", "CASIM_JOINED_FETCH", "CASIM_POOL_THREADS")}}))
