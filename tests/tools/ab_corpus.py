"""Runs a corpus of scale-up batches through whichever libcasim the environment selects (CASIM_LIB_PATH) and prints a hash
of every result array: the two builds of tests/test_gpu_ab_structurizer.py must print the same line."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _ffi, workloads  # noqa: E402
from harness import GroupSpec, Scenario, encode, encode_batch, run_gpu, run_gpu_tables  # noqa: E402


def main():
    ctx = kaa.Context(0)
    h = hashlib.sha256()
    n = 0

    def feed(res):
        nonlocal n
        for f in ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum",
                  "order", "placed"):
            h.update(np.ascontiguousarray(getattr(res, f)).tobytes())
        n += 1

    def scen(w, fastpath=False, device_csr=False):
        return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None if device_csr else g.pegs) for g in w.groups],
                        existing=w.existing, lanes=w.lanes, fastpath=fastpath, device_csr=device_csr)
    for base, cnt, kw, fast, dcsr in ((0, 60, dict(rich=False), False, False), (1000, 200, {}, False, False), (2000, 60, {}, True, False),
                                      (3000, 60, {}, False, True), (5000, 60, dict(max_groups=3, max_pegs=48), False, False)):
        for seed in range(cnt):
            sc = scen(workloads.fuzz(base + seed, **kw), fast, dcsr)
            enc = encode(sc)
            for generic in (False, True, 2):     # register store on int32 lanes, LDS store, register store on int64 lanes
                res, _ = run_gpu(enc, ctx, fastpath=fast, generic=generic)
                feed(res)
            enc.close()
    for name in ("C0", "C1", "C2", "C3", "C4"):
        w = workloads.CONFIGS[name]()
        enc = encode(scen(w, device_csr=True))
        res, _ = run_gpu(enc, ctx)
        feed(res)
        enc.close()
    for seed in range(10):     # batches of simulations, tiled
        scs = [scen(workloads.fuzz(7000 + 10 * seed + k, max_groups=5, max_pegs=14), device_csr=True) for k in range(4)]
        for sc in scs:
            sc.existing = []
        enc, ts, _ = encode_batch(scs)
        res, exp = run_gpu_tables(ts.tile(40), ctx, kinds=[0])
        feed(res)
        h.update(exp["packed"].tobytes())
        enc.close()
    ctx.close()
    print(json.dumps({"lib": _ffi.LIB_PATH, "batches": n, "sha256": h.hexdigest()}))


if __name__ == "__main__":
    main()
