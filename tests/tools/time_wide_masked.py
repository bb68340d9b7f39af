"""Register packer with exclusion words at 16 slots per lane (pack_fast_kernel<R,16,2>: ~250 VGPRs, 1-2 waves per SIMD) against the generic
LDS packer on the same batch: C4-shaped simulations (pod anti-affinity) with node limits of 600 per group, tiled.  usage (GPU box):
python tests/tools/time_wide_masked.py"""
import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.tables import TableSet
import bench

ctx = kaa.Context(0)
for cap in (50, 200, 600):
    w = workloads.config_c4(0, cap=cap, pods_per_peg=25 if cap <= 200 else 120)
    enc = bench.encode_workload(w, kaa.Encoder)
    ts = TableSet.from_encoder(enc).as_one_simulation().tile(256)   # 256 simulations x 20 groups (mask widths differ between seeds)
    enc.close()
    pegs, groups = ts.structs()
    out = {"node_limit": cap, "groups": ts.n_groups}
    for generic in (False, True):
        with kaa.Problem(ctx, pegs, groups, force_generic_packer=generic) as p:
            p.run(); res = p.fetch()
            tot, k = p.time(iters=10)
            out["generic" if generic else "register"] = {"pack_ms": round(k["pack_ms"], 4), "info": p.info(), "nodes": int(np.asarray(res.node_count).sum())}
    print(json.dumps(out))
