#!/usr/bin/env python3
"""debug: schedulable PEG lists per group, feas_stream_kernel vs feas_sim_kernel vs the oracle, on small batches (GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from harness import GroupSpec, Scenario, encode_batch, run_gpu_tables, run_oracle


def _scenario(seed, groups=5):
    w = workloads.fuzz(seed, max_groups=groups, max_pegs=14)
    return Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], device_csr=True)


ctx = kaa.Context(0)
shown = 0
for seed in range(1, 12):
    scs = [_scenario(1000 * seed + k) for k in range(1 + seed % 6)]
    enc, ts, bases = encode_batch(scs)
    os.environ.pop("CASIM_NO_FEAS_STREAM", None)
    new, _ = run_gpu_tables(ts, ctx)
    os.environ["CASIM_NO_FEAS_STREAM"] = "1"
    old, _ = run_gpu_tables(ts, ctx)
    os.environ.pop("CASIM_NO_FEAS_STREAM", None)
    gi = 0
    for si, (sc, (pb, gb)) in enumerate(zip(scs, bases)):
        want = run_oracle(sc)
        for j, (est, ids) in enumerate(want):
            a, b = int(new.offsets[gi]), int(new.offsets[gi + 1])
            got = sorted(int(x) - pb for x in new.order[a:b])
            a2, b2 = int(old.offsets[gi]), int(old.offsets[gi + 1])
            got_old = sorted(int(x) - pb for x in old.order[a2:b2])
            if got != sorted(ids) and shown < 12:
                shown += 1
                print(f"seed {seed}: sim {si} (of {len(scs)}; {len(sc.pegs)} PEGs, {len(sc.groups)} groups, peg base {pb}, group base {gb}) group {j} (global {gi}): "
                      f"stream {got}  old {got_old}  oracle {sorted(ids)}")
            gi += 1
    enc.close()
print("done, mismatching groups shown:", shown)
ctx.close()
