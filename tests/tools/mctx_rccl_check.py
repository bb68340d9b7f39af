"""casim_mctx with RCCL on the hardware, in a process whose HIP runtime and RCCL come from ONE place: torch is imported first,
so libcasim binds to the libamdhip64 / librccl torch ships (on this pool's boxes the system RCCL of ROCm 7.2 fails its own
topology discovery inside the container: "alt_rsmi: Could not read node").  Prints one JSON line."""
import json
import os
import sys

import torch  # noqa: F401  (first: see above)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _abi  # noqa: E402
from harness import encode_batch, run_gpu_tables  # noqa: E402
from test_gpu_round2 import _oracle_of, _scenario  # noqa: E402
from harness import assert_matches_oracle  # noqa: E402

devices = list(range(kaa.device_count()))
ctx = kaa.Context(0)
checked = 0
with kaa.MultiContext(devices, use_rccl=True) as m:
    for seed in range(12):
        scs = [_scenario(3100 + 10 * seed + k, groups=7) for k in range(1 + seed % 4)]
        enc, ts, bases = encode_batch(scs)
        pegs, groups = ts.structs()
        for kinds in ([_abi.EXPANDER_LEAST_NODES], [_abi.EXPANDER_MOST_PODS]):
            got, exp = m.estimate_batch(pegs, groups, kinds=kinds)
            assert_matches_oracle(got, _oracle_of(scs, bases), f"mctx rccl seed {seed}")
            _, one = run_gpu_tables(ts, ctx, kinds=kinds)
            assert list(exp["best"]) == list(one["best"]) and list(exp["packed"]) == list(one["packed"])
            assert m.info()["last_reduce_by_rccl"]
            checked += 1
        enc.close()
    info = m.info()
ctx.close()
print(json.dumps({"devices": devices, "batches": checked, "rccl": info["rccl"], "last_reduce_by_rccl": info["last_reduce_by_rccl"]}))
