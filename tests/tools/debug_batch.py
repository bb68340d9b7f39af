import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kubernetes_autoscaler_amd as kaa
from test_gpu_round2 import _scenario
from harness import encode_batch, run_emu_tables, run_gpu_tables
ctx = kaa.Context(0)
for seed in range(0, 40):
    scs = [_scenario(1000 * seed + k) for k in range(1 + seed % 6)]
    enc, ts, bases = encode_batch(scs)
    for rep in range(3):
        g, _ = run_gpu_tables(ts, ctx)
        e, _ = run_emu_tables(ts)
        bad = [f for f in ("offsets", "order", "placed", "node_count", "pods_scheduled") if list(getattr(g, f)) != list(getattr(e, f))]
        if bad:
            print("seed", seed, "rep", rep, "differs in", bad, "groups", ts.n_groups, "dims", ts.dims)
            print(" gpu offsets", list(g.offsets)); print(" emu offsets", list(e.offsets))
            for i in range(ts.n_groups):
                a, b = g.group(i), e.group(i)
                if list(a[0]) != list(b[0]) or list(a[1]) != list(b[1]):
                    print("  group", i, "lo/hi", ts.peg_lo[i], ts.peg_hi[i], "gpu", list(a[0]), list(a[1]), "emu", list(b[0]), list(b[1]))
                    sc = ts.pegs["req"]; al = ts.groups["alloc"][i]
                    print("   scores", [(int(x), float(sc[x][0]) / al[0] + float(sc[x][1]) / al[1]) for x in b[0]])
            break
    enc.close()
print("done")
