import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from harness import GroupSpec, Scenario, encode
ctx = kaa.Context(0)
for name, mk in (("C2", workloads.config_c2), ("C4", workloads.config_c4)):
    w = workloads.batch_of(mk, 16)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes, device_csr=False)
    t0 = time.time(); enc = encode(sc); te = time.time() - t0
    with kaa.Problem(ctx, enc.pegs, enc.groups) as p:
        p.run(); res = p.fetch()
        tot, k = p.time(iters=10)
        print(json.dumps({"config": name, "groups": len(w.groups), "pegs": len(w.pegs), "encode_s": te, "pipeline_ms": tot, **k, "info": p.info(),
                          "groups_per_s": len(w.groups) / (tot * 1e-3)}))
