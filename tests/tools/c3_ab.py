"""The batched C3 step (512 simulations, 8 seeds tiled; bench.py headline_rows.c3_resident) with the orderer ranking once per allocatable pair
(default for this shape) and with the per-group sort (CASIM_RANK_ONCE=0): ms per resident step, results compared.  Also C2 (the headline's shape:
the automatic rule keeps the per-group sort; forced on for the record).  Usage on the GPU box: python tests/tools/c3_ab.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.tables import TableSet
kinds = [_abi.EXPANDER_LEAST_NODES]
rows = []
for cfg, n_sims, n_seeds, modes in (("C3", 512, 8, (None, "0")), ("C3", 512, 8, (None, "share1")), ("C2", 4096, 64, (None, "1"))):
    ts = bench.simulation_tables(workloads.CONFIGS[cfg], range(n_seeds), kaa.Encoder, TableSet).tile((n_sims + n_seeds - 1) // n_seeds).head(n_sims)
    row = {"config": cfg, "sims": n_sims}
    keep = {}
    for mode in modes:
        os.environ.pop("CASIM_RANK_ONCE", None); os.environ.pop("CASIM_RANK_SHARE", None)
        if mode == "share1":
            os.environ["CASIM_RANK_SHARE"] = "1"
        elif mode is not None:
            os.environ["CASIM_RANK_ONCE"] = mode
        stream = torch.cuda.Stream(device=0)
        b = kaa.StreamedBatch(0, ts, n_streams=4, stream=stream.cuda_stream)
        for _ in range(5):
            b.run(); b.best_option_sims(kinds, fetch=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            b.run(); b.best_option_sims(kinds, fetch=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100
        res = b.fetch()
        name = "default" if mode is None else ("CASIM_RANK_SHARE=1" if mode == "share1" else f"CASIM_RANK_ONCE={mode}")
        row[name] = {"ms_per_step": dt * 1e3, "ranked_orderer": bool(b.prob.info()["ranked_orderer"])}
        keep[name] = res
        b.close()
    a, c = list(keep.values())
    row["same_results"] = bool(all(np.array_equal(getattr(a, f), getattr(c, f)) for f in ("node_count", "pods_scheduled", "last_index_out", "order", "placed", "offsets")))
    rows.append(row)
os.environ.pop("CASIM_RANK_ONCE", None); os.environ.pop("CASIM_RANK_SHARE", None)
print(json.dumps(rows))
