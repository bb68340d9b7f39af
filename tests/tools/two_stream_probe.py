"""Experiment: the contract bench's step as ONE problem on one stream vs the same simulations as TWO half problems on two
HIP streams (the latency-bound feasibility / order kernels of one half overlap the issue-bound packer of the other).
usage (GPU box): python tests/tools/two_stream_probe.py [batch=4096] [steps=200]"""
import sys, time
sys.path.insert(0, ".")
import torch
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.tables import TableSet
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed_set = bench.simulation_tables(workloads.CONFIGS["C2"], range(64), kaa.Encoder, TableSet)
kinds = [_abi.EXPANDER_LEAST_NODES]


def run(n_parts):
    streams = [torch.cuda.Stream(device=0) for _ in range(n_parts)]
    ctxs = [kaa.Context(0, stream=s.cuda_stream) for s in streams]
    per = B // n_parts
    ts = seed_set.tile((per + 63) // 64).head(per)
    pegs, groups = ts.structs()
    probs = [kaa.Problem(c, pegs, groups) for c in ctxs]
    keys = [torch.full((ts.n_sims,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device="cuda:0") for _ in range(n_parts)]

    def step():
        for p, k in zip(probs, keys):
            p.run()
            p.best_option_sims(kinds, per_sim=True, fetch=False, dev_packed_ptr=k.data_ptr(), n_sims=ts.n_sims)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    w = [int(k.cpu().numpy().sum() & 0xffffffff) for k in keys]
    for p in probs: p.close()
    for c in ctxs: c.close()
    return dt * 1e3, w


for n in (1, 2, 3, 4, 5, 6, 8):
    ms, w = run(n)
    print(f"{n} stream(s): {ms:.4f} ms per step of {B} simulations  (key checksum {w[0]})")
