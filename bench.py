#!/usr/bin/env python3
"""bench.py — scale-up simulation throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch resident in HBM: B independent C1-shaped
scale-up simulations per GPU (BASELINE config[1]: 10k pending pods x 256 candidate nodes, CPU+mem,
200 PEGs x 50 pods, one node group each; distinct seeds) through
   order kernel -> pack kernel -> expander reduce kernel [-> one RCCL collective when N > 1].
value = predicate checks per second = N * B * (pods x node cap) / time, the metric's unit; the
closed-form packer does not enumerate them one by one — see DESIGN.md §Measurement.
One JSON line is printed by rank 0.  `roofline` describes the dominant kernel (pack), timed with
HIP events inside libcasim on the launch stream; `cpu_baseline` times the CPU oracle (a C
restatement of the reference, NOT the Go reference: no Go toolchain here) on a bounded sample of
the same simulations, single thread."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate


def build_batch(workloads, Encoder, B, seed_base, n_pegs, pods_per_peg, cap):
    """B independent C1 simulations (seeds seed_base .. seed_base+B-1) in one encoder: the PEGs go in
    through the bulk ABI entry (same (cpu, mem) pairs as workloads.config_c1(seed))."""
    import numpy as np
    enc = Encoder()
    checks = 0
    pegs_total = 0
    tmpl = workloads.config_c1(seed_offset=0, n_pegs=1, pods_per_peg=1, cap=cap).groups[0].template
    counts = np.full(n_pegs, pods_per_peg, np.int32)
    for b in range(B):
        pairs = np.array(workloads.c1_pairs(seed_base + b, n_pegs), dtype=np.int64)
        ids = enc.add_resource_pegs(pairs, counts)
        enc.add_group(tmpl, max_nodes=cap, existing_nodes=0, last_index=0, pegs=list(ids))
        checks += n_pegs * pods_per_peg * cap
        pegs_total += n_pegs
    enc.finalize()
    return enc, checks, pegs_total


def algorithmic_bytes_pack(pegs, groups, nnz, fast):
    """sum_NG (G_NG * Bp) + NG * Bn + sum_NG (8 + 8 * G_NG), with the record sizes of the
    packer that actually runs.  Bp = PEG record read per (group, PEG): request lanes + count + flags +
    order entry (+ masks); Bn = node-group record; written: placed per PEG + 40 B of counters per group.
    The register-resident packer reads gcd-scaled int32 lanes (4 B each), the generic one int64."""
    R = pegs.n_res
    lane_bytes = 4 if fast else 8
    wsum = pegs.w_taint + pegs.w_label + 2 * pegs.w_excl + 2 * pegs.w_zone
    Bp = lane_bytes * R + 4 + 4 + 4 + 8 * wsum
    Bn = lane_bytes * R + 4 * 6 + 8 * (pegs.w_excl + 2 * pegs.w_zone)
    return nnz * Bp + groups.n_groups * Bn + groups.n_groups * 40 + 4 * nnz, Bp, Bn


def cpu_baseline(workloads, seed_base, n_pegs, pods_per_peg, cap, budget_s=12.0, max_sims=4096):
    """Times orc_estimate (oracle/casim_oracle.c, single thread) on the first simulations of the
    batch until ~budget_s of CPU work has been spent.  Scenario construction is not timed."""
    from oracle_driver import OracleScenario
    sims = []
    t_build = time.time()
    for b in range(max_sims):
        w = workloads.config_c1(seed_offset=seed_base + b, n_pegs=n_pegs, pods_per_peg=pods_per_peg, cap=cap)
        s = OracleScenario()
        sims.append((s, s.node(w.groups[0].template), w))
        if len(sims) >= 64 and time.time() - t_build > 20.0:
            break
        if len(sims) >= 256:
            break
    checks = 0
    elapsed = 0.0
    n = 0
    filter_runs = 0
    rounds = 0
    while elapsed < budget_s and rounds < 1000:
        for s, tmpl, w in sims:
            t0 = time.perf_counter()
            r = s.estimate(tmpl, w.pegs, max_nodes=w.groups[0].max_nodes)
            elapsed += time.perf_counter() - t0
            checks += w.checks()
            filter_runs += r.filter_runs
            n += 1
        rounds += 1
    for s, _, _ in sims:
        s.close()
    return {"value": checks / elapsed, "unit": "checks/s", "cores": 1, "kind": "port",
            "sample": f"{n} C1 simulations ({len(sims)} distinct seeds x {rounds} rounds), {elapsed:.1f} s of orc_estimate, "
                      f"{filter_runs / max(n, 1):.0f} real Filter runs per simulation",
            "sims_per_s": n / elapsed, "host_cores_available": os.cpu_count()}


def cpu_worker(seed_base, n_sims, n_pegs, pods_per_peg, cap, budget_s):
    """One process of the multi-core CPU leg: builds its own scenarios, waits for the start line on stdin, runs
    orc_estimate for ~budget_s and prints {"n": simulations, "s": seconds}.  No torch / HIP in this process."""
    from kubernetes_autoscaler_amd import workloads
    from oracle_driver import OracleScenario
    sims = []
    for b in range(n_sims):
        w = workloads.config_c1(seed_offset=seed_base + b, n_pegs=n_pegs, pods_per_peg=pods_per_peg, cap=cap)
        s = OracleScenario()
        sims.append((s, s.node(w.groups[0].template), w))
    print("ready", flush=True)
    sys.stdin.readline()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for s, tmpl, w in sims:
            s.estimate(tmpl, w.pegs, max_nodes=w.groups[0].max_nodes)
            n += 1
    print(json.dumps({"n": n, "s": time.perf_counter() - t0}), flush=True)


def cpu_baseline_all_cores(n_pegs, pods_per_peg, cap, budget_s=4.0, max_procs=64, sims_per_proc=4):
    """The same oracle on the host's cores at once: independent Python processes (one simulation stream each, like the
    node-group-parallel CPU variant SURVEY §8d asks for), started together, aggregate simulations per second."""
    import subprocess
    procs_n = max(1, min(max_procs, (os.cpu_count() or 1)))
    procs = []
    try:
        for i in range(procs_n):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(100000 + i * sims_per_proc),
                                           str(sims_per_proc), str(n_pegs), str(pods_per_peg), str(cap), str(budget_s)],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        deadline = time.time() + 120.0
        for p in procs:
            line = p.stdout.readline()
            if line.strip() != "ready" or time.time() > deadline:
                raise RuntimeError("cpu worker did not start")
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        total = 0
        for p in procs:
            total += json.loads(p.stdout.readline())["n"]
        wall = time.perf_counter() - t0
        for p in procs:
            p.wait(timeout=30)
        sims_per_s = total / wall
        return {"value": sims_per_s * n_pegs * pods_per_peg * cap, "unit": "checks/s", "cores": procs_n, "kind": "port",
                "sample": f"{total} C1 simulations by {procs_n} processes in {wall:.1f} s (each loops over {sims_per_proc} seeds)",
                "sims_per_s": sims_per_s}
    except Exception as e:  # never take the bench line down
        for p in procs:
            try:
                p.kill()
            except Exception:
                pass
        return {"error": str(e)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        a = sys.argv[2:]
        cpu_worker(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]))
        return None
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16384, help="independent C1 simulations per GPU per step")
    ap.add_argument("--pegs", type=int, default=200)
    ap.add_argument("--pods-per-peg", type=int, default=50)
    ap.add_argument("--cap", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the TrySchedulePods / node-removal side measurements")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import kubernetes_autoscaler_amd as kaa
    from kubernetes_autoscaler_amd import _abi, workloads
    from kubernetes_autoscaler_amd.distributed import global_best_option

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: kubernetes_autoscaler_amd has no CPU path")
    # CASIM_BENCH_SELFTEST=1: exercise the N > 1 control flow on a 1-GPU box (every rank on cuda:0, gloo
    # for the collectives).  Numbers from this mode are meaningless; the driver never sets it.
    selftest = os.environ.get("CASIM_BENCH_SELFTEST") == "1"
    dev_index = 0 if selftest else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))

    B = args.batch
    t0 = time.time()
    enc, checks_per_step, pegs_total = build_batch(workloads, kaa.Encoder, B, rank * B, args.pegs, args.pods_per_peg, args.cap)
    t_encode = time.time() - t0

    stream = torch.cuda.current_stream().cuda_stream
    ctx = kaa.Context(dev_index, stream=stream)
    prob = kaa.Problem(ctx, enc.pegs, enc.groups)
    key = torch.full((10,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=f"cuda:{dev_index}")
    kinds = [_abi.EXPANDER_LEAST_NODES]
    base = rank * B

    def step():
        prob.run()
        return global_best_option(prob, kinds, base, key)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        best = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if selftest else f"cuda:{dev_index}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * checks_per_step / (dt / args.steps)
        # dominant kernel: per-kernel HIP-event timing on the launch stream (libcasim: casim_problem_time)
        total_ms, kms = prob.time(iters=max(5, min(args.steps, 20)))
        nnz, _ = prob.csr()
        info = prob.info()
        bytes_pack, Bp, Bn = algorithmic_bytes_pack(enc.pegs, enc.groups, nnz, info["fast_packer_slots_per_lane"] > 0)
        achieved = bytes_pack / (kms["pack_ms"] * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "pack_fast_kernel<%d,%d>" % (info["fast_packer_lanes"], info["fast_packer_slots_per_lane"]) if info["fast_packer_slots_per_lane"] else "pack_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": bytes_pack, "bytes_per_peg_record": Bp, "bytes_per_group_record": Bn,
                    "kernel_ms": kms["pack_ms"],
                    "note": "packer is integer-ALU/latency bound (sequential per-PEG dependency), not HBM bound; see DESIGN.md"}
        # HBM bytes per launch from the PMC passes of the same command (tools/gpu_round.sh -> tools/pmc_traffic.py):
        # counters cannot be collected from inside the timed run, so the committed figure is used when it was taken
        # on the same kernel and launch size
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pack_traffic.json")))
            # same kernel family and launch size (the profile names the full instantiation, e.g. pack_fast_kernel<2, 4, 0>)
            if tr.get("waves_per_launch") == B and roofline["kernel"].replace(" ", "").rstrip(">") in tr.get("kernel", "").replace(" ", ""):
                roofline["traffic"] = tr["traffic_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/pack_traffic.json (%s: FETCH_SIZE x2 per the gfx950 rule + WRITE_SIZE, per launch)" % tr.get("run", "?")
        except (OSError, ValueError, KeyError):
            pass
        extra = {"kernel_ms": kms, "pipeline_ms_hip_events": total_ms, "encode_s": t_encode,
                 "sims_per_s": world * B / (dt / args.steps), "best_group": best}
        # single-simulation latency (B = 1), the north-star "< 50 ms / iteration" figure
        enc1, checks1, _ = build_batch(workloads, kaa.Encoder, 1, 1 << 20, args.pegs, args.pods_per_peg, args.cap)
        with kaa.Problem(ctx, enc1.pegs, enc1.groups) as p1:
            p1.run(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                p1.run()
            torch.cuda.synchronize()
            extra["single_sim_latency_ms"] = (time.perf_counter() - t1) / 50 * 1e3
            t1 = time.perf_counter()
            for _ in range(20):
                with kaa.Problem(ctx, enc1.pegs, enc1.groups) as p2:
                    p2.run(); p2.fetch()
            extra["single_sim_upload_run_fetch_ms"] = (time.perf_counter() - t1) / 20 * 1e3
        # streaming form of the same predicates: dense per-pod x per-node check (HBM-facing kernel)
        if not args.no_dense and world == 1:   # side measurements: N = 1 only
            try:
                # bounded probe: 256 simulations' pods x (256 groups x 16 nodes) = 2.56 M x 4096 -> 1.3 GB of bits
                rep = 16
                encd, _, _ = build_batch(workloads, kaa.Encoder, 256, 1 << 21, args.pegs, args.pods_per_peg, args.cap)
                with kaa.Problem(ctx, encd.pegs, encd.groups) as pd:
                    ms, nr, nc = pd.time_dense(rep, iters=5)
                R = enc.pegs.n_res
                bp = 8 * R + 4 + 4 + 8 * 4
                bn = 8 * 2 * R + 8 + 8 * 4
                dbytes = nr * (bp + 4) + (nc // rep) * bn + nr * ((nc + 63) // 64) * 8
                extra["roofline_dense_check"] = {"bound": "hbm", "kernel": "dense_check_kernel", "rows_pods": nr, "cols_nodes": nc,
                                                 "checks_per_s": nr * nc / (ms * 1e-3), "achieved": dbytes / (ms * 1e-3) / 1e9,
                                                 "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                 "kernel_ms": ms, "algorithmic_bytes_per_launch": dbytes}
            except Exception as e:  # the probe must never take the headline number down
                extra["roofline_dense_check"] = {"error": str(e)}
        # the callers either side of the path, one mid-size case each: resident tables, HIP-event time
        if not args.no_next_rows and world == 1:
            try:
                from kubernetes_autoscaler_amd.scheduling import encode_pending_pods
                w1 = workloads.pending_scale(5000, 50000, 64, 2)
                e1, pc1 = encode_pending_pods(w1.nodes, w1.pods)
                _, _, _, ns1 = ctx.try_schedule_pods(e1.pegs, e1.groups, pc1)
                _, ms1 = ctx.try_schedule_pods(e1.pegs, e1.groups, pc1, time_iters=5)
                e1.close()
                extra["try_schedule_pods"] = {"workload": w1.name, "nodes": len(w1.nodes), "pending_pods": len(w1.pods), "scheduled": int(ns1),
                                              "kernels_ms": ms1, "pods_per_s": len(w1.pods) / (ms1 * 1e-3)}
                w2 = workloads.removal_scale(5000, pods_per_node=12, frac_candidates=0.3, seed=1)
                e2 = kaa.Encoder(explicit_self_exclusion=True)
                cls, pcl, off = {}, [], [0]
                for c in w2.candidates:
                    for p in w2.nodes[c].pods:
                        k = p.spec_key()
                        if k not in cls:
                            cls[k] = e2.add_peg(kaa.PodEquivalenceGroup(pods=[p]))
                        pcl.append(cls[k])
                    off.append(len(pcl))
                for info in w2.nodes:
                    e2.add_group(info, pegs=[])
                e2.finalize()
                r2 = ctx.simulate_node_removals(e2.pegs, e2.groups, w2.candidates, off, pcl)
                _, ms2 = ctx.simulate_node_removals(e2.pegs, e2.groups, w2.candidates, off, pcl, time_iters=5)
                e2.close()
                extra["node_removals"] = {"workload": w2.name, "nodes": len(w2.nodes), "candidates": len(w2.candidates),
                                          "removable": int((r2.removable == 1).sum()), "kernels_ms": ms2,
                                          "candidates_per_s": len(w2.candidates) / (ms2 * 1e-3)}
            except Exception as e:  # must never take the headline number down
                extra["next_rows_error"] = str(e)
        try:
            extra["copy_bandwidth_gbps"] = ctx.copy_bandwidth_gbps(1 << 30, 10)
        except Exception as e:
            extra["copy_bandwidth_gbps"] = str(e)
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # the CPU legs are timed at N = 1 only (other ranks would idle in the barrier)
            cpu = cpu_baseline(workloads, 0, args.pegs, args.pods_per_peg, args.cap)
            extra["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.pegs, args.pods_per_peg, args.cap)
        out = {"metric": "scale-up simulation predicate checks/s (pods x nodes)", "value": value, "unit": "checks/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
               "config": {"workload": f"C1 x {B} simulations per GPU per step (10k pending pods x 256 candidate nodes, "
                                      f"{args.pegs} PEGs x {args.pods_per_peg} pods, CPU+mem, 1 node group each)",
                          "batch_per_gpu": B, "pegs_per_sim": args.pegs, "pods_per_sim": args.pegs * args.pods_per_peg,
                          "node_cap": args.cap, "expander": "least-nodes",
                          "reduce": "rccl all_reduce(min) on packed int64 key" if world > 1 else "device kernel only"},
               "roofline": roofline, "cpu_baseline": cpu}
        out.update(extra)
        print(json.dumps(out))
    prob.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
