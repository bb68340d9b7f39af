#!/usr/bin/env python3
"""bench.py — scale-up simulation throughput on MI355X (BASELINE.json metric: pods x nodes predicate checks / s).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-launches itself under torch.distributed.run)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Headline workload = BASELINE config[2] ("C2": 10k pending pods x 1k candidate nodes across 20 node groups with
taints / tolerations + nodeSelector), B independent simulations per GPU resident in HBM (S distinct seeds, tiled).
A "step" = one pass of the whole hot path over that batch:
   feasibility (SchedulablePodGroups) -> CSR compaction -> order (DecreasingPodOrderer) -> pack (Estimate)
   -> expander reduce per simulation [-> ONE RCCL all-reduce(min) over the per-simulation keys when N > 1].
N > 1 (weak scaling): B * N simulations; the node groups of every simulation are block-partitioned over the N ranks
(PEG table replicated, no data-path collective), the only exchange is the expander's min over packed keys.
value = N-rank total of sum_NG(P_NG x Ncap_NG) per step / time, P_NG = pods of the PEGs schedulable on NG.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel of the step (HIP events on the launch stream, inside libcasim), HBM bytes and — because
                the packer is bound by instruction issue, not bandwidth — the VALU/SALU issue fraction from the PMC profile;
  cpu_baseline  the CPU oracle (C restatement of the reference, `kind: port`; no Go toolchain in this image) on the SAME
                C2 simulations, one thread, bounded sample; the all-cores figure next to it;
  configs       one row per BASELINE config C0..C4: ONE simulation, enter -> return wall time of casim_estimate_batch
                (H2D + kernels + D2H), its phases, the host encode time, the oracle on the same input, bit-exact flag;
  c3_sharded    BASELINE config[3]: 64 node groups block-partitioned over the ranks, one all-reduce(min) per simulation."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# One hardware queue per internal stream of a streamed batch: the HOST's setting, made here — before anything
# initialises the HIP runtime — because libcasim no longer edits the process environment by itself.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

SHIM_MIN_DEVICE_WORK = 150000   # gpubinpacking.DefaultRouting.MinDeviceWork (integration/go/gpubinpacking/estimator.go)
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate
SIMDS, CLOCK_HZ = 1024, 2.4e9


# ------------------------------------------------------------------------------------------------------------------
# workloads -> tables (product encoder only; nothing here touches the oracle)
# ------------------------------------------------------------------------------------------------------------------
def encode_workload(w, Encoder):
    """One simulation through the host encoder the way a shim would: every PEG, every node-group template, schedulable
    subsets left to the device (feasibility kernel)."""
    enc = Encoder(lanes=w.lanes)
    for pg in w.pegs:
        enc.add_peg(pg)
    for info in w.existing:
        for p in info.pods:
            enc.add_existing_pod(p, info.node.labels)
    for g in w.groups:
        enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=len(w.existing), last_index=g.last_index,
                      pegs=list(g.pegs) if g.pegs is not None else None)
    enc.finalize()
    return enc


def simulation_tables(make, seeds, Encoder, TableSet):
    """S distinct simulations of one config as one batch (each group sees only its own simulation's PEGs)."""
    sets = []
    for s in seeds:
        enc = encode_workload(make(seed_offset=s), Encoder)
        sets.append(TableSet.from_encoder(enc).as_one_simulation())
        enc.close()
    return TableSet.concat(sets)


def checks_of(ts, res):
    """sum_NG P_NG x Ncap_NG over the groups of a table set, P_NG from the schedulable subsets the device derived."""
    import numpy as np
    cnt = ts.pegs["count"][:, 0].astype(np.int64)
    cap = np.maximum(ts.groups["max_nodes"][:, 0].astype(np.int64), 0)
    pods = np.add.reduceat(np.concatenate([cnt[res.order], [0]]), res.offsets[:-1].astype(np.int64))
    pods[res.offsets[1:] == res.offsets[:-1]] = 0
    return int((pods * cap).sum()), int(res.offsets[-1])


def algorithmic_bytes_pack(dims, n_groups, nnz, fast):
    """sum_NG (G_NG * Bp) + NG * Bn + sum_NG (8 + 8 * G_NG) with the record sizes of the packer that runs.
    Bp = PEG record read per (group, PEG); Bn = node-group record; written: placed per PEG + 40 B of counters per group.
    Register packer: ONE record of 32 B (<= 2 lanes) or 64 B (<= 4 lanes) per PEG — count, flags + fresh-node capacity, the
    gcd-scaled int32 requests and their reciprocals (csrc/casim_types.h) — plus the order entry when mask tables are indexed.
    Generic packer: int64 request lanes + count + flags + order entry + mask words."""
    R = dims["n_res"]
    wsum = dims["w_taint"] + dims["w_label"] + 2 * dims["w_excl"] + 2 * dims["w_zone"]
    masks = dims["w_excl"] + dims["w_zone"]
    if fast:
        Bp = (32 if R <= 2 else 64) + (4 + 8 * 2 * masks if masks else 0)
        Bn = 4 * R + 4 * 6 + 8 * (dims["w_excl"] + 2 * dims["w_zone"])
    else:
        Bp = 8 * R + 4 + 4 + 4 + 8 * wsum
        Bn = 8 * R + 4 * 6 + 8 * (dims["w_excl"] + 2 * dims["w_zone"])
    return nnz * Bp + n_groups * Bn + n_groups * 40 + 4 * nnz, Bp, Bn


# ------------------------------------------------------------------------------------------------------------------
# CPU legs (the oracle as the BASELINE, never as the product): only these functions import anything under oracle/
# ------------------------------------------------------------------------------------------------------------------
def oracle_simulation(workloads, make, seed):
    """Builds one simulation inside the oracle; returns run(collect) -> (per-group results or None, seconds of oracle work).
    One run = ONE native call (orc_scale_up_simulation: SchedulablePodGroups + Estimate per node group), so the CPU leg is
    not charged for Python / ctypes overhead."""
    from oracle_driver import OracleScenario
    w = make(seed_offset=seed) if seed is not None else make()
    s = OracleScenario(lanes=w.lanes)
    for info in w.existing:
        s.add_existing(info)
    tmpls = [s.node(g.template) for g in w.groups]
    if all(g.pegs is None for g in w.groups):
        native = s.prepare_simulation(tmpls, w.pegs, [g.max_nodes for g in w.groups], [g.last_index for g in w.groups])

        def run(collect=True):
            t0 = time.perf_counter()
            out, runs = native(collect)
            return out, time.perf_counter() - t0, runs
    else:   # explicit PEG lists per group
        def run(collect=True):
            t0 = time.perf_counter()
            out = [(s.estimate(tmpl, [w.pegs[i] for i in g.pegs], max_nodes=g.max_nodes, last_index=g.last_index, node_pods_cap=0), list(g.pegs))
                   for g, tmpl in zip(w.groups, tmpls)]
            return out, time.perf_counter() - t0, sum(e.filter_runs for e, _ in out)
    return w, s, run


def cpu_baseline(workloads, make, seeds, checks_per_sim, budget_s=12.0):
    sims = [oracle_simulation(workloads, make, sd) for sd in seeds]
    n, elapsed, filter_runs = 0, 0.0, 0
    while elapsed < budget_s:
        for _, _, run in sims:
            _, dt, runs = run(False)
            elapsed += dt
            filter_runs += runs
            n += 1
            if elapsed >= budget_s:
                break
    for _, s, _ in sims:
        s.close()
    return {"value": n * checks_per_sim / elapsed, "unit": "checks/s", "cores": 1, "kind": "port",
            "label": "C restatement, not the Go reference (no Go toolchain on the box): a lower bound on the Go reference's time",
            "sample": f"{n} C2 simulations ({len(sims)} distinct seeds, the first of the GPU batch), {elapsed:.1f} s of oracle work "
                      f"(orc_scale_up_simulation: CheckPredicates per PEG x group + Estimate per group, one native call per simulation), "
                      f"{filter_runs / max(n, 1):.0f} real Filter runs per simulation",
            "sims_per_s": n / elapsed, "ms_per_sim": elapsed / n * 1e3, "host_cores_available": os.cpu_count()}


def verify_headline(workloads, make, n_seeds, tables, res):
    """the timed batch itself against the oracle.  Every DISTINCT simulation of the batch (`n_seeds` seeds, tiled)
    is run through the oracle once (orc_scale_up_simulation, results collected) and EVERY group of the rank's batch — all tiles — is
    compared with it bit for bit: schedulable PEG list, PEG order, pods placed per PEG, node count, pods, nodes added, limiter
    grants, lastIndex, request sums.  `tables` is the rank's TableSet (sim_offsets: the groups of simulation s, which is seed s % n_seeds
    of the tiling; global_id: the group's index inside its simulation — a sharded rank holds some groups of every simulation; peg_lo:
    the first PEG of its simulation), `res` the results fetched after the timed loop."""
    import numpy as np
    t0 = time.perf_counter()
    want = []
    for sd in range(n_seeds):
        _, s, run = oracle_simulation(workloads, make, sd)
        out, _, _ = run(True)
        want.append(out)
        s.close()
    so = np.asarray(tables.sim_offsets, np.int64)
    sim = np.searchsorted(so, np.arange(tables.n_groups, dtype=np.int64), side="right") - 1
    local = tables.global_id.astype(np.int64) if tables.global_id is not None else np.arange(tables.n_groups, dtype=np.int64) - so[sim]
    bad, first = 0, None
    for j in range(tables.n_groups):
        est, ids = want[int(sim[j]) % n_seeds][int(local[j])]
        a, b = int(res.offsets[j]), int(res.offsets[j + 1])
        base = int(tables.peg_lo[j])
        ok = (int(res.status[j]) == 0 and b - a == len(est.order) and
              np.array_equal(res.order[a:b] - base, np.asarray(ids, np.int64)[est.order]) and np.array_equal(res.placed[a:b], est.placed) and
              (int(res.node_count[j]), int(res.pods_scheduled[j]), int(res.nodes_added[j]), int(res.limiter_nodes[j]), int(res.last_index_out[j]),
               int(res.req_cpu_sum[j]), int(res.req_mem_sum[j])) ==
              (est.node_count, est.pods_scheduled, est.nodes_added, est.limiter_nodes, est.last_index_out, est.req_cpu_sum, est.req_mem_sum))
        if not ok:
            bad += 1
            first = first if first is not None else j
    return {"headline_bit_exact": bad == 0, "groups_compared": int(tables.n_groups), "simulations_compared": int(len(np.unique(sim))),
            "distinct_simulations_in_the_oracle": n_seeds, "groups_differing": bad, "first_differing_group": first,
            "what": "results of the LAST timed step, every group of the batch vs orc_scale_up_simulation of its seed (order, placed, node count, "
                    "pods, nodes added, limiter grants, lastIndex, request sums)", "verify_s": time.perf_counter() - t0}


def cpu_worker(config, seed, budget_s):
    """One process of the multi-core CPU leg: one simulation of `config`, waits for the start line, loops ~budget_s."""
    from kubernetes_autoscaler_amd import workloads
    _, s, run = oracle_simulation(workloads, workloads.CONFIGS[config], None if config == "C0" else seed)
    print("ready", flush=True)
    sys.stdin.readline()
    spins = _alu_spin(0.25)   # (all workers at once: do the cores the processes were given really run in parallel?)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        run(False)
        n += 1
    print(json.dumps({"n": n, "s": time.perf_counter() - t0, "spins": spins}), flush=True)
    s.close()


def _cgroup_cpus():
    """CPUs the cgroup grants (cpu.max quota / period), None when unlimited or unreadable"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def _alu_spin(seconds):
    """iterations of a register-only loop in `seconds`: a worker's share of a core, whatever the memory system does"""
    n, x, t_end = 0, 1, time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(2000):
            x = (x * 1103515245 + 12345) & 0x7fffffff
        n += 1
    return n


def cpu_baseline_all_cores(config, checks_per_sim, budget_s=5.0, max_procs=128):
    """The same oracle on every host core at once: independent processes, one simulation stream each (the node-group /
    simulation-parallel CPU variant of SURVEY 8d), aggregate rate."""
    procs_n = max(1, min(max_procs, (os.cpu_count() or 1)))
    granted = _cgroup_cpus()
    if granted:   # (r10f: the GPU box reports 256 CPUs and grants 16 of them, cpu.max = "1600000 100000": more processes than that only take turns)
        procs_n = max(1, min(procs_n, int(granted + 0.999)))
    procs = []
    try:
        for i in range(procs_n):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", config, str(100000 + i), str(budget_s)],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
        deadline = time.time() + 180.0
        for p in procs:
            line = p.stdout.readline()
            if line.strip() != "ready" or time.time() > deadline:
                raise RuntimeError("cpu worker did not start")
        alone = _alu_spin(0.25)   # one process with the machine to itself
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        answers = [json.loads(p.stdout.readline()) for p in procs]
        total = sum(a["n"] for a in answers)
        wall = time.perf_counter() - t0
        for p in procs:
            p.wait(timeout=30)
        # why N processes are not N times one: the register-only spin every worker runs first tells whether the cores are
        # really there (cgroup quota, affinity mask, SMT siblings); what is missing beyond that is the memory system under pointer-chasing
        quota = None
        try:
            quota = open("/sys/fs/cgroup/cpu.max").read().strip()
        except OSError:
            pass
        alu_scaling = sum(a.get("spins", 0) for a in answers) / max(alone, 1)
        return {"value": total / wall * checks_per_sim, "unit": "checks/s", "cores": procs_n, "kind": "port", "label": "C restatement, not the Go reference",
                "sample": f"{total} {config} simulations by {procs_n} processes in {wall:.1f} s", "sims_per_s": total / wall,
                "scaling_evidence": {"register_only_spin_aggregate_over_one_process": alu_scaling, "cgroup_cpu_max": quota,
                                     "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "os_cpu_count": os.cpu_count(),
                                     "cgroup_cpus_granted": granted,
                                     "reading": "processes = min(128, os.cpu_count(), the cgroup's CPUs); spin scaling ~ processes: the cores are there and what the "
                                                "oracle lacks beyond it is its memory traffic; spin scaling well below: quota / affinity / SMT"}}
    except Exception as e:  # never take the bench line down
        for p in procs:
            try:
                p.kill()
            except Exception:
                pass
        return {"error": str(e)}


def config_rows(kaa, ctx, workloads, kinds, iters=20):
    """One simulation of every BASELINE config through the whole boundary, enter -> return."""
    import numpy as np
    from kubernetes_autoscaler_amd.engine import estimate_batch_timed
    from kubernetes_autoscaler_amd.tables import TableSet
    from harness import assert_matches_oracle
    from kubernetes_autoscaler_amd.engine import finish_results
    rows = []
    loop_tables = {}   # filled by row C2 (native --shim replay), reported by row C2-per-call

    def c2_one_group(seed_offset=0):
        """What ONE Estimate() call carries when the shim does not batch: the first node group of C2 and every PEG."""
        w = workloads.config_c2(seed_offset)
        w.groups = w.groups[:1]
        w.name = "C2, one node group per call"
        return w
    # R1 / R2 = the reference's own benchmark regimes for this path: BenchmarkRunOnceScaleUp — 10 000
    # singleton PEGs -> one 200-node group (core/bench/benchmark_runonce_test.go:395-418,493-503) — and
    # BenchmarkBinpackingEstimate — 2595 nodes / 51 000 pods (estimator/binpacking_estimator_test.go:256-303)
    for name in ("C0", "C1", "C2", "C3", "C4", "C2-per-call", "R1", "R2"):
        make = workloads.CONFIGS.get(name, c2_one_group)
        row = {"config": name}
        try:
            t0 = time.perf_counter()
            w = make()
            enc = encode_workload(w, kaa.Encoder)
            row["encode_ms_python_mirror"] = (time.perf_counter() - t0) * 1e3
            ts = TableSet.from_encoder(enc).as_one_simulation() if all(g.pegs is None for g in w.groups) else TableSet.from_encoder(enc)
            if ts.sim_offsets is None:
                ts.sim_offsets = np.array([0, ts.n_groups], np.int32)
            pegs, groups = ts.structs()
            row.update({"pods": w.n_pods, "pegs": len(w.pegs), "node_groups": len(w.groups),
                        "node_cap_total": int(sum(max(g.max_nodes, 0) for g in w.groups))})
            estimate_batch_timed(ctx, pegs, groups, kinds)          # first call: LDS opt-ins, code objects
            ph_acc, walls = None, []
            for _ in range(iters):
                arrs, ph, exp = estimate_batch_timed(ctx, pegs, groups, kinds)
                ph_acc = ph if ph_acc is None else {k: ph_acc[k] + v for k, v in ph.items()}
            for _ in range(iters):   # the plain call a shim makes (no drain between phases) + the expander reduce
                t1 = time.perf_counter()
                with kaa.Problem(ctx, pegs, groups) as p:
                    p.run()
                    res = p.fetch()
                    best = p.best_option_sims(kinds, per_sim=True, n_sims=1)
                walls.append((time.perf_counter() - t1) * 1e3)
            row["phases_ms"] = {k: v / iters for k, v in ph_acc.items()}
            row["wall_ms"] = float(np.median(walls))
            row["wall_ms_min"] = float(np.min(walls))
            row["checks"], row["schedulable_peg_group_pairs"] = checks_of(ts, res)
            row["checks_per_s_single_sim"] = row["checks"] / (row["wall_ms"] * 1e-3)
            row["best_group"] = int(best["best"][0])
            # the oracle on the same input (CPU, one thread) and the bit-exact comparison
            if name == "C2-per-call":
                row["note"] = ("per-call mode: one casim_estimate_batch per Estimate(), 20 of these make one C2 loop iteration; "
                               "the batch row above is the same work in ONE call")
            if name in ("R1", "R2"):
                # one group, G dependent PEG steps in ONE wave: what a step costs when nothing runs beside it
                row["pegs_in_the_one_group"] = len(w.pegs)
                row["pack_us_per_peg_step"] = row["phases_ms"]["pack_ms"] * 1e3 / max(len(w.pegs), 1)
                row["expected_by_the_reference"] = {"R1": "target size 200 (verifyTargetSize)", "R2": "2595 nodes, 51000 pods"}[name]
                row["got"] = [int(res.node_count[0]), int(res.pods_scheduled[0])]
            _, s, run = oracle_simulation(workloads, make, None if name in ("C0", "R1", "R2") else 0)
            want, osec, _ = run()
            _, osec2, _ = run(False)
            s.close()
            row["oracle_ms"] = min(osec, osec2) * 1e3
            try:
                assert_matches_oracle(res, want, name)
                row["bit_exact"] = True
            except AssertionError as e:
                row["bit_exact"] = False
                row["mismatch"] = str(e)[:200]
            row["speedup_vs_oracle_wall"] = row["oracle_ms"] / row["wall_ms"]
            # which path the Go shim takes for this call (gpubinpacking.Routing: per-call Estimates with pods x node bound below MinDeviceWork
            # go to the reference estimator; batches from the prefetch fill always go to the device) and the ratio ON THAT PATH
            if len(w.groups) == 1:
                bound = w.groups[0].max_nodes if w.groups[0].max_nodes > 0 else w.n_pods
                work = w.n_pods * min(bound, w.n_pods)
                row["shim_route"] = {"work_pods_x_node_bound": int(work), "min_device_work": SHIM_MIN_DEVICE_WORK,
                                     "path": "reference estimator" if work < SHIM_MIN_DEVICE_WORK else "device"}
                row["speedup_on_the_path_the_shim_takes"] = 1.0 if work < SHIM_MIN_DEVICE_WORK else row["speedup_vs_oracle_wall"]
            else:
                row["shim_route"] = {"path": "device (one batch per loop: the prefetch fill)"}
                row["speedup_on_the_path_the_shim_takes"] = row["speedup_vs_oracle_wall"]
            enc.close()
            # the same simulation through tools/casim_native: plain C++ over the C ABI, no Python between the calls —
            # encode (all casim_enc_* calls + finalize), tables -> HBM, kernels, results -> host
            import native_trace
            tpath = os.path.join("/tmp", f"casim_{name}.trace")
            native_trace.trace_estimate(w, tpath, kinds=kinds, iters=iters).close()
            nrc, nat = native_trace.run_native(tpath, shim=(name == "C2"))
            if name == "C2" and isinstance(nat.get("shim"), dict):
                # per-call mode of the shim on the LOOP's tables: the loop is encoded once (this row's encode_ms, every PEG
                # and every group), an Estimate() that misses the prefetch cache is casim_enc_group_rows + casim_estimate_batch — no encoder work
                sh = nat["shim"]
                loop_tables["per_call_on_loop_tables_ms"] = sh.get("per_call_on_loop_tables_ms")
                loop_tables["loop_encode_ms"] = nat.get("encode_ms")
                loop_tables["groups"] = sh.get("groups")
                loop_tables["failed_checks"] = sh.get("failed_checks")
            keep = ("enc_calls", "encode_calls_ms", "finalize_ms", "encode_ms", "upload_ms", "feasibility_csr_ms", "order_ms", "pack_ms",
                    "expander_ms", "fetch_ms", "timed_wall_ms", "wall_ms", "best_group", "engine_error")
            row["native"] = {k: nat[k] for k in keep if k in nat}
            row["native"]["exit_code"] = nrc
            # the row's encode figures are the shim's sequence since ABI 11 (the PEGs through casim_enc_add_pods); the same workload pod by pod beside it
            try:
                pbp = native_trace.encode_only(w, tpath + ".pod_by_pod", bulk=False)
                row["native"]["encode_pod_by_pod"] = dict(pbp, same_tables=pbp.get("tables_fnv") == nat.get("tables_fnv"))
            except Exception as e:
                row["native"]["encode_pod_by_pod"] = {"error": f"{type(e).__name__}: {e}"}
            if "wall_ms" in nat and "encode_ms" in nat:
                row["native"]["encode_plus_call_ms"] = nat["encode_ms"] + nat["wall_ms"]
                row["native"]["same_winner_as_python_path"] = nat.get("best_group") == row["best_group"]
            if name == "C2-per-call" and loop_tables.get("per_call_on_loop_tables_ms"):
                # what ONE Estimate() of a C2 loop costs in per-call mode now: the call on the loop's tables + its share of the loop's one encode
                lt = loop_tables
                per = lt["per_call_on_loop_tables_ms"] + (lt["loop_encode_ms"] or 0.0) / max(lt["groups"] or 1, 1)
                row["native"]["on_loop_tables"] = dict(lt, encode_share_plus_call_ms=per, speedup_vs_oracle=row["oracle_ms"] / per,
                                                       what="casim_enc_group_rows + casim_estimate_batch on the tables the loop encoded once (row C2's encode_ms / 20 groups): "
                                                            "how the Go shim serves an Estimate() that misses the prefetch cache (integration/go/gpubinpacking/prefetch.go estimateOnLoopTables); "
                                                            "encode_plus_call_ms above is a call that encodes its 400 PEGs itself (no loop tables: the analyser path)")
        except Exception as e:  # a side table must never take the headline down
            row["error"] = f"{type(e).__name__}: {e}"
        rows.append(row)
    return rows


def c3_sharded(kaa, ctx, workloads, kinds, rank, world, dist, torch, dev_index, allreduce, collective, iters=200):
    """BASELINE config[3]: one C3 simulation (50k pods x 4k nodes, 64 node groups); node groups block-partitioned over
    the ranks, PEG table replicated, ONE all-reduce(min) on the packed key.  Strong scaling: the work is fixed."""
    import numpy as np
    from kubernetes_autoscaler_amd.tables import TableSet
    enc = encode_workload(workloads.config_c3(), kaa.Encoder)
    full = TableSet.from_encoder(enc).as_one_simulation()
    mine = full.shard(rank, world, rotate=False)
    pegs, groups = mine.structs()
    key = torch.full((1,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=f"cuda:{dev_index}")
    with kaa.Problem(ctx, pegs, groups) as p:
        def it():
            p.run()
            p.best_option_sims(kinds, per_sim=True, fetch=False, dev_packed_ptr=key.data_ptr(), n_sims=1)
            if collective:
                allreduce(key, dist.ReduceOp.MIN)
        for _ in range(10):
            it()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        winner = int(key.item())
        # the collective alone (same tensor, nothing else enqueued)
        red_ms = None
        if collective:
            torch.cuda.synchronize(); dist.barrier()
            t1 = time.perf_counter()
            for _ in range(iters):
                allreduce(key, dist.ReduceOp.MIN)
            torch.cuda.synchronize()
            red_ms = (time.perf_counter() - t1) / iters * 1e3
            tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev_index}")
            allreduce(tt, dist.ReduceOp.MAX)
            dt = float(tt.item())
        res = p.fetch()
        chk, _ = checks_of(mine, res)
    enc.close()
    if collective:
        tc = torch.tensor([chk], dtype=torch.int64, device=f"cuda:{dev_index}")
        allreduce(tc, dist.ReduceOp.SUM)
        chk = int(tc.item())
    # what does NOT shrink when the groups are spread over more GPUs: the expander reduce + the collective, and the chain of the
    # longest group (one wave walks its PEGs one after the other) — measured here as the reduce share and the kernel share
    with kaa.Problem(ctx, pegs, groups) as p2:
        p2.run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(iters):
            p2.best_option_sims(kinds, per_sim=True, fetch=False, dev_packed_ptr=key.data_ptr(), n_sims=1)
            if collective:
                allreduce(key, dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - t1) / iters * 1e3
        _, kms = p2.time(iters=10)
    serial = {"reduce_ms_per_simulation": reduce_ms, "kernels_ms_per_simulation": sum(kms.values()), "kernel_ms": kms,
              "reduce_share_of_iteration": reduce_ms / (dt / iters * 1e3),
              "note": "one simulation = 64 waves of dependent PEG steps: already concurrent on ONE GPU, so N GPUs shorten only the "
                      "feasibility / order launches; expected strong-scaling ceiling ~1.0-1.2x (DESIGN.md section 6)"}
    return {"workload": "C3: 50k pods x 4k nodes, 64 node groups, resident tables", "scaling": "strong", "ranks": world, "serial_fraction": serial,
            "groups_on_rank0": mine.n_groups, "ms_per_simulation": dt / iters * 1e3, "checks": chk,
            "checks_per_s": chk / (dt / iters), "winner_group": -1 if winner == 0x7FFFFFFFFFFFFFFF else winner & 0xFFFFF,
            "winner_nodes": None if winner == 0x7FFFFFFFFFFFFFFF else winner >> 20,
            "collective": (f"all_reduce(min), 1 x int64, {world} rank(s)" if collective else "none (1 rank)"), "all_reduce_ms": red_ms}


# ------------------------------------------------------------------------------------------------------------------
def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start the ranks the way the driver would."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------------------------------------------------------------------
# the contract line: ONE compact JSON line, the LAST line of stdout; everything else is a side table
# ------------------------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096
SIDE_FILE = "bench_side.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _sig(x, digits=6):
    """floats to `digits` significant figures (the line has to stay under COMPACT_LIMIT bytes), containers walked"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(out, side_file=SIDE_FILE):
    """The bench contract's line from the full result dict: the contract fields, `roofline` and `cpu_baseline` reduced to their
    numbers, the wall / int64 regimes of the same step.  Everything else (configs, f1 / f4 rows, encoder rows, C3, ...) goes to the
    side table (`side_file`, also on stderr)."""
    cfg = out.get("config") or {}
    roof = out.get("roofline") or {}
    rows = out.get("headline_rows") or {}
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = _pick(cfg, ("workload", "batch_per_gpu", "streams", "distinct_seeds", "checks_per_simulation", "expander", "partition", "reduce"))
    r = _pick(roof, ("bound", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "achieved", "peak", "unit", "frac", "launch_groups"))
    if isinstance(roof.get("device_to_itself"), dict):
        r["alone"] = _pick(roof["device_to_itself"], ("kernel_ms", "achieved", "frac"))
    if isinstance(roof.get("issue_roofline"), dict):
        r["issue_roofline"] = _pick(roof["issue_roofline"], ("bound", "frac", "valu_insts_per_launch", "salu_insts_per_launch"))
    line["roofline"] = r
    if isinstance(out.get("roofline_feasibility"), dict):
        line["roofline_feasibility"] = _pick(out["roofline_feasibility"], ("bound", "kernel", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "achieved",
                                                                          "peak", "unit", "frac", "workload", "kernel_ms_in_loop", "bit_exact"))
    if isinstance(out.get("roofline_feasibility_c3"), dict):
        line["roofline_feasibility_c3"] = _pick(out["roofline_feasibility_c3"], ("kernel", "kernel_ms", "algorithmic_bytes_per_launch", "traffic", "achieved", "frac", "bit_exact"))
    cpu = out.get("cpu_baseline")
    line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "label", "sample", "sims_per_s")) if isinstance(cpu, dict) else None
    allc = out.get("cpu_baseline_all_cores")
    if isinstance(allc, dict) and "value" in allc:
        line["cpu_baseline_all_cores"] = _pick(allc, ("value", "cores", "sims_per_s"))
    for k in ("sims_per_s", "timed_region_s", "value_wall", "ms_per_step_wall", "ms_per_step_wall_median", "table_bytes_in", "value_wall_req32", "ms_per_step_wall_req32",
              "ms_per_step_wall_req32_median", "table_bytes_in_req32", "wall_req32_bit_equal", "value_wall_shared_pegs", "ms_per_step_wall_shared_pegs", "ms_per_step_wall_shared_pegs_median",
              "table_bytes_in_shared_pegs", "wall_shared_pegs_bit_equal", "value_wall_every_list", "ms_per_step_wall_every_list", "headline_bit_exact"):
        if k in out:
            line[k] = out[k]
    i64 = rows.get("int64") if isinstance(rows, dict) else None
    if isinstance(i64, dict) and "checks_per_s" in i64:
        line["value_int64"], line["ms_per_step_int64"] = i64["checks_per_s"], i64.get("ms_per_step")
    for name in ("c1_resident", "c3_resident", "c4_resident"):
        r = rows.get(name) if isinstance(rows, dict) else None
        if isinstance(r, dict) and "checks_per_s" in r:
            line.setdefault("other_configs", {})[name] = _pick(r, ("ms_per_step", "checks_per_s", "sims_per_s", "bit_exact"))
    mg = out.get("multi_gpu")
    if isinstance(mg, dict) and mg.get("all_reduce_ms") is not None:
        line["multi_gpu"] = _pick(mg, ("rccl_world_size", "collective_backend", "all_reduce_ms", "all_reduce_share_of_step"))
    line["side_tables"] = side_file
    line = _sig(line)
    # the sample / workload sentences are the only unbounded strings: cut them rather than lose the line
    over = len(json.dumps(line)) - (COMPACT_LIMIT - 64)
    if over > 0 and isinstance(line.get("cpu_baseline"), dict) and "sample" in line["cpu_baseline"]:
        smp = line["cpu_baseline"]["sample"]
        line["cpu_baseline"]["sample"] = smp[:max(40, len(smp) - over)]
    over = len(json.dumps(line)) - (COMPACT_LIMIT - 64)
    if over > 0 and "workload" in line["config"]:
        line["config"]["workload"] = line["config"]["workload"][:max(60, len(line["config"]["workload"]) - over)]
    return line


def emit(out, side_path=None):
    """Side tables first (file + stderr), then the contract line as the LAST line of stdout.  C stdio is flushed in between:
    RCCL prints its version banner through printf, which a pipe buffers until exit — i.e. BEHIND a line Python printed (how
    BENCH_r04's line got lost)."""
    side_path = side_path or os.path.join(ROOT, SIDE_FILE)
    try:
        with open(side_path, "w") as f:
            json.dump(out, f)
            f.write("\n")
    except OSError:
        pass
    sys.stderr.write("bench side tables: " + json.dumps(out) + "\n")
    sys.stderr.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    line = json.dumps(compact_line(out, os.path.basename(side_path)))
    assert len(line) < COMPACT_LIMIT, len(line)
    sys.stdout.flush()
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2], int(sys.argv[3]), float(sys.argv[4]))
        return None
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="independent C2 simulations per GPU per step")
    ap.add_argument("--seeds", type=int, default=64, help="distinct simulations the batch is tiled from")
    ap.add_argument("--streams", type=int, default=4,
                    help="the batch of a GPU runs as this many sub-batches on HIP streams of their own (one casim context each): the "
                         "latency-bound feasibility / order kernels of one sub-batch overlap the issue-bound packer of another")
    ap.add_argument("--config", default="C2", choices=["C1", "C2", "C3", "C4"], help="headline config (C2 = BASELINE config[2])")
    ap.add_argument("--expander", default="least-nodes", choices=["least-nodes", "least-waste", "most-pods"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the timed batch (headline_bit_exact)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config C0..C4 wall-time table")
    ap.add_argument("--no-dense", action="store_true", help="(accepted for old scripts; the dense probe kernel was retired in round 2)")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the TrySchedulePods / node-removal side measurements")
    ap.add_argument("--no-c3", action="store_true")
    ap.add_argument("--no-feasibility-row", action="store_true", help="skip the batched feasibility launches of roofline_feasibility")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        relaunch_under_torchrun(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist
    import kubernetes_autoscaler_amd as kaa
    from kubernetes_autoscaler_amd import _abi, workloads
    from kubernetes_autoscaler_amd.tables import TableSet

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: kubernetes_autoscaler_amd has no CPU path")
    # CASIM_BENCH_ONE_GPU=1: every rank on cuda:0 (exercises the N > 1 path, RCCL included, on a 1-GPU box; the numbers
    # of such a run say nothing about scaling and the JSON line says so).  The driver never sets it.
    one_gpu = os.environ.get("CASIM_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    backend = None
    # CASIM_BENCH_FORCE_DIST=1 (under a launcher): build the process group and run the per-step collective even with ONE
    # rank — how the RCCL calls are exercised on a 1-GPU box (RCCL refuses two ranks on one device).
    collective = world > 1 or (os.environ.get("CASIM_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CASIM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    kinds = [{"least-nodes": _abi.EXPANDER_LEAST_NODES, "least-waste": _abi.EXPANDER_LEAST_WASTE, "most-pods": _abi.EXPANDER_MOST_PODS}[args.expander]]
    make = workloads.CONFIGS[args.config]
    B, S = args.batch, max(1, min(args.seeds, args.batch))
    t0 = time.time()
    seed_set = simulation_tables(make, range(S), kaa.Encoder, TableSet)       # S distinct simulations, one table set
    t_encode = time.time() - t0
    total_sims = B * world
    full = seed_set.tile((total_sims + S - 1) // S).head(total_sims)
    mine = full.shard(rank, world) if world > 1 else full
    n_sims = mine.n_sims
    K = max(1, min(args.streams, n_sims))

    # ONE casim context on an explicit torch stream (torch's default stream has handle 0, which casim_ctx_create takes as "create
    # your own": the kernels would then run unordered with torch's work).  The sub-batching lives inside libcasim
    # (casim_options.n_streams): the context's internal streams fork from this stream at every run and join into it when the
    # expander writes the keys to a device pointer — the per-step RCCL all-reduce runs on it.
    side_stream = torch.cuda.Stream(device=dev_index)
    torch.cuda.set_stream(side_stream)
    assert side_stream.cuda_stream != 0
    t0 = time.time()
    batch = kaa.StreamedBatch(dev_index, mine, n_streams=K, stream=side_stream.cuda_stream)
    ctx, prob = batch.ctx, batch.prob
    K = batch.parts
    t_upload = time.time() - t0
    keys = torch.full((n_sims,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device=f"cuda:{dev_index}")

    def allreduce(t, op):
        """RCCL reduces device tensors in place; the gloo self-test backend goes through the host."""
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu(); dist.all_reduce(h, op=op); t.copy_(h)

    # N > 1: the step's collective runs on a stream of its own, and ONLY that stream waits for the keys (casim_option_query.join_stream): the
    # context's stream stays idle, so the next run does not fork from it and the sub-batches keep running ahead of each other exactly as at
    # N = 1 — with the default join every internal stream waited for all the others' previous step, the four chains ran in lock-step (all in
    # the feasibility phase, then all in the packer) and the step took 1.45 ms instead of 1.03 (1-rank RCCL group, profiles/r05j_*).
    # The steps are independent batches; keys are double-buffered and the host waits for the all-reduce of two steps ago before a buffer is
    # written again.
    comm_stream = torch.cuda.Stream(device=dev_index) if collective else None
    keys2 = [keys, torch.full_like(keys, 0x7FFFFFFFFFFFFFFF)] if collective else [keys]
    reduced = [None, None]
    step_no = [0]

    def make_step(b):
        if collective:
            def step():
                i = step_no[0] & 1
                step_no[0] += 1
                if reduced[i] is not None:
                    reduced[i].synchronize()                                                 # host: this buffer's previous all-reduce (two steps ago) is done
                b.run()
                b.best_option_sims(kinds, dev_packed_ptr=keys2[i].data_ptr(), fetch=False, join_stream=comm_stream.cuda_stream)
                with torch.cuda.stream(comm_stream):
                    allreduce(keys2[i], dist.ReduceOp.MIN)
                    reduced[i] = torch.cuda.Event(); reduced[i].record(comm_stream)
        else:
            def step():
                b.run()
                b.best_option_sims(kinds, fetch=False)   # keys stay in the problem's own buffers: nothing on the context's stream, steps overlap
        return step

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t_start

    dt = timed(make_step(batch), args.steps, args.warmup)
    res_all = prob.fetch()   # (the results of the LAST timed step: verify_headline compares exactly these with the oracle)
    my_checks, my_nnz = checks_of(batch.tables, res_all)
    part0_groups = int(batch.tables.sim_offsets[(n_sims * 1) // K]) if K > 1 else mine.n_groups
    part0_nnz = int(res_all.offsets[part0_groups])
    final = prob.best_option_sims(kinds, per_sim=True, n_sims=n_sims)    # (host fetch: the winners of the last step)
    checks_per_step = my_checks
    if collective:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev_index}")
        allreduce(tt, dist.ReduceOp.MAX)
        dt = float(tt.item())
        tc = torch.tensor([my_checks], dtype=torch.int64, device=f"cuda:{dev_index}")
        allreduce(tc, dist.ReduceOp.SUM)
        checks_per_step = int(tc.item())
    # the collective of a step alone (same tensor, nothing else enqueued): what the step pays that does not shrink with N
    all_reduce_ms = None
    if collective:
        torch.cuda.synchronize(); dist.barrier()
        t1 = time.perf_counter()
        for _ in range(50):
            allreduce(keys, dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        all_reduce_ms = (time.perf_counter() - t1) / 50 * 1e3

    out = None
    side = {}
    if not args.no_c3:   # every rank takes part (collective inside)
        try:
            side["c3_sharded"] = c3_sharded(kaa, ctx, workloads, kinds, rank, world, dist, torch, dev_index, allreduce, collective)
        except Exception as e:
            side["c3_sharded"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = checks_per_step / (dt / args.steps)
        winners = keys2[(step_no[0] - 1) & 1].cpu().numpy() if collective else final["packed"]   # (the last step's reduced keys)
        have = winners != 0x7FFFFFFFFFFFFFFF
        # per-kernel HIP-event timing on the launch stream (libcasim), twice: in the regime of the timed region — the other
        # internal streams keep running their parts while part 0 is timed (casim_problem_run_marked), which is what a kernel
        # trace of this command sees — and with the device to itself (casim_problem_time: what the serialising PMC passes see)
        n_time = max(5, min(args.steps, 20))
        if K > 1:
            for _ in range(48):
                prob.run_marked()
                prob.best_option_sims(kinds, per_sim=True, fetch=False, n_sims=n_sims)
            total_ms, kms, _n = prob.marked_ms()
            torch.cuda.synchronize()
            alone_ms, kms_alone = prob.time(iters=n_time)
        else:
            total_ms, kms = prob.time(iters=n_time)
            alone_ms, kms_alone = total_ms, kms
        info = prob.info()
        fast = info["fast_packer_slots_per_lane"] > 0
        # (one launch = one sub-batch: bytes, durations and the PMC figures below are all per launch of sub-batch 0)
        bytes_pack, Bp, Bn = algorithmic_bytes_pack(mine.dims, part0_groups, part0_nnz, fast)
        achieved = bytes_pack / (kms["pack_ms"] * 1e-3) / 1e9
        build_info = ctx.pack_build_info()   # which of the library's two builds of the register packer ran (self-check verdict)
        kname = ("pack_fast_kernel<%d,%d,%d,%d>" % (info["fast_packer_lanes"], info["fast_packer_slots_per_lane"],
                                                     2 if (mine.dims["w_excl"] or mine.dims["w_zone"]) else 0,
                                                     1 if build_info["build"] == "plain" else 0)) if fast else "pack_kernel"
        roofline = {"bound": "hbm", "effective_bound": "scalar instruction issue (see issue_roofline): the >= 40 % HBM target of BASELINE.json is not reachable by this "
                                                       "algorithm — counter traffic is 1.07x the algorithmic bytes, there is nothing left to fetch faster",
                    "kernel": kname, "packer_build_self_check": build_info, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": bytes_pack, "bytes_per_peg_record": Bp, "bytes_per_group_record": Bn,
                    "kernel_ms": kms["pack_ms"], "share_of_step": kms["pack_ms"] / max(total_ms, 1e-9),
                    "launch": f"sub-batch 0 of {K}: {(n_sims * 1) // K if K > 1 else n_sims} simulations, {part0_groups} node groups (one wave each); "
                              f"kernel_ms = its average duration over 48 more steps of the timed loop (events recorded, not waited for): the kernels of the {K} streams time-share the device",
                    "device_to_itself": {"kernel_ms": kms_alone["pack_ms"], "achieved": bytes_pack / (kms_alone["pack_ms"] * 1e-3) / 1e9,
                                         "frac": bytes_pack / (kms_alone["pack_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                         "kernel_ms_all": kms_alone, "pipeline_ms": alone_ms},
                    "note": "the packer is bound by scalar / vector instruction ISSUE (sequential per-PEG dependency), not by HBM: "
                            "see issue_roofline (counters and duration of the launch with the device to itself); DESIGN.md section 4"}
        # PMC figures of the same command (separate rocprofv3 --pmc passes, tools/gpu_round.sh -> tools/pmc_*.py): counters
        # cannot be collected from inside the timed run; the committed figures are used when they were taken on the same
        # kernel instantiation and launch size
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "pack_traffic.json")))
            if tr.get("waves_per_launch") == part0_groups and kname.replace(" ", "").rstrip(">") in tr.get("kernel", "").replace(" ", ""):
                # FETCH_SIZE corrected by the probe of the kernel's own access path when the PMC pass carried one (the register
                # packer reads its records with scalar loads: stream_probe_scalar_kernel), else by the x2 rule of wide vector streams
                roofline["traffic"] = tr.get("traffic_bytes_per_launch_by_scalar_probe", tr["traffic_bytes_per_launch"])
                roofline["traffic_source"] = "profiles/pack_traffic.json (%s%s)" % (tr.get("run", "?"), ", FETCH_SIZE calibrated on 32-byte scalar loads"
                                                                                   if "traffic_bytes_per_launch_by_scalar_probe" in tr else "")
                if "valu_insts_per_launch" in tr:
                    # issue roofline: one VALU / SALU wave-instruction holds its SIMD's issue port ~4 cycles (measured:
                    # SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycle); 1024 SIMDs x 2.4 GHz
                    cyc = tr.get("cycles_per_valu", 4.0)
                    # a SIMD issues at most one VALU and one scalar instruction per ~4 cycles (different waves): the port with
                    # more instructions bounds the kernel
                    valu, salu = tr["valu_insts_per_launch"], tr.get("salu_insts_per_launch") or 0
                    port = "salu_issue" if salu > valu else "valu_issue"
                    t_issue = max(valu, salu) * cyc / (SIMDS * CLOCK_HZ)
                    roofline["issue_roofline"] = {"bound": port, "valu_insts_per_launch": valu, "salu_insts_per_launch": salu,
                                                  "cycles_per_inst": cyc, "issue_time_ms": t_issue * 1e3,
                                                  "frac": t_issue / (kms_alone["pack_ms"] * 1e-3), "clock_ghz_assumed": CLOCK_HZ / 1e9,
                                                  "source": tr.get("run", "?")}
                    if tr.get("effective_clock_ghz"):   # the chip clocks to its power budget: the same fraction at the measured clock
                        ec = tr["effective_clock_ghz"]
                        roofline["issue_roofline"]["effective_clock_ghz"] = ec
                        roofline["issue_roofline"]["frac_at_effective_clock"] = max(valu, salu) * cyc / (SIMDS * ec * 1e9) / (kms_alone["pack_ms"] * 1e-3)
        except (OSError, ValueError, KeyError):
            pass
        extra = {"kernel_ms": kms, "pipeline_ms_hip_events": total_ms,
                 "kernel_ms_note": f"HIP events around each kernel class of sub-batch 0 ({(n_sims * 1) // K if K > 1 else n_sims} of the {n_sims} simulations) in the timed loop's "
                                   f"own regime ({K} streams time-sharing the device: per stream the durations add up to the step); "
                                   f"roofline.device_to_itself has the same launches alone", "encode_s_python_mirror": t_encode, "upload_s": t_upload,
                 "sims_per_step": total_sims, "sims_per_s": total_sims / (dt / args.steps),
                 "timed_region_s": dt, "winners": {"simulations_with_an_option": int(have.sum()),
                                                   "mean_nodes_of_winner": float((winners[have] >> 20).mean()) if have.any() else None}}
        checks_per_sim = checks_per_step / total_sims
        # ---- the headline three ways, same batch, same context -------------------------------------------
        rows = {"resident": {"what": "tables resident in HBM, results stay on the device (the `value` of this line)", "dtype": "int32" if fast else "int64",
                             "ms_per_step": ms_per_step, "checks_per_s": value, "sims_per_s": total_sims / (dt / args.steps), "steps": args.steps,
                             "one_casim_ctx": True, "streams_inside_libcasim": K}}
        if world == 1:
            rows["enter_return"] = _try(lambda: enter_return_row(kaa, ctx, batch.tables, kinds, K, checks_per_step, max(3, min(args.steps, 10)), res_all, final))
            rows["enter_return_every_list"] = _try(lambda: enter_return_row(kaa, ctx, batch.tables, kinds, K, checks_per_step, max(3, min(args.steps, 10)), res_all, final,
                                                                            winners_only=False))
            # the caller's tables in page-locked memory (casim_host_alloc): no staging memcpy on the host, the columns travel from where they lie
            def _pinned_row():
                r = enter_return_row(kaa, ctx, batch.tables.pinned(), kinds, K, checks_per_step, max(3, min(args.steps, 10)), res_all, final)
                r["what"] = "enter_return with the caller's tables in page-locked host memory (casim_host_alloc): columns of >= 1 MiB are copied to the device where they lie"
                return r
            rows["enter_return_pinned_tables"] = _try(_pinned_row)
            # the Go-compat form (every list comes back) with tables AND result lists page-locked: the prefetch fill of a shim that allocates
            # its flat arrays through casim_host_alloc
            def _pinned_all_row():
                r = enter_return_row(kaa, ctx, batch.tables.pinned(), kinds, K, checks_per_step, max(3, min(args.steps, 10)), res_all, final, winners_only=False,
                                     pinned_results=True)
                r["what"] = "enter_return_every_list with the caller's tables and its order / placed arrays in page-locked host memory (casim_host_alloc)"
                return r
            rows["enter_return_every_list_pinned"] = _try(_pinned_all_row)
            # PEG tables shipped once per DISTINCT simulation, a node-group table per simulation (TableSet.tile_groups: what a sweep of limiter /
            # template variants over the same pending pods hands over) — the same 4096 simulations, the same answers
            # the requests as 32-bit multiples of a per-lane unit (casim_pegs.req32 / req_unit, ABI 10: milli-cpu, MiB — what a Go shim holds anyway)
            def _req32_row():
                r = enter_return_row(kaa, ctx, batch.tables, kinds, K, checks_per_step, max(3, min(args.steps, 10)), res_all, final, narrow=True)
                r["what"] = ("enter_return with the requests handed over as casim_pegs.req32 + req_unit (req = NULL): 8 bytes per PEG less on the link, no gcd "
                             "pass (a device round trip in the middle of every part's upload), the int64 table rebuilt on the device")
                return r
            rows["enter_return_req32"] = _try(_req32_row)
            def _shared_row():
                shared = seed_set.tile_groups((total_sims + S - 1) // S).head(total_sims)
                r = enter_return_row(kaa, ctx, shared, kinds, K, checks_per_step, max(3, min(args.steps, 10)), res_all, final, order_mod=seed_set.n_pegs, narrow=True)
                r["what"] = (f"enter_return_req32 with the PEG tables of the {S} distinct simulations shipped ONCE and a node-group table per simulation (casim_groups.peg_lo / "
                             "peg_hi of the tiles point into the same PEG rows): what a sweep of limiter / template variants over the same pending pods hands over")
                r["peg_rows_shipped"] = shared.n_pegs
                return r
            rows["enter_return_shared_pegs"] = _try(_shared_row)
            rows["int64"] = _try(lambda: int64_row(kaa, dev_index, batch.tables, kinds, K, checks_per_step, max(5, min(args.steps, 50)), res_all, final, torch, packer=2))
            rows["int64_lds_store"] = _try(lambda: int64_row(kaa, dev_index, batch.tables, kinds, K, checks_per_step, max(5, min(args.steps, 50)), res_all, final, torch, packer=1))
        extra["headline_rows"] = rows
        # the §8(d) wall-clock form of the same metric at top level (VERDICT r3 next #1b): `value` is the resident regime the bench
        # contract asks for (inputs in HBM when the timed region starts); value_wall is casim_estimate_batch_query enter -> return
        er = rows.get("enter_return") or {}
        extra["value_wall"] = er.get("checks_per_s")
        extra["ms_per_step_wall"] = er.get("ms_per_step")
        extra["sims_per_s_wall"] = er.get("sims_per_s")
        # (the winners-only regime is not what a shim's prefetch fill needs — the every-list form next to it, top level)
        extra["table_bytes_in"] = er.get("table_bytes_in")
        extra["ms_per_step_wall_median"] = er.get("ms_per_step_median")   # (ten calls: one host hiccup of 6-10 ms moves the mean by 10-20 %)
        rq = rows.get("enter_return_req32") or {}
        extra["value_wall_req32"] = rq.get("checks_per_s")
        extra["ms_per_step_wall_req32"] = rq.get("ms_per_step")
        extra["ms_per_step_wall_req32_median"] = rq.get("ms_per_step_median")
        extra["table_bytes_in_req32"] = rq.get("table_bytes_in")
        extra["wall_req32_bit_equal"] = rq.get("bit_equal_to_resident")
        sh = rows.get("enter_return_shared_pegs") or {}
        extra["value_wall_shared_pegs"] = sh.get("checks_per_s")
        extra["ms_per_step_wall_shared_pegs"] = sh.get("ms_per_step")
        extra["ms_per_step_wall_shared_pegs_median"] = sh.get("ms_per_step_median")
        extra["table_bytes_in_shared_pegs"] = sh.get("table_bytes_in")
        extra["wall_shared_pegs_bit_equal"] = sh.get("bit_equal_to_resident")
        el = rows.get("enter_return_every_list") or {}
        extra["value_wall_every_list"] = el.get("checks_per_s")
        extra["ms_per_step_wall_every_list"] = el.get("ms_per_step")
        extra["value_regimes"] = {"value": "resident: tables in HBM, results stay on the device (bench contract)",
                                  "value_wall": "SURVEY 8(d): host-side enter -> return of casim_estimate_batch_query every step, H2D + kernels + expander + D2H (PCIe inclusive); "
                                                "results = every group's scalars + the winners' PEG lists (SURVEY 8e; headline_rows.enter_return_every_list ships all lists)"}
        if not args.no_verify:
            extra["headline_check"] = _try(lambda: verify_headline(workloads, make, S, batch.tables, res_all))
            extra["headline_bit_exact"] = bool((extra["headline_check"] or {}).get("headline_bit_exact", False))
        extra["multi_gpu"] = {"rccl_world_size": world if (collective and backend == "nccl") else (0 if not collective else None),
                              "collective_backend": backend if collective else None, "all_reduce_ms": all_reduce_ms,
                              "all_reduce_operand": f"{n_sims} packed int64 keys per step" if collective else None,
                              "all_reduce_share_of_step": (all_reduce_ms / ms_per_step) if all_reduce_ms else None}
        if world == 1:   # side measurements and CPU legs at N = 1 only (other ranks would idle in a barrier)
            extra["copy_bandwidth_gbps"] = _try(lambda: ctx.copy_bandwidth_gbps(1 << 30, 10))
            extra["read_stream_gbps"] = _try(lambda: {"4B_per_lane": ctx.stream_probe_gbps(1 << 30, 4, 5), "16B_per_lane": ctx.stream_probe_gbps(1 << 30, 16, 5),
                                                             "32B_scalar_load_per_wave": ctx.stream_probe_gbps(1 << 30, 0, 5)})
            if not args.no_configs:
                extra["configs"] = config_rows(kaa, ctx, workloads, kinds)
            if not args.no_next_rows:
                extra.update(_try(lambda: next_rows(kaa, ctx, workloads)) or {})
            if not args.no_c3:
                extra["c3_in_process_multi_device"] = _try(lambda: in_process_multi_device(kaa, workloads, kinds))
            if not args.no_configs:
                extra["per_call_crossover"] = _try(lambda: per_call_crossover(kaa, ctx, workloads))
                extra["chained_loop"] = _try(lambda: chained_loop_row(kaa, ctx, workloads))
                # the other BASELINE configs that fit one GPU as verified batched throughput rows (C2 is the headline itself)
                torch.cuda.synchronize()
                extra["headline_rows"]["c4_resident"] = _try(lambda: batched_config_row(kaa, torch, dev_index, workloads, TableSet, "C4", 2048, 32, kinds, K, verify=not args.no_verify))
                extra["headline_rows"]["c3_resident"] = _try(lambda: batched_config_row(kaa, torch, dev_index, workloads, TableSet, "C3", 512, 8, kinds, K, verify=not args.no_verify))
                extra["headline_rows"]["c1_resident"] = _try(lambda: batched_config_row(kaa, torch, dev_index, workloads, TableSet, "C1", 4096, 32, kinds, K, verify=not args.no_verify))
            if not args.no_feasibility_row:
                # the HBM roofline on the kernel that can carry it: the batched feasibility launch, C2 and C3 shapes
                torch.cuda.synchronize()
                extra["roofline_feasibility"] = _try(lambda: feasibility_roofline(kaa, ctx, workloads, TableSet, "C2", 16384, 64, verify=not args.no_verify))
                extra["roofline_feasibility_c3"] = _try(lambda: feasibility_roofline(kaa, ctx, workloads, TableSet, "C3", 1024, 8, verify=not args.no_verify))
                if isinstance(extra.get("roofline_feasibility"), dict):
                    extra["roofline_feasibility"]["kernel_ms_in_loop"] = kms.get("feasibility_csr_ms")
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(workloads, make, range(min(S, 8)), checks_per_sim)
            extra["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.config, checks_per_sim)
        desc = {"C1": "10k pending pods x 256 candidate nodes, CPU+mem, 1 node group",
                "C2": "10k pending pods x 1k candidate nodes across 20 node groups, taints / tolerations + nodeSelector",
                "C4": "10k pending pods x 1k candidate nodes, 20 node groups, pod anti-affinity"}[args.config]
        out = {"metric": "scale-up simulation predicate checks/s (pods x nodes)", "value": value, "unit": "checks/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               # the arithmetic type of the kernels that ran: the register packer and the batch feasibility kernel compute on int32 lanes
               # (every resource lane divided by the gcd of its values on the host: exact), the generic packer on the boundary's int64
               "dtype": "int32" if fast else "int64", "data": "synthetic",
               "config": {"workload": f"{args.config} x {B} simulations per GPU per step ({desc}; {S} distinct seeds tiled), "
                                      f"`value` = RESIDENT regime: tables resident in HBM, results stay on the device (value_wall = enter -> return, PCIe inclusive); "
                                      f"step = feasibility + CSR + order + pack + expander reduce per simulation, "
                                      f"the batch as {K} sub-batches on {K} HIP streams",
                          "batch_per_gpu": B, "distinct_seeds": S, "checks_per_simulation": checks_per_sim,
                          "streams": K, "forks_from_the_context_stream": info.get("forks"), "streams_parked_by_the_lane_probe": info.get("parked_streams"), "streams_where": "inside libcasim (casim_options.n_streams): ONE casim_ctx, one casim_problem",
                          "simulations_per_stream": [(n_sims * (i + 1)) // K - (n_sims * i) // K for i in range(K)],
                          "node_groups_per_rank": mine.n_groups, "schedulable_peg_group_pairs_per_rank": my_nnz,
                          "expander": args.expander,
                          "partition": ("node groups of every simulation block-partitioned over the ranks (rotated), PEG table replicated"
                                        if world > 1 else "single GPU"),
                          "reduce": (f"{'rccl' if backend == 'nccl' else backend} all_reduce(min) on {n_sims} packed int64 keys per step, {world} rank(s)"
                                     if collective else "device kernel only"),
                          "all_ranks_on_one_gpu": bool(one_gpu and world > 1)},
               "roofline": roofline, "cpu_baseline": cpu}
        out.update(extra)
        out.update(side)
    batch.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        emit(out)   # the contract line is the last thing this process writes to stdout
    return out


def batched_config_row(kaa, torch, dev_index, workloads, TableSet, config, n_sims, n_seeds, kinds, K, steps=50, verify=True):
    """a verified THROUGHPUT figure for the other BASELINE configs that fit one GPU — `n_sims` simulations of `config`
    resident in HBM as one streamed casim_problem (the headline's regime: `steps` steps of run + expander reduce, nothing fetched), the
    kernels of sub-batch 0 timed with the device to itself, and every group of the last step compared with the oracle."""
    make = workloads.CONFIGS[config]
    ts = simulation_tables(make, range(n_seeds), kaa.Encoder, TableSet).tile((n_sims + n_seeds - 1) // n_seeds).head(n_sims)
    stream = torch.cuda.Stream(device=dev_index)
    b = kaa.StreamedBatch(dev_index, ts, n_streams=K, stream=stream.cuda_stream)
    try:
        for _ in range(5):
            b.run(); b.best_option_sims(kinds, fetch=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            b.run(); b.best_option_sims(kinds, fetch=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = b.prob.fetch()
        checks, nnz = checks_of(b.tables, res)
        alone_ms, kms = b.prob.time(iters=5)
        info = b.prob.info()
        row = {"what": f"{config} x {n_sims} simulations per step, resident ({n_seeds} distinct seeds tiled), {b.parts} sub-batches on {b.parts} streams", "ms_per_step": dt * 1e3,
               "checks_per_s": checks / dt, "sims_per_s": n_sims / dt, "checks_per_simulation": checks / n_sims, "node_groups": int(ts.n_groups), "pegs": int(ts.n_pegs),
               "schedulable_pairs": nnz, "kernel_ms_sub_batch_0_alone": kms, "packer": {"lanes": info["fast_packer_lanes"], "slots_per_lane": info["fast_packer_slots_per_lane"]},
               "dtype": "int32" if info["fast_packer_slots_per_lane"] > 0 and info["fast_packer_lanes"] != 8 else "int64", "steps": steps}
        row["ranked_orderer"] = bool(info.get("ranked_orderer"))
        if verify:
            chk = verify_headline(workloads, make, n_seeds, b.tables, res)
            row["bit_exact"] = bool(chk["headline_bit_exact"]); row["groups_compared"] = chk["groups_compared"]; row["verify_s"] = chk["verify_s"]
        # rocprofv3 of the same step (tools/config_prof.sh -> tools/config_counters.py -> profiles/config_counters.json, committed): the step's kernels
        # inside the 4-stream loop, and the dominant one's duration and instruction counters with the device to itself
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "config_counters.json"))).get(config)
            if prof:
                d = prof.get("dominant") or {}
                row["rocprof"] = {"run": prof.get("run"), "step_kernels_in_loop": prof.get("step_kernels_in_loop", [])[:4], "dominant": d}
                if d.get("valu_insts") and d.get("salu_insts") and d.get("kernel_us_alone"):
                    port = max(d["valu_insts"], d["salu_insts"])
                    cyc = d.get("cycles_per_valu") or 4.14
                    row["rocprof"]["issue_roofline"] = {"bound": "salu_issue" if d["salu_insts"] > d["valu_insts"] else "valu_issue",
                                                        "frac": port * cyc / (SIMDS * CLOCK_HZ) / (d["kernel_us_alone"] * 1e-6)}
        except (OSError, ValueError, KeyError):
            pass
        return row
    finally:
        b.close()


def per_call_crossover(kaa, ctx, workloads, iters=60):
    """where does ONE Estimate() per call stop paying on the device?  C1-shaped single-group calls (CPU + memory, PEGs of
    10 pods, the caller's own PEG list: what the shim's per-call path sends) over a sweep of (pods, node cap); per point the enter -> return
    wall time of casim_estimate_batch_query (median of `iters` calls, tables already encoded: the shim's per-call path reuses the loop's
    tables) next to the oracle's Estimate of the same call (one native call, best of a few).  `work` = pods x node bound is the quantity
    gpubinpacking.Routing compares with MinDeviceWork; `crossover_work` = the smallest work from which on the device wins at every larger
    point of the sweep."""
    from kubernetes_autoscaler_amd.engine import BatchCall
    from harness import GroupSpec, Scenario, encode
    from oracle_driver import OracleScenario
    rows = []
    for pods, cap in ((50, 5), (100, 10), (200, 10), (200, 40), (500, 20), (500, 100), (1000, 30), (1000, 100), (2000, 50), (2000, 256), (5000, 100), (10000, 256)):
        w = workloads.config_c1(n_pegs=max(1, pods // 10), pods_per_peg=10, cap=cap)
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, list(range(len(w.pegs)))) for g in w.groups], existing=w.existing,
                      lanes=w.lanes, device_csr=False)
        enc = encode(sc)
        bc = BatchCall(ctx, enc.pegs, enc.groups)
        for _ in range(10):
            bc.call_raw()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter(); bc.call_raw(); ts.append(time.perf_counter() - t0)
        ts.sort()
        o = OracleScenario(lanes=w.lanes)
        tmpl = o.node(w.groups[0].template)
        native = o.prepare_simulation([tmpl], w.pegs, [w.groups[0].max_nodes], [0])
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); native(False); best = min(best, time.perf_counter() - t0)
        o.close(); enc.close()
        n_pods = sum(len(pg.pods) for pg in w.pegs)
        rows.append({"pods": n_pods, "node_cap": cap, "work": n_pods * min(cap, n_pods), "device_call_us": ts[len(ts) // 2] * 1e6, "oracle_us": best * 1e6,
                     "device_over_oracle": ts[len(ts) // 2] / best})
    rows.sort(key=lambda r: r["work"])
    cross = None
    for i, r in enumerate(rows):
        if all(x["device_over_oracle"] < 1.0 for x in rows[i:]):
            cross = r["work"]
            break
    return {"what": "one Estimate() per call, C1-shaped (one group, PEGs of 10 pods, the caller's PEG list), device enter -> return vs the oracle's native Estimate (one EPYC core, "
                    "a C restatement: the Go reference is slower, i.e. its crossover lies lower)",
            "rows": rows, "crossover_work": cross, "shim_default_min_device_work": SHIM_MIN_DEVICE_WORK,
            "note": "gpubinpacking.Routing (integration/go/gpubinpacking/estimator.go) hands calls with pods x node bound below MinDeviceWork to the reference estimator"}


def chained_loop_row(kaa, ctx, workloads, replicas=16, iters=12):
    """the Go shim's prefetch is ONE simulation with a group per node group, chained (casim_options.chain_last_index) — a
    cluster with a few hundred node groups used to enqueue a few hundred fix-up passes, nearly all of them empty launches.  Config C2's 20
    groups `replicas` times over (320 node groups, 400 PEGs, one chain): enter -> return of casim_estimate_batch_query with the chain stopped
    at its fixed point (default) against the whole bound enqueued (CASIM_CHAIN_ASYNC_MAX raised), identical results, and against the
    oracle's chained loop on one core."""
    import numpy as np
    from kubernetes_autoscaler_amd.engine import BatchCall
    from harness import GroupSpec, Scenario, encode, run_oracle, assert_matches_oracle
    w = workloads.config_c2()
    base = [GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups]
    sc = Scenario(pegs=w.pegs, groups=base * replicas, existing=w.existing, lanes=w.lanes, device_csr=True)
    enc = encode(sc)
    out = {"workload": f"C2's {len(base)} node groups x {replicas} = {len(base) * replicas} groups in ONE chained simulation, {len(w.pegs)} PEGs (the Go shim's prefetch fill)"}
    results = {}
    for label, env in (("stop_at_fixed_point", None), ("whole_bound_enqueued", "1000000")):
        if env is None:
            os.environ.pop("CASIM_CHAIN_ASYNC_MAX", None)
        else:
            os.environ["CASIM_CHAIN_ASYNC_MAX"] = env
        bc = BatchCall(ctx, enc.pegs, enc.groups, chain_last_index=True)
        for _ in range(3):
            bc.call_raw()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter(); bc.call_raw(); ts.append(time.perf_counter() - t0)
        ts.sort()
        res, _ = bc.call()
        results[label] = res
        out[label] = {"call_ms": ts[len(ts) // 2] * 1e3, "chain": kaa.Context.last_chain_info()}
    os.environ.pop("CASIM_CHAIN_ASYNC_MAX", None)
    a, b = results["stop_at_fixed_point"], results["whole_bound_enqueued"]
    out["identical_results"] = bool(all(np.array_equal(getattr(a, f), getattr(b, f)) for f in ("node_count", "pods_scheduled", "nodes_added", "last_index_out", "status", "order", "placed")))
    t0 = time.perf_counter()
    want = run_oracle(sc, chain=True)
    out["oracle_ms"] = (time.perf_counter() - t0) * 1e3
    try:
        assert_matches_oracle(a, want, "chained loop")
        out["bit_exact"] = True
    except AssertionError as e:
        out["bit_exact"] = False
        out["first_difference"] = str(e)[:200]
    enc.close()
    return out


def feasibility_bytes(ts, lean):
    """Algorithmic bytes of ONE feas_stream_kernel launch over the table set (P x Bp + N x Bn + ceil(P x N / 8) with the record
    sizes of the kernel that runs): per PEG the columns a cell needs as the kernel reads them — 4 B per narrowed
    request lane, 4 B flags, one 8-byte word per mask kind; per group its 64-byte record; per (group, 64 PEGs) one 8-byte ballot word."""
    d = ts.dims
    lanes = min(d["n_res"], 2 if lean else 4)
    Bp = 4 * lanes + 4 + 8 * (1 if d["w_taint"] else 0) + 8 * (1 if d["w_label"] else 0) + (0 if lean else 16 * (1 if d["w_excl"] else 0) + 8 * (1 if d["w_zone"] else 0))
    Bn = 64
    import numpy as np
    words = (np.asarray(ts.peg_hi, np.int64) - np.asarray(ts.peg_lo, np.int64) + 63) // 64
    out = int(words.sum()) * 8
    return ts.n_pegs * Bp + ts.n_groups * Bn + out, Bp, Bn, out


def feasibility_roofline(kaa, ctx, workloads, TableSet, config, n_sims, n_seeds, iters=50, verify=True):
    """The HBM-roofline row BASELINE.md section 4 asks for, on the kernel that can carry it (VERDICT r4 missing #5): the SchedulablePodGroups
    matrix of a BATCHED launch — `n_sims` simulations of `config` (n_seeds distinct ones, tiled), one launch of feas_stream_kernel — timed
    alone (casim_problem_time_feasibility: `iters` launches back to back between two HIP events), algorithmic bytes per launch from the
    tables, and the whole problem's results checked against the oracle (every group of every tile)."""
    make = workloads.CONFIGS[config]
    ts = simulation_tables(make, range(n_seeds), kaa.Encoder, TableSet).tile((n_sims + n_seeds - 1) // n_seeds).head(n_sims)
    pegs, groups = ts.structs()
    with kaa.Problem(ctx, pegs, groups) as prob:
        ms, info = prob.time_feasibility(iters)
        row = {"bound": "hbm", "workload": f"{config} x {n_sims} simulations in ONE launch ({ts.n_pegs} PEGs, {ts.n_groups} node groups; {n_seeds} distinct seeds tiled)",
               "kernel": ("feas_stream_kernel<%s, %s, %s>" % ("lean" if info["lean"] else "full", "lo" if info["narrow_masks"] else "hi", "bit" if info["unsched_on_spare_bit"] else "term"))
                         if info["stream"] else "feas_sim_kernel",
               "kernel_ms": ms, "workgroups": info["workgroups"], "launches_timed": iters, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "traffic": None}
        b, Bp, Bn, out = feasibility_bytes(ts, info["lean"])
        row.update({"algorithmic_bytes_per_launch": b, "bytes_per_peg": Bp, "bytes_per_group_record": Bn, "bytes_written": out,
                    "achieved": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "cells_per_launch": int(((ts.peg_hi - ts.peg_lo).astype("int64")).sum()),
                    "bytes_note": "per PEG the columns the cell needs as the kernel reads them (narrowed int32 request lanes, flags, one word per mask kind), "
                                  "per group its 64-byte record, one bit per cell; the int64 boundary tables (16 B of requests per PEG instead of 8) are narrowed once per problem at init"})
        if verify:
            prob.run()
            res = prob.fetch()
            chk = verify_headline(workloads, make, n_seeds, ts, res)
            row["bit_exact"] = bool(chk["headline_bit_exact"]); row["groups_compared"] = chk["groups_compared"]
    try:   # counters of the same kernel and launch size, when a PMC pass has been committed (tools/feas_traffic.py)
        tr = json.load(open(os.path.join(ROOT, "profiles", "feas_traffic.json")))
        for r in tr.get("rows", []):
            if r.get("workgroups") == row["workgroups"] and r.get("kernel_tag") == row["kernel"]:
                row["traffic"] = r.get("traffic_bytes_per_launch"); row["traffic_source"] = "profiles/feas_traffic.json (%s)" % tr.get("run", "?")
                for k in ("valu_insts_per_launch", "salu_insts_per_launch", "kernel_ms_rocprof"):
                    if k in r:
                        row[k] = r[k]
    except (OSError, ValueError, KeyError):
        pass
    return row


def _same_results(a, b):
    """bit-equality of two BatchResults + expander answers (resident vs enter -> return vs int64)"""
    import numpy as np
    ra, ea = a; rb, eb = b
    nnz = int(ra.offsets[-1])
    ok = all(np.array_equal(getattr(ra, f), getattr(rb, f)) for f in ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes",
                                                                      "last_index_out", "status", "req_cpu_sum", "req_mem_sum"))
    ok = ok and np.array_equal(ra.order[:nnz], rb.order[:nnz]) and np.array_equal(ra.placed[:nnz], rb.placed[:nnz])
    return bool(ok and np.array_equal(ea["best"], eb["best"]) and np.array_equal(ea["packed"], eb["packed"]))


def _same_winners(a, b, order_mod=None):
    """a = (BatchResult, exp) of a winners_only call, b = the full answer: scalars, offsets, expander answer equal, and the compact lists are
    the winners' slices of the full lists.  order_mod: a's PEG ids are those of a shared PEG table of that many rows (TableSet.tile_groups), b's
    the tiled batch's (the same rows, shifted by a multiple of it per tile)"""
    import numpy as np
    ra, ea = a; rb, eb = b
    ok = all(np.array_equal(getattr(ra, f), getattr(rb, f)) for f in ("offsets", "node_count", "pods_scheduled", "nodes_added", "limiter_nodes",
                                                                      "last_index_out", "status", "req_cpu_sum", "req_mem_sum"))
    ok = ok and np.array_equal(ea["best"], eb["best"]) and np.array_equal(ea["packed"], eb["packed"])
    if not ok:
        return False
    w = ra.winner_offsets
    best = np.asarray(ea["best"], np.int64)
    has = np.nonzero(best >= 0)[0]
    # gather the winners' slices of the full lists in one go
    starts = rb.offsets[best[has]].astype(np.int64); lens = (rb.offsets[best[has] + 1] - rb.offsets[best[has]]).astype(np.int64)
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens) + np.arange(int(lens.sum()))
    n = int(w[-1])
    want_order = rb.order[idx] if order_mod is None else rb.order[idx] % order_mod
    return bool(n == int(lens.sum()) and np.array_equal(ra.order[:n], want_order) and np.array_equal(ra.placed[:n], rb.placed[idx]))


def link_table_bytes(tables, narrow=False, fastpath=False):
    """bytes of the caller's columns that cross the link in one call: every column the library ships (the fastpath chooser's four only with the
    fastpath; with casim_pegs.req32 the 4-byte requests + the units instead of the int64 table)"""
    n = 0
    for k, v in tables.pegs.items():
        if v is None or (k in ("fp_cpu", "fp_mem") and not fastpath):
            continue
        n += v.nbytes // 2 + 8 * v.shape[1] if (k == "req" and narrow) else v.nbytes
    for k, v in tables.groups.items():
        if v is None or (k in ("cap_cpu", "cap_mem") and not fastpath):
            continue
        n += v.nbytes
    for a in (tables.peg_lo, tables.peg_hi, tables.global_id, tables.sim_offsets):
        n += 0 if a is None else a.nbytes
    return int(n)


def enter_return_row(kaa, ctx, tables, kinds, K, checks_per_step, steps, res_resident, exp_resident, winners_only=True, pinned_results=False, order_mod=None, narrow=False):
    """SURVEY 8d's wall time: casim_estimate_batch_query enter -> return — fresh tables packed into pinned memory and copied to
    HBM, kernels, expander reduce, results copied back, EVERY step; the parts of the batch run end to end on the context's internal
    streams (upload of one part under the kernels of another).  winners_only: the per-group scalars and offsets of every
    group, PEG order / pods placed of the winning group of every simulation only (compacted on the device); False = every list, what
    a shim that serves all Estimate() calls from the batch fetches."""
    from kubernetes_autoscaler_amd.engine import BatchCall
    pegs, groups = tables.structs(narrow_requests=narrow)
    if winners_only:
        call = BatchCall(ctx, pegs, groups, kinds=kinds, n_streams=K, winners_only=True)
        for _ in range(8):                # first calls: lanes' pools and pinned buffers grow to this call's sizes (a 15 ms call still showed up
            call.call_raw()               # among the first seven after three warm-up calls: tests/tools/enter_return_outliers.py)
        # (the interpreter's cyclic GC is parked for the timed calls: this process holds millions of workload objects, and a full collection
        # in the middle of a call showed up as one 10-16 ms call in ten — the harness's pause, not the library's)
        import gc
        gc.collect(); gc.disable()
        seq = []
        try:
            for _ in range(steps):
                t0 = time.perf_counter()
                call.call_raw()
                seq.append(time.perf_counter() - t0)
        finally:
            gc.enable()
        dt = sum(seq) / steps
        res, exp = call.call()
        bytes_in = link_table_bytes(tables, narrow)
        return {"what": "casim_estimate_batch_query enter -> return every step, casim_options.winners_only: H2D of fresh tables from pinned staging + kernels + "
                        "expander + winners' lists compacted on the device + D2H of every group's scalars / offsets and the winners' order / placed", "dtype": "int32",
                "ms_per_step": dt * 1e3, "checks_per_s": checks_per_step / dt, "sims_per_s": tables.n_sims / dt, "steps": steps,
                "ms_per_call_sequence": [round(x * 1e3, 3) for x in seq], "ms_per_step_median": sorted(seq)[len(seq) // 2] * 1e3,
                "table_bytes_in": bytes_in, "result_bytes_out": 8 * int(res.winner_offsets[-1]) + 52 * tables.n_groups + 16 * tables.n_sims,
                "pcie_inclusive": True, "bit_equal_to_resident": _same_winners((res, exp), (res_resident, exp_resident), order_mod)}
    import gc
    call = BatchCall(ctx, pegs, groups, kinds=kinds, n_streams=K, pinned_results=pinned_results)
    call.call_raw()                       # first call: lanes, pools, pinned buffers
    gc.collect(); gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(steps):
            call.call_raw()
        dt = (time.perf_counter() - t0) / steps
        one = BatchCall(ctx, pegs, groups, kinds=kinds, n_streams=0)
        one.call_raw()
        t0 = time.perf_counter()
        for _ in range(max(1, steps // 2)):
            one.call_raw()
        dt1 = (time.perf_counter() - t0) / max(1, steps // 2)
    finally:
        gc.enable()
    res, exp = call.call()
    bytes_in = link_table_bytes(tables, narrow)
    nnz = int(res.offsets[-1])
    return {"what": "casim_estimate_batch_query enter -> return every step: H2D of fresh tables from pinned staging + kernels + expander + D2H of "
                    "scalars / order / placed, parts overlapped on the internal streams", "dtype": "int32",
            "ms_per_step": dt * 1e3, "checks_per_s": checks_per_step / dt, "sims_per_s": tables.n_sims / dt, "steps": steps,
            "ms_per_step_one_stream": dt1 * 1e3, "table_bytes_in": bytes_in, "result_bytes_out": 8 * nnz + 48 * tables.n_groups,
            "pcie_inclusive": True, "bit_equal_to_resident": _same_results((res, exp), (res_resident, exp_resident))}


def int64_row(kaa, dev_index, tables, kinds, K, checks_per_step, steps, res_resident, exp_resident, torch, packer=1):
    """The resident step on int64 lanes end to end (the boundary's own type) — what a batch pays when the exact gcd narrowing to int32
    does not apply.  packer = 2 (casim_options.force_generic_packer == 2): the register store on two int64 lanes (round 4: what such a
    batch takes by itself); packer = 1: the LDS store's generic packer (more than two lanes, negative requests; every such batch until
    round 4)."""
    with kaa.StreamedBatch(dev_index, tables, n_streams=K, force_generic_packer=packer) as b:
        for _ in range(3):
            b.run(); b.best_option_sims(kinds, fetch=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            b.run(); b.best_option_sims(kinds, fetch=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res = b.fetch()
        exp = b.best_option_sims(kinds)
        info = b.prob.info()
    return {"what": "resident step, int64 lanes: " + ("register store on two int64 lanes (pack_fast64_kernel)" if packer == 2 else "force_generic_packer (int64 MemStore packer, node state in LDS)"),
            "dtype": "int64", "ms_per_step": dt * 1e3, "packer_lanes": info["fast_packer_lanes"],
            "checks_per_s": checks_per_step / dt, "sims_per_s": tables.n_sims / dt, "steps": steps,
            "register_packer": info["fast_packer_slots_per_lane"] > 0,
            "bit_equal_to_resident": _same_results((res, exp), (res_resident, exp_resident))}


def in_process_multi_device(kaa, workloads, kinds, iters=20):
    """SURVEY 8e inside ONE process (the shape a Go estimator has): casim_mctx over every visible device, one C3 simulation
    per call, enter -> return (tables -> every device, kernels, RCCL all-reduce(min) of the key, results back)."""
    import numpy as np
    from kubernetes_autoscaler_amd.tables import TableSet
    n = kaa.device_count()
    enc = encode_workload(workloads.config_c3(), kaa.Encoder)
    ts = TableSet.from_encoder(enc).as_one_simulation()
    pegs, groups = ts.structs()
    out = {"workload": "C3: 50k pods x 4k nodes, 64 node groups, one call = upload + kernels + reduce + fetch", "devices": n}
    with kaa.MultiContext(list(range(n)), use_rccl=True) as m:
        res, exp = m.estimate_batch(pegs, groups, kinds=kinds)
        walls = []
        for _ in range(iters):
            t0 = time.perf_counter()
            res, exp = m.estimate_batch(pegs, groups, kinds=kinds)
            walls.append((time.perf_counter() - t0) * 1e3)
        info = m.info()
        out.update({"wall_ms": float(np.median(walls)), "winner_group": int(exp["best"][0]), "reduce": "rccl all_reduce(min)" if info["last_reduce_by_rccl"] else "host",
                    "groups_per_device": info["groups_per_device"]})
    enc.close()
    return out


def _try(f):
    try:
        return f()
    except Exception as e:  # side probes must never take the headline number down
        return {"error": f"{type(e).__name__}: {e}"}


def sched_issue_roofline(row, kernels_ms):
    """Issue roofline of K_sched for a bench row from the committed PMC figures (profiles/sched_counters.json, tools/sched_counters.sh: separate
    rocprofv3 --pmc passes of the SAME workloads): the pass is ONE workgroup, so the hardware it can use is one CU = 4 SIMDs, each issuing
    one vector and one scalar instruction per ~4 cycles.  frac = busier port's wave-instructions / SIMDs in use x cycles per instruction /
    clock / kernel time; frac_of_device says what that is of the whole chip's 1024 SIMDs (the sequential pass cannot use them)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "sched_counters.json")))[row]
        c = rec["counters"]
        valu, salu = c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"]
        waves = int(round(c.get("SQ_WAVES", rec["workgroup_threads"] / 64)))
        simds = min(4, max(1, waves))
        cyc = rec.get("cycles_per_valu", 4.14)
        port = "salu_issue" if salu > valu else "valu_issue"
        t_issue = max(valu, salu) / simds * cyc / CLOCK_HZ
        if waves == 1:   # ONE wave issues one instruction at a time, whatever its port: every instruction counts
            port = "single_wave_issue"
            t_issue = (valu + salu + c.get("SQ_INSTS_LDS", 0.0) + c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0)) * cyc / CLOCK_HZ
        return {"bound": port, "valu_insts": valu, "salu_insts": salu, "waves": waves, "simds_in_use": simds, "cycles_per_inst": cyc,
                "issue_time_ms": t_issue * 1e3, "frac": t_issue / (kernels_ms * 1e-3), "frac_of_device": t_issue * simds / SIMDS / (kernels_ms * 1e-3),
                "wait_share_of_wave_cycles": (c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAIT_INST_ANY") and c.get("SQ_WAVE_CYCLES") else None,
                "wait_any_share_of_wave_cycles": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAIT_ANY") and c.get("SQ_WAVE_CYCLES") else None,
                "scope": ("one wave (the chain of candidates is sequential: lastIndex is a list position that every commit shifts, DESIGN 17e)" if waves == 1 else
                          "one workgroup on one CU (the pass is sequential by definition: every pod sees the placements before it)"),
                "source": f"profiles/sched_counters.json ({rec.get('run', '?')}), kernel {rec['kernel'][:60]}"}
    except (OSError, ValueError, KeyError, ZeroDivisionError) as e:
        return {"error": f"no committed counters for this row: {type(e).__name__}: {e}"}


def run_filters_row(kaa, ctx, OracleScenario):
    import numpy as np
    from kubernetes_autoscaler_amd.objects import NodeInfo, build_test_node, build_test_pod, with_labels, with_pod_hostname_anti_affinity
    from kubernetes_autoscaler_amd.scheduling import encode_pending_pods
    pod = build_test_pod("p", 100, 1000, with_pod_hostname_anti_affinity({"app": "p"}), with_labels({"app": "p"}))
    nodes = []
    for i in range(5000):
        info = NodeInfo(build_test_node(f"n-{i}", 10, 1000))
        info.pods.extend(build_test_pod(f"p-{i}-{j}", 1, 1, with_labels({"app": "p"})) for j in range(10))
        nodes.append(info)
    nodes.append(NodeInfo(build_test_node("n-5000", 1000, 1000)))
    enc, pc = encode_pending_pods(nodes, [pod])
    rc, node_out, li, ns = ctx.try_schedule_pods(enc.pegs, enc.groups, pc)
    _, ms = ctx.try_schedule_pods(enc.pegs, enc.groups, pc, time_iters=20)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.try_schedule_pods(enc.pegs, enc.groups, pc)
    call_ms = (time.perf_counter() - t0) / 5 * 1e3
    with kaa.ResidentCluster(ctx, enc.pegs, enc.groups) as cl:
        cl.try_schedule_pods(pc, commit=False)
        t0 = time.perf_counter()
        for _ in range(20):
            cl.try_schedule_pods(pc, commit=False)
        resident_ms = (time.perf_counter() - t0) / 20 * 1e3
    enc.close()
    s = OracleScenario()
    for info in nodes:
        s.add_existing(info)
    s.pod(pod)
    want = s.try_schedule_pods([pod], None, None, None, False, 0)
    oracle_ms = s.last_native_s * 1e3
    for _ in range(4):
        s.try_schedule_pods([pod], None, None, None, False, 0)
        oracle_ms = min(oracle_ms, s.last_native_s * 1e3)
    s.close()
    exact = bool(rc == 0 and np.array_equal(np.asarray(node_out), want[0]) and int(li) == want[1] and int(ns) == want[2])
    return {"workload": "BenchmarkRunFiltersUntilPassingNode: 1 pod x 5001 label-less nodes (50 000 running pods), hostname anti-affinity term",
            "status": int(rc), "passing_node": int(node_out[0]), "reference_answer": {"passing_node": 5000, "matches": int(node_out[0]) == 5000,
                                                                                    "source": "plugin_runner_test.go:524-583: \"Last node is the only one that can fit the pod\""},
            "kernels_ms": ms, "call_ms_tables_uploaded": call_ms, "call_ms_resident_cluster": resident_ms, "oracle_ms": oracle_ms, "bit_exact": exact,
            "cpu_baseline": {"kind": "port", "cores": 1, "what": "orc_try_schedule_pods, the native call alone (best of 5)"}}


def removal_row(kaa, ctx, OracleScenario, w2, counters_row, ext_capacity=None):
    """One scale-down removal workload on the device — the kernel the library picks, then K_sched's general loop forced — and through
    the oracle's native call: kernel times, the A/B, bit-exactness in every field.  counters_row: name of the profiles/sched_counters.json row or None."""
    import numpy as np
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
    kw = {} if ext_capacity is None else {"ext_capacity": ext_capacity}
    r2 = ctx.simulate_node_removals(e2.pegs, e2.groups, w2.candidates, off, pcl, **kw)
    kern2 = ctx.last_removals_info()
    _, ms2 = ctx.simulate_node_removals(e2.pegs, e2.groups, w2.candidates, off, pcl, time_iters=5, **kw)
    # A/B: the same call through K_sched's general transaction loop (CASIM_NO_LEAN_REMOVALS is read when the call is prepared)
    prev = os.environ.get("CASIM_NO_LEAN_REMOVALS")
    os.environ["CASIM_NO_LEAN_REMOVALS"] = "1"
    try:
        r2b = ctx.simulate_node_removals(e2.pegs, e2.groups, w2.candidates, off, pcl, **kw)
        _, ms2b = ctx.simulate_node_removals(e2.pegs, e2.groups, w2.candidates, off, pcl, time_iters=5, **kw)
    finally:
        if prev is None:
            del os.environ["CASIM_NO_LEAN_REMOVALS"]
        else:
            os.environ["CASIM_NO_LEAN_REMOVALS"] = prev
    same2 = bool(np.array_equal(r2.removable, r2b.removable) and np.array_equal(r2.node_out, r2b.node_out) and int(r2.last_index) == int(r2b.last_index))
    e2.close()
    s = OracleScenario()
    for info in w2.nodes:
        s.add_existing(info)
    lists = [list(w2.nodes[c].pods) for c in w2.candidates]
    for lst in lists:
        for p in lst:
            s.pod(p)
    want2 = s.simulate_node_removals(w2.candidates, lists, None, None, True, 0, None, ext_capacity, 0, None)
    oracle2_ms = s.last_native_s * 1e3
    s.close()
    ext_same = [tuple(x) for x in zip(r2.ext_candidate.tolist(), r2.ext_pod.tolist(), r2.ext_node.tolist())] == [tuple(x) for x in want2["ext"]]
    exact2 = bool(np.array_equal(np.asarray(r2.removable), want2["removable"]) and np.array_equal(np.asarray(r2.node_out), want2["node_out"]) and
                  int(r2.last_index) == want2["last_index"] and int(r2.n_processed) == want2["n_processed"] and ext_same)
    row = {"workload": w2.name, "nodes": len(w2.nodes), "candidates": len(w2.candidates),
           "removable": int((r2.removable == 1).sum()), "pods_moved": len(pcl), "pods_listed_again": len(want2["ext"]),
           "kernel": "removals_lean_kernel (one wave over per-class fit masks)" if kern2["lean"] else "sched_kernel (K_sched's transaction loop)",
           "kernels_ms": ms2, "candidates_per_s": len(w2.candidates) / (ms2 * 1e-3),
           "kernels_ms_k_sched": ms2b, "same_results_as_k_sched": same2,
           "oracle_ms": oracle2_ms, "bit_exact": exact2, "speedup_vs_oracle_kernels": oracle2_ms / ms2,
           "cpu_baseline": {"kind": "port", "cores": 1, "what": "orc_simulate_node_removals, the native call alone"}}
    if counters_row:
        row["issue_roofline"] = sched_issue_roofline(counters_row, ms2)
    # the same call through tools/casim_native (plain C++ over the C ABI): encode of the whole snapshot the way the Go shim hands it over (one spec record
    # per running pod; the plain ones in ONE casim_enc_add_running_pods over an interned string table: encode.go runningPods), then
    # casim_simulate_node_removals enter -> return with its tables uploaded
    try:
        import native_trace
        tpath = os.path.join("/tmp", f"casim_{w2.name}.trace")
        native_trace.trace_removals(w2, tpath, iters=5, bulk=True)[0].close()
        nrc, nat = native_trace.run_native(tpath)
        row["native"] = dict({k: nat[k] for k in ("enc_calls", "encode_calls_ms", "finalize_ms", "encode_ms", "wall_ms", "kernels_ms", "removable", "n_ext", "engine_error") if k in nat}, exit_code=nrc,
                             what="encode_ms: every casim_enc_* call (running pods through casim_enc_add_running_pods) + finalize; wall_ms: casim_simulate_node_removals enter -> return (tables uploaded by the call, ext_capacity 2 * pods + 64)")
    except Exception as e:
        row["native"] = {"error": f"{type(e).__name__}: {e}"}
    return row


def next_rows(kaa, ctx, workloads):
    """The callers either side of the path, one mid-size case each: resident tables, HIP-event time; the oracle on the
    same input beside it (one host core, the native call alone), bit-exact flag, issue roofline of K_sched."""
    import numpy as np
    from kubernetes_autoscaler_amd.scheduling import encode_pending_pods
    from oracle_driver import OracleScenario
    out = {}
    w1 = workloads.pending_scale(5000, 50000, 64, 2)
    e1, pc1 = encode_pending_pods(w1.nodes, w1.pods)
    _, node_out1, li1, ns1 = ctx.try_schedule_pods(e1.pegs, e1.groups, pc1)
    _, ms1 = ctx.try_schedule_pods(e1.pegs, e1.groups, pc1, time_iters=5)
    s = OracleScenario()
    for info in w1.nodes:
        s.add_existing(info)
    canon = {}
    opods = [canon.setdefault(p.spec_key(), p) for p in w1.pods]
    for p in canon.values():
        s.pod(p)
    want1 = s.try_schedule_pods(opods, None, None, None, False, 0)
    oracle1_ms = s.last_native_s * 1e3
    s.close()
    exact1 = bool(np.array_equal(np.asarray(node_out1), want1[0]) and int(li1) == want1[1] and int(ns1) == want1[2])
    # the same call enter -> return: every table uploaded by the call vs the node table resident (casim_cluster_*)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.try_schedule_pods(e1.pegs, e1.groups, pc1)
    call_ms = (time.perf_counter() - t0) / 5 * 1e3
    with kaa.ResidentCluster(ctx, e1.pegs, e1.groups) as cl:
        cl.try_schedule_pods(pc1, commit=False)
        t0 = time.perf_counter()
        for _ in range(5):
            cl.try_schedule_pods(pc1, commit=False)
        resident_ms = (time.perf_counter() - t0) / 5 * 1e3
    out["try_schedule_pods"] = {"workload": w1.name, "nodes": len(w1.nodes), "pending_pods": len(w1.pods), "scheduled": int(ns1),
                                "kernels_ms": ms1, "pods_per_s": len(w1.pods) / (ms1 * 1e-3), "call_ms_tables_uploaded": call_ms,
                                "call_ms_resident_cluster": resident_ms, "oracle_ms": oracle1_ms, "bit_exact": exact1,
                                "speedup_vs_oracle_call": oracle1_ms / call_ms, "cpu_baseline": {"kind": "port", "cores": 1, "what": "orc_try_schedule_pods, the native call alone"},
                                "issue_roofline": sched_issue_roofline("try_schedule_pods", ms1)}
    e1.close()
    out["node_removals"] = removal_row(kaa, ctx, OracleScenario, workloads.removal_scale(5000, pods_per_node=12, frac_candidates=0.3, seed=1), "node_removals")
    # R3 = the reference's own benchmark of this loop: BenchmarkRunOnceScaleDown (core/bench/benchmark_runonce_test.go:505-521), 400 nodes at 40 %,
    # every node a candidate, verifyToBeDeleted(240) — tests/golden/reference_vectors.json: benchmark_runonce_scale_down
    try:
        r3 = removal_row(kaa, ctx, OracleScenario, workloads.runonce_scale_down(400), None, ext_capacity=64 * 1024)
        r3["reference_answer"] = {"to_be_deleted": 240, "matches": r3["removable"] == 240,
                                  "source": "BenchmarkRunOnceScaleDown: verifyToBeDeleted(240), core/bench/benchmark_runonce_test.go:505-521"}
        out["node_removals_runonce_scale_down"] = r3
    except Exception as e:  # a side table must never take the headline down
        out["node_removals_runonce_scale_down"] = {"error": f"{type(e).__name__}: {e}"}
    # R4 = BenchmarkRunFiltersUntilPassingNode (simulator/clustersnapshot/predicate/plugin_runner_test.go:524-583): ONE pod with a hostname
    # anti-affinity term against 5 001 nodes built by BuildTestNode (no labels: the term is inert), one node with room — a single
    # RunFiltersUntilPassingNode, the smallest unit of the path.  Device: one K_sched launch on the resident node table; a call that uploads its
    # tables first is dominated by the upload (the shim would not send ONE pod: Routing)
    try:
        out["run_filters_until_passing_node"] = run_filters_row(kaa, ctx, OracleScenario)
    except Exception as e:  # a side table must never take the headline down
        out["run_filters_until_passing_node"] = {"error": f"{type(e).__name__}: {e}"}
    out["group_pods"] = group_pods_row()
    out["incremental_encode"] = incremental_encode_row()
    return out


def group_pods_row():
    """SURVEY 8 f2, first half: equivalence.BuildPodGroups behind the C ABI (casim_enc_group_pods) at 150 000 pending pods, host only,
    through tools/casim_group_bench (plain C++ over include/casim.h)."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "casim_group_bench")
    if not os.path.exists(exe):
        return {"error": "tools/casim_group_bench not built"}
    r = subprocess.run([exe, "150000", "1500", "3", "5"], capture_output=True, text=True, timeout=300)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return {"reference": "CA/core/scaleup/equivalence/groups.go:39-104", "rows": rows} if rows else {"error": r.stderr[-300:]}


def incremental_encode_row():
    """second-iteration encode of a 15 000-node / 150 000-pod cluster with 1 % of the nodes changed (casim_enc_begin_update /
    _group_reset / _refinalize) next to the full encode of the same cluster, host only, through tools/casim_incr_bench (plain C++)."""
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "casim_incr_bench")
    if not os.path.exists(exe):
        return {"error": "tools/casim_incr_bench not built"}
    r = subprocess.run([exe, "15000", "10", "128", "1", "3"], capture_output=True, text=True, timeout=600)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return {"reference": "CA/simulator/clustersnapshot/store/delta.go:235-246,292-323 (the snapshot's O(1) fork + in-place AddPod)", "rows": rows} if rows else {"error": r.stderr[-300:]}


if __name__ == "__main__":
    main()
