"""Seeded synthetic scale-up workloads: BASELINE.md §3 configs C0..C4 plus a feature-mix fuzzer.

PRNG = splitmix64, seed 0xCA5CADE0 + config id.  All request values are exact
integers (no fractional milli), so Quantity rounding never engages.  Generators return plain
objects (kubernetes_autoscaler_amd.objects); nothing here touches the device or any checker."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

from .objects import (GiB, LABEL_HOSTNAME, LABEL_ZONE, MiB, ContainerPort, Node, NodeInfo, Pod, PodAffinityTerm,
                      PodEquivalenceGroup, Taint, Toleration, build_test_node, build_test_pod)

SEED_BASE = 0xCA5CADE0
MASK = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        return self.next() % n

    def pick(self, seq):
        return seq[self.below(len(seq))]

    def chance(self, num: int, den: int) -> bool:
        return self.below(den) < num

    def sample(self, seq, k: int):
        pool = list(seq)
        out = []
        for _ in range(min(k, len(pool))):
            out.append(pool.pop(self.below(len(pool))))
        return out


@dataclass
class GroupPlan:
    template: NodeInfo
    max_nodes: int = 0
    last_index: int = 0
    pegs: Optional[Sequence[int]] = None


@dataclass
class Workload:
    name: str
    pegs: List[PodEquivalenceGroup]
    groups: List[GroupPlan]
    existing: List[NodeInfo] = field(default_factory=list)
    lanes: Sequence[str] = ("cpu", "memory")

    @property
    def n_pods(self) -> int:
        return sum(len(p.pods) for p in self.pegs)

    def checks(self, pegs_per_group=None) -> int:
        """pods x nodes predicate checks of one simulation: sum_NG P_NG * Ncap_NG."""
        total = 0
        for gi, g in enumerate(self.groups):
            ids = pegs_per_group[gi] if pegs_per_group is not None else (g.pegs if g.pegs is not None else range(len(self.pegs)))
            total += sum(len(self.pegs[i].pods) for i in ids) * max(g.max_nodes, 0)
        return total


def _node(name, cpu_m, mem, pods, labels=None, taints=None) -> Node:
    cap = {"cpu": cpu_m, "memory": mem, "pods": pods}
    lab = {LABEL_HOSTNAME: name}
    lab.update(labels or {})
    return Node(name=name, labels=lab, taints=list(taints or []), capacity=dict(cap), allocatable=dict(cap))


def _peg(name, cpu_m, mem, count, **kw) -> PodEquivalenceGroup:
    pod = Pod(name=name, namespace=kw.pop("namespace", "default"), labels=kw.pop("labels", {"app": name}),
              requests={"cpu": cpu_m, "memory": mem}, **kw)
    return PodEquivalenceGroup(pods=[pod] * count)


def _score(cpu, mem, acpu, amem):
    return float(cpu) / float(acpu) + float(mem) / float(amem)


def _distinct_scores(rng, n, shapes, draw, max_rejects=None):
    """Draw n (cpu, mem) pairs whose orderer score is pairwise distinct on EVERY template shape
    (Go's sort.Slice is unstable; distinct scores make the reference order well defined).
    Raises ValueError when the draw space runs out of distinct scores (it used to spin forever: VERDICT r2 weak #9)."""
    out, seen = [], [set() for _ in shapes]
    rejects, limit = 0, (max_rejects if max_rejects is not None else 64 * n + 4096)
    while len(out) < n:
        cpu, mem = draw(rng)
        keys = [_score(cpu, mem, a, b) for a, b in shapes]
        if any(k in s for k, s in zip(keys, seen)):
            rejects += 1
            if rejects > limit:
                raise ValueError(f"_distinct_scores: only {len(out)} of {n} pairwise distinct scores after {rejects} rejected draws "
                                 f"(the draw space is exhausted; use a wider draw or allow ties)")
            continue
        for k, s in zip(keys, seen):
            s.add(k)
        out.append((cpu, mem))
    return out


def config_c0() -> Workload:
    """100 pods x 10 identical nodes, CPU+mem only (plumbing)."""
    tmpl = NodeInfo(_node("c0-template", 4000, 16 * GiB, 110))
    shapes = [(1500, 2 * GiB), (1000, 1 * GiB), (500, 4 * GiB), (250, 512 * MiB)]
    pegs = [_peg(f"c0-peg{i}", c, m, 25) for i, (c, m) in enumerate(shapes)]
    return Workload("C0", pegs, [GroupPlan(tmpl, max_nodes=10)])


def _draw_c1(rng):
    return 50 * (1 + rng.below(80)), 64 * MiB * (1 + rng.below(256))


C1_ALLOC = (32000, 128 * GiB)


def c1_pairs(seed_offset: int = 0, n_pegs: int = 200):
    """The (cpu milli, memory bytes) request pairs of one C1 simulation (same stream as config_c1)."""
    rng = SplitMix64(SEED_BASE + 1 + (seed_offset << 8))
    return _distinct_scores(rng, n_pegs, [C1_ALLOC], _draw_c1)


def config_c1(seed_offset: int = 0, n_pegs: int = 200, pods_per_peg: int = 50, cap: int = 256) -> Workload:
    """10k pods x 256 candidate nodes, CPU+mem only: 1 group, 32 cores / 128 GiB / 110 pods."""
    acpu, amem = C1_ALLOC
    tmpl = NodeInfo(_node("c1-template", acpu, amem, 110))
    pairs = c1_pairs(seed_offset, n_pegs)
    pegs = [_peg(f"c1-peg{i}", c, m, pods_per_peg) for i, (c, m) in enumerate(pairs)]
    return Workload("C1", pegs, [GroupPlan(tmpl, max_nodes=cap)])


TAINT_KEYS = [f"dedicated-{i}" for i in range(8)]
LABEL_KEYS = [f"pool-{i}" for i in range(8)]
SHAPES = [8, 16, 32, 64, 96]


def _c2_groups(rng, n_groups, cap, with_taints=True):
    groups = []
    for gi in range(n_groups):
        cores = SHAPES[gi % len(SHAPES)]
        labels = {LABEL_ZONE: f"zone-{gi % 3}", "shape": f"c{cores}"}
        for k in rng.sample(LABEL_KEYS, 4 + rng.below(5)):
            labels[k] = f"v{rng.below(4)}"
        taints = []
        if with_taints:
            for k in rng.sample(TAINT_KEYS, rng.below(3)):
                taints.append(Taint(k, f"t{rng.below(2)}", rng.pick(["NoSchedule", "NoExecute"])))
        node = _node(f"ng{gi}-template", cores * 1000, cores * 4 * GiB, 110, labels, taints)
        groups.append(GroupPlan(NodeInfo(node), max_nodes=cap[gi] if isinstance(cap, list) else cap))
    return groups


def _c2_pegs(rng, n_pegs, pods_per_peg, groups, prefix, with_tolerations=True, with_selectors=True):
    shapes = sorted({(g.template.node.allocatable["cpu"], g.template.node.allocatable["memory"]) for g in groups})
    pairs = _distinct_scores(rng, n_pegs, shapes, lambda r: (50 * (1 + r.below(80)), 64 * MiB * (1 + r.below(256))))
    pegs = []
    for i, (c, m) in enumerate(pairs):
        tols, sel = [], {}
        if with_tolerations:
            for k in rng.sample(TAINT_KEYS, 2 + rng.below(5)):
                if rng.chance(1, 2):
                    tols.append(Toleration(key=k, operator="Exists"))
                else:
                    tols.append(Toleration(key=k, operator="Equal", value=f"t{rng.below(2)}", effect=rng.pick(["", "NoSchedule", "NoExecute"])))
        if with_selectors and rng.chance(1, 2):
            for k in rng.sample(LABEL_KEYS, 1 + rng.below(2)):
                sel[k] = f"v{rng.below(4)}"
        pegs.append(_peg(f"{prefix}-peg{i}", c, m, pods_per_peg, tolerations=tols, node_selector=sel))
    return pegs


def config_c2(seed_offset: int = 0, n_groups: int = 20, n_pegs: int = 400, pods_per_peg: int = 25, cap: int = 50) -> Workload:
    """10k pods x 1k nodes across 20 node groups with taints/tolerations + nodeSelector."""
    rng = SplitMix64(SEED_BASE + 2 + (seed_offset << 8))
    groups = _c2_groups(rng, n_groups, cap)
    pegs = _c2_pegs(rng, n_pegs, pods_per_peg, groups, "c2")
    return Workload("C2", pegs, groups)


def config_c3(seed_offset: int = 0, n_groups: int = 64, n_pegs: int = 1000, pods_per_peg: int = 50) -> Workload:
    """50k pods x 4k nodes, 64 node groups (caps 62-63, sum 4000): sharded over the GPUs of one node."""
    rng = SplitMix64(SEED_BASE + 3 + (seed_offset << 8))
    total = (4000 * n_groups) // 64
    caps = [total // n_groups + (1 if i < total % n_groups else 0) for i in range(n_groups)]
    groups = _c2_groups(rng, n_groups, caps)
    pegs = _c2_pegs(rng, n_pegs, pods_per_peg, groups, "c3")
    return Workload("C3", pegs, groups)


def config_c4(seed_offset: int = 0, n_groups: int = 20, n_pegs: int = 400, pods_per_peg: int = 25, cap: int = 50) -> Workload:
    """10k pods x 1k nodes with pod anti-affinity: 50 % of the PEGs self-anti-affine on hostname,
    10 % anti-affine to another PEG's label."""
    rng = SplitMix64(SEED_BASE + 4 + (seed_offset << 8))
    groups = _c2_groups(rng, n_groups, cap, with_taints=False)
    pegs = _c2_pegs(rng, n_pegs, pods_per_peg, groups, "c4", with_tolerations=False)
    for i, pg in enumerate(pegs):
        pod = pg.pods[0]
        r = rng.below(10)
        if r < 5:
            pod.anti_affinity = [PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": pod.labels["app"]})]
        elif r == 5:
            other = pegs[rng.below(len(pegs))].pods[0]
            pod.anti_affinity = [PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": other.labels["app"]})]
    return Workload("C4", pegs, groups)


def config_r1(nodes: int = 200, pods_per_node: int = 50, node_cpu: int = 10000, node_mem: int = 10000, max_ng_size: int = 10000) -> Workload:
    """BenchmarkRunOnceScaleUp (CA/core/bench/benchmark_runonce_test.go:395-418,493-503): nodes * 50 controller-less pods
    (cpu = mem = node / 50), i.e. one SINGLETON PodEquivalenceGroup per pod (equivalence/groups.go:69-73), against ONE
    node group whose template is BuildTestNode("n-template", 10000, 10000) (pods capacity 100, test_utils.go:367-400), scale-up
    from zero, limiter = min(MaxNodesPerScaleUp, group max size, MaxNodesTotal) = 10000.  The reference verifies target size
    `nodes` (verifyTargetSize(200)).  Every PEG has the same score: the canonical tie rule (input order) applies, and because
    the pods are identical the result does not depend on it."""
    tmpl = NodeInfo(build_test_node("n-template", node_cpu, node_mem))
    cpu, mem = node_cpu // pods_per_node, node_mem // pods_per_node
    pegs = [PodEquivalenceGroup(pods=[build_test_pod(f"pod-{i}", cpu, mem)]) for i in range(nodes * pods_per_node)]
    return Workload("R1", pegs, [GroupPlan(tmpl, max_nodes=max_ng_size)])


def config_r2() -> Workload:
    """BenchmarkBinpackingEstimate (CA/estimator/binpacking_estimator_test.go:256-303): template 1000m / 5000 MiB / 100 pods,
    50 000 pods of (50m, 100 B) + 1 000 pods of (95m, 190 B), limiter 3000, one pre-existing node ("oldnode", E = 1).
    Known answer of the reference: 2595 nodes, 51 000 pods."""
    from .objects import make_node, make_pod_equivalence_group
    tmpl = NodeInfo(make_node(1000, 5000, 100, "template", "zone-mars"))
    old = NodeInfo(make_node(100, 100, 10, "oldnode", "zone-jupiter"))
    from .objects import with_labels, with_namespace
    mk = lambda cpu, mem: build_test_pod("estimatee", cpu, mem, with_namespace("universe"), with_labels({"app": "estimatee"}))
    pegs = [make_pod_equivalence_group(mk(50, 100), 50000), make_pod_equivalence_group(mk(95, 190), 1000)]
    return Workload("R2", pegs, [GroupPlan(tmpl, max_nodes=3000)], existing=[old])


def config_many_pegs(seed_offset: int = 0, n_pegs: int = 5000, cap: int = 200, max_count: int = 3) -> Workload:
    """One group with thousands of PEGs (the HBM-slab sort of order_kernel, list bound > 1024): mostly singletons
    (controller-less pods) with a few small controllers in between; scores tie freely (canonical rule =
    input order, decreasing_pod_orderer.go:46-88 + SURVEY N8)."""
    rng = SplitMix64(SEED_BASE + 0x51 + (seed_offset << 8))
    tmpl = NodeInfo(_node("many-template", 16000, 64 * GiB, 110))
    pegs = []
    for i in range(n_pegs):
        cpu, mem = 50 * (1 + rng.below(40)), 64 * MiB * (1 + rng.below(64))
        pegs.append(_peg(f"many-peg{i}", cpu, mem, 1 if rng.chance(3, 4) else 1 + rng.below(max_count)))
    return Workload(f"MANY{n_pegs}", pegs, [GroupPlan(tmpl, max_nodes=cap)])


def config_retry_mix() -> Workload:
    """Node BOUNDS beyond the register packer's 1024 slots in one launch: a roomy template (tens of nodes) and a tiny one
    (> 1024 nodes), unlimited and limited — the tiny ones are packed again by the generic packer's retry launch."""
    pegs = [_peg(f"p{i}", 100 + 10 * (i % 7), 128 * MiB * (1 + i % 3), 40 + 13 * (i % 5)) for i in range(60)]
    groups = [GroupPlan(NodeInfo(_node("roomy", 64000, 256 * GiB, 110)), max_nodes=0),
              GroupPlan(NodeInfo(_node("tiny", 300, 1 * GiB, 110)), max_nodes=0),
              GroupPlan(NodeInfo(_node("tiny-capped", 300, 1 * GiB, 110)), max_nodes=1500),
              GroupPlan(NodeInfo(_node("roomy-capped", 64000, 256 * GiB, 110)), max_nodes=3000, last_index=2)]
    return Workload("retry", pegs, groups)


CONFIGS = {"C0": config_c0, "C1": config_c1, "C2": config_c2, "C3": config_c3, "C4": config_c4, "R1": config_r1, "R2": config_r2}


def batch_of(make, n: int, **kw) -> Workload:
    """n independent simulations of one config (distinct seeds) concatenated into one batch: every
    group only sees the PEGs of its own simulation."""
    pegs, groups, name = [], [], None
    for b in range(n):
        w = make(seed_offset=b, **kw) if make is not config_c0 else make()
        name = w.name
        base = len(pegs)
        pegs.extend(w.pegs)
        for g in w.groups:
            ids = g.pegs if g.pegs is not None else range(len(w.pegs))
            groups.append(GroupPlan(g.template, g.max_nodes, g.last_index, [base + i for i in ids]))
    return Workload(f"{name}x{n}", pegs, groups)


# ---------------------------------------------------------------------------------------------
# feature-mix fuzzer (tests): small scenarios touching every encoded predicate and every exit
# of the packer; scores may tie here (the canonical tie rule = input order is part of parity)
# ---------------------------------------------------------------------------------------------
def fuzz(seed: int, max_groups: int = 4, max_pegs: int = 12, rich: bool = True) -> Workload:
    rng = SplitMix64(0xF0220000 + seed)
    n_groups = 1 + rng.below(max_groups)
    n_pegs = 1 + rng.below(max_pegs)
    n_existing = rng.below(4)
    ports = [5555, 8080, 9090]
    groups = []
    for gi in range(n_groups):
        cpu = rng.pick([1000, 2000, 4000, 8000])
        mem = rng.pick([1, 2, 8, 64]) * GiB
        labels = {LABEL_ZONE: f"zone-{rng.below(2)}"} if rng.chance(3, 4) else {}
        for k in rng.sample(LABEL_KEYS[:4], rng.below(4)):
            labels[k] = f"v{rng.below(2)}"
        taints = [Taint(k, f"t{rng.below(2)}", rng.pick(["NoSchedule", "NoExecute", "PreferNoSchedule"]))
                  for k in rng.sample(TAINT_KEYS[:3], rng.below(3))] if rich else []
        node = _node(f"fz{seed}-ng{gi}", cpu, mem, rng.pick([3, 10, 110]), labels, taints)
        if rich and rng.chance(1, 12):
            node.unschedulable = True
        pre = []
        if rich and rng.chance(1, 3):
            ds = Pod(name=f"ds{gi}", namespace="kube-system", labels={"app": "ds"}, requests={"cpu": 100, "memory": 64 * MiB})
            if rng.chance(1, 2):
                ds.host_ports = [ContainerPort(rng.pick(ports))]
            pre.append(ds)
        groups.append(GroupPlan(NodeInfo(node, pre), max_nodes=rng.pick([0, 0, 1, 3, 7, 64, -1]), last_index=rng.below(6)))
    pegs = []
    for i in range(n_pegs):
        cpu = rng.pick([0, 50, 100, 250, 500, 1000, 1500, 3000])
        mem = rng.pick([0, 64 * MiB, 256 * MiB, 1 * GiB, 3 * GiB])
        count = rng.pick([1, 1, 2, 3, 7, 20, 64, 130])
        kw = {}
        if rich:
            if rng.chance(1, 2):
                kw["tolerations"] = [Toleration(key=k, operator=rng.pick(["Exists", "Equal", ""]), value=f"t{rng.below(2)}",
                                                effect=rng.pick(["", "NoSchedule", "NoExecute"])) for k in rng.sample(TAINT_KEYS[:3], 1 + rng.below(3))]
                if rng.chance(1, 6):
                    kw["tolerations"].append(Toleration(operator="Exists"))  # tolerates everything
            if rng.chance(1, 4):
                kw["node_selector"] = {k: f"v{rng.below(2)}" for k in rng.sample(LABEL_KEYS[:4], 1)}
            if rng.chance(1, 5):
                kw["host_ports"] = [ContainerPort(rng.pick(ports), host_ip=rng.pick(["", "", "10.0.0.1"]), protocol=rng.pick(["", "TCP", "UDP"]))]
        app = f"app{rng.below(max(2, n_pegs // 2))}"
        pg = _peg(f"fz{seed}-peg{i}", cpu, mem, count, labels={"app": app}, **kw)
        pod = pg.pods[0]
        if rich and rng.chance(1, 3):
            target = rng.pick([app, f"app{rng.below(max(2, n_pegs // 2))}"])
            key = LABEL_HOSTNAME if rng.chance(3, 4) else LABEL_ZONE
            pod.anti_affinity = [PodAffinityTerm(key, match_labels={"app": target})]
        pegs.append(pg)
    existing = [NodeInfo(_node(f"fz{seed}-old{i}", 1000, 1 * GiB, 10, {LABEL_ZONE: f"zone-{rng.below(2)}"})) for i in range(n_existing)]
    return Workload(f"fuzz{seed}", pegs, groups, existing)


def fuzz_lean(seed: int, max_groups: int = 5, max_pegs: int = 16) -> Workload:
    """Small estimates of the shape the lean register packer takes in a batch (pack_fast_kernel<2, 1, 0>): cpu + memory requests (zero ones
    among them), taints / tolerations / selectors, DaemonSet pods on the template, unschedulable templates next to pods that tolerate
    everything, pod limits of 1 / 3 / 10 / 110, limiter values of every sign — and nothing that needs exclusion words, no PEG of more
    than 255 pods, no group whose node bound exceeds 64."""
    rng = SplitMix64(0x1EA70000 + seed)
    n_groups = 1 + rng.below(max_groups)
    n_pegs = 1 + rng.below(max_pegs)
    counts = [rng.pick([1, 1, 1, 2, 3, 7, 20, 25, 64, 130, 255]) for _ in range(n_pegs)]
    few = sum(max(c, 1) for c in counts) <= 64
    groups = []
    for gi in range(n_groups):
        cpu = rng.pick([1000, 2000, 4000, 8000, 16000])
        mem = rng.pick([1, 2, 8, 64]) * GiB
        labels = {LABEL_ZONE: f"zone-{rng.below(2)}"} if rng.chance(3, 4) else {}
        for k in rng.sample(LABEL_KEYS[:4], rng.below(4)):
            labels[k] = f"v{rng.below(2)}"
        taints = [Taint(k, f"t{rng.below(2)}", rng.pick(["NoSchedule", "NoExecute", "PreferNoSchedule"])) for k in rng.sample(TAINT_KEYS[:3], rng.below(3))]
        node = _node(f"ln{seed}-ng{gi}", cpu, mem, rng.pick([1, 3, 10, 110]), labels, taints)
        if rng.chance(1, 8):
            node.unschedulable = True
        pre = []
        if rng.chance(1, 3):
            pre.append(Pod(name=f"ds{gi}", namespace="kube-system", labels={"app": "ds"}, requests={"cpu": rng.pick([100, 900]), "memory": 64 * MiB}))
        limits = [1, 2, 3, 7, 19, 40, 50, 64, -1] + ([0, 0] if few else [])
        groups.append(GroupPlan(NodeInfo(node, pre), max_nodes=rng.pick(limits), last_index=rng.pick([0, 1, 2, 5, 9, 70])))
    pegs = []
    for i in range(n_pegs):
        cpu = rng.pick([0, 50, 100, 250, 500, 1000, 1500, 3000])
        mem = rng.pick([0, 64 * MiB, 256 * MiB, 1 * GiB, 3 * GiB])
        kw = {}
        if rng.chance(1, 2):
            kw["tolerations"] = [Toleration(key=k, operator=rng.pick(["Exists", "Equal", ""]), value=f"t{rng.below(2)}",
                                            effect=rng.pick(["", "NoSchedule", "NoExecute"])) for k in rng.sample(TAINT_KEYS[:3], 1 + rng.below(3))]
            if rng.chance(1, 3):
                kw["tolerations"].append(Toleration(operator="Exists"))  # tolerates everything, node.kubernetes.io/unschedulable included
        if rng.chance(1, 4):
            kw["node_selector"] = {k: f"v{rng.below(2)}" for k in rng.sample(LABEL_KEYS[:4], 1)}
        pegs.append(_peg(f"ln{seed}-peg{i}", cpu, mem, counts[i], labels={"app": f"app{i}"}, **kw))
    existing = [NodeInfo(_node(f"ln{seed}-old{i}", 1000, 1 * GiB, 10, {LABEL_ZONE: f"zone-{rng.below(2)}"})) for i in range(rng.below(4))]
    return Workload(f"lean{seed}", pegs, groups, existing)


def fuzz_singleton_runs(seed: int, max_groups: int = 3) -> Workload:
    """Runs of ADJACENT identical controller-less pods — one PodEquivalenceGroup each (equivalence/groups.go:69-73; the shape
    of BenchmarkRunOnceScaleUp) — between ordinary PEGs, against templates with limits of every sign, existing nodes, entry
    lastIndex values, an unschedulable template now and then: what casim_pipeline.h merges into one row per run (SingletonRuns).
    The CPU checker of the tests estimates every singleton on its own."""
    rng = SplitMix64(0x51A61E00 + seed)
    groups = []
    for gi in range(1 + rng.below(max_groups)):
        node = _node(f"sr{seed}-ng{gi}", rng.pick([1000, 2000, 4000, 8000]), rng.pick([2, 4, 8, 32]) * GiB, rng.pick([2, 3, 5, 10, 110]),
                     {LABEL_ZONE: f"zone-{rng.below(2)}", "pool": f"p{rng.below(2)}"},
                     [Taint("dedicated", "x", "NoSchedule")] if rng.chance(1, 6) else [])
        if rng.chance(1, 10):
            node.unschedulable = True
        groups.append(GroupPlan(NodeInfo(node, []), max_nodes=rng.pick([0, 0, 2, 5, 9, 40, -1]), last_index=rng.below(7)))
    pegs = []
    shapes = [(rng.pick([100, 250, 500, 1000]), rng.pick([128 * MiB, 512 * MiB, 1 * GiB, 2 * GiB])) for _ in range(3)]
    n = 0
    for _ in range(2 + rng.below(6)):
        kind = rng.below(4)
        if kind <= 1:                       # a run of identical singletons
            cpu, mem = rng.pick(shapes)
            kw = {}
            if rng.chance(1, 4):
                kw["tolerations"] = [Toleration(key="dedicated", operator="Exists")]
            if rng.chance(1, 5):
                kw["node_selector"] = {"pool": f"p{rng.below(2)}"}
            for _k in range(rng.pick([1, 2, 3, 5, 9, 17, 40])):
                pegs.append(_peg(f"sr{seed}-s{n}", cpu, mem, 1, labels={"app": "solo"}, **{k: (list(v) if isinstance(v, list) else dict(v)) for k, v in kw.items()})); n += 1
        elif kind == 2:                     # an ordinary PEG in between
            pegs.append(_peg(f"sr{seed}-p{n}", rng.pick([100, 300, 700]), rng.pick([256 * MiB, 1 * GiB]), rng.pick([2, 4, 11, 30]), labels={"app": f"c{n}"})); n += 1
        else:                               # singletons that differ (no run), one with a host port (never merged)
            for _k in range(1 + rng.below(3)):
                kw = {"host_ports": [ContainerPort(8080)]} if rng.chance(1, 3) else {}
                pegs.append(_peg(f"sr{seed}-d{n}", rng.pick([150, 350, 550]), rng.pick([128 * MiB, 384 * MiB]), 1, labels={"app": f"d{n}"}, **kw)); n += 1
    existing = [NodeInfo(_node(f"sr{seed}-old{i}", 1000, 1 * GiB, 10, {LABEL_ZONE: "zone-0"})) for i in range(rng.below(4))]
    return Workload(f"singleton_runs{seed}", pegs, groups, existing)


# ---------------------------------------------------------------------------------------------
# filter-out-schedulable workloads: pending pods against the nodes already in the cluster
# ---------------------------------------------------------------------------------------------
@dataclass
class PendingWorkload:
    name: str
    nodes: List[NodeInfo]
    pods: List[Pod]
    hints: Optional[List[int]] = None        # node index per pod or -1
    acceptable: Optional[List[int]] = None   # per node
    break_on_failure: bool = False
    last_index: int = 0


def filter_out_schedulable_benchmark(n_nodes: int, n_scheduled: int, n_pending: int) -> PendingWorkload:
    """BenchmarkFilterOutSchedulable (CA/core/podlistprocessor/filter_out_schedulable_test.go:212-300): nodes
    2000m / 200000 B, scheduled pods 1000m / 200000 B assigned round-robin, pending pods 1000m / 2000000 B (memory
    never fits: every pending pod stays pending — the reference runs pods x nodes Filters, its worst case)."""
    nodes = [NodeInfo(build_test_node(f"n-{i}", 2000, 200000)) for i in range(n_nodes)]
    for i in range(n_scheduled):
        nodes[i % n_nodes].pods.append(build_test_pod(f"s-{i}", 1000, 200000))
    pods = [build_test_pod(f"p-{i}", 1000, 2000000) for i in range(n_pending)]
    return PendingWorkload(f"fos_{n_nodes}n_{n_scheduled}s_{n_pending}p", nodes, pods)


def pending_scale(n_nodes: int, n_pending: int, n_classes: int = 32, seed: int = 1) -> PendingWorkload:
    """A packing-heavy variant: heterogeneous half-full nodes, `n_classes` controller specs whose pods mostly DO fit."""
    rng = SplitMix64(0x5C4ED000 + seed)
    nodes = []
    for i in range(n_nodes):
        cpu = rng.pick([4000, 8000, 16000, 32000])
        mem = rng.pick([16, 32, 64, 128]) * GiB
        info = NodeInfo(_node(f"n-{i}", cpu, mem, 110, {LABEL_ZONE: f"zone-{i % 3}", "pool": f"p{rng.below(4)}"}))
        for j in range(rng.below(6)):
            info.pods.append(Pod(name=f"r-{i}-{j}", labels={"app": "running"}, requests={"cpu": rng.pick([250, 500, 1000]), "memory": rng.pick([1, 2, 4]) * GiB}))
        nodes.append(info)
    specs = []
    for c in range(n_classes):
        kw = {}
        if rng.chance(1, 4):
            kw["node_selector"] = {"pool": f"p{rng.below(4)}"}
        specs.append(dict(labels={"app": f"c{c}"}, requests={"cpu": rng.pick([100, 250, 500, 1000, 2000]), "memory": rng.pick([128 * MiB, 512 * MiB, 1 * GiB, 4 * GiB])},
                          controller_uid=f"rs-{c}", **kw))
    pods = []
    while len(pods) < n_pending:
        c = rng.below(n_classes)
        for _ in range(min(rng.pick([1, 3, 10, 40, 200]), n_pending - len(pods))):
            pods.append(Pod(name=f"p-{len(pods)}", **{k: (dict(v) if isinstance(v, dict) else v) for k, v in specs[c].items()}))
    return PendingWorkload(f"pending_{n_nodes}n_{n_pending}p_{n_classes}c", nodes, pods)


def fuzz_pending(seed: int, max_nodes: int = 40, max_pods: int = 120) -> PendingWorkload:
    """Random small filter-out-schedulable scenario inside the encoded predicate subset (hostname anti-affinity,
    host ports, taints, selectors, unschedulable / unacceptable nodes, hints, runs of identical pods)."""
    rng = SplitMix64(0xF1F1000 + seed)
    n_nodes = 1 + rng.below(max_nodes) if not rng.chance(1, 8) else 60 + rng.below(140)
    ports = [5555, 8080, 9090]
    apps = [f"app{i}" for i in range(4)]
    nodes = []
    for i in range(n_nodes):
        labels = {k: f"v{rng.below(2)}" for k in rng.sample(LABEL_KEYS[:3], rng.below(3))}
        taints = [Taint(k, f"t{rng.below(2)}", rng.pick(["NoSchedule", "NoExecute", "PreferNoSchedule"]))
                  for k in rng.sample(TAINT_KEYS[:2], rng.below(2))] if rng.chance(1, 3) else []
        node = _node(f"fp{seed}-n{i}", rng.pick([500, 1000, 2000, 4000]), rng.pick([1, 2, 8]) * GiB, rng.pick([2, 5, 110]), labels, taints)
        if rng.chance(1, 10):
            node.unschedulable = True
        info = NodeInfo(node)
        for j in range(rng.below(3)):
            p = Pod(name=f"run{i}-{j}", labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([0, 100, 300]), "memory": rng.pick([0, 128 * MiB, 512 * MiB])})
            if rng.chance(1, 5):
                p.host_ports = [ContainerPort(rng.pick(ports))]
            if rng.chance(1, 6):
                p.anti_affinity = [PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": rng.pick(apps)})]
            info.pods.append(p)
        nodes.append(info)
    n_specs = 1 + rng.below(6)
    specs = []
    for c in range(n_specs):
        kw = dict(labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([0, 50, 100, 250, 500, 1000]), "memory": rng.pick([0, 64 * MiB, 256 * MiB, 1 * GiB])})
        if rng.chance(1, 3):
            kw["tolerations"] = [Toleration(key=k, operator=rng.pick(["Exists", "Equal", ""]), value=f"t{rng.below(2)}",
                                            effect=rng.pick(["", "NoSchedule", "NoExecute"])) for k in rng.sample(TAINT_KEYS[:2], 1 + rng.below(2))]
            if rng.chance(1, 4):
                kw["tolerations"].append(Toleration(operator="Exists"))
        if rng.chance(1, 4):
            kw["node_selector"] = {k: f"v{rng.below(2)}" for k in rng.sample(LABEL_KEYS[:3], 1)}
        if rng.chance(1, 5):
            kw["host_ports"] = [ContainerPort(rng.pick(ports), host_ip=rng.pick(["", "", "10.0.0.1"]), protocol=rng.pick(["", "TCP", "UDP"]))]
        if rng.chance(1, 4):
            kw["anti_affinity"] = [PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": rng.pick(apps)})]
        if rng.chance(1, 2):
            kw["controller_uid"] = f"ctrl-{c}"
        specs.append(kw)
    n_pods = 1 + rng.below(max_pods)
    pods, hints = [], []
    while len(pods) < n_pods:
        c = rng.below(n_specs)
        for _ in range(min(rng.pick([1, 1, 2, 5, 20, 70]), n_pods - len(pods))):
            kw = specs[c]
            pods.append(Pod(name=f"pend{len(pods)}", labels=dict(kw["labels"]), requests=dict(kw["requests"]),
                            tolerations=list(kw.get("tolerations", [])), node_selector=dict(kw.get("node_selector", {})),
                            host_ports=list(kw.get("host_ports", [])), anti_affinity=list(kw.get("anti_affinity", [])),
                            controller_uid=kw.get("controller_uid", "")))
            hints.append(rng.below(n_nodes) if rng.chance(1, 5) else -1)
    acceptable = [0 if rng.chance(1, 6) else 1 for _ in range(n_nodes)] if rng.chance(1, 3) else None
    return PendingWorkload(f"fuzz_pending{seed}", nodes, pods, hints if rng.chance(2, 3) else None, acceptable,
                           break_on_failure=rng.chance(1, 5), last_index=rng.below(n_nodes + 2))


# ---------------------------------------------------------------------------------------------
# scale-down workloads: which nodes can go once their pods are packed elsewhere
# ---------------------------------------------------------------------------------------------
@dataclass
class RemovalWorkload:
    name: str
    nodes: List[NodeInfo]
    candidates: List[int]                       # node indices, planner order
    destination: Optional[List[int]] = None
    hints: Optional[dict] = None                # id(pod) -> node index
    persist: bool = True
    max_removable: int = 0
    last_index: int = 0


def fuzz_removals(seed: int, max_nodes: int = 30) -> RemovalWorkload:
    """Random small cluster with under-used nodes; candidates in a random order."""
    rng = SplitMix64(0xD0D0000 + seed)
    n_nodes = 2 + rng.below(max_nodes) if not rng.chance(1, 10) else 70 + rng.below(120)
    ports = [5555, 8080]
    apps = [f"app{i}" for i in range(4)]
    n_specs = 1 + rng.below(5)
    specs = []
    for c in range(n_specs):
        kw = dict(labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([50, 100, 250, 500, 1000]), "memory": rng.pick([64 * MiB, 256 * MiB, 1 * GiB])})
        if rng.chance(1, 6):
            kw["host_ports"] = [ContainerPort(rng.pick(ports))]
        if rng.chance(1, 6):
            kw["anti_affinity"] = [PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": rng.pick(apps)})]
        if rng.chance(1, 5):
            kw["node_selector"] = {"pool": f"p{rng.below(2)}"}
        if rng.chance(1, 5):
            kw["tolerations"] = [Toleration(key="dedicated", operator="Exists")]
        specs.append(kw)
    nodes = []
    fill = rng.pick([5, 5, 12, 30])   # attempts to place a running pod per node: from nearly empty to crowded clusters
    for i in range(n_nodes):
        taints = [Taint("dedicated", "x", "NoSchedule")] if rng.chance(1, 8) else []
        node = _node(f"sd{seed}-n{i}", rng.pick([1000, 2000, 4000]), rng.pick([2, 4, 8]) * GiB, rng.pick([4, 8, 110]), {"pool": f"p{rng.below(2)}"}, taints)
        if rng.chance(1, 15):
            node.unschedulable = True
        info = NodeInfo(node)
        # fill greedily with random specs that fit the node on their own terms (the snapshot is a legal cluster state)
        cpu = mem = 0
        used_ports, apps_here = set(), []
        for _ in range(rng.below(fill)):
            kw = specs[rng.below(n_specs)]
            rq = kw["requests"]
            if cpu + rq["cpu"] > node.allocatable["cpu"] or mem + rq["memory"] > node.allocatable["memory"] or len(info.pods) >= node.allocatable["pods"]:
                continue
            pp = tuple(h.host_port for h in kw.get("host_ports", []))
            if any(x in used_ports for x in pp):
                continue
            aa = kw.get("anti_affinity", [])
            if any(t.match_labels["app"] in apps_here for t in aa):
                continue
            if any(q.anti_affinity and q.anti_affinity[0].match_labels["app"] == kw["labels"]["app"] for q in info.pods):
                continue
            if "node_selector" in kw and kw["node_selector"]["pool"] != node.labels["pool"]:
                continue
            if taints and "tolerations" not in kw:
                continue
            info.pods.append(Pod(name=f"r{i}-{len(info.pods)}", labels=dict(kw["labels"]), requests=dict(rq), host_ports=list(kw.get("host_ports", [])),
                                 anti_affinity=list(aa), node_selector=dict(kw.get("node_selector", {})), tolerations=list(kw.get("tolerations", [])),
                                 controller_uid=f"rs-{specs.index(kw)}"))
            cpu += rq["cpu"]; mem += rq["memory"]; used_ports.update(pp); apps_here.append(kw["labels"]["app"])
        if rng.chance(1, 6):
            info.pods.append(Pod(name=f"ds{i}", namespace="kube-system", labels={"app": "ds"}, requests={"cpu": 50, "memory": 32 * MiB}, daemonset=True))
        nodes.append(info)
    order = rng.sample(list(range(n_nodes)), 1 + rng.below(n_nodes))
    destination = [0 if rng.chance(1, 8) else 1 for _ in range(n_nodes)] if rng.chance(1, 3) else None
    hints = {}
    if rng.chance(1, 2):
        for c in order:
            for p in nodes[c].pods:
                if not p.daemonset and rng.chance(1, 4):
                    hints[id(p)] = rng.below(n_nodes)
    return RemovalWorkload(f"fuzz_removals{seed}", nodes, order, destination, hints or None, persist=not rng.chance(1, 5),
                           max_removable=rng.pick([0, 0, 0, 1, 3]), last_index=rng.below(n_nodes + 1))


def fuzz_removals_plain(seed: int, max_nodes: int = 40) -> RemovalWorkload:
    """fuzz_removals without host ports / anti-affinity (the shape the one-wave removal kernel takes, csrc/casim_sched.h removals_lean_kernel):
    resources, pod slots, node selectors, taints / tolerations, unschedulable nodes — crowded clusters so that simulations fail and are
    reverted, up to 40 pod specs, candidates that are destinations too (pods listed again), hints (some stale), a destination subset, now
    and then a cluster of more than 64 mask words' worth of nodes and a candidate with more pods than the kernel's LDS ring of placements."""
    rng = SplitMix64(0x1EA2000 + seed)
    big = rng.chance(1, 25)
    n_nodes = (4100 + rng.below(300)) if big else (2 + rng.below(max_nodes) if not rng.chance(1, 8) else 70 + rng.below(200))
    n_specs = 1 + rng.below(40 if rng.chance(1, 4) else 6)
    specs = []
    for c in range(n_specs):
        kw = dict(labels={"app": f"a{c % 5}"}, requests={"cpu": rng.pick([0, 50, 100, 250, 500, 1000]), "memory": rng.pick([0, 64 * MiB, 256 * MiB, 1 * GiB])})
        if rng.chance(1, 5):
            kw["node_selector"] = {"pool": f"p{rng.below(2)}"}
        if rng.chance(1, 5):
            kw["tolerations"] = [Toleration(key="dedicated", operator="Exists")]
        specs.append(kw)
    many = (not big) and rng.chance(1, 12)   # one node with hundreds of tiny pods
    fill = rng.pick([3, 5, 12, 30, 60])
    nodes = []
    for i in range(n_nodes):
        taints = [Taint("dedicated", "x", "NoSchedule")] if rng.chance(1, 8) else []
        node = _node(f"lr{seed}-n{i}", rng.pick([1000, 2000, 4000]), rng.pick([2, 4, 8]) * GiB, rng.pick([4, 8, 110]), {"pool": f"p{rng.below(2)}"}, taints)
        if many and i == 1:
            node = _node(f"lr{seed}-n{i}", 64000, 256 * GiB, 600, {"pool": "p0"})
        if rng.chance(1, 15):
            node.unschedulable = True
        info = NodeInfo(node)
        cpu = mem = 0
        for _ in range(rng.below((3 if big else fill) + 1) if not (many and i == 1) else 300 + rng.below(100)):
            kw = specs[rng.below(n_specs)] if not (many and i == 1) else specs[0]
            rq = kw["requests"] if not (many and i == 1) else {"cpu": 10, "memory": 8 * MiB}
            if cpu + rq["cpu"] > node.allocatable["cpu"] or mem + rq["memory"] > node.allocatable["memory"] or len(info.pods) >= node.allocatable["pods"]:
                continue
            if "node_selector" in kw and kw["node_selector"]["pool"] != node.labels["pool"]:
                continue
            if taints and "tolerations" not in kw:
                continue
            info.pods.append(Pod(name=f"r{i}-{len(info.pods)}", labels=dict(kw["labels"]), requests=dict(rq), node_selector=dict(kw.get("node_selector", {})),
                                 tolerations=list(kw.get("tolerations", [])), controller_uid=f"rs-{specs.index(kw)}"))
            cpu += rq["cpu"]; mem += rq["memory"]
        if rng.chance(1, 6):
            info.pods.append(Pod(name=f"ds{i}", namespace="kube-system", labels={"app": "ds"}, requests={"cpu": 50, "memory": 32 * MiB}, daemonset=True))
        nodes.append(info)
    if big:
        order = rng.sample(list(range(n_nodes)), 20 + rng.below(60))
    else:
        order = rng.sample(list(range(n_nodes)), 1 + rng.below(n_nodes))
        if many and 1 not in order:
            order.insert(rng.below(len(order) + 1), 1)
    destination = [0 if rng.chance(1, 8) else 1 for _ in range(n_nodes)] if rng.chance(1, 3) else None
    hints = {}
    if rng.chance(1, 2):
        for c in order:
            for p in nodes[c].pods:
                if not p.daemonset and rng.chance(1, 4):
                    hints[id(p)] = rng.below(n_nodes + 2)   # (now and then a node that is not there any more)
    return RemovalWorkload(f"fuzz_removals_plain{seed}", nodes, order, destination, hints or None, persist=not rng.chance(1, 5),
                           max_removable=rng.pick([0, 0, 0, 1, 3]), last_index=rng.below(n_nodes + 1))


def removal_scale(n_nodes: int, pods_per_node: int = 12, frac_candidates: float = 0.3, seed: int = 1) -> RemovalWorkload:
    """An under-used cluster: every node ~35 % full with controller pods; the emptiest nodes are candidates
    (utilization order, like the planner's eligibility + sorting processors)."""
    rng = SplitMix64(0x5CA1E000 + seed)
    nodes = []
    for i in range(n_nodes):
        info = NodeInfo(_node(f"n-{i}", 16000, 64 * GiB, 110, {LABEL_ZONE: f"zone-{i % 3}"}))
        for j in range(rng.below(pods_per_node + 1)):
            c = rng.below(24)
            info.pods.append(Pod(name=f"p-{i}-{j}", labels={"app": f"c{c}"}, requests={"cpu": 250 * (1 + c % 4), "memory": (1 + c % 3) * GiB},
                                 controller_uid=f"rs-{c}"))
        nodes.append(info)
    util = sorted(range(n_nodes), key=lambda i: (sum(p.requests["cpu"] for p in nodes[i].pods), i))
    cands = util[:max(1, int(n_nodes * frac_candidates))]
    return RemovalWorkload(f"removal_{n_nodes}n", nodes, cands)


def fuzz_removals_runs(seed: int) -> RemovalWorkload:
    """Plain clusters whose nodes run REPLICAS: a few pod specs, each node a handful of runs of 1-50 identical pods (what BenchmarkRunOnceScaleDown
    is made of, and what the one-wave removal kernel places a word of nodes at a time: csrc/casim_sched.h schedule_run).  Tight and loose clusters
    (runs that come round the node list, runs that fail half way and are reverted), pod-slot limits that bite before the resources do, node
    selectors / taints that thin the passing nodes out, clusters of several mask words, every node a candidate now and then (pods that arrived are
    listed again — in runs of their own), some hints, a destination subset, any lastIndex."""
    rng = SplitMix64(0x2B0B5000 + seed)
    big = rng.chance(1, 30)                    # more than 64 mask words: the walk crosses blocks of words
    n_nodes = (4100 + rng.below(200)) if big else rng.pick([3, 6, 12, 30, 64, 65, 100, 200]) + rng.below(5)
    n_specs = 1 + rng.below(rng.pick([1, 2, 4, 6]))
    specs = []
    for c in range(n_specs):
        kw = dict(labels={"app": f"a{c}"}, requests={"cpu": rng.pick([0, 10, 50, 100, 250]), "memory": rng.pick([0, 16 * MiB, 64 * MiB, 256 * MiB])})
        if rng.chance(1, 6):
            kw["node_selector"] = {"pool": f"p{rng.below(2)}"}
        if rng.chance(1, 6):
            kw["tolerations"] = [Toleration(key="dedicated", operator="Exists")]
        specs.append(kw)
    load = rng.pick([0.2, 0.4, 0.6, 0.8])      # how full the nodes start: 0.8 leaves room for few removals
    slots = rng.pick([20, 60, 110])
    nodes = []
    for i in range(n_nodes):
        taints = [Taint("dedicated", "x", "NoSchedule")] if rng.chance(1, 10) else []
        cpu_cap, mem_cap = rng.pick([2000, 4000, 8000]), rng.pick([4, 8, 16]) * GiB
        node = _node(f"rr{seed}-n{i}", cpu_cap, mem_cap, slots, {"pool": f"p{rng.below(2)}"}, taints)
        if rng.chance(1, 20):
            node.unschedulable = True
        info = NodeInfo(node)
        cpu = mem = 0
        for _ in range(1 + rng.below(4)):
            kw = specs[rng.below(n_specs)]
            if "node_selector" in kw and kw["node_selector"]["pool"] != node.labels["pool"]:
                continue
            if taints and "tolerations" not in kw:
                continue
            rq = kw["requests"]
            for _ in range(1 + rng.below(rng.pick([3, 10, 50]))):
                if cpu + rq["cpu"] > load * cpu_cap or mem + rq["memory"] > load * mem_cap or len(info.pods) >= load * slots:
                    break
                info.pods.append(Pod(name=f"r{i}-{len(info.pods)}", labels=dict(kw["labels"]), requests=dict(rq), node_selector=dict(kw.get("node_selector", {})),
                                     tolerations=list(kw.get("tolerations", [])), controller_uid=f"rs-{specs.index(kw)}"))
                cpu += rq["cpu"]; mem += rq["memory"]
        nodes.append(info)
    if big:
        order = rng.sample(list(range(n_nodes)), 20 + rng.below(40))
    else:
        order = list(range(n_nodes)) if rng.chance(1, 3) else rng.sample(list(range(n_nodes)), 1 + rng.below(n_nodes))
    destination = [0 if rng.chance(1, 8) else 1 for _ in range(n_nodes)] if rng.chance(1, 4) else None
    if big and rng.chance(1, 2):               # few destinations: runs come round a long list
        destination = [1 if rng.chance(1, 40) else 0 for _ in range(n_nodes)]
    hints = {}
    if rng.chance(1, 4):
        for c in order:
            for p in nodes[c].pods:
                if rng.chance(1, 10):
                    hints[id(p)] = rng.below(n_nodes + 1)
    return RemovalWorkload(f"fuzz_removals_runs{seed}", nodes, order, destination, hints or None, persist=not rng.chance(1, 6),
                           max_removable=rng.pick([0, 0, 0, 2, 5]), last_index=rng.below(n_nodes + 1))


def runonce_scale_down(n_nodes: int = 400, pods_per_node: int = 40) -> RemovalWorkload:
    """R3 = the reference's own benchmark of the scale-down path, BenchmarkRunOnceScaleDown (CA/core/bench/benchmark_runonce_test.go:505-521) on
    setupScaleDown60Percent(400) (:424-452): `n_nodes` nodes BuildTestNode(.., 10000, 10000) with 100 pod slots, pods_per_node pods of 1 % of a
    node each (pod i on node i % n_nodes, safe-to-evict, no controller), every node under the utilisation threshold and therefore a candidate, in
    list order, simulations persisted (the planner's NewRemovalSimulator(.., true)).  The benchmark's verify step holds the answer:
    verifyToBeDeleted(240) (:473-491) — 60 % of the nodes go, the other 160 end up exactly full."""
    nodes = [NodeInfo(_node(f"ng1-node-{i}", 10000, 10000, 100)) for i in range(n_nodes)]
    for j in range(n_nodes * pods_per_node):
        nodes[j % n_nodes].pods.append(Pod(name=f"pod-{j}", requests={"cpu": 100, "memory": 100}))
    return RemovalWorkload(f"runonce_scale_down_{n_nodes}n", nodes, list(range(n_nodes)))


def fuzz_pending_domains(seed: int, max_nodes: int = 40, max_pods: int = 90) -> PendingWorkload:
    """Like fuzz_pending, with the Filters that look at a node's topology DOMAIN: PodTopologySpread constraints
    (hostname / zone / rack keys, maxSkew 1-3, minDomains, selectors that do or do not match the pod itself, node
    selectors that make only part of the cluster eligible) and required anti-affinity on non-hostname keys, in pending
    AND running pods; some nodes lack a topology label."""
    from .objects import TopologySpreadConstraint
    rng = SplitMix64(0xD0A1A000 + seed)
    n_nodes = 2 + rng.below(max_nodes) if not rng.chance(1, 8) else 70 + rng.below(200)
    apps = [f"app{i}" for i in range(4)]
    keys = [LABEL_HOSTNAME, LABEL_ZONE, "rack"]
    n_zones, n_racks = 1 + rng.below(4), 1 + rng.below(6)
    nodes = []
    for i in range(n_nodes):
        labels = {"pool": f"p{rng.below(2)}"}
        if not rng.chance(1, 10):
            labels[LABEL_ZONE] = f"z{rng.below(n_zones)}"
        if not rng.chance(1, 6):
            labels["rack"] = f"r{rng.below(n_racks)}"
        node = _node(f"fd{seed}-n{i}", rng.pick([1000, 2000, 4000]), rng.pick([2, 8]) * GiB, rng.pick([3, 8, 110]), labels)
        if rng.chance(1, 12):
            node.unschedulable = True
        info = NodeInfo(node)
        for j in range(rng.below(3)):
            p = Pod(name=f"run{i}-{j}", labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([0, 100, 300]), "memory": rng.pick([0, 256 * MiB])})
            if rng.chance(1, 6):
                p.anti_affinity = [PodAffinityTerm(rng.pick(keys[1:]), match_labels={"app": rng.pick(apps)})]
            info.pods.append(p)
        nodes.append(info)
    n_specs = 1 + rng.below(5)
    specs = []
    for c in range(n_specs):
        kw = dict(labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([0, 50, 100, 250, 500]), "memory": rng.pick([0, 64 * MiB, 256 * MiB])})
        if rng.chance(1, 4):
            kw["node_selector"] = {"pool": f"p{rng.below(2)}"}
        if rng.chance(1, 2):
            n_c = 1 + rng.below(2)
            kw["spread_constraints"] = [TopologySpreadConstraint(max_skew=1 + rng.below(3), topology_key=rng.pick(keys), min_domains=rng.pick([0, 0, 1, 2, 4]),
                                                                 match_labels=({"app": kw["labels"]["app"]} if rng.chance(2, 3) else ({"app": rng.pick(apps)} if rng.chance(3, 4) else {})))
                                        for _ in range(n_c)]
        if rng.chance(1, 4):
            kw["anti_affinity"] = [PodAffinityTerm(rng.pick(keys), match_labels={"app": rng.pick(apps)})]
        if rng.chance(1, 3):
            kw["controller_uid"] = f"ctrl-{c}"
        specs.append(kw)
    n_pods = 1 + rng.below(max_pods)
    pods, hints = [], []
    while len(pods) < n_pods:
        c = rng.below(n_specs)
        for _ in range(min(rng.pick([1, 1, 2, 5, 20]), n_pods - len(pods))):
            kw = specs[c]
            pods.append(Pod(name=f"pend{len(pods)}", labels=dict(kw["labels"]), requests=dict(kw["requests"]),
                            node_selector=dict(kw.get("node_selector", {})), anti_affinity=list(kw.get("anti_affinity", [])),
                            spread_constraints=list(kw.get("spread_constraints", [])), topology_spread=bool(kw.get("spread_constraints")),
                            controller_uid=kw.get("controller_uid", "")))
            hints.append(rng.below(n_nodes) if rng.chance(1, 6) else -1)
    acceptable = [0 if rng.chance(1, 6) else 1 for _ in range(n_nodes)] if rng.chance(1, 4) else None
    return PendingWorkload(f"fuzz_pending_domains{seed}", nodes, pods, hints if rng.chance(1, 2) else None, acceptable,
                           break_on_failure=rng.chance(1, 6), last_index=rng.below(n_nodes + 2))


def fuzz_removals_domains(seed: int, max_nodes: int = 30) -> RemovalWorkload:
    """fuzz_removals on a cluster with topology labels whose running pods carry spread constraints and zone-level
    anti-affinity: removing a node takes its pods out of the domain counters, its ghost stays a domain for the
    simulation, a committed removal drops the node from its domains."""
    from .objects import TopologySpreadConstraint
    rng = SplitMix64(0xD0D1000 + seed)
    n_nodes = 2 + rng.below(max_nodes) if not rng.chance(1, 10) else 70 + rng.below(100)
    apps = [f"app{i}" for i in range(3)]
    keys = [LABEL_HOSTNAME, LABEL_ZONE, "rack"]
    n_zones, n_racks = 1 + rng.below(3), 1 + rng.below(5)
    n_specs = 1 + rng.below(5)
    specs = []
    for c in range(n_specs):
        kw = dict(labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([50, 100, 250, 500]), "memory": rng.pick([64 * MiB, 256 * MiB])})
        if rng.chance(1, 2):
            kw["spread_constraints"] = [TopologySpreadConstraint(max_skew=1 + rng.below(3), topology_key=rng.pick(keys), min_domains=rng.pick([0, 0, 2, 3]),
                                                                 match_labels=({"app": kw["labels"]["app"]} if rng.chance(3, 4) else {"app": rng.pick(apps)}))]
        if rng.chance(1, 4):
            kw["anti_affinity"] = [PodAffinityTerm(rng.pick(keys), match_labels={"app": rng.pick(apps)})]
        if rng.chance(1, 5):
            kw["node_selector"] = {"pool": f"p{rng.below(2)}"}
        specs.append(kw)
    nodes = []
    fill = rng.pick([3, 5, 8])
    for i in range(n_nodes):
        labels = {"pool": f"p{rng.below(2)}"}
        if not rng.chance(1, 12):
            labels[LABEL_ZONE] = f"z{rng.below(n_zones)}"
        if not rng.chance(1, 8):
            labels["rack"] = f"r{rng.below(n_racks)}"
        node = _node(f"sdd{seed}-n{i}", rng.pick([1000, 2000, 4000]), rng.pick([2, 4]) * GiB, rng.pick([4, 8, 110]), labels)
        info = NodeInfo(node)
        cpu = 0
        for _ in range(rng.below(fill)):
            kw = specs[rng.below(n_specs)]
            if cpu + kw["requests"]["cpu"] > node.allocatable["cpu"] or len(info.pods) >= node.allocatable["pods"]:
                continue
            if "node_selector" in kw and kw["node_selector"]["pool"] != labels["pool"]:
                continue
            info.pods.append(Pod(name=f"r{i}-{len(info.pods)}", labels=dict(kw["labels"]), requests=dict(kw["requests"]),
                                 anti_affinity=list(kw.get("anti_affinity", [])), node_selector=dict(kw.get("node_selector", {})),
                                 spread_constraints=list(kw.get("spread_constraints", [])), topology_spread=bool(kw.get("spread_constraints")),
                                 controller_uid=f"rs-{specs.index(kw)}"))
            cpu += kw["requests"]["cpu"]
        nodes.append(info)
    order = rng.sample(list(range(n_nodes)), 1 + rng.below(n_nodes))
    destination = [0 if rng.chance(1, 8) else 1 for _ in range(n_nodes)] if rng.chance(1, 3) else None
    return RemovalWorkload(f"fuzz_removals_domains{seed}", nodes, order, destination, None, persist=not rng.chance(1, 5),
                           max_removable=rng.pick([0, 0, 0, 2]), last_index=rng.below(n_nodes + 1))


def fuzz_estimate_domains(seed: int) -> Workload:
    """One node group whose PEGs carry PodTopologySpread constraints (hostname / zone / rack) and zone-level
    anti-affinity, next to a cluster whose nodes have room and topology labels of their own: exercises the
    estimator's hostname-spread retry (binpacking_estimator.go:212-227), which may place pods on cluster nodes."""
    from .objects import TopologySpreadConstraint
    rng = SplitMix64(0xE57D0000 + seed)
    apps = [f"app{i}" for i in range(3)]
    keys = [LABEL_HOSTNAME, LABEL_HOSTNAME, LABEL_ZONE, "rack"]
    n_zones = 1 + rng.below(3)
    existing = []
    for i in range(rng.below(6)):
        labels = {}
        if not rng.chance(1, 6):
            labels[LABEL_ZONE] = f"z{rng.below(n_zones)}"
        if rng.chance(1, 2):
            labels["rack"] = f"r{rng.below(3)}"
        info = NodeInfo(_node(f"fe{seed}-old{i}", rng.pick([200, 1000, 4000]), rng.pick([1, 4]) * GiB, rng.pick([3, 10, 110]), labels))
        if rng.chance(1, 8):
            del info.node.labels[LABEL_HOSTNAME]
        for j in range(rng.below(3)):
            p = Pod(name=f"old{i}-{j}", labels={"app": rng.pick(apps)}, requests={"cpu": rng.pick([0, 50, 100]), "memory": 64 * MiB})
            if rng.chance(1, 5):
                p.anti_affinity = [PodAffinityTerm(rng.pick([LABEL_ZONE, "rack"]), match_labels={"app": rng.pick(apps)})]
            info.pods.append(p)
        existing.append(info)
    tlabels = {}
    if not rng.chance(1, 8):
        tlabels[LABEL_ZONE] = f"z{rng.below(n_zones + 1)}"
    if rng.chance(1, 2):
        tlabels["rack"] = f"r{rng.below(4)}"
    tmpl = NodeInfo(_node(f"fe{seed}-tmpl", rng.pick([1000, 2000, 4000]), rng.pick([2, 8]) * GiB, rng.pick([4, 10, 110]), tlabels))
    if rng.chance(1, 4):
        tmpl.pods.append(Pod(name="ds", namespace="kube-system", labels={"app": rng.pick(apps)}, requests={"cpu": 100, "memory": 64 * MiB}))
    pegs = []
    for i in range(1 + rng.below(5)):
        app = rng.pick(apps)
        pod = Pod(name=f"fe{seed}-p{i}", labels={"app": app}, requests={"cpu": rng.pick([20, 100, 250, 500]), "memory": rng.pick([64 * MiB, 256 * MiB, 1 * GiB])})
        if rng.chance(2, 3):
            pod.spread_constraints = [TopologySpreadConstraint(max_skew=1 + rng.below(3), topology_key=rng.pick(keys), min_domains=rng.pick([0, 0, 1, 2, 3]),
                                                               match_labels=({"app": app} if rng.chance(3, 4) else {"app": rng.pick(apps)}))
                                      for _ in range(1 + rng.below(2))]
            pod.topology_spread = True
        if rng.chance(1, 5):
            pod.anti_affinity = [PodAffinityTerm(rng.pick([LABEL_HOSTNAME, LABEL_ZONE]), match_labels={"app": rng.pick(apps)})]
        if rng.chance(1, 6):
            pod.host_ports = [ContainerPort(8080)]
        pegs.append(PodEquivalenceGroup(pods=[pod] * rng.pick([1, 2, 5, 9, 14])))
    return Workload(f"fuzz_estimate_domains{seed}", pegs, [GroupPlan(tmpl, max_nodes=rng.pick([0, 0, 2, 5, -1]), last_index=rng.below(8))], existing)


# ---------------------------------------------------------------------------------------------
# required node affinity with several nodeSelectorTerms (ORed) and matchFields on metadata.name
# ---------------------------------------------------------------------------------------------
def add_random_node_affinity_terms(seed: int, pods: Sequence[Pod], nodes: Sequence[NodeInfo], allow_per_node: bool = True) -> int:
    """Decorates `pods` (same spec -> same terms, so equivalence classes survive) with random
    requiredDuringSchedulingIgnoredDuringExecution.nodeSelectorTerms over the label keys / values / names of `nodes`:
    0-3 terms, empty terms, every operator incl. Gt / Lt on a numeric `gen` label (added to some nodes here), operators
    that do not parse (a bad operator, In without values, Exists with values, matchFields with two values or Exists),
    and - when allow_per_node - matchFields on metadata.name and expressions on kubernetes.io/hostname.
    Returns the number of pods decorated."""
    from .objects import NodeSelectorTerm, Requirement
    rng = SplitMix64(0x7E4A5000 + seed)
    for info in nodes:
        if rng.chance(2, 3):
            info.node.labels["gen"] = str(rng.below(5))
    label_values: Dict[str, List[str]] = {}
    for info in nodes:
        for k, v in info.node.labels.items():
            if k != LABEL_HOSTNAME or allow_per_node:
                vs = label_values.setdefault(k, [])
                if v not in vs:
                    vs.append(v)
    keys = list(label_values) or ["gen"]
    names = [info.node.name for info in nodes] or ["nobody"]

    def expression():
        k = rng.pick(keys + ["absent-key"])
        vals = label_values.get(k, []) + ["other"]
        op = rng.pick(["In", "In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt", "In", "NotIn"])
        if rng.chance(1, 25):
            return Requirement(k, rng.pick(["Equals", "", "in"]), [rng.pick(vals)])          # not an operator
        if op in ("In", "NotIn"):
            return Requirement(k, op, rng.sample(vals, rng.below(3) if rng.chance(1, 12) else 1 + rng.below(2)))
        if op in ("Exists", "DoesNotExist"):
            return Requirement(k, op, [rng.pick(vals)] if rng.chance(1, 15) else [])
        return Requirement("gen" if rng.chance(4, 5) else k, op, [rng.pick(["0", "2", "3", "-1", "x", "+1"])] * (2 if rng.chance(1, 15) else 1))

    def field_requirement():
        key = rng.pick(["metadata.name", "metadata.name", "metadata.name", "metadata.namespace"])
        op = rng.pick(["In", "In", "NotIn", "Exists"])
        return Requirement(key, op, [rng.pick(names + ["nobody", ""]) for _ in range(2 if rng.chance(1, 10) else 1)])

    by_spec: Dict[tuple, List[Pod]] = {}
    for p in pods:
        by_spec.setdefault(p.spec_key(), []).append(p)
    decorated = 0
    for group in by_spec.values():
        if group[0].node_affinity or not rng.chance(3, 5):
            continue
        terms = []
        for _ in range(rng.pick([0, 1, 1, 2, 2, 3])):
            t = NodeSelectorTerm()
            for _ in range(rng.pick([0, 1, 1, 2])):
                t.match_expressions.append(expression())
            if allow_per_node and rng.chance(1, 3):
                for _ in range(rng.pick([1, 1, 2])):
                    t.match_fields.append(field_requirement())
            terms.append(t)
        for p in group:
            p.node_affinity_terms = [NodeSelectorTerm(list(t.match_expressions), list(t.match_fields)) for t in terms]
        decorated += len(group)
    return decorated


# ---------------------------------------------------------------------------------------------
# namespaceSelector of required anti-affinity terms
# ---------------------------------------------------------------------------------------------
def add_random_namespace_selectors(seed: int, pods: Sequence[Pod], hostname_only: bool = False) -> Dict[str, Dict[str, str]]:
    """Spreads `pods` (pending and running alike; same spec -> same treatment) over four namespaces with different labels,
    gives some anti-affinity terms a namespaceSelector (empty = every namespace, In / NotIn / Exists / DoesNotExist), drops
    the explicit namespace list of some, and adds a selector-only term to some pods that had none.  Returns the namespace
    table to install as the lister (objects.namespaces)."""
    from .objects import Requirement
    rng = SplitMix64(0x4E5AA000 + seed)
    table = {"default": {"team": "core"}, "ns-a": {"team": "a", "tier": "prod"}, "ns-b": {"team": "b"}, "ns-c": {}}
    names = list(table)
    selectors = [[], [Requirement("team", "In", ["a"])], [Requirement("team", "In", ["a", "b"])], [Requirement("tier", "Exists", [])],
                 [Requirement("team", "NotIn", ["a"])], [Requirement("team", "DoesNotExist", [])],
                 [Requirement("team", "Exists", []), Requirement("tier", "DoesNotExist", [])]]
    by_spec: Dict[tuple, List[Pod]] = {}
    for p in pods:
        by_spec.setdefault(p.spec_key(), []).append(p)
    apps = sorted({p.labels.get("app", "") for p in pods if p.labels.get("app")}) or ["app0"]
    for group in by_spec.values():
        ns = rng.pick(names)
        terms = [PodAffinityTerm(t.topology_key, dict(t.match_labels), list(t.match_expressions), tuple(t.namespaces), t.namespace_selector)
                 for t in group[0].anti_affinity]
        for t in terms:
            if rng.chance(1, 2):
                t.namespace_selector = list(rng.pick(selectors))
                t.namespaces = tuple(rng.sample(names, rng.below(3))) if rng.chance(1, 2) else ()
            elif rng.chance(1, 3):
                t.namespaces = tuple(rng.sample(names, 1 + rng.below(2)))
        if not terms and rng.chance(1, 4):
            terms.append(PodAffinityTerm(LABEL_HOSTNAME if hostname_only or rng.chance(2, 3) else LABEL_ZONE, match_labels={"app": rng.pick(apps)},
                                         namespace_selector=list(rng.pick(selectors))))
        # required AFFINITY terms: the incoming pod's own terms, resolved through the namespace lister (plugin.go:144-157)
        aterms = [PodAffinityTerm(t.topology_key, dict(t.match_labels), list(t.match_expressions), tuple(t.namespaces), t.namespace_selector)
                  for t in getattr(group[0], "affinity", [])]
        for t in aterms:
            if rng.chance(1, 2):
                t.namespace_selector = list(rng.pick(selectors))
                t.namespaces = tuple(rng.sample(names, rng.below(3))) if rng.chance(1, 2) else ()
            elif rng.chance(1, 3):
                t.namespaces = tuple(rng.sample(names, 1 + rng.below(2)))
        for p in group:
            p.namespace = ns
            p.anti_affinity = [PodAffinityTerm(t.topology_key, dict(t.match_labels), list(t.match_expressions), tuple(t.namespaces),
                                               None if t.namespace_selector is None else list(t.namespace_selector)) for t in terms]
            if aterms:
                p.affinity = [PodAffinityTerm(t.topology_key, dict(t.match_labels), list(t.match_expressions), tuple(t.namespaces),
                                              None if t.namespace_selector is None else list(t.namespace_selector)) for t in aterms]
    return table


# ---------------------------------------------------------------------------------------------
# required pod affinity (InterPodAffinity, V/.../interpodaffinity/filtering.go:234-272,382-409)
# ---------------------------------------------------------------------------------------------
def add_random_pod_affinity(seed: int, pods: Sequence[Pod], frac: float = 0.5, keys: Sequence[str] = (LABEL_HOSTNAME, LABEL_ZONE, "rack"),
                            apps: Sequence[str] = ("app0", "app1", "app2", "app3")) -> int:
    """Decorates `pods` (same spec -> same terms, so equivalence classes survive) with 1-2 REQUIRED pod-affinity terms: towards
    the pod's own app label (the self-affine series with its first-pod exception), towards another app (only nodes / zones
    that already hold such a pod qualify), sometimes through a selector nobody matches.  Returns the number of decorated specs."""
    import random
    rng = random.Random(0xAFF1 + seed)
    plan = {}
    for p in pods:
        k = p.spec_key()
        if k not in plan:
            terms = []
            if rng.random() < frac:
                for _ in range(1 if rng.random() < 0.7 else 2):
                    own = p.labels.get("app", apps[0])
                    r = rng.random()
                    target = own if r < 0.5 else (rng.choice(list(apps)) if r < 0.9 else "nobody")
                    terms.append(PodAffinityTerm(rng.choice(list(keys)), match_labels={"app": target}))
            plan[k] = terms
        p.affinity = list(plan[k])
    return sum(1 for v in plan.values() if v)
