"""Required pod affinity (a18 complete): InterPodAffinity's getIncomingAffinityAntiAffinityCounts / satisfyPodAffinity
(V/.../interpodaffinity/filtering.go:234-272,382-409) as domain rules of kind 2 in K_sched (TrySchedulePods, the removal loop)
and K_est (Estimate on the snapshot), product kernels under the wave emulator vs the object-level oracle."""
import pytest

from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.objects import LABEL_HOSTNAME, LABEL_ZONE, GiB, MiB, NodeInfo, Pod, PodAffinityTerm, PodEquivalenceGroup
from kubernetes_autoscaler_amd.workloads import _node, add_random_pod_affinity
from harness import (EmuContext, GroupSpec, RemovalCase, Scenario, SchedCase, assert_cluster_estimate_matches, assert_removal_matches,
                     assert_sched_matches, cluster_estimate_emu, encode, removal_device, removal_oracle, run_emu, run_oracle, sched_emu,
                     sched_oracle)


def sched_case(w):
    return SchedCase(nodes=w.nodes, pods=w.pods, hints=w.hints, acceptable=w.acceptable, break_on_failure=w.break_on_failure, last_index=w.last_index)


@pytest.mark.parametrize("seed", range(300))
def test_try_schedule_pods_with_required_pod_affinity(seed):
    w = workloads.fuzz_pending_domains(7000 + seed)
    n = add_random_pod_affinity(seed, w.pods + [p for info in w.nodes for p in info.pods], frac=0.6)
    case = sched_case(w)
    want = sched_oracle(case)
    for lds in (0, 4096):
        got = sched_emu(case, lds_budget=lds)
        assert_sched_matches(got, want, f"{w.name} lds={lds} ({n} specs with affinity)")


@pytest.mark.parametrize("seed", range(200))
def test_removal_loop_with_required_pod_affinity(seed):
    w = workloads.fuzz_removals_domains(7000 + seed)
    add_random_pod_affinity(seed, [p for info in w.nodes for p in info.pods], frac=0.5, apps=("app0", "app1", "app2"))
    case = RemovalCase(nodes=w.nodes, candidates=w.candidates, destination=w.destination, hints=w.hints, persist=w.persist,
                       max_removable=w.max_removable, last_index=w.last_index)
    want = removal_oracle(case)
    for lds in (0, 4096):
        got = removal_device(case, EmuContext(lds))
        assert_removal_matches(got, want, f"{w.name} lds={lds}")


@pytest.mark.parametrize("seed", range(300))
def test_estimate_on_the_snapshot_with_required_pod_affinity(seed):
    w = workloads.fuzz_estimate_domains(7000 + seed)
    add_random_pod_affinity(seed, [pg.pods[0] for pg in w.pegs] + [p for info in w.existing for p in info.pods] + list(w.groups[0].template.pods),
                            frac=0.6, apps=("app0", "app1", "app2"))
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)
    if cluster_estimate_emu(sc)[0] == 1:
        pytest.skip("delegated (hostname anti-affinity next to an unnamed node)")
    est, ids = run_oracle(sc)[0]
    for lds in (0, 64):
        assert_cluster_estimate_matches(cluster_estimate_emu(sc, 0, lds_budget=lds), est, ids, f"{w.name} lds={lds}")


def test_template_mode_keeps_the_hostname_series_in_the_batch():
    """Until round 4 the one affinity verdict that changes while an Estimate runs — the self-affine series on a HOSTNAME key that got in by
    the first-pod exception: the rest of the series has to join the first pod's node — made its groups CASIM_NG_UNSUPPORTED.  Now it is a
    node bit of NEED polarity (casim_pegs.excl_polarity, ABI 8) that the PEG marks itself, and the packer walks the record twice (first pod
    with the bit waived, the rest with it in force).  The same series on a zone key was and is a no-op for the whole Estimate."""
    from harness import assert_matches_oracle
    tmpl = NodeInfo(_node("t", 4000, 8 * GiB, 110, {LABEL_ZONE: "z0"}))
    plain = Pod(name="plain", labels={"app": "x"}, requests={"cpu": 100, "memory": 64 * MiB})
    for key in (LABEL_HOSTNAME, LABEL_ZONE):
        for n_web, want_web, want_nodes_added in ((5, 5, 1), (8, 8, 2), (20, 8, 2)) if key == LABEL_HOSTNAME else ((5, 5, 1), (20, 20, 3)):
            web = Pod(name="web", labels={"app": "web"}, requests={"cpu": 500, "memory": 256 * MiB}, affinity=[PodAffinityTerm(key, match_labels={"app": "web"})])
            sc = Scenario(pegs=[PodEquivalenceGroup([web] * n_web), PodEquivalenceGroup([plain] * 3)], groups=[GroupSpec(tmpl, 0, 0, None), GroupSpec(tmpl, 0, 0, [1])])
            enc = encode(sc)
            assert not (enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED) and not (enc.pegs.flags[1] & _abi.PEG_UNSUPPORTED)
            assert bool(enc.pegs.excl_polarity) == (key == LABEL_HOSTNAME)
            oracle = run_oracle(sc)
            for generic in (False, True):
                res, _ = run_emu(enc, generic=generic)
                assert [int(x) for x in res.status] == [_abi.NG_OK, _abi.NG_OK]
                assert_matches_oracle(res, oracle, f"self-affine series on {key}, {n_web} pods, generic={generic}")
                # hostname: ONE node takes what fits (8 pods by cpu); the node the reference then opens in vain for the ninth (:257-263) is the
                # one the three plain pods behind it land on
                order, placed = res.group(0)
                assert (list(order), int(placed[0]), int(placed[1]), int(res.nodes_added[0])) == ([0, 1], want_web, 3, want_nodes_added)
            enc.close()


def test_hostname_affinity_towards_a_partner_of_the_batch():
    """app=web has to sit next to app=cache (hostname key, no exception: web does not match its own term): with the caller's lists web waits
    until cache has opened nodes, fills up next to it and what finds no room there stays pending; SchedulablePodGroups (device-derived
    lists) never lists it — its sample pod fails on a fresh node."""
    from harness import assert_matches_oracle
    tmpl = NodeInfo(_node("t", 4000, 8 * GiB, 110, {LABEL_ZONE: "z0"}))
    cache = Pod(name="cache", labels={"app": "cache"}, requests={"cpu": 1500, "memory": 256 * MiB})
    web = Pod(name="web", labels={"app": "web"}, requests={"cpu": 400, "memory": 256 * MiB}, affinity=[PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": "cache"})])
    big = Pod(name="big", labels={"app": "big"}, requests={"cpu": 3000, "memory": 256 * MiB})
    for n_web in (2, 6, 30):
        for lists in ([[0, 1, 2], [1, 0], [1, 2]], None):
            groups = [GroupSpec(tmpl, 0, li, None if lists is None else lists[k]) for k, li in enumerate((0, 1, 0))]
            sc = Scenario(pegs=[PodEquivalenceGroup([cache] * 3), PodEquivalenceGroup([web] * n_web), PodEquivalenceGroup([big] * 2)], groups=groups,
                          device_csr=lists is None)
            enc = encode(sc)
            assert not any(enc.pegs.flags[i] & _abi.PEG_UNSUPPORTED for i in range(3))
            oracle = run_oracle(sc)
            for generic in (False, True, 2):
                res, _ = run_emu(enc, generic=generic)
                assert_matches_oracle(res, oracle, f"partner of the batch, {n_web} web pods, lists={lists}, generic={generic}")
            if lists is not None:
                assert int(res.pods_scheduled[0]) > 3      # web found its partner
            enc.close()


def test_self_affine_series_on_hostname_packs_one_node():
    """The first pod passes by the exception (no matching pod anywhere, it matches its own term), the others must join it:
    a hostname-level self-affine PEG fills ONE node and the rest stays pending — oracle and device agree."""
    nodes = [NodeInfo(_node(f"n{i}", 1000, 4 * GiB, 110, {LABEL_ZONE: "z0"})) for i in range(3)]
    pods = [Pod(name=f"p{i}", labels={"app": "db"}, requests={"cpu": 300, "memory": 64 * MiB}, controller_uid="db",
                affinity=[PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": "db"})]) for i in range(5)]
    case = SchedCase(nodes=nodes, pods=pods)
    want = sched_oracle(case)
    assert list(want[0]) == [1, 1, 1, -1, -1]      # (the walk starts behind lastIndex = 0)
    assert_sched_matches(sched_emu(case), want, "self-affine hostname series")
    # towards another app that runs on node 2 only
    nodes[2].pods.append(Pod(name="cache", labels={"app": "cache"}, requests={"cpu": 100, "memory": 64 * MiB}))
    pods2 = [Pod(name=f"q{i}", labels={"app": "web"}, requests={"cpu": 300, "memory": 64 * MiB}, controller_uid="web",
                 affinity=[PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": "cache"})]) for i in range(4)]
    case2 = SchedCase(nodes=nodes, pods=pods2)
    want2 = sched_oracle(case2)
    assert list(want2[0]) == [2, 2, 2, -1]
    assert_sched_matches(sched_emu(case2), want2, "affinity to a running pod")


# ---- namespaceSelector on required AFFINITY terms (VERDICT r2 missing #4, second half) -----------------------------------------
# The term is the incoming pod's: a non-empty selector selects among the namespaces the lister knows (plugin.go:144-157), an empty
# one selects every namespace; the encoder resolves it at finalize from casim_enc_add_namespace / _namespace_add_label.
@pytest.mark.parametrize("seed", range(200))
def test_try_schedule_pods_with_affinity_namespace_selectors(seed):
    from kubernetes_autoscaler_amd.objects import namespaces
    w = workloads.fuzz_pending_domains(7600 + seed)
    everybody = w.pods + [p for info in w.nodes for p in info.pods]
    add_random_pod_affinity(seed, everybody, frac=0.6)
    table = workloads.add_random_namespace_selectors(seed, everybody)
    n_sel = sum(1 for p in everybody for t in p.affinity if t.namespace_selector is not None)
    with namespaces(table):
        case = sched_case(w)
        want = sched_oracle(case)
        got = sched_emu(case)
        assert got[0] == 0, f"affinity terms with a namespaceSelector must not be delegated any more (status {got[0]})"
        assert_sched_matches(got, want, f"{w.name} ({n_sel} affinity terms with a namespaceSelector)")


@pytest.mark.parametrize("seed", range(150))
def test_estimate_on_the_snapshot_with_affinity_namespace_selectors(seed):
    from kubernetes_autoscaler_amd.objects import namespaces
    w = workloads.fuzz_estimate_domains(7600 + seed)
    everybody = [pg.pods[0] for pg in w.pegs] + [p for info in w.existing for p in info.pods] + list(w.groups[0].template.pods)
    add_random_pod_affinity(seed, everybody, frac=0.6, apps=("app0", "app1", "app2"))
    # (the PEGs' other pods are copies of the exemplar: same namespace and terms)
    table = workloads.add_random_namespace_selectors(seed, [p for pg in w.pegs for p in pg.pods] + [p for info in w.existing for p in info.pods] + list(w.groups[0].template.pods))
    with namespaces(table):
        sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)
        if cluster_estimate_emu(sc)[0] == 1:
            pytest.skip("delegated (hostname anti-affinity next to an unnamed node, or a namespaceSelector next to an unlisted namespace)")
        est, ids = run_oracle(sc)[0]
        assert_cluster_estimate_matches(cluster_estimate_emu(sc, 0), est, ids, w.name)


def test_an_affinity_selector_only_sees_listed_namespaces():
    """team=a selects ns-a; a partner in an UNLISTED namespace with the same labels is not selected (the lister does not know it)"""
    from kubernetes_autoscaler_amd.objects import Requirement, namespaces
    nodes = [NodeInfo(_node(f"n{i}", 2000, 4 * GiB, 110, {LABEL_ZONE: f"z{i}"})) for i in range(3)]
    nodes[1].pods.append(Pod(name="partner-listed", namespace="ns-a", labels={"app": "db"}, requests={"cpu": 100, "memory": 64 * MiB}))
    nodes[2].pods.append(Pod(name="partner-unlisted", namespace="ns-ghost", labels={"app": "db"}, requests={"cpu": 100, "memory": 64 * MiB}))
    web = [Pod(name=f"web{i}", namespace="default", labels={"app": "web"}, requests={"cpu": 300, "memory": 64 * MiB}, controller_uid="web",
               affinity=[PodAffinityTerm(LABEL_ZONE, match_labels={"app": "db"}, namespace_selector=[Requirement("team", "In", ["a"])])]) for i in range(3)]
    with namespaces({"default": {"team": "core"}, "ns-a": {"team": "a"}}):
        case = SchedCase(nodes=nodes, pods=web)
        want = sched_oracle(case)
        assert list(want[0]) == [1, 1, 1]                     # only the zone of the listed partner qualifies
        assert_sched_matches(sched_emu(case), want, "affinity namespaceSelector, listed namespaces only")


# ---- required pod affinity in TEMPLATE mode when the verdict is static (VERDICT r2 missing #4, first half: the common shape) ----------------
# Non-hostname keys and no possible partner inside the batch: every (PEG, group) verdict is fixed by the existing cluster and the
# template's preloaded pods, and the group runs in the template-mode packer (casim_estimate_batch) instead of the snapshot path.
def _static_affinity_workload(seed):
    from kubernetes_autoscaler_amd.workloads import SplitMix64
    rng = SplitMix64(0xAFF17000 + seed)
    w = workloads.fuzz(9300 + seed, max_groups=5, max_pegs=10, rich=seed % 2 == 0)
    zones = ["zone-0", "zone-1", "zone-9"]
    # partners live in the existing cluster only: cache / db pods on some existing nodes (their zone is their node's), and sometimes a
    # DaemonSet-like cache pod preloaded on a template
    for k, info in enumerate(w.existing):
        for _ in range(rng.below(3)):
            info.pods.append(Pod(name=f"part-{seed}-{k}", namespace=rng.pick(["default", "infra"]), labels={"app": rng.pick(["cache", "db"])},
                                 requests={"cpu": 50, "memory": 64 * MiB}))
    if rng.chance(1, 3) and w.groups:
        w.groups[rng.below(len(w.groups))].template.pods.append(Pod(name=f"ds-cache-{seed}", namespace="default", labels={"app": "cache"},
                                                                    requests={"cpu": 50, "memory": 32 * MiB}))
    n = 0
    for pg in w.pegs:
        if rng.chance(1, 2):
            terms = [PodAffinityTerm(LABEL_ZONE, match_labels={"app": rng.pick(["cache", "db"])},
                                     namespaces=tuple(rng.sample(["default", "infra"], 1 + rng.below(2))) if rng.chance(1, 2) else ())]
            if rng.chance(1, 4):
                terms.append(PodAffinityTerm("pool", match_labels={"app": terms[0].match_labels["app"]}))   # a key most templates lack
            for p in pg.pods:
                p.affinity = [PodAffinityTerm(t.topology_key, dict(t.match_labels), list(t.match_expressions), tuple(t.namespaces), None) for t in terms]
            n += 1
    return w, n


@pytest.mark.parametrize("device_csr", [False, True])
@pytest.mark.parametrize("seed", range(150))
def test_template_mode_packs_groups_whose_affinity_verdict_is_static(seed, device_csr):
    w, n_aff = _static_affinity_workload(seed)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                  device_csr=device_csr)
    enc = encode(sc)
    assert not any(enc.pegs.flags[i] & _abi.PEG_UNSUPPORTED for i in range(enc.pegs.n_pegs) if w.pegs[i].pods and w.pegs[i].pods[0].affinity and
                   not any(t.topology_key == LABEL_HOSTNAME for t in w.pegs[i].pods[0].affinity)), "static affinity must not be delegated"
    for generic in (False, True):
        res, _ = run_emu(enc, generic=generic)
        if any(int(s) != 0 for s in res.status):
            pytest.skip("another predicate of the fuzz family is outside the template subset")
        from harness import assert_matches_oracle
        assert_matches_oracle(res, run_oracle(sc), f"static affinity seed {seed} ({n_aff} PEGs) generic={generic}")
    enc.close()


def test_static_affinity_changes_the_answer():
    """near the cache in zone-0: the zone-0 template takes the pods, the zone-1 template none — without the term both would"""
    cache = Pod(name="cache", labels={"app": "cache"}, requests={"cpu": 100, "memory": 64 * MiB})
    old = NodeInfo(_node("old", 1000, 2 * GiB, 10, {LABEL_ZONE: "zone-0"}), [cache])
    t0 = NodeInfo(_node("t0", 4000, 8 * GiB, 110, {LABEL_ZONE: "zone-0"}))
    t1 = NodeInfo(_node("t1", 4000, 8 * GiB, 110, {LABEL_ZONE: "zone-1"}))
    web = [Pod(name=f"web{i}", labels={"app": "web"}, requests={"cpu": 500, "memory": 256 * MiB},
               affinity=[PodAffinityTerm(LABEL_ZONE, match_labels={"app": "cache"})]) for i in range(9)]
    sc = Scenario(pegs=[PodEquivalenceGroup(web)], groups=[GroupSpec(t0, 0, 0, None), GroupSpec(t1, 0, 0, None)], existing=[old], device_csr=True)
    enc = encode(sc)
    assert not (enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED)
    res, _ = run_emu(enc)
    assert [int(x) for x in res.status] == [0, 0] and [int(x) for x in res.pods_scheduled] == [9, 0] and int(res.node_count[0]) == 2
    from harness import assert_matches_oracle
    assert_matches_oracle(res, run_oracle(sc), "near the cache")
    enc.close()


# Partners INSIDE the batch, self-affine series and hostname keys: an Estimate only sees the PEGs whose sample pod passes on a fresh
# template node at snapshot time, and counts only grow, so every verdict but one (the hostname series that got in by the first-pod
# exception) is fixed per (PEG, group) — encoder block (4a).
def _batch_affinity_workload(seed, keys=(LABEL_HOSTNAME, LABEL_ZONE, LABEL_ZONE, "pool")):
    from kubernetes_autoscaler_amd.workloads import SplitMix64
    rng = SplitMix64(0xAFF18000 + seed)
    w = workloads.fuzz(9700 + seed, max_groups=5, max_pegs=10, rich=seed % 2 == 0)
    apps = ["app0", "app1", "app2"]
    for pg in w.pegs:   # the batch's own pods carry the labels the terms select
        a = rng.pick(apps)
        for p in pg.pods:
            p.labels = dict(p.labels, app=a)
    for k, info in enumerate(w.existing):
        for _ in range(rng.below(2)):
            info.pods.append(Pod(name=f"part-{seed}-{k}", labels={"app": rng.pick(apps)}, requests={"cpu": 50, "memory": 64 * MiB}))
    if rng.chance(1, 3) and w.groups:
        w.groups[rng.below(len(w.groups))].template.pods.append(Pod(name=f"ds-{seed}", labels={"app": rng.pick(apps)}, requests={"cpu": 50, "memory": 32 * MiB}))
    n = add_random_pod_affinity(seed, [p for pg in w.pegs for p in pg.pods], frac=0.6, keys=keys, apps=apps)
    return w, n


def _dynamic(pg):
    """what may be left to the snapshot path: a hostname term (the partner has to sit on the SAME node; whether a verdict really is
    dynamic depends on the cluster and the batch — the encoder decides, this is the necessary condition)"""
    p = pg.pods[0]
    return bool(p.affinity) and any(t.topology_key == LABEL_HOSTNAME for t in p.affinity)


@pytest.mark.parametrize("seed", range(300))
def test_template_mode_with_partners_inside_the_batch(seed):
    w, n_aff = _batch_affinity_workload(seed)
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes)
    enc = encode(sc)
    for i, pg in enumerate(w.pegs):
        if pg.pods and pg.pods[0].affinity and not _dynamic(pg) and (enc.pegs.flags[i] & _abi.PEG_UNSUPPORTED):
            other = [p for p in pg.pods[:1] if p.spread_constraints or p.unsupported_reason]
            assert other, f"PEG {i}: an affinity verdict that is static was delegated"
    oracle = run_oracle(sc)
    compared = 0
    for generic in (False, True):
        res, _ = run_emu(enc, generic=generic)
        for gi, (est, ids) in enumerate(oracle):
            if int(res.status[gi]) != 0:
                continue   # (a delegated PEG on the group's list)
            order, placed = res.group(gi)
            tag = f"batch affinity seed {seed} group {gi} generic={generic}"
            assert list(order) == [ids[k] for k in est.order], f"{tag}: PEG order"
            assert list(placed) == list(est.placed), f"{tag}: placed per PEG\n got {list(placed)}\n want {list(est.placed)}"
            assert (int(res.node_count[gi]), int(res.pods_scheduled[gi]), int(res.nodes_added[gi]), int(res.last_index_out[gi])) == \
                (est.node_count, est.pods_scheduled, est.nodes_added, est.last_index_out), tag
            compared += 1
    enc.close()
    if not compared:
        pytest.skip("every group holds a delegated PEG")


@pytest.mark.parametrize("device_csr", [False, True])
@pytest.mark.parametrize("seed", range(300))
def test_template_mode_zone_affinity_never_delegates(seed, device_csr):
    """Non-hostname keys only: every verdict is static or a group bit of NEED polarity (the PEG waits for a partner of the batch),
    nothing goes to the snapshot path.  device_csr: the lists come from K_feas, which reads the same bits — SchedulablePodGroups
    drops a PEG whose partner is not there yet; with the caller's lists (every group sees every PEG) the PEG waits its turn."""
    w, n_aff = _batch_affinity_workload(seed, keys=(LABEL_ZONE, LABEL_ZONE, "pool-0", "pool"))
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                  device_csr=device_csr)
    enc = encode(sc)
    for i, pg in enumerate(w.pegs):
        if pg.pods and pg.pods[0].affinity and (enc.pegs.flags[i] & _abi.PEG_UNSUPPORTED):
            assert pg.pods[0].spread_constraints or pg.pods[0].unsupported_reason, f"PEG {i}: zone-level affinity was delegated"
    oracle = run_oracle(sc)
    from harness import assert_matches_oracle
    for generic in (False, True):
        res, _ = run_emu(enc, generic=generic)
        if any(int(x) != 0 for x in res.status):
            pytest.skip("another predicate of the fuzz family is outside the template subset")
        assert_matches_oracle(res, oracle, f"zone affinity seed {seed} ({n_aff} PEGs) generic={generic} device_csr={device_csr}")
    enc.close()


@pytest.mark.parametrize("device_csr", [False, True])
@pytest.mark.parametrize("seed", range(400))
def test_template_mode_hostname_affinity_never_delegates(seed, device_csr):
    """Hostname keys (VERDICT r3 missing #3): partners inside the batch, self-affine series that enter by the first-pod exception, series
    next to host ports / hostname anti-affinity / zone terms, templates that carry the partner — every verdict is a node bit of NEED
    polarity now, nothing goes to the snapshot path; all three packers (register store on int32 and int64 lanes, LDS store), with and
    without tryFastPath."""
    from harness import assert_matches_oracle
    w, n_aff = _batch_affinity_workload(20000 + seed, keys=(LABEL_HOSTNAME, LABEL_HOSTNAME, LABEL_HOSTNAME, LABEL_ZONE))
    fast = seed % 4 == 3
    sc = Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, g.pegs) for g in w.groups], existing=w.existing, lanes=w.lanes,
                  device_csr=device_csr, fastpath=fast)
    enc = encode(sc)
    for i, pg in enumerate(w.pegs):
        if pg.pods and pg.pods[0].affinity and (enc.pegs.flags[i] & _abi.PEG_UNSUPPORTED):
            assert pg.pods[0].spread_constraints or pg.pods[0].unsupported_reason, f"PEG {i}: hostname-level affinity was delegated"
    oracle = run_oracle(sc)
    for generic in (False, True, 2):
        res, _ = run_emu(enc, generic=generic, fastpath=fast)
        if any(int(x) != 0 for x in res.status):
            pytest.skip("another predicate of the fuzz family is outside the template subset")
        assert_matches_oracle(res, oracle, f"hostname affinity seed {seed} ({n_aff} PEGs) generic={generic} device_csr={device_csr} fastpath={fast}")
    enc.close()


@pytest.mark.parametrize("seed", range(20))
def test_hostname_affinity_in_batches_of_simulations(seed):
    """simulations side by side in one table set (one encoder, one polarity row; TableSet carries casim_pegs.excl_polarity through its
    views), fixed-stride lists, stream parts: against the oracle's simulation of each scenario"""
    from harness import assert_matches_oracle, encode_batch, run_emu_streams, run_emu_tables
    import numpy as np
    scs = []
    for k in range(2 + seed % 5):
        w, _ = _batch_affinity_workload(21000 + 13 * seed + k, keys=(LABEL_HOSTNAME, LABEL_HOSTNAME, LABEL_ZONE))
        scs.append(Scenario(pegs=w.pegs, groups=[GroupSpec(g.template, g.max_nodes, g.last_index, None) for g in w.groups], existing=[], lanes=w.lanes, device_csr=True))
    enc, ts, bases = encode_batch(scs)
    want = []
    for sc, (pb, _) in zip(scs, bases):
        want.extend([(est, [pb + i for i in ids]) for est, ids in run_oracle(sc)])
    one, _ = run_emu_tables(ts)
    if any(int(s) != 0 for s in one.status):
        pytest.skip("another predicate of the fuzz family is outside the template subset")
    assert_matches_oracle(one, want, f"hostname affinity, {len(scs)} simulations")
    cut, _, parts = run_emu_streams(ts, 3)
    for f in ("node_count", "pods_scheduled", "nodes_added", "last_index_out", "status", "order", "placed"):
        assert np.array_equal(getattr(one, f), getattr(cut, f)), f
    enc.close()
