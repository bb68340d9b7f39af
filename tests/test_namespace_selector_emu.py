"""namespaceSelector of required anti-affinity terms (AffinityTerm.Matches, V/kube-scheduler/framework/types.go:390-395;
resolved by InterPodAffinity.PreFilter through the namespace lister, interpodaffinity/plugin.go:144-169).  The encoder
resolves the selectors into namespace sets, so the kernels are unchanged: product encoder + kernels under the wave emulator
against the object-level oracle through TrySchedulePods (f1), the removal loop (f4) and the estimator."""
import pytest

import test_kernels_emu_fuzz as kf
import test_removal_emu as rm
import test_sched_emu as se
from harness import SchedCase, assert_matches_oracle, encode, run_emu, run_oracle, sched_emu, sched_encode
from kubernetes_autoscaler_amd import _abi, workloads
from kubernetes_autoscaler_amd.objects import (LABEL_HOSTNAME, NodeInfo, PodAffinityTerm, Requirement, build_test_node,
                                               build_test_pod, namespaces)
from oracle_driver import OracleScenario


def _pod(name, ns, app, *terms):
    p = build_test_pod(name, 100, 0)
    p.namespace, p.labels, p.anti_affinity = ns, {"app": app}, list(terms)
    return p


def _term(app, selector=None, nss=()):
    return PodAffinityTerm(LABEL_HOSTNAME, match_labels={"app": app}, namespaces=nss, namespace_selector=selector)


def test_selector_picks_namespaces_by_label():
    table = {"default": {}, "prod": {"tier": "prod"}, "dev": {"tier": "dev"}}
    nodes = [NodeInfo(build_test_node(f"n{i}", 1000, 1 << 30)) for i in range(3)]
    for info in nodes:
        info.node.labels[LABEL_HOSTNAME] = info.node.name
    nodes[0].pods.append(_pod("web-prod", "prod", "web"))
    nodes[1].pods.append(_pod("web-dev", "dev", "web"))
    nodes[2].pods.append(_pod("guard", "default", "guard", _term("db", [Requirement("tier", "In", ["dev"])])))   # keeps dev's db pods away
    pods = [
        _pod("a", "default", "x", _term("web", [Requirement("tier", "In", ["prod"])])),      # avoids web in prod only: n1 (after n0)
        _pod("b", "default", "x", _term("web", [])),                                         # every namespace: n2
        _pod("c", "default", "x", _term("web")),                                             # own namespace only: n0 is fine
        _pod("d", "dev", "db"),                                                              # guard's term selects dev: not n2
        _pod("e", "prod", "db"),                                                             # prod is not selected: n2 is fine
        _pod("f", "default", "x", _term("web", [Requirement("tier", "In", ["dev"])], ("prod",))),   # listed OR selected: n2 only
    ]
    with namespaces(table):
        sc = SchedCase(nodes=nodes, pods=pods, last_index=2)   # every search starts at n0 while lastIndex stays 2
        node_out, _, n = se.check(sc, "namespace selector")
    assert n == 6
    assert list(node_out[:2]) == [1, 2]
    assert node_out[3] != 2 and node_out[5] == 2


def test_unlisted_namespace_is_delegated_and_the_oracle_keeps_the_reference_asymmetry():
    """A namespace the lister does not know: an arriving pod's selector cannot list it (plugin.go:148-154), a resident pod's
    selector sees it as unlabelled (plugin.go:161-169) - DoesNotExist then matches.  The encoder hands the case back."""
    table = {"default": {}}
    sel = [Requirement("tier", "DoesNotExist", [])]

    def node(pods):
        info = NodeInfo(build_test_node("n0", 1000, 1 << 30), pods)
        info.node.labels[LABEL_HOSTNAME] = "n0"
        return info
    with namespaces(table):
        # resident pod owns the term, arriving pod lives in the unlisted namespace: blocked
        s = OracleScenario()
        a = s.add_existing(node([_pod("res", "default", "r", _term("x", sel))]))
        ok, plugin, _ = s.run_filters_on_node(a, _pod("arr", "ghost", "x"))
        assert not ok and plugin == "InterPodAffinity"
        s.close()
        # arriving pod owns the term, resident pod lives in the unlisted namespace: not blocked
        s = OracleScenario()
        a = s.add_existing(node([_pod("res", "ghost", "x")]))
        ok, _, _ = s.run_filters_on_node(a, _pod("arr", "default", "r", _term("x", sel)))
        assert ok
        s.close()
        enc, _ = sched_encode(SchedCase(nodes=[node([_pod("res", "ghost", "x")])], pods=[_pod("arr", "default", "r", _term("x", sel))]))
        assert enc.pegs.flags[0] & _abi.PEG_UNSUPPORTED
        enc.close()


@pytest.mark.parametrize("seed", range(300))
def test_fuzz_try_schedule_pods(seed):
    w = workloads.fuzz_pending(seed, max_nodes=24, max_pods=60)
    table = workloads.add_random_namespace_selectors(seed, list(w.pods) + [p for info in w.nodes for p in info.pods], hostname_only=True)
    with namespaces(table):
        se.check(se.case_of(w), w.name)


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_try_schedule_pods_with_domain_rules(seed):
    w = workloads.fuzz_pending_domains(seed, max_nodes=24, max_pods=50)
    table = workloads.add_random_namespace_selectors(seed, list(w.pods) + [p for info in w.nodes for p in info.pods])
    with namespaces(table):
        se.check(se.case_of(w), w.name)


@pytest.mark.parametrize("seed", range(200))
def test_fuzz_removals(seed):
    w = workloads.fuzz_removals(seed)
    table = workloads.add_random_namespace_selectors(seed, [p for info in w.nodes for p in info.pods], hostname_only=True)
    with namespaces(table):
        rm.check(rm.case_of(w), w.name)


@pytest.mark.parametrize("seed", range(150))
def test_fuzz_estimate_template_mode(seed):
    w = workloads.fuzz(8000 + seed)
    pods = [p for pg in w.pegs for p in pg.pods] + [p for g in w.groups for p in g.template.pods] + [p for info in w.existing for p in info.pods]
    table = workloads.add_random_namespace_selectors(seed, pods)
    with namespaces(table):
        sc = kf.scenario_of(w)
        res, _ = run_emu(encode(sc))
        assert_matches_oracle(res, run_oracle(sc), f"seed {seed}")
