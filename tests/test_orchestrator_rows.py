"""The reference's orchestrator tests as known answers for the CHAIN SchedulablePodGroups -> Estimate per node group -> expander input
-> status sets: core/scaleup/orchestrator/orchestrator_test.go rows TestScaleUpOK, TestMixedScaleUp,
TestZeroOrMaxNodeScaling (3), TestScaleUpCapToMaxTotalNodesLimit (2), the three GPU-pool tests (taint + toleration + extended resource
inside a scale-up: the closest thing the reference holds to a known answer for config C2), TestAllOrNothing, TestBinpackingLimiter,
TestScaleUpNoHelp.  CPU tier: (1) the oracle, per call AND through orc_scale_up_simulation — the native loop bench.py's cpu_baseline
times —, (2) the product kernels under the wave emulator.  The MI355X runs the same rows in tests/test_gpu_round4.py."""
import pytest

from harness import encode, run_emu, run_oracle
from oracle_driver import OracleScenario
from orchestrator_rows import ROWS, Row, per_group_of_batch, per_group_of_oracle


@pytest.mark.parametrize("row", ROWS, ids=[r["name"] for r in ROWS])
def test_oracle_reproduces_the_reference_row_per_call_and_as_one_native_simulation(row):
    r = Row(row)
    sc = r.scenario()
    per_call = run_oracle(sc)
    r.check(r.decide(per_group_of_oracle(per_call)), "oracle, per call:")
    # the same through ONE native call (orc_scale_up_simulation): what cpu_baseline times must be the reference's answer too
    s = OracleScenario(lanes=sc.lanes)
    for info in sc.existing:
        s.add_existing(info)
    tmpls = [s.node(g.template) for g in sc.groups]
    native = s.prepare_simulation(tmpls, sc.pegs, [g.max_nodes for g in sc.groups], [g.last_index for g in sc.groups])
    out, _ = native(True)
    s.close()
    assert [(list(e.order), list(e.placed), e.node_count, e.pods_scheduled, ids) for e, ids in out] == \
           [(list(e.order), list(e.placed), e.node_count, e.pods_scheduled, ids) for e, ids in per_call]
    r.check(r.decide(per_group_of_oracle(out)), "oracle, orc_scale_up_simulation:")


@pytest.mark.parametrize("generic", [False, True], ids=["register-packer", "int64-packer"])
@pytest.mark.parametrize("row", ROWS, ids=[r["name"] for r in ROWS])
def test_product_kernels_reproduce_the_reference_row_under_the_emulator(row, generic):
    r = Row(row)
    enc = encode(r.scenario())
    res, _ = run_emu(enc, generic=generic)
    enc.close()
    assert all(int(s) == 0 for s in res.status)
    r.check(r.decide(per_group_of_batch(res)), "emulator:")


def _sanitized_template():
    import json, os
    from kubernetes_autoscaler_amd.objects import Node, NodeInfo, Pod, Taint
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json")))["sanitized_node_info"]
    t = G["template"]
    cap = {"cpu": t["cpu"], "memory": t["mem"], "pods": t["pods_capacity"]}
    node = Node(name=t["name"], labels=dict(t["labels"]), taints=[Taint(*x) for x in t["taints"]], allocatable=dict(cap), capacity=dict(cap))
    return G, NodeInfo(node, [Pod(name=n, requests={"cpu": c, "memory": m}) for n, c, m in t["pods"]])


@pytest.mark.parametrize("device", ["oracle", "emulator-register", "emulator-int64"])
def test_a_simulated_node_inherits_what_the_reference_says_it_inherits(device):
    """simulator/node_info_utils_test.go:395-428 (TestSanitizedNodeInfo) as seen through Estimate (row a8): every node the estimate adds
    carries ALL the template's taints (nothing is sanitized a second time — not even ToBeDeleted), its labels, and the template's pods
    (their requests and their pod slots)."""
    from harness import GroupSpec, Scenario
    from kubernetes_autoscaler_amd.objects import Pod, PodEquivalenceGroup, Toleration
    G, tmpl = _sanitized_template()
    exp = G["expect"]
    everything = [Toleration(operator="Exists")]
    two_of_three = [Toleration(key="startup-taint", operator="Exists"), Toleration(key="a", operator="Equal", value="b")]
    free = 1000 - exp["requested_cpu"]
    pegs = [PodEquivalenceGroup(pods=[Pod(name="untolerating", requests={"cpu": 10, "memory": 0})] * 3),
            PodEquivalenceGroup(pods=[Pod(name="misses-the-to-be-deleted-taint", requests={"cpu": 10, "memory": 0}, tolerations=two_of_three)] * 3),
            PodEquivalenceGroup(pods=[Pod(name="exactly-what-is-left", requests={"cpu": free, "memory": 0}, tolerations=everything)] * 2),
            PodEquivalenceGroup(pods=[Pod(name="one-milli-too-much", requests={"cpu": free + 1, "memory": 0}, tolerations=everything)] * 2),
            PodEquivalenceGroup(pods=[Pod(name="slots-only", requests={"cpu": 0, "memory": 0}, tolerations=everything)] * 200)]
    want = {0: (0, 0), 1: (0, 0), 2: (2, 2), 3: (0, 0), 4: (-(-200 // (100 - exp["pod_count"])), 200)}   # PEG -> (nodes, pods): 98 free slots per node
    for k, (nodes, pods) in want.items():
        sc = Scenario(pegs=[pegs[k]], groups=[GroupSpec(tmpl, 0, 0, [0])], existing=[])
        if device == "oracle":
            est, _ = run_oracle(sc)[0]
            got = (est.node_count, est.pods_scheduled)
        else:
            enc = encode(sc)
            res, _ = run_emu(enc, generic=device.endswith("int64"))
            enc.close()
            got = (int(res.node_count[0]), int(res.pods_scheduled[0]))
        assert got == (nodes, pods), (pegs[k].pods[0].name, got)
