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
