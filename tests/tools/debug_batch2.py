import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kubernetes_autoscaler_amd as kaa
from test_gpu_round2 import _scenario
from test_reasons_emu import emu_reasons
from harness import encode_batch, run_emu_tables, run_gpu_tables, run_emu_feasibility
ctx = kaa.Context(0)
seed = 17
scs = [_scenario(1000 * seed + k) for k in range(1 + seed % 6)]
enc, ts, bases = encode_batch(scs)
pegs, groups = ts.structs()
gb = ctx.feasibility(pegs, groups)
print("gpu bits", [hex(int(x)) for x in gb[:, 0]])
import ctypes as C
from harness import emu_lib
from kubernetes_autoscaler_amd import _abi
L = emu_lib()
eb = np.zeros((ts.n_groups, 1), np.uint64)
L.emu_feasibility(C.byref(pegs), C.byref(groups), eb.ctypes.data_as(_abi.u64p))
print("emu bits", [hex(int(x)) for x in eb[:, 0]])
gr = ctx.feasibility_reasons(pegs, groups, enc.port_block)
er = emu_reasons(ts, enc.port_block)
print("gpu reasons g7", list(gr[7])); print("emu reasons g7", list(er[7]))
i, g = 7, 17
print("peg17", {k: (v[g].tolist() if v is not None else None) for k, v in ts.pegs.items()})
print("group7", {k: (v[i].tolist() if v is not None else None) for k, v in ts.groups.items()})
for rep in range(3):
    gb2 = ctx.feasibility(pegs, groups)
    print("rep", rep, [hex(int(x)) for x in gb2[:, 0]])
