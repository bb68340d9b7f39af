#!/bin/bash
# Phase profile of the register packer (s_memtime ticks per phase, mean over the groups) on the bench workload:
#   make -C kubernetes_autoscaler_amd/csrc OUT=../libcasim_prof.so OBJDIR=../../build/obj_prof EXTRA=-DCASIM_PACK_PROF
# usage: tools/prof_pack.sh [config=C2] [batch=1024]
export CASIM_LIB_PATH=$PWD/kubernetes_autoscaler_amd/libcasim_prof.so CASIM_PACK_PROF_DUMP=1
python - "${1:-C2}" "${2:-1024}" <<'PY' 2>&1 | grep -vE "^\s*$" | tail -40
import sys
sys.path.insert(0, ".")
import numpy as np
import kubernetes_autoscaler_amd as kaa
from kubernetes_autoscaler_amd import workloads
from kubernetes_autoscaler_amd.tables import TableSet
import bench
name, batch = sys.argv[1], int(sys.argv[2])
ctx = kaa.Context(0)
ts = bench.simulation_tables(workloads.CONFIGS[name], range(8), kaa.Encoder, TableSet).tile(batch // 8)
pegs, groups = ts.structs()
prob = kaa.Problem(ctx, pegs, groups)
prob.run()
r = prob.fetch()
print("info", prob.info(), file=sys.stderr)
na = np.asarray(r.nodes_added)
print("workload", name, "groups", len(na), file=sys.stderr)
print("nodes_added: mean %.1f max %d hist/8 %s" % (na.mean(), na.max(), np.bincount(np.minimum(na, 64) // 8).tolist()), file=sys.stderr)
d = np.diff(np.asarray(r.offsets))
print("pegs per group: mean %.1f max %d" % (d.mean(), d.max()), file=sys.stderr)
pl = np.asarray(r.placed)
print("pegs with placed>0: %.3f" % ((pl > 0).mean()), file=sys.stderr)
PY
