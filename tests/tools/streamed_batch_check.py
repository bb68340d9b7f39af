"""casim_options.n_streams on the hardware against ONE unstreamed problem over the same batch, in a process of its own: the
packed keys land in a torch tensor through the device-pointer form (what bench.py all-reduces at N > 1; the context runs on a
torch stream, the internal streams join into it), and torch must be imported before libcasim so that both use one HIP
runtime.  Prints one JSON line."""
import json
import os
import sys

import torch  # noqa: F401  (first: see above)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kubernetes_autoscaler_amd as kaa  # noqa: E402
from kubernetes_autoscaler_amd import _abi  # noqa: E402
from harness import assert_matches_oracle, encode_batch, run_gpu_tables  # noqa: E402
from test_gpu_round2 import _oracle_of, _scenario  # noqa: E402

ctx = kaa.Context(0)
scs = [_scenario(900 + k) for k in range(11)]
enc, ts, bases = encode_batch(scs)
kinds = [_abi.EXPANDER_LEAST_NODES, _abi.EXPANDER_LEAST_WASTE]
whole, wexp = run_gpu_tables(ts, ctx, kinds=kinds)
assert_matches_oracle(whole, _oracle_of(scs, bases), "whole batch")
checked = []
for k in (1, 3, 4, 16):
    stream = torch.cuda.Stream(device=0)
    with torch.cuda.stream(stream), kaa.StreamedBatch(0, ts, n_streams=k, stream=stream.cuda_stream) as sb:
        assert (2 <= sb.parts <= min(k, ts.n_sims)) if k > 1 else sb.parts == 1   # (as many lanes as the runtime has concurrent hardware queues for)
        keys = torch.full((ts.n_sims,), 0x7FFFFFFFFFFFFFFF, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        for _ in range(3):   # resident: several passes, same answer; the device-pointer form joins the internal streams into `stream`
            sb.run()
            sb.best_option_sims(kinds, dev_packed_ptr=keys.data_ptr(), fetch=False)
            doubled = keys * 2          # torch work on the context's stream: must see this step's keys
        exp = sb.best_option_sims(kinds, dev_packed_ptr=keys.data_ptr())
        res = sb.fetch()
        torch.cuda.synchronize()
        for name in ("node_count", "pods_scheduled", "nodes_added", "limiter_nodes", "last_index_out", "status", "req_cpu_sum", "req_mem_sum"):
            assert list(getattr(res, name)) == list(getattr(whole, name)), (k, name)
        assert list(res.offsets) == list(whole.offsets) and list(res.order) == list(whole.order) and list(res.placed) == list(whole.placed)
        assert list(exp["best"]) == list(wexp["best"]) and list(exp["n_best"]) == list(wexp["n_best"])
        assert list(exp["best_set"]) == list(wexp["best_set"]) and exp["keys"].tolist() == wexp["keys"].tolist()
        assert list(exp["packed"]) == list(wexp["packed"]) == keys.cpu().numpy().tolist()
        assert (doubled.cpu().numpy() == 2 * keys.cpu().numpy()).all()
        checked.append(k)
enc.close()
ctx.close()
print(json.dumps({"streams_checked": checked, "simulations": ts.n_sims, "groups": ts.n_groups}))
