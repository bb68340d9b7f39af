"""A batch of independent simulations as sub-batches on HIP streams of their own.

The feasibility / ordering kernels of the scale-up path wait on memory, the packer on instruction issue: run back to back they
leave each other's resource idle, and every launch ends in a tail.  Cut by simulation (TableSet.sim_slice), each part a
casim_problem on its own context (= its own stream and memory pool), the parts overlap: 1.23 -> 1.05 ms per 4096 C2
simulations on one MI355X.  Nothing here is more than bookkeeping over engine.Context / engine.Problem —
it is the shape a Go shim with several contexts on one device has."""
from typing import List, Optional, Sequence

import numpy as np

from .engine import BatchResult, Context, Problem
from .tables import TableSet


class StreamedBatch:
    """`n_streams` casim contexts on `device`, the simulations of `tables` spread evenly over them.

    streams: optional raw hipStream_t handles (ints), one per part — e.g. torch streams, so that torch work can be ordered
    against the kernels; without them every context creates its own stream."""

    def __init__(self, device: int, tables: TableSet, n_streams: int = 4, streams: Optional[Sequence[int]] = None, **problem_kw):
        one = tables if tables.peg_lo is not None else tables.as_one_simulation()
        n = one.n_sims
        k = max(1, min(int(n_streams), n))
        if streams is not None and len(streams) < k:
            raise ValueError("one stream handle per part is needed")
        self.cuts = [(n * i) // k for i in range(k + 1)]
        self.parts: List[TableSet] = [one.sim_slice(self.cuts[i], self.cuts[i + 1]) for i in range(k)] if k > 1 else [one]
        self.peg_base = [int(one.peg_lo[int(one.sim_offsets[c])]) if c < n else one.n_pegs for c in self.cuts[:-1]]
        self.ctxs = [Context(device, stream=(streams[i] if streams is not None else None)) for i in range(k)]
        self.probs = [Problem(c, *p.structs(), **problem_kw) for c, p in zip(self.ctxs, self.parts)]
        self.n_sims = n

    def close(self):
        for p in self.probs:
            p.close()
        for c in self.ctxs:
            c.close()
        self.probs, self.ctxs = [], []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self):
        """Enqueues feasibility -> CSR -> order -> pack of every part on its stream; nothing waits."""
        for p in self.probs:
            p.run()

    def best_option_sims(self, kinds: Sequence[int], dev_packed_ptr: Optional[int] = None, fetch: bool = True):
        """The expander chain once per simulation.  dev_packed_ptr: device array of n_sims int64 — every part writes the packed
        keys of its own simulations into its slice (the operand of ONE all-reduce(min) over ranks).  fetch=True returns the
        per-simulation results of all parts concatenated (best = group index inside the WHOLE batch)."""
        outs = []
        gbase = 0
        for i, (p, part) in enumerate(zip(self.probs, self.parts)):
            ptr = dev_packed_ptr + 8 * self.cuts[i] if dev_packed_ptr else None
            o = p.best_option_sims(kinds, per_sim=True, fetch=fetch, dev_packed_ptr=ptr, n_sims=part.n_sims)
            if fetch:
                o["best"] = np.where(o["best"] >= 0, o["best"] + gbase, o["best"]).astype(np.int32)
                outs.append(o)
            gbase += part.n_groups
        if not fetch:
            return None
        return {k: np.concatenate([o[k] for o in outs], axis=0) for k in ("best", "n_best", "best_set", "keys", "packed")}

    def fetch(self) -> BatchResult:
        """Results of every part as ONE batch result: group arrays concatenated, CSR offsets shifted, PEG ids back in the
        numbering of the whole table set."""
        rs = [p.fetch() for p in self.probs]
        offs, order, base = [np.zeros(1, np.int64)], [], 0
        for r, pb in zip(rs, self.peg_base):
            offs.append(r.offsets[1:].astype(np.int64) + base)
            order.append(r.order[:int(r.offsets[-1])] + pb)
            base += int(r.offsets[-1])
        cat = lambda name: np.concatenate([getattr(r, name) for r in rs])   # noqa: E731
        return BatchResult(offsets=np.concatenate(offs).astype(np.int32), node_count=cat("node_count"), pods_scheduled=cat("pods_scheduled"),
                           nodes_added=cat("nodes_added"), limiter_nodes=cat("limiter_nodes"), last_index_out=cat("last_index_out"),
                           status=cat("status"), req_cpu_sum=cat("req_cpu_sum"), req_mem_sum=cat("req_mem_sum"),
                           order=np.concatenate(order).astype(np.int32),
                           placed=np.concatenate([r.placed[:int(r.offsets[-1])] for r in rs]))
