"""A batch of independent simulations as sub-batches on HIP streams of their own.

The overlap lives INSIDE libcasim since round 3 (casim_options.n_streams, csrc/casim_streams.h): ONE context, one casim_problem
whose parts run on the context's internal streams.  This wrapper only keeps the old call shape for bench.py and the tests."""
from typing import Optional, Sequence

from .engine import BatchResult, Context, Problem
from .tables import TableSet


class StreamedBatch:
    """One casim context on `device` (on `stream`, a raw hipStream_t handle, when given) and one streamed casim_problem."""

    def __init__(self, device: int, tables: TableSet, n_streams: int = 4, stream: Optional[int] = None, **problem_kw):
        self.tables = tables if tables.peg_lo is not None else tables.as_one_simulation()
        self.n_sims = self.tables.n_sims
        self.ctx = Context(device, stream=stream)
        self._structs = self.tables.structs()
        self.prob = Problem(self.ctx, *self._structs, n_streams=n_streams, **problem_kw)
        self.parts = self.prob.info()["parts"]

    def close(self):
        if self.prob is not None:
            self.prob.close(); self.ctx.close()
            self.prob = self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def run(self):
        self.prob.run()

    def best_option_sims(self, kinds: Sequence[int], dev_packed_ptr: Optional[int] = None, fetch: bool = True, join_stream: Optional[int] = None):
        return self.prob.best_option_sims(kinds, per_sim=True, fetch=fetch, dev_packed_ptr=dev_packed_ptr, n_sims=self.n_sims, join_stream=join_stream)

    def fetch(self) -> BatchResult:
        return self.prob.fetch()
