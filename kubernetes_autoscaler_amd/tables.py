"""Flat table sets (casim_pegs + casim_groups) as numpy arrays: slicing, tiling and sharding of encoded batches.

The encoder hands out views into its own memory; a TableSet owns copies, so a batch can be re-shaped on the host
without touching a single pod object again: B simulations tiled from S encoded ones (bench), the node groups of every
simulation block-partitioned over the GPUs of a node, one simulation cut out of a batch (tests)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _abi

_PEG_COLS = {  # name -> (dtype, width field or constant per record)
    "req": (np.int64, "n_res"), "count": (np.int32, 1), "flags": (np.uint32, 1),
    "tol_mask": (np.uint64, "w_taint"), "sel_mask": (np.uint64, "w_label"), "excl_block": (np.uint64, "w_excl"),
    "excl_mark": (np.uint64, "w_excl"), "zone_block": (np.uint64, "w_zone"), "zone_mark": (np.uint64, "w_zone"),
    "fp_cpu": (np.float64, 1), "fp_mem": (np.float64, 1),
}
_GROUP_COLS = {
    "alloc": (np.int64, "n_res"), "init_req": (np.int64, "n_res"), "allowed_pods": (np.int32, 1), "init_pods": (np.int32, 1),
    "flags": (np.uint32, 1), "taint_mask": (np.uint64, "w_taint"), "label_mask": (np.uint64, "w_label"),
    "init_excl": (np.uint64, "w_excl"), "init_zone": (np.uint64, "w_zone"), "zone_valid": (np.uint64, "w_zone"),
    "max_nodes": (np.int32, 1), "existing_nodes": (np.int32, 1), "last_index": (np.int32, 1),
    "cap_cpu": (np.float64, 1), "cap_mem": (np.float64, 1), "waste_cpu": (np.int64, 1), "waste_mem": (np.int64, 1),
}
_CT = {np.int64: C.c_int64, np.int32: C.c_int32, np.uint32: C.c_uint32, np.uint64: C.c_uint64, np.float64: C.c_double, np.uint8: C.c_uint8}


def _view(ptr, n, width, dtype):
    if not ptr or n * width == 0:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n * width,)).astype(dtype, copy=True).reshape(n, width)


@dataclass
class TableSet:
    dims: Dict[str, int]                      # n_res, w_taint, w_label, w_excl, w_zone
    pegs: Dict[str, Optional[np.ndarray]]     # [G][width] per column
    groups: Dict[str, Optional[np.ndarray]]   # [NG][width] per column
    peg_lo: Optional[np.ndarray] = None       # [NG] candidate PEG range of each group (device-side subsets)
    peg_hi: Optional[np.ndarray] = None
    peg_offsets: Optional[np.ndarray] = None  # explicit per-group PEG lists (host-side subsets)
    peg_index: Optional[np.ndarray] = None
    global_id: Optional[np.ndarray] = None    # [NG]
    sim_offsets: Optional[np.ndarray] = None  # [S+1]
    zone_polarity: Optional[np.ndarray] = None  # [w_zone] group bits of NEED polarity (casim_pegs.zone_polarity), one row per batch
    excl_polarity: Optional[np.ndarray] = None  # [w_excl] node bits of NEED polarity (casim_pegs.excl_polarity), one row per batch
    _keep: List[object] = field(default_factory=list)

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def from_structs(cls, pegs: _abi.Pegs, groups: _abi.Groups) -> "TableSet":
        dims = {k: int(getattr(pegs, k)) for k in ("n_res", "w_taint", "w_label", "w_excl", "w_zone")}
        G, NG = pegs.n_pegs, groups.n_groups
        w = lambda spec: dims[spec] if isinstance(spec, str) else spec
        pc = {k: _view(getattr(pegs, k), G, w(wd), dt) for k, (dt, wd) in _PEG_COLS.items()}
        gc = {k: _view(getattr(groups, k), NG, w(wd), dt) for k, (dt, wd) in _GROUP_COLS.items()}
        ts = cls(dims, pc, gc)
        if pegs.zone_polarity and dims["w_zone"] > 0:
            ts.zone_polarity = np.ctypeslib.as_array(pegs.zone_polarity, shape=(dims["w_zone"],)).astype(np.uint64, copy=True)
        if pegs.excl_polarity and dims["w_excl"] > 0:
            ts.excl_polarity = np.ctypeslib.as_array(pegs.excl_polarity, shape=(dims["w_excl"],)).astype(np.uint64, copy=True)
        if groups.peg_offsets:
            ts.peg_offsets = np.ctypeslib.as_array(groups.peg_offsets, shape=(NG + 1,)).astype(np.int32, copy=True)
            nnz = int(ts.peg_offsets[NG]) if NG else 0
            ts.peg_index = (np.ctypeslib.as_array(groups.peg_index, shape=(nnz,)).astype(np.int32, copy=True) if nnz
                            else np.zeros(0, np.int32))
        if groups.peg_lo:
            ts.peg_lo = np.ctypeslib.as_array(groups.peg_lo, shape=(NG,)).astype(np.int32, copy=True)
            ts.peg_hi = np.ctypeslib.as_array(groups.peg_hi, shape=(NG,)).astype(np.int32, copy=True)
        if groups.global_id:
            ts.global_id = np.ctypeslib.as_array(groups.global_id, shape=(NG,)).astype(np.int32, copy=True)
        if groups.n_sims > 0:
            ts.sim_offsets = np.ctypeslib.as_array(groups.sim_offsets, shape=(groups.n_sims + 1,)).astype(np.int32, copy=True)
        return ts

    @classmethod
    def from_encoder(cls, enc) -> "TableSet":
        return cls.from_structs(enc.pegs, enc.groups)

    @property
    def n_pegs(self) -> int:
        return int(self.pegs["count"].shape[0]) if self.pegs["count"] is not None else 0

    @property
    def n_groups(self) -> int:
        return int(self.groups["max_nodes"].shape[0]) if self.groups["max_nodes"] is not None else 0

    @property
    def n_sims(self) -> int:
        return 0 if self.sim_offsets is None else len(self.sim_offsets) - 1

    # ---- re-shaping -----------------------------------------------------------------------------
    def as_one_simulation(self) -> "TableSet":
        """Every group sees every PEG (device-side subsets) and the whole set is one simulation.  A table set that carries explicit
        SchedulablePodGroups lists (peg_offsets / peg_index) cannot be re-shaped this way: the lists would silently be replaced
        by "every group sees every PEG" and the estimates would change."""
        if self.peg_offsets is not None and int(self.peg_offsets[-1]) > 0:   # (all-empty lists: the per-node tables of casim_try_schedule_pods, nothing to lose)
            raise ValueError("this table set carries explicit per-group PEG lists (peg_offsets): head / sim_slice / shard / StreamedBatch "
                             "work on device-derived subsets only")
        G, NG = self.n_pegs, self.n_groups
        return TableSet(self.dims, self.pegs, self.groups, np.zeros(NG, np.int32), np.full(NG, G, np.int32), None, None,
                        np.arange(NG, dtype=np.int32), np.array([0, NG], np.int32), zone_polarity=self.zone_polarity, excl_polarity=self.excl_polarity)

    @staticmethod
    def concat(sets: Sequence["TableSet"]) -> "TableSet":
        """Independent simulations side by side: PEG ids and group ids are shifted, every group keeps seeing only its own
        simulation's PEGs.  All sets must share the dictionaries (same encoder) or at least the mask widths."""
        d = sets[0].dims
        if any(s.dims != d for s in sets):
            sets = TableSet._widened(sets)
            d = sets[0].dims
        cat = lambda cols: None if cols[0] is None else np.concatenate(cols, axis=0)
        pc = {k: cat([s.pegs[k] for s in sets]) for k in _PEG_COLS}
        gc = {k: cat([s.groups[k] for s in sets]) for k in _GROUP_COLS}
        lo, hi, gid, so = [], [], [], [0]
        pbase = gbase = 0
        for s in sets:
            one = s if s.peg_lo is not None else s.as_one_simulation()
            lo.append(one.peg_lo + pbase); hi.append(one.peg_hi + pbase)
            gid.append((one.global_id if one.global_id is not None else np.arange(s.n_groups, dtype=np.int32)))
            offs = one.sim_offsets if one.sim_offsets is not None else np.array([0, s.n_groups], np.int32)
            so.extend((offs[1:] + gbase).tolist())
            pbase += s.n_pegs; gbase += s.n_groups
        pols = [s.zone_polarity for s in sets if s.zone_polarity is not None]   # (one dictionary, one polarity row)
        if any(not np.array_equal(pz, pols[0]) for pz in pols) or (pols and len(pols) != len(sets)):
            raise ValueError("table sets with different zone polarity rows cannot share one batch")
        xpols = [s.excl_polarity for s in sets if s.excl_polarity is not None]
        if any(not np.array_equal(px, xpols[0]) for px in xpols) or (xpols and len(xpols) != len(sets)):
            raise ValueError("table sets with different node polarity rows cannot share one batch")
        return TableSet(d, pc, gc, np.concatenate(lo).astype(np.int32), np.concatenate(hi).astype(np.int32), None, None,
                        np.concatenate(gid).astype(np.int32), np.array(so, np.int32), zone_polarity=pols[0] if pols else None, excl_polarity=xpols[0] if xpols else None)

    @staticmethod
    def _widened(sets: Sequence["TableSet"]) -> List["TableSet"]:
        """Simulations encoded by encoders of their OWN (one dictionary each: C4's pairwise exclusion bits differ from seed to seed) side by
        side: every mask column padded with zero words to the widest set's width.  Exact — a simulation's groups only ever meet their own
        simulation's PEGs, so the bit numbering never crosses a simulation, and a zero word neither blocks nor marks anything.  Only for sets
        without NEED polarity rows (one row serves the whole batch) and with equal resource lanes."""
        if any(s.dims["n_res"] != sets[0].dims["n_res"] for s in sets):
            raise ValueError("table sets with different resource lanes cannot share one batch")
        for s in sets:
            if (s.zone_polarity is not None and s.zone_polarity.any()) or (s.excl_polarity is not None and s.excl_polarity.any()):
                raise ValueError("table sets with different mask widths AND polarity rows cannot share one batch")
        dims = dict(sets[0].dims)
        for k in ("w_taint", "w_label", "w_excl", "w_zone"):
            dims[k] = max(s.dims[k] for s in sets)

        def pad(col, n, spec, dtype):
            w = dims[spec] if isinstance(spec, str) else spec
            if col is None:
                return np.zeros((n, w), dtype) if isinstance(spec, str) and w > 0 else None
            return col if col.shape[1] == w else np.concatenate([col, np.zeros((col.shape[0], w - col.shape[1]), col.dtype)], axis=1)
        out = []
        for s in sets:
            pc = {k: pad(s.pegs[k], s.n_pegs, wd, dt) for k, (dt, wd) in _PEG_COLS.items()}
            gc = {k: pad(s.groups[k], s.n_groups, wd, dt) for k, (dt, wd) in _GROUP_COLS.items()}
            # (optional scalar columns must be all there or all absent)
            out.append(TableSet(dims, pc, gc, s.peg_lo, s.peg_hi, s.peg_offsets, s.peg_index, s.global_id, s.sim_offsets,
                                zone_polarity=np.zeros(dims["w_zone"], np.uint64) if dims["w_zone"] else None,
                                excl_polarity=np.zeros(dims["w_excl"], np.uint64) if dims["w_excl"] else None))
        return out

    def tile(self, times: int) -> "TableSet":
        """The batch repeated `times` times (distinct memory, same simulations)."""
        return TableSet.concat([self] * times) if times > 1 else self

    def tile_groups(self, times: int) -> "TableSet":
        """`times` copies of every simulation's node-group table over ONE copy of the PEG tables: the simulations of copy k see the same PEG rows
        as those of copy 0 (casim_groups.peg_lo / peg_hi point into the shared rows).  What a caller that sweeps limiter / template variants
        over the same pending pods hands over: the pods travel once.  PEG ids in the results are the shared table's."""
        if times <= 1:
            return self
        one = self if self.peg_lo is not None else self.as_one_simulation()
        gc = {k: (None if v is None else np.concatenate([v] * times, axis=0)) for k, v in one.groups.items()}
        ng = one.n_groups
        so = np.concatenate([[0]] + [one.sim_offsets[1:] + k * ng for k in range(times)]).astype(np.int32)
        gid = one.global_id if one.global_id is not None else np.arange(ng, dtype=np.int32)
        return TableSet(one.dims, one.pegs, gc, np.concatenate([one.peg_lo] * times).astype(np.int32), np.concatenate([one.peg_hi] * times).astype(np.int32),
                        None, None, np.concatenate([gid] * times).astype(np.int32), so, zone_polarity=one.zone_polarity, excl_polarity=one.excl_polarity)

    def pinned(self) -> "TableSet":
        """The same tables with every column in page-locked host memory (engine.pinned_copy): enter -> return calls then upload them
        without the library's staging copy."""
        from .engine import pinned_copy
        pc = {k: (None if v is None else pinned_copy(v)) for k, v in self.pegs.items()}
        gc = {k: (None if v is None else pinned_copy(v)) for k, v in self.groups.items()}
        return TableSet(self.dims, pc, gc, self.peg_lo, self.peg_hi, self.peg_offsets, self.peg_index, self.global_id, self.sim_offsets,
                        zone_polarity=self.zone_polarity, excl_polarity=self.excl_polarity)

    def head(self, n_sims: int) -> "TableSet":
        """The first n_sims simulations (their groups; the PEG table is cut after the last PEG they can see)."""
        one = self if self.peg_lo is not None else self.as_one_simulation()
        if n_sims >= one.n_sims:
            return one
        ng = int(one.sim_offsets[n_sims])
        gp = int(one.peg_hi[:ng].max()) if ng else 0
        pc = {k: (None if v is None else v[:gp]) for k, v in one.pegs.items()}
        gc = {k: (None if v is None else v[:ng]) for k, v in one.groups.items()}
        return TableSet(one.dims, pc, gc, one.peg_lo[:ng], one.peg_hi[:ng], None, None,
                        None if one.global_id is None else one.global_id[:ng], one.sim_offsets[:n_sims + 1].copy(), zone_polarity=one.zone_polarity, excl_polarity=one.excl_polarity)

    def sim_slice(self, a: int, b: int) -> "TableSet":
        """Simulations [a, b) as a table set of their own (their groups and the PEG rows they can see, re-based to 0): how a
        batch is cut into sub-batches that run on different HIP streams of one device."""
        one = self if self.peg_lo is not None else self.as_one_simulation()
        a, b = max(0, a), min(b, one.n_sims)
        g0, g1 = int(one.sim_offsets[a]), int(one.sim_offsets[b])
        p0 = int(one.peg_lo[g0:g1].min()) if g1 > g0 else 0
        p1 = int(one.peg_hi[g0:g1].max()) if g1 > g0 else 0
        pc = {k: (None if v is None else v[p0:p1]) for k, v in one.pegs.items()}
        gc = {k: (None if v is None else v[g0:g1]) for k, v in one.groups.items()}
        return TableSet(one.dims, pc, gc, one.peg_lo[g0:g1] - p0, one.peg_hi[g0:g1] - p0, None, None,
                        None if one.global_id is None else one.global_id[g0:g1], (one.sim_offsets[a:b + 1] - g0).astype(np.int32),
                        zone_polarity=one.zone_polarity, excl_polarity=one.excl_polarity)

    def select_groups(self, keep: np.ndarray) -> "TableSet":
        """The groups `keep` (ascending indices) of every simulation: how one GPU's shard of a batch is cut out.  PEG table
        replicated, the groups keep their simulation-wide ids in expander keys."""
        keep = np.asarray(keep, np.int64)
        one = self if self.peg_lo is not None else self.as_one_simulation()
        gc = {k: (None if v is None else v[keep]) for k, v in one.groups.items()}
        so = np.searchsorted(keep, one.sim_offsets, side="left").astype(np.int32)
        gid = one.global_id if one.global_id is not None else np.arange(one.n_groups, dtype=np.int32)
        return TableSet(one.dims, one.pegs, gc, one.peg_lo[keep], one.peg_hi[keep], None, None, gid[keep], so, zone_polarity=one.zone_polarity, excl_polarity=one.excl_polarity)

    def shard(self, rank: int, world: int, rotate: bool = True) -> "TableSet":
        """Node groups of every simulation block-partitioned over `world` GPUs.  With `rotate` the block
        boundaries of simulation s start at rank s % world, so that group counts which do not divide evenly (20 groups on
        8 GPUs) still balance over a batch."""
        one = self if self.peg_lo is not None else self.as_one_simulation()
        keep = []
        for s in range(one.n_sims):
            a, b = int(one.sim_offsets[s]), int(one.sim_offsets[s + 1])
            n = b - a
            r = (rank + (s if rotate else 0)) % world
            lo, hi = (n * r) // world, (n * (r + 1)) // world
            keep.extend(range(a + lo, a + hi))
        return one.select_groups(np.array(keep, np.int64))

    # ---- ctypes ---------------------------------------------------------------------------------
    def narrowed_requests(self):
        """(req32 [G][R] int32, req_unit [R] int64): the request table as 32-bit multiples of a per-lane unit (the lane's gcd) — what a caller
        that knows its units (milli-cpu, MiB) holds in the first place (casim_pegs.req32 / req_unit, ABI 10).  ValueError when a lane does not fit."""
        req = self.pegs["req"]
        R = self.dims["n_res"]
        unit = np.ones(R, np.int64)
        out = np.zeros(req.shape, np.int32)
        for r in range(R):
            col = req[:, r]
            g = int(np.gcd.reduce(np.abs(col))) if col.size else 0
            unit[r] = g if g > 0 else 1
            q = col // unit[r]
            if q.size and (q.max() > 0x7fffffff or q.min() < -0x7fffffff):
                raise ValueError(f"lane {r} does not narrow to 32 bits")
            out[:, r] = q
        return out, unit

    def structs(self, narrow_requests: bool = False):
        """(casim_pegs, casim_groups) over this set's arrays; the set must outlive every use of the structs.
        narrow_requests: hand the requests over as casim_pegs.req32 + req_unit (req = NULL): half the request bytes for the link."""
        keep = []

        def ptr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(_CT[dt]))
        p = _abi.Pegs(n_pegs=self.n_pegs, **self.dims)
        for k, (dt, _) in _PEG_COLS.items():
            setattr(p, k, ptr(self.pegs[k], dt))
        if narrow_requests:
            r32, unit = self.narrowed_requests()
            p.req = None
            p.req32 = ptr(r32, np.int32); p.req_unit = ptr(unit, np.int64)
        if self.zone_polarity is not None:
            p.zone_polarity = ptr(self.zone_polarity, np.uint64)
        if self.excl_polarity is not None:
            p.excl_polarity = ptr(self.excl_polarity, np.uint64)
        g = _abi.Groups(n_groups=self.n_groups)
        for k, (dt, _) in _GROUP_COLS.items():
            setattr(g, k, ptr(self.groups[k], dt))
        if self.peg_offsets is not None:
            g.peg_offsets = ptr(self.peg_offsets, np.int32)
            g.peg_index = ptr(self.peg_index if self.peg_index.size else np.zeros(1, np.int32), np.int32)
        elif self.peg_lo is not None:
            g.peg_lo = ptr(self.peg_lo, np.int32); g.peg_hi = ptr(self.peg_hi, np.int32)
        if self.global_id is not None:
            g.global_id = ptr(self.global_id, np.int32)
        if self.sim_offsets is not None:
            g.n_sims = len(self.sim_offsets) - 1
            g.sim_offsets = ptr(self.sim_offsets, np.int32)
        self._keep = keep
        return p, g
