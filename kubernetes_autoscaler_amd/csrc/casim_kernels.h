// casim_kernels.h — the hot-path kernels of the scale-up simulation engine (gfx950, wave64).
//
// Reference path restated here (closed forms, not a translation):
//   K_feas   `fits(peg, fresh node)` for every PEG x node group  — CheckPredicates as used by
//            SchedulablePodGroups, CA/core/scaleup/orchestrator/orchestrator.go:535-570
//   K_csr*   compaction of the feasibility bit-matrix into per-group PEG lists
//   K_order  DecreasingPodOrderer.Order  CA/estimator/decreasing_pod_orderer.go:46-88 and
//            determineBestPEGToFastpath  CA/estimator/binpacking_estimator.go:433-473
//   K_pack   BinpackingNodeEstimator.Estimate  CA/estimator/binpacking_estimator.go:102-342
//            (tryToScheduleOnExistingNodes :163, tryToScheduleOnNewNodes :190, tryFastPath :274,
//            limiter CA/estimator/threshold_based_limiter.go:57-69, node order
//            CA/simulator/clustersnapshot/scheduling_opts.go:54-59) with the Filter arithmetic
//            of NodeResourcesFit (V/.../noderesources/fit.go:678-765), TaintToleration,
//            NodeAffinity/nodeSelector, NodeUnschedulable, NodePorts and hostname / group-wide
//            InterPodAffinity folded into integer compares and bitmask tests.
//   K_option expander filter chain  CA/expander/{leastnodes,waste,mostpods}/*.go
//
// All integer work: no MFMA.  One wavefront simulates one node group; every simulated node is
// owned by exactly one lane (node m <-> lane m & 63), so all node state is lane-private and the
// only cross-lane traffic is ballots / wave reductions (no barriers in K_pack).
#pragma once
#include "casim_device.h"
#include "casim_types.h"

#include "../../include/casim.h"

namespace casim {

// ------------------------------------------------------------------------------------------
// shared predicate pieces
// ------------------------------------------------------------------------------------------
// Static (template-level) Filters of PEG g against group ng:
//   TaintToleration  : every NoSchedule/NoExecute taint tolerated  (taint & ~tol) == 0
//   NodeAffinity     : every required label requirement satisfied  (sel & ~label) == 0
//   NodeUnschedulable: unschedulable template needs the toleration flag
CS_DEVICE bool static_filters_pass(const DevTables& t, int g, int ng) {
    const uint64_t* tol = t.tol + (int64_t)g * t.Wt;
    const uint64_t* tnt = t.taint + (int64_t)ng * t.Wt;
    for (int w = 0; w < t.Wt; ++w)
        if (tnt[w] & ~tol[w]) return false;
    const uint64_t* sel = t.sel + (int64_t)g * t.Wl;
    const uint64_t* lab = t.label + (int64_t)ng * t.Wl;
    for (int w = 0; w < t.Wl; ++w)
        if (sel[w] & ~lab[w]) return false;
    if ((t.gflags[ng] & CASIM_NG_UNSCHEDULABLE) && !(t.pflags[g] & CASIM_PEG_TOLERATES_UNSCHEDULABLE)) return false;
    return true;
}

// How many pods with request `req` fit into (free, slots), clamped to `clampk`.
// fitsRequest: pod count first, then every lane with req > 0 needs req <= alloc - requested
// (fit.go:681-765); k pods fit iff k <= slots and k*req <= free for each such lane.
// `rq` (optional) = 1.0 / (double)req[r], precomputed once per PEG: the quotient is then ONE f64
// multiply + an exact +-1 fix-up instead of a ~100-instruction emulated 64-bit division.
// Exactness: f < 2^53 converts exactly; the quotient d < c <= 2^31, so the estimate's absolute error
// is < 2^-20 and trunc() is within +-1 of floor(f/q); the remainder test restores the exact floor.
CS_DEVICE uint32_t capacity_of(const int64_t* fr, int stride, int32_t slots, int R, const int64_t* req, uint32_t clampk,
                               const double* rq = nullptr) {
    if (slots <= 0) return 0;
    uint32_t c = (uint32_t)slots < clampk ? (uint32_t)slots : clampk;
    for (int r = 0; r < CASIM_KMAX_RES; ++r) {
        if (r >= R) break;
        const int64_t q = req[r];
        if (q > 0) {
            const int64_t f = fr[(int64_t)r * stride];
            if (f < q) return 0;
            // division only when the running bound does not already fit: c*q <= f  => floor(f/q) >= c
            const unsigned __int128 cq = (unsigned __int128)c * (uint64_t)q;
            if (cq > (unsigned __int128)(uint64_t)f) {
                uint64_t d;
                if (rq && f < (1ll << 53)) {
                    uint32_t e = (uint32_t)((double)f * rq[r]);
                    const int64_t rem = f - (int64_t)((uint64_t)e * (uint64_t)q);
                    if (rem < 0) e -= 1;
                    else if (rem >= q) e += 1;
                    d = e;
                } else d = (uint64_t)f / (uint64_t)q;
                c = (uint32_t)d;  // d < c here
            }
        }
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// K_feas: feasibility bit-matrix [NG][ceil(G/64)]
// ------------------------------------------------------------------------------------------
// grid = (ceil(G/256), NG), block = 256.  Thread -> one PEG; a wave's ballot is one output word.
CS_DEVICE bool fits_fresh_node(const DevTables& t, int g, int ng) {
    if (t.pflags[g] & CASIM_PEG_UNSUPPORTED) return false;
    if (!static_filters_pass(t, g, ng)) return false;
    int64_t fr[CASIM_KMAX_RES];
    for (int r = 0; r < CASIM_KMAX_RES; ++r)
        fr[r] = r < t.R ? t.alloc[(int64_t)ng * t.R + r] - t.init_req[(int64_t)ng * t.R + r] : 0;
    const int32_t slots = t.allowed[ng] - t.init_pods[ng];
    if (capacity_of(fr, 1, slots, t.R, t.req + (int64_t)g * t.R, 1u) == 0) return false;
    const uint64_t* xb = t.xblock + (int64_t)g * t.Wx;
    const uint64_t* ix = t.init_excl + (int64_t)ng * t.Wx;
    for (int w = 0; w < t.Wx; ++w)
        if (xb[w] & ix[w]) return false;
    const uint64_t* zb = t.zblock + (int64_t)g * t.Wz;
    const uint64_t* iz = t.init_zone + (int64_t)ng * t.Wz;
    for (int w = 0; w < t.Wz; ++w)
        if (zb[w] & iz[w]) return false;
    return true;
}

CS_GLOBAL void feas_kernel(DevTables t, uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg) {
    const int ng = cs::bid_y();
    const int g = cs::bid() * cs::nthreads() + cs::tid();
    bool ok = false;
    if (g < t.G) ok = fits_fresh_node(t, g, ng);
    const uint64_t b = cs::ballot(ok);
    if (cs::lane() == 0 && (g >> 6) < Wg) bits[(int64_t)ng * Wg + (g >> 6)] = b;
}

// K_csr_count: nnz per group; one block (256 threads) per group.
CS_GLOBAL void csr_count_kernel(const uint64_t* CS_RESTRICT bits, int Wg, int32_t* CS_RESTRICT counts) {
    const int ng = cs::bid();
    uint32_t c = 0;
    for (int w = cs::tid(); w < Wg; w += cs::nthreads()) c += (uint32_t)cs::popc64(bits[(int64_t)ng * Wg + w]);
    c = cs::wave_sum_u32(c);
    uint32_t* sm = (uint32_t*)cs::dyn_smem();
    const int wave = cs::tid() >> 6, nw = (cs::nthreads() + 63) >> 6;
    if (cs::lane() == 0) sm[wave] = c;
    cs::sync();
    if (cs::tid() == 0) {
        uint32_t s = 0;
        for (int i = 0; i < nw; ++i) s += sm[i];
        counts[ng] = (int32_t)s;
    }
}
// K_csr_scan: exclusive scan of counts -> offsets[NG+1]; single block, serial chunks per thread.
CS_GLOBAL void csr_scan_kernel(const int32_t* CS_RESTRICT counts, int NG, int32_t* CS_RESTRICT offsets) {
    // NG is small (<= a few thousand): one thread scans; launch-latency bound either way.
    if (cs::bid() == 0 && cs::tid() == 0) {
        int32_t s = 0;
        for (int i = 0; i < NG; ++i) { offsets[i] = s; s += counts[i]; }
        offsets[NG] = s;
    }
}
// K_csr_fill: PEG ids of each group in ascending order; one wave per group, 64 words per step.
CS_GLOBAL void csr_fill_kernel(const uint64_t* CS_RESTRICT bits, int Wg, const int32_t* CS_RESTRICT offsets,
                               int32_t* CS_RESTRICT idx) {
    const int ng = cs::bid();
    const int lane = cs::lane();
    int32_t base = offsets[ng];
    for (int w0 = 0; w0 < Wg; w0 += 64) {
        const int w = w0 + lane;
        uint64_t word = w < Wg ? bits[(int64_t)ng * Wg + w] : 0ull;
        const uint32_t pc = (uint32_t)cs::popc64(word);
        // exclusive prefix of pc over lanes (log-step scan through wave exchange)
        uint32_t incl = pc;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)cs::readlane_u64(incl, lane >= d ? lane - d : lane);
            if (lane >= d) incl += o;
        }
        int32_t pos = base + (int32_t)(incl - pc);
        while (word) {
            const int b = cs::ffs64(word);
            word &= word - 1;
            idx[pos++] = w * 64 + b;
        }
        base += (int32_t)cs::readlane_u64(incl, 63);
    }
}

// ------------------------------------------------------------------------------------------
// K_order: PEG order per group (+ fastpath chooser)
// ------------------------------------------------------------------------------------------
// score = cpuReq/cpuAlloc + memReq/memAlloc in IEEE double, each term only if alloc > 0
// (decreasing_pod_orderer.go:76-81).  Sorted descending; ties keep input order (canonical rule:
// Go's sort.Slice is unstable, insertion-sort-stable for <= 12 elements).
CS_DEVICE double peg_score(const DevTables& t, int g, int ng) {
    if (t.count[g] <= 0) return 0.0;  // Exemplar() == nil
    const int64_t ca = t.alloc[(int64_t)ng * t.R + 0], ma = t.alloc[(int64_t)ng * t.R + 1];
    double s = 0.0;
    if (ca > 0) s += (double)t.req[(int64_t)g * t.R + 0] / (double)ca;
    if (ma > 0) s += (double)t.req[(int64_t)g * t.R + 1] / (double)ma;
    return s;
}
CS_DEVICE uint64_t desc_key(double s) {
    uint64_t u = cs::double_bits(s);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // order-preserving map of doubles
    return ~u;                                          // ascending key == descending score
}
// simulationsSaved of determineBestPEGToFastpath (:436-466); -1 = not eligible
CS_DEVICE int32_t fastpath_saved(const DevTables& t, int g, int ng) {
    const int32_t n = t.count[g];
    if (n <= 0 || !(t.pflags[g] & CASIM_PEG_FASTPATH_OK)) return -1;
    int32_t by_aa = (t.pflags[g] & CASIM_PEG_FASTPATH_AA_SELF) ? n : 0;
    int32_t by_cpu = 0, by_mem = 0;
    if (t.fp_cpu && t.cap_cpu) by_cpu = (int32_t)ceil((double)n * t.fp_cpu[g] / t.cap_cpu[ng]);
    if (t.fp_mem && t.cap_mem) by_mem = (int32_t)ceil((double)n * t.fp_mem[g] / t.cap_mem[ng]);
    int32_t nodes = by_aa > by_cpu ? by_aa : by_cpu;
    nodes = nodes > by_mem ? nodes : by_mem;
    return nodes > 0 ? n - n / nodes : 0;
}

// One block (256 threads) per group.  Bitonic sort of (key, position) pairs in LDS, or in an
// HBM scratch slab when the group's PEG list does not fit (kLds == false).
template <bool kLds>
CS_GLOBAL void order_kernel(DevTables t, DevResults res, OrderScratch os) {
    const int ng = cs::bid();
    const int off = t.peg_off[ng];
    const int Gn = t.peg_off[ng + 1] - off;
    const int tid = cs::tid(), nt = cs::nthreads();
    int npad = 1;
    while (npad < Gn) npad <<= 1;
    char* base = kLds ? cs::dyn_smem() : os.gbuf + os.off[ng];
    uint64_t* keys = (uint64_t*)base;           // [npad]
    int32_t* pos = (int32_t*)(keys + npad);     // [npad]
    for (int i = tid; i < npad; i += nt) {
        if (i < Gn) {
            keys[i] = desc_key(peg_score(t, t.peg_idx[off + i], ng));
            pos[i] = i;
        } else {
            keys[i] = ~0ull;
            pos[i] = 0x7fffffff;
        }
    }
    cs::sync();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npad; i += nt) {
                const int l = i ^ j;
                if (l > i) {
                    const uint64_t ki = keys[i], kl = keys[l];
                    const int32_t pi = pos[i], pl = pos[l];
                    const bool gt = ki > kl || (ki == kl && pi > pl);  // (i) sorts after (l)
                    const bool up = (i & k) == 0;
                    if (gt == up) { keys[i] = kl; keys[l] = ki; pos[i] = pl; pos[l] = pi; }
                }
            }
            cs::sync();
        }
    }
    // fastpath: the eligible PEG with the largest simulationsSaved, last one on ties, goes last
    int best = -1;
    if (t.fastpath && Gn > 0) {
        int64_t* red = (int64_t*)(pos + npad + (npad & 1));  // [nt] 8-byte aligned
        int64_t mine = -1;
        for (int i = tid; i < Gn; i += nt) {
            const int32_t sv = fastpath_saved(t, t.peg_idx[off + pos[i]], ng);
            if (sv >= 0) {
                const int64_t v = ((int64_t)sv << 32) | (uint32_t)i;
                mine = v > mine ? v : mine;
            }
        }
        red[tid] = mine;
        cs::sync();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) { const int64_t o = red[tid + s]; if (o > red[tid]) red[tid] = o; }
            cs::sync();
        }
        const int64_t top = red[0];
        if (top >= 0) best = (int)(uint32_t)(top & 0xffffffffll);
        cs::sync();
    }
    for (int i = tid; i < Gn; i += nt) {
        int src = i;
        if (best >= 0) {
            if (i == Gn - 1) src = best;
            else if (i >= best) src = i + 1;
        }
        // Emit the PEG record in PROCESSING order (structure of arrays): the packer then streams its
        // group's records with coalesced loads, 64 records per wave-load, and never chases indices.
        // The template-level Filters (taints, nodeSelector / affinity, unschedulable) are constant per
        // (PEG, group): evaluate them once here and hand them over as one flag bit.
        const int g = t.peg_idx[off + pos[src]];
        res.order[off + i] = g;
        res.s_count[off + i] = t.count[g];
        res.s_flags[off + i] = (t.pflags[g] & ~CASIM_KFLAG_STATIC_OK) | (static_filters_pass(t, g, ng) ? CASIM_KFLAG_STATIC_OK : 0u);
        for (int r = 0; r < t.R; ++r) res.s_req[(int64_t)(off + i) * t.R + r] = t.req[(int64_t)g * t.R + r];
    }
    if (tid == 0) res.fast_last[ng] = best >= 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// K_pack: one wavefront = one Estimate()  -> casim_pack.h (included at the end of this file)
// ------------------------------------------------------------------------------------------
#if 0  // round-1 first version (LDS state, shuffle reductions); kept until casim_pack.h has soaked
struct PackCtx {
    int64_t* sfree;   // [R][cap]
    uint64_t* sexcl;  // [Wx][cap]
    int32_t* sslots;  // [cap]
    int32_t* snpods;  // [cap]
    int32_t* sctmp;   // [cap]
    uint64_t* szone;  // [Wz][64] one private copy per lane
    int cap;
};

CS_DEVICE uint32_t node_capacity(const PackCtx& c, const DevTables& t, int m, const int64_t* req, const uint64_t* xblock,
                                 uint32_t clampk, bool selfx) {
    for (int w = 0; w < t.Wx; ++w)
        if (c.sexcl[(int64_t)w * c.cap + m] & xblock[w]) return 0;  // NodePorts / hostname anti-affinity
    uint32_t k = capacity_of(c.sfree + m, c.cap, c.sslots[m], t.R, req, clampk);
    if (selfx && k > 1) k = 1;
    return k;
}
// place x pods of the PEG on node m (NodeInfo.AddPodInfo / update, types.go:361-371,439-463)
CS_DEVICE void node_commit(const PackCtx& c, const DevTables& t, int m, uint32_t x, const int64_t* req, const uint64_t* xmark) {
    for (int r = 0; r < CASIM_KMAX_RES; ++r) {
        if (r >= t.R) break;
        c.sfree[(int64_t)r * c.cap + m] -= (int64_t)x * req[r];
    }
    c.sslots[m] -= (int32_t)x;
    c.snpods[m] += (int32_t)x;
    for (int w = 0; w < t.Wx; ++w) c.sexcl[(int64_t)w * c.cap + m] |= xmark[w];
}
CS_DEVICE uint32_t sat_add(uint32_t a, uint32_t b, uint32_t cap1) {  // min(a + b, cap1), a,b <= 2^31
    const uint32_t s = a + b;
    return s > cap1 ? cap1 : s;
}

template <bool kLds>
CS_GLOBAL void pack_kernel(DevTables t, DevResults res, PackScratch ps) {
    const int ng = cs::bid();
    const int lane = cs::lane();
    const int R = t.R, Wx = t.Wx, Wz = t.Wz;
    const int off = t.peg_off[ng];
    const int Gn = t.peg_off[ng + 1] - off;

    PackCtx c;
    c.cap = ps.node_cap[ng];
    {
        char* base = kLds ? cs::dyn_smem() : ps.gstate + ps.state_off[ng];
        c.sfree = (int64_t*)base;
        c.sexcl = (uint64_t*)(c.sfree + (int64_t)R * c.cap);
        c.szone = c.sexcl + (int64_t)Wx * c.cap;
        c.sslots = (int32_t*)(c.szone + 64 * (Wz > 0 ? Wz : 1));
        c.snpods = c.sslots + c.cap;
        c.sctmp = c.snpods + c.cap;
    }

    // a group carrying a PEG outside the encoded predicate subset is delegated (status only)
    {
        bool bad = false;
        for (int i = lane; i < Gn; i += 64) bad |= (t.pflags[t.peg_idx[off + i]] & CASIM_PEG_UNSUPPORTED) != 0;
        if (cs::ballot(bad)) {
            for (int i = lane; i < Gn; i += 64) res.placed[off + i] = 0;
            if (lane == 0) {
                res.node_count[ng] = 0; res.pods[ng] = 0; res.nodes_added[ng] = 0; res.limiter_nodes[ng] = 0;
                res.last_index_out[ng] = t.last_index[ng]; res.status[ng] = CASIM_NG_UNSUPPORTED;
                res.cpu_sum[ng] = 0; res.mem_sum[ng] = 0;
            }
            return;
        }
    }

    // group constants (wave-uniform)
    int64_t ffree[CASIM_KMAX_RES];  // free vector of a fresh node: alloc - requested-by-preloaded-pods
    for (int r = 0; r < CASIM_KMAX_RES; ++r) ffree[r] = r < R ? t.alloc[(int64_t)ng * R + r] - t.init_req[(int64_t)ng * R + r] : 0;
    const int32_t fslots = t.allowed[ng] - t.init_pods[ng];
    const uint64_t* fexcl = t.init_excl + (int64_t)ng * Wx;
    const int32_t maxn = t.max_nodes[ng];
    const int32_t E = t.existing[ng];
    const bool fast_last = t.fastpath && res.fast_last[ng];
    const bool group_unschedulable = (t.gflags[ng] & CASIM_NG_UNSCHEDULABLE) != 0;
    const uint64_t* zvalid = t.zone_valid + (int64_t)ng * Wz;
    for (int w = 0; w < Wz; ++w) c.szone[w * 64 + lane] = t.init_zone[(int64_t)ng * Wz + w];

    int32_t M = 0;                        // simulated nodes so far (estimationState.newNodeNameIndex)
    int32_t last_index = t.last_index[ng];// lastIndexOrderMapping.lastIndex
    int32_t granted = 0;                  // limiter.nodes
    bool more = true;                     // newNodesAvailable
    int32_t fakes = 0;                    // fastpath fake nodes
    int32_t total_placed = 0;
    int64_t cpu_sum = 0, mem_sum = 0;

    for (int k = 0; k < Gn; ++k) {
        const int g = res.order[off + k];
        const int32_t cnt = t.count[g];
        const uint32_t pf = t.pflags[g];
        const bool selfx = (pf & CASIM_PEG_SELF_EXCL_NODE) != 0;
        bool zselfx = (pf & CASIM_PEG_SELF_EXCL_ZONE) != 0;
        int64_t req[CASIM_KMAX_RES];
        for (int r = 0; r < CASIM_KMAX_RES; ++r) req[r] = r < R ? t.req[(int64_t)g * R + r] : 0;
        const uint64_t* xblock = t.xblock + (int64_t)g * Wx;
        const uint64_t* xmark = t.xmark + (int64_t)g * Wx;
        const uint64_t* zblock = t.zblock + (int64_t)g * Wz;
        const uint64_t* zmark = t.zmark + (int64_t)g * Wz;
        const bool static_ok = static_filters_pass(t, g, ng);

        bool zblocked = false;
        for (int w = 0; w < Wz; ++w) {
            zblocked |= (c.szone[w * 64 + lane] & zblock[w]) != 0;
            zselfx |= (zblock[w] & zmark[w] & zvalid[w]) != 0;  // the PEG excludes itself group-wide
        }

        int32_t placed = 0;
        uint32_t on_last = 0;  // pods of THIS PEG that a2 put on the newest node (self-exclusion has no node bit)

        // ---- a2: tryToScheduleOnExistingNodes (:163-186), closed form over the cyclic node order ----
        // k identical pods visit the nodes round-robin from lastIndex+1 (MarkMatch moves the start
        // to the matched node); after t full rounds node j holds min(c_j, t) pods  (SURVEY N3).
        const uint32_t keff = (uint32_t)(zselfx ? (cnt > 0 ? 1 : 0) : cnt);
        // RunFiltersUntilPassingNode skips Spec.Unschedulable nodes before any Filter runs, tolerated or not
        // (plugin_runner.go:108-110); every simulated node clones the template's flag.
        if (M > 0 && keff > 0 && static_ok && !zblocked && !group_unschedulable) {
            const int S = (M + 63) >> 6;
            const uint32_t cap1 = keff + 1;
            uint32_t tot = 0, cmax = 0;
            for (int s = 0; s < S; ++s) {
                const int m = s * 64 + lane;
                uint32_t cj = 0;
                if (m < M) cj = node_capacity(c, t, m, req, xblock, keff, selfx);
                c.sctmp[m] = (int32_t)cj;
                tot = sat_add(tot, cs::wave_sum_u32(cj), cap1);
                const uint32_t mx = cs::wave_max_u32(cj);
                cmax = mx > cmax ? mx : cmax;
            }
            if (tot > 0) {
                uint32_t T, Rr;
                if (tot <= keff) { T = cmax; Rr = 0; placed = (int32_t)tot; }
                else {
                    uint32_t lo = 0, hi = cmax, slo = 0;  // S(lo) <= keff < S(hi)
                    while (hi - lo > 1) {
                        const uint32_t mid = lo + ((hi - lo) >> 1);
                        uint32_t sm = 0;
                        for (int s = 0; s < S; ++s) {
                            const uint32_t cj = (uint32_t)c.sctmp[s * 64 + lane];
                            sm = sat_add(sm, cs::wave_sum_u32(cj < mid ? cj : mid), cap1);
                        }
                        if (sm <= keff) { lo = mid; slo = sm; } else hi = mid;
                    }
                    T = lo; Rr = keff - slo; placed = (int32_t)keff;
                }
                const uint32_t Tf = Rr > 0 ? T + 1 : T;  // last round: candidates have c >= Tf
                // rotated order starts at list position (lastIndex + 1) % n; positions < E are the
                // pre-existing cluster nodes (never acceptable)
                const int32_t n = E + M;
                const int32_t o = (int32_t)(((int64_t)last_index + 1) % n);
                const int32_t m0 = o > E ? o - E : 0;
                int32_t A = 0, Tot = 0;
                for (int s = 0; s < S; ++s) {
                    const uint64_t b = cs::ballot((uint32_t)c.sctmp[s * 64 + lane] >= Tf);
                    Tot += cs::popc64(b);
                    A += cs::popc64(b & cs::low_mask(m0 - s * 64));
                }
                const int32_t target = Rr > 0 ? (int32_t)Rr - 1 : Tot - 1;
                int32_t basec = 0, new_last = last_index;
                uint32_t x_mine_last = 0;
                for (int s = 0; s < S; ++s) {
                    const int m = s * 64 + lane;
                    const uint32_t cj = (uint32_t)c.sctmp[m];
                    const bool cand = cj >= Tf;
                    const uint64_t b = cs::ballot(cand);
                    const int32_t pex = basec + cs::mbcnt(b);
                    const int32_t rot = m >= m0 ? pex - A : (Tot - A) + pex;
                    uint32_t x = cj < T ? cj : T;
                    if (Rr > 0 && cand && rot < (int32_t)Rr) x += 1;
                    const uint64_t hit = cs::ballot(cand && rot == target);
                    if (hit) new_last = E + s * 64 + cs::ffs64(hit);
                    if (x > 0) node_commit(c, t, m, x, req, xmark);
                    if (m == M - 1) x_mine_last = x;
                    basec += cs::popc64(b);
                }
                on_last = (uint32_t)cs::readlane_u64(x_mine_last, (M - 1) & 63);
                last_index = new_last;
                for (int w = 0; w < Wz; ++w) c.szone[w * 64 + lane] |= zmark[w] & zvalid[w];
            }
        }

        // ---- a3 / a4: tryToScheduleOnNewNodes (:190-269) or tryFastPath (:274-324) ----
        int32_t rem = cnt - placed;
        if (rem > 0 && more) {
            zblocked = false;
            for (int w = 0; w < Wz; ++w) zblocked |= (c.szone[w * 64 + lane] & zblock[w]) != 0;
            bool blocked = !static_ok || zblocked;
            // capacity of a FRESH node for this PEG
            uint32_t cfresh = 0;
            {
                bool xb = false;
                for (int w = 0; w < Wx; ++w) xb |= (fexcl[w] & xblock[w]) != 0;
                if (!xb) cfresh = capacity_of(ffree, 1, fslots, R, req, (uint32_t)rem);
                if ((selfx || zselfx) && cfresh > 1) cfresh = 1;
            }
            // lane that owns node m writes its fresh state + x pods
            auto create_nodes = [&](int32_t first, int32_t nadd, uint32_t per, int32_t pods_total) {
                // node first+i gets min(per, pods_total - i*per) pods
                for (int32_t m = first + ((lane - first) & 63); m < first + nadd; m += 64) {
                    const int32_t i = m - first;
                    int64_t left = (int64_t)pods_total - (int64_t)i * per;
                    const uint32_t x = left <= 0 ? 0u : (left < (int64_t)per ? (uint32_t)left : per);
                    for (int r = 0; r < CASIM_KMAX_RES; ++r) { if (r >= R) break; c.sfree[(int64_t)r * c.cap + m] = ffree[r] - (int64_t)x * req[r]; }
                    c.sslots[m] = fslots - (int32_t)x;
                    c.snpods[m] = (int32_t)x;
                    for (int w = 0; w < Wx; ++w) c.sexcl[(int64_t)w * c.cap + m] = fexcl[w] | (x > 0 ? xmark[w] : 0ull);
                }
            };
            auto permission_left = [&]() -> int64_t {  // nodes the limiter would still grant
                if (maxn < 0) return 0;
                if (maxn == 0) return 0x7fffffffll;
                return maxn > granted ? (int64_t)(maxn - granted) : 0;
            };
            bool marked = false;

            if (fast_last && k == Gn - 1) {
                // tryFastPath: one simulated node, the rest by arithmetic
                if (permission_left() <= 0) more = false;
                else {
                    granted++;
                    const uint32_t per = blocked ? 0u : (cfresh < (uint32_t)rem ? cfresh : (uint32_t)rem);
                    create_nodes(M, 1, per, (int32_t)per);
                    M++;
                    if (per > 0) {
                        marked = true;
                        placed += (int32_t)per;
                        const int32_t size = (int32_t)(((int64_t)rem + per - 1) / per);  // scaleUpSize
                        const int64_t left = permission_left();
                        const int32_t want = size - 1;
                        const int32_t nf = want < left ? want : (int32_t)left;
                        const int64_t fp = (int64_t)nf * per;
                        placed += (int32_t)(nf == want ? (int64_t)rem - per : fp);
                        fakes += nf; granted += nf;
                        if (nf < want) more = false;
                    }
                }
            } else {
                // next-fit on the newest node (:198-209)
                if (M > 0) {
                    const int lm = M - 1, owner = lm & 63;
                    uint32_t cl = 0;
                    if (!blocked && !(selfx && on_last > 0) && lane == owner)
                        cl = node_capacity(c, t, lm, req, xblock, (uint32_t)rem, selfx || zselfx);
                    cl = (uint32_t)cs::readlane_u64(cl, owner);
                    if (cl > 0) {
                        if (lane == owner) node_commit(c, t, lm, cl, req, xmark);
                        placed += (int32_t)cl; rem -= (int32_t)cl; marked = true;
                        if (zselfx) blocked = true;
                    }
                }
                bool stop = rem == 0;
                if (!stop && M > 0) {
                    // newest node still empty and the pod does not fit it: a new one would not help (:234-236)
                    const int lm = M - 1, owner = lm & 63;
                    int32_t np = lane == owner ? c.snpods[lm] : 0;
                    np = (int32_t)cs::readlane_u64((uint32_t)np, owner);
                    if (np == 0) stop = true;
                }
                while (!stop) {
                    const uint32_t cn = blocked ? 0u : cfresh;
                    if (cn == 0 || zselfx) {
                        if (permission_left() <= 0) { more = false; break; }       // :244-246
                        granted++;
                        const uint32_t x = cn < (uint32_t)rem ? cn : (uint32_t)rem;  // 0 or 1
                        create_nodes(M, 1, x, (int32_t)x);
                        M++;
                        if (x == 0) break;                                          // :257-263 node stays, PEG abandoned
                        placed += (int32_t)x; rem -= (int32_t)x; marked = true;
                        blocked = true;                                             // zselfx: the group now holds one
                        if (rem == 0) break;
                    } else {
                        const int64_t need = ((int64_t)rem + cn - 1) / cn;
                        const int64_t left = permission_left();
                        const int32_t nadd = (int32_t)(need < left ? need : left);
                        const int64_t fit = (int64_t)nadd * cn;
                        const int32_t pl = (int32_t)(fit < rem ? fit : rem);
                        if (nadd > 0) {
                            create_nodes(M, nadd, cn, pl);
                            M += nadd; granted += nadd; placed += pl; rem -= pl; marked = true;
                        }
                        if (need > left) more = false;
                        break;
                    }
                }
            }
            if (marked)
                for (int w = 0; w < Wz; ++w) c.szone[w * 64 + lane] |= zmark[w] & zvalid[w];
        }

        if (lane == 0) res.placed[off + k] = placed;
        total_placed += placed;
        cpu_sum += (int64_t)placed * req[0];
        mem_sum += (int64_t)placed * (R > 1 ? req[1] : 0);
    }

    // len(newNodesWithPods) (:160)
    int32_t with_pods = 0;
    for (int s = 0; s < ((M + 63) >> 6); ++s) {
        const int m = s * 64 + lane;
        with_pods += cs::popc64(cs::ballot(m < M && c.snpods[m] > 0));
    }
    if (lane == 0) {
        res.node_count[ng] = with_pods + fakes;
        res.pods[ng] = total_placed;
        res.nodes_added[ng] = M;
        res.limiter_nodes[ng] = granted;
        res.last_index_out[ng] = last_index;
        res.status[ng] = CASIM_NG_OK;
        res.cpu_sum[ng] = cpu_sum;
        res.mem_sum[ng] = mem_sum;
    }
}

#endif  // round-1 first version of pack_kernel

// ------------------------------------------------------------------------------------------
// K_option: expander filter chain over the groups of one launch
// ------------------------------------------------------------------------------------------
// Each filter keeps the options whose metric equals the best one among the current survivors
// (leastnodes.go:35-61, mostpods.go:33-53, waste.go:37-73), then chainStrategy stops as soon as
// one survivor is left (chain.go:36-45).  Single block; NG is small.
struct OptionArgs {
    const int32_t* node_count; const int32_t* pods; const int32_t* status;
    const int64_t* cpu_sum; const int64_t* mem_sum;
    const int64_t* waste_cpu; const int64_t* waste_mem;
    int32_t NG;
    int32_t kinds[8]; int32_t n_kinds;
    int32_t group_id_base;
    uint8_t* best_set;      // [NG] out
    int32_t* out;           // [2]: best index (local), survivors
    int64_t* key_out;       // [10]: key block of the winner (see option_kernel)
};

CS_DEVICE uint64_t option_metric(const OptionArgs& a, int kind, int i) {
    // smaller is better, as an order-preserving uint64
    if (kind == CASIM_EXPANDER_LEAST_NODES) return (uint64_t)(uint32_t)a.node_count[i];
    if (kind == CASIM_EXPANDER_MOST_PODS) return (uint64_t)(0x7fffffff - a.pods[i]);
    // least-waste: (availCPU-reqCPU)/availCPU + (availMem-reqMem)/availMem, avail = capacity*count (waste.go:48-54)
    const int64_t avc = a.waste_cpu ? a.waste_cpu[i] * (int64_t)a.node_count[i] : 0;
    const int64_t avm = a.waste_mem ? a.waste_mem[i] * (int64_t)a.node_count[i] : 0;
    const double wc = (double)(avc - a.cpu_sum[i]) / (double)avc;
    const double wm = (double)(avm - a.mem_sum[i]) / (double)avm;
    const double sc = wc + wm;
    uint64_t u = cs::double_bits(sc);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

CS_GLOBAL void option_kernel(OptionArgs a) {
    const int tid = cs::tid(), nt = cs::nthreads();
    uint64_t* red = (uint64_t*)cs::dyn_smem();  // [nt]
    // valid options: something was scheduled on at least one node (orchestrator.go:1057-1063)
    for (int i = tid; i < a.NG; i += nt)
        a.best_set[i] = (a.status[i] == CASIM_NG_OK && a.node_count[i] > 0 && a.pods[i] > 0) ? 1 : 0;
    cs::sync();
    for (int f = 0; f < a.n_kinds; ++f) {
        uint64_t mine = ~0ull;
        for (int i = tid; i < a.NG; i += nt)
            if (a.best_set[i]) { const uint64_t m = option_metric(a, a.kinds[f], i); mine = m < mine ? m : mine; }
        red[tid] = mine;
        cs::sync();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s && red[tid + s] < red[tid]) red[tid] = red[tid + s];
            cs::sync();
        }
        const uint64_t best = red[0];
        cs::sync();
        for (int i = tid; i < a.NG; i += nt)
            if (a.best_set[i] && option_metric(a, a.kinds[f], i) != best) a.best_set[i] = 0;
        cs::sync();
        // survivors
        uint32_t sv = 0;
        for (int i = tid; i < a.NG; i += nt) sv += a.best_set[i];
        red[tid] = sv;
        cs::sync();
        for (int s = nt >> 1; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; cs::sync(); }
        const uint64_t survivors = red[0];
        cs::sync();
        if (survivors <= 1) break;
    }
    // lowest surviving index + count
    uint64_t first = ~0ull; uint32_t sv = 0;
    for (int i = tid; i < a.NG; i += nt) if (a.best_set[i]) { if ((uint64_t)i < first) first = (uint64_t)i; sv++; }
    red[tid] = first;
    cs::sync();
    for (int s = nt >> 1; s > 0; s >>= 1) { if (tid < s && red[tid + s] < red[tid]) red[tid] = red[tid + s]; cs::sync(); }
    const uint64_t bi = red[0];
    cs::sync();
    red[tid] = sv;
    cs::sync();
    for (int s = nt >> 1; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; cs::sync(); }
    if (tid == 0) {
        a.out[0] = bi == ~0ull ? -1 : (int32_t)bi;
        a.out[1] = (int32_t)red[0];
        // key block (10 x int64, smaller = better) for the cross-GPU reduce:
        //   [0]      packed (first filter's metric << 20 | global group id): one all-reduce(min) is exact
        //            for the integer metrics (least-nodes / most-pods, < 2^43)
        //   [1..8]   the winner's metric under each filter of the chain, order-preserving int64
        //            (the chain == lexicographic min over (m_1, .., m_k, id))
        //   [9]      global group id of the winner
        const int64_t none = 0x7fffffffffffffffll;
        for (int i = 0; i < 10; ++i) a.key_out[i] = none;
        if (bi != ~0ull) {
            for (int f = 0; f < a.n_kinds; ++f)
                a.key_out[1 + f] = (int64_t)(option_metric(a, a.kinds[f], (int)bi) ^ 0x8000000000000000ull);
            const int64_t gid = (int64_t)a.group_id_base + (int64_t)bi;
            a.key_out[9] = gid;
            if (a.n_kinds > 0) {
                uint64_t m = option_metric(a, a.kinds[0], (int)bi);
                if (m > 0x7ffffffffffull) m = 0x7ffffffffffull;
                a.key_out[0] = (int64_t)((m << 20) | (uint64_t)(gid & 0xfffff));
            } else a.key_out[0] = gid;
        }
    }
}

}  // namespace casim

#include "casim_pack.h"
