// casim_pack_wide.h — K_pack_wide: one LANE simulates one BinpackingNodeEstimator.Estimate()
// (CA/estimator/binpacking_estimator.go:102-342; `CA/` = /root/reference/cluster-autoscaler/), 64 estimates per wavefront.
//
// pack_fast_kernel (casim_pack.h) gives an estimate a whole wave: the simulated nodes are its lanes and every PEG step pays its
// wave-uniform control flow on the scalar port — ~117 wave-instructions per step that places pods, 14 per step that does not, and a
// batch of thousands of small estimates (BASELINE config C2 x 4096: 81 920 of them, <= 50 nodes and ~116 PEGs each) is bound by
// exactly that issue rate.  Here an estimate is ONE lane: its simulated nodes live in LDS (node j of lane l at [j][l]: any per-lane
// node index is bank-conflict free), the PEG loop runs in lock-step over 64 estimates, and nothing crosses lanes — no ballot, no
// reduction, no scalar verdict.  A step costs the wave one scan of the node slots (the fit mask of the PEG as a 64-bit word per lane)
// plus one iteration per pod that a2 places in the busiest lane; the closed forms of a2 are not needed (pods are walked one by one,
// bounded by the eligibility rule below), those of a3 / a4 are kept (node creation is arithmetic).
//
// Same semantics as pack_body on RegStore<2, 1, 0> (the lean register store), restated per lane:
//   a2  tryToScheduleOnExistingNodes (:163-186): pods visit the simulated nodes in the cyclic order that starts right after lastIndex
//       (scheduling_opts.go:54-59; MarkMatch moves the start to the matched node, plugin_runner.go:138) — the next set bit of the fit
//       mask at or after the pointer, wrapping; a node leaves the mask when the next pod no longer fits it (NodeResourcesFit,
//       fit.go:678-765: pod slots, then every requested lane).
//   a3  tryToScheduleOnNewNodes (:190-269): next-fit on the newest node (only when a2 did not walk it: template unschedulable),
//       then ceil(r / c_new) fresh nodes gated by the limiter (threshold_based_limiter.go:57-69), with the three exits of SURVEY N2.
//   a4  tryFastPath (:274-324): one simulated node + arithmetic for the rest (last PEG only).
//
// Eligibility (casim_pipeline.h, ProblemT::init): a batch of simulations on fixed-stride lists whose packer would be
// pack_fast_kernel<2, 1, 0> (<= 2 narrowed int32 lanes, no exclusion words, no group-wide words, node bound <= 64), no PEG of more
// than kWideMaxCount pods (a2 walks pods one by one), no node_pods output, enough groups to fill waves.  Fix-up passes of
// casim_options.chain_last_index (a handful of marked groups) stay with pack_fast_kernel: the two kernels write identical results.
#pragma once
#include "casim_device.h"
#include "casim_types.h"
#include "../../include/casim.h"

namespace casim {

constexpr int32_t kWideMaxCount = 255;      // pods per PEG the one-by-one walk of a2 accepts
constexpr int32_t kWideMaxNodes = 64;       // node bound per group: the fit mask is one 64-bit word
constexpr int32_t kWideFull = (int32_t)0x80000000;   // free[0] of a node without a pod slot left (or not created yet): fits nothing, ever

struct WideScratch {
    const int32_t* fresh32;  // [NG][R] gcd-scaled free amounts of an empty node (FastScratch::fresh32)
    const int64_t* scale;    // [R]     gcd per lane
    const int32_t* perm;     // [n_slots] group of lane slot i (groups sorted by list length: a wave's lanes finish together), or null = identity
    int32_t cap;             // node slots per estimate in LDS (largest node bound of the launch)
    int32_t n_slots;         // lane slots in use (= NG)
};
// node slots in LDS: the scan walks them four at a time
CS_HOST_DEVICE int32_t casim_wide_cap4(int32_t cap) { return (cap + 3) & ~3; }
static inline size_t casim_wide_smem(int32_t cap) { return (size_t)casim_wide_cap4(cap) * 64u * 12u; }

struct WidePair { int32_t x, y; };   // (free[0], free[1]) of a node: one ds_read_b64

CS_GLOBAL CS_LAUNCH_BOUNDS(64, 1) void pack_wide_kernel(DevTables t, DevResults res, WideScratch ws) {
    const int lane = cs::tid();
    const int cap4 = casim_wide_cap4(ws.cap);
    WidePair* sfree = (WidePair*)cs::dyn_smem() + lane;                        // node j of this lane: sfree[j * 64]
    int32_t* sslots = (int32_t*)((WidePair*)cs::dyn_smem() + (size_t)cap4 * 64) + lane;   // sslots[j * 64]
    for (int j = 0; j < cap4; ++j) { sfree[j * 64] = WidePair{kWideFull, 0}; sslots[j * 64] = 0; }
    // Every lane stays for the whole loop (a lane without an estimate — past the last group, or a group a fix-up pass of
    // casim_options.chain_last_index leaves alone — idles with an empty list): the loop bounds are wave-uniform, ballots see 64 lanes
    const int slot = cs::bid() * 64 + lane;
    int ng = 0;
    bool live = slot < ws.n_slots;
    if (live) { ng = ws.perm ? ws.perm[slot] : slot; if (t.chain_redo && !t.chain_redo[ng]) live = false; }
    const int off = t.peg_off[ng];
    const int Gn = !live ? 0 : (t.peg_cnt ? t.peg_cnt[ng] : t.peg_off[ng + 1] - off);
    const int Gmax = (int)cs::wave_max_u32((uint32_t)Gn);
    const int32_t maxn = t.max_nodes[ng];
    const int32_t E = t.existing[ng];
    // the limiter as one number: 0 = grants nothing (max_nodes < 0), INT32_MAX = no limit (max_nodes == 0)
    const int32_t grant_bound = maxn < 0 ? 0 : (maxn == 0 ? 0x7fffffff : maxn);
    const int fast_k = (t.fastpath && res.fast_last[ng]) ? Gn - 1 : -1;
    const int R = t.R;
    const int32_t fresh0 = ws.fresh32[(int64_t)ng * R], fresh1 = R > 1 ? ws.fresh32[(int64_t)ng * R + 1] : 0;
    const int32_t fresh_slots = t.allowed[ng] - t.init_pods[ng];
    int32_t M = 0, last_index = t.last_index[ng], granted = 0, fakes = 0, total_placed = 0;
    bool more = true, bad = false;
    int64_t acc0 = 0, acc1 = 0;
    const uint32_t* recp = res.rec + (int64_t)off * 8;
    int32_t* placed_out = res.placed + off;
    int Mw = 0;   // wave-uniform, a multiple of 4: no lane has more simulated nodes (the scan's bound)

    // node m := a fresh node holding x pods of the current PEG
    auto create = [&](int m, uint32_t x, int32_t q0, int32_t q1) {
        const int32_t s = fresh_slots - (int32_t)x;
        sfree[m * 64] = WidePair{s > 0 ? fresh0 - (int32_t)x * q0 : kWideFull, fresh1 - (int32_t)x * q1};
        sslots[m * 64] = s;
    };

    cs::Words<4> nxt = cs::load4(recp);   // (a lane without records reads somebody's first record and ignores it)
    for (int k = 0; k < Gmax; ++k) {
        const cs::Words<4> w = nxt;
        { const int kn = k + 1 < Gn ? k + 1 : 0; nxt = cs::load4(recp + (int64_t)kn * 8); }   // record k + 1 travels while step k runs
        const bool act = k < Gn && !bad;
        const int32_t cnt = (int32_t)w.w[0];
        const uint32_t pf = w.w[1];
        if (act && (pf & CASIM_PEG_UNSUPPORTED) != 0) bad = true;
        const bool go = act && !bad;
        const int32_t q0 = (int32_t)w.w[2], q1 = (int32_t)w.w[3];
        // a lane nobody asks for fits whatever the node holds (fit.go:731-763 skips zero requests): compared against the smallest
        // value a free amount can take (kWideFull + 1; kWideFull itself marks a node that takes nothing)
        const int32_t q0e = q0 > 0 ? q0 : kWideFull + 1, q1e = q1 > 0 ? q1 : kWideFull + 1;
        const bool selfx = (pf & CASIM_PEG_SELF_EXCL_NODE) != 0;
        int32_t placed = 0;

        // ---- a2: the cyclic first-fit over the simulated nodes, pod by pod ----
        const bool a2 = go && M > 0 && (pf & CASIM_REC_A2_OK) != 0;
        if (cs::ballot(a2) != 0) {
            // the PEG's fit mask over the node slots, every lane its own (slots past a lane's M hold kWideFull): loads four at a time
            uint32_t flo = 0, fhi = 0;
            for (int jb = 0; jb < Mw; jb += 4) {
                WidePair f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) f[u] = sfree[(jb + u) * 64];
                uint32_t b = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) b |= ((f[u].x >= q0e) & (f[u].y >= q1e)) ? (1u << u) : 0u;
                if (jb < 32) flo |= b << jb; else fhi |= b << (jb - 32);
            }
            uint64_t F = a2 ? (((uint64_t)fhi << 32) | flo) : 0ull;
            if (F != 0) {
                // rotated order starts at list position (lastIndex + 1) % n; positions < E are the pre-existing cluster nodes (never
                // acceptable): first simulated node of the rotated order
                const int32_t n = E + M;
                const uint32_t li1 = (uint32_t)last_index + 1u;
                const int32_t o = last_index < 0 ? 0 : (int32_t)(li1 < (uint32_t)n ? li1 : li1 % (uint32_t)n);
                int p = o > E ? o - E : 0;
                int lastj = -1;
                while (F != 0 && placed < cnt) {
                    const uint64_t hiF = p < 64 ? (F >> p) << p : 0ull;
                    const int j = cs::ffs64(hiF != 0 ? hiF : F);
                    const WidePair f = sfree[j * 64];
                    const int32_t f0 = f.x - q0, f1 = f.y - q1, s = sslots[j * 64] - 1;
                    ++placed; lastj = j;
                    if (selfx || s <= 0 || f0 < q0e || f1 < q1e) F &= ~(1ull << j);
                    sfree[j * 64] = WidePair{s > 0 ? f0 : kWideFull, f1};
                    sslots[j * 64] = s;
                    p = j + 1;
                }
                last_index = E + lastj;
            }
        }

        // ---- a3 / a4: tryToScheduleOnNewNodes (:190-269) or tryFastPath (:274-324) ----
        int32_t rem = cnt - placed;
        if (go && more && rem > 0) {
            const uint32_t cf = (pf >> CASIM_REC_FRESH_SHIFT) & CASIM_REC_FRESH_MAX;   // pods that fit an EMPTY node (0: the template-level Filters fail)
            uint32_t cfresh = cf < (uint32_t)rem ? cf : (uint32_t)rem;
            if (selfx && cfresh > 1u) cfresh = 1u;
            const bool static_ok = (pf & CASIM_KFLAG_STATIC_OK) != 0;
            if (k == fast_k) {
                // tryFastPath: one simulated node, the rest by arithmetic
                if (grant_bound - granted <= 0) more = false;
                else {
                    granted++;
                    const uint32_t per = static_ok ? cfresh : 0u;
                    create(M, per, q0, q1);
                    M++;
                    if (per > 0) {
                        placed += (int32_t)per;
                        const int32_t size = (int32_t)(((uint32_t)rem + per - 1u) / per);  // scaleUpSize
                        const int32_t left = grant_bound - granted;
                        const int32_t want = size - 1;
                        const int32_t nf = want < left ? want : left;
                        placed += nf == want ? rem - (int32_t)per : (int32_t)((int64_t)nf * per);
                        fakes += nf; granted += nf;
                        if (nf < want) more = false;
                    }
                }
            } else {
                bool open = true;   // new nodes are worth asking for
                if (M > 0) {
                    const int m1 = (M - 1) * 64;
                    // next-fit on the newest node (:198-209).  When a2 ran for this PEG it walked EVERY simulated node, the newest included,
                    // with the same predicate, and left pods over: only a PEG whose a2 gate is closed (template unschedulable) asks by name
                    if ((pf & CASIM_REC_A2_OK) == 0 && static_ok) {
                        const WidePair f = sfree[m1];
                        int32_t f0 = f.x, f1 = f.y, s = sslots[m1], cl = 0;
                        while (cl < rem && f0 >= q0e && f1 >= q1e && !(selfx && cl >= 1)) {
                            f0 -= q0; f1 -= q1; --s; ++cl;
                            if (s <= 0) f0 = kWideFull;
                        }
                        if (cl > 0) { sfree[m1] = WidePair{f0, f1}; sslots[m1] = s; placed += cl; rem -= cl; }
                    }
                    // newest node still empty and the pod does not fit it: a new one would not help (:234-236)
                    open = rem != 0 && fresh_slots - sslots[m1] != 0;
                }
                if (open) {
                    const uint32_t cn = cfresh;
                    if (cn == 0) {
                        if (grant_bound - granted <= 0) more = false;                      // :244-246
                        else { granted++; create(M, 0u, q0, q1); M++; }                    // :257-263 node stays, PEG abandoned
                    } else {
                        const int32_t need = (int32_t)(((uint32_t)rem + cn - 1u) / cn);
                        const int32_t left = grant_bound - granted;
                        const int32_t nadd = need < left ? need : left;
                        if (nadd > 0) {
                            const uint32_t fit = (uint32_t)nadd * cn;
                            const int32_t pl = fit < (uint32_t)rem ? (int32_t)fit : rem;
                            for (int i = 0; i < nadd; ++i) {
                                const int32_t lf = pl - i * (int32_t)cn;
                                create(M + i, lf <= 0 ? 0u : ((uint32_t)lf < cn ? (uint32_t)lf : cn), q0, q1);
                            }
                            M += nadd; granted += nadd; placed += pl;
                        }
                        if (need > left) more = false;
                    }
                }
            }
        }
        while (cs::ballot(M > Mw) != 0) Mw += 4;
        if (go) {
            placed_out[k] = placed;
            total_placed += placed;
            acc0 += (int64_t)placed * (int64_t)q0;
            acc1 += (int64_t)placed * (int64_t)q1;
        }
    }

    if (!live) return;
    if (bad) {   // a group carrying a PEG outside the encoded predicate subset is delegated (status only): pack_unsupported
        for (int k = 0; k < Gn; ++k) placed_out[k] = 0;
        res.node_count[ng] = 0; res.pods[ng] = 0; res.nodes_added[ng] = 0; res.limiter_nodes[ng] = 0;
        res.last_index_out[ng] = t.last_index[ng]; res.status[ng] = CASIM_NG_UNSUPPORTED;
        res.cpu_sum[ng] = 0; res.mem_sum[ng] = 0;
        return;
    }
    int32_t with_pods = 0;   // len(newNodesWithPods) (:160)
    for (int j = 0; j < M; ++j) with_pods += sslots[j * 64] != fresh_slots ? 1 : 0;
    res.node_count[ng] = with_pods + fakes;
    res.pods[ng] = total_placed;
    res.nodes_added[ng] = M;
    res.limiter_nodes[ng] = granted;
    res.last_index_out[ng] = last_index;
    res.status[ng] = CASIM_NG_OK;
    res.cpu_sum[ng] = acc0 * (ws.scale ? ws.scale[0] : 1);
    res.mem_sum[ng] = acc1 * ((ws.scale && R > 1) ? ws.scale[1] : 1);
}

}  // namespace casim
