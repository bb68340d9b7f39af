// casim_sched.h — K_sched: HintingSimulator.TrySchedulePods on the device (SURVEY §8 row f1,
// CA/simulator/scheduling/hinting_simulator.go:53-135; `CA/` = /root/reference/cluster-autoscaler/).
//
// The reference walks the pending pods one by one; for each it tries the hinted node
// (tryScheduleUsingHints :86-110), else the first passing node in the cyclic order that starts right
// after lastIndex (SchedulePodOnAnyNodeMatching -> RunFiltersUntilPassingNode, plugin_runner.go:54-143),
// commits the pod to the snapshot and moves lastIndex to the matched node.  The pass is sequential in
// the pods (every placement changes what the next pod sees) and data-parallel in the nodes, so:
//
//   K_sched_static  grid (ceil(N/64), C): one wave = 64 nodes x one pod class; the state-independent
//                   Filters (TaintToleration, NodeAffinity / nodeSelector, NodeUnschedulable) as one
//                   ballot word per (class, 64 nodes).  Fully parallel, streams the mask tables once.
//   K_sched         ONE workgroup of T = 64..512 threads (the pass is one sequential process); node m is owned by
//                   thread m % T, chunk m / T, its state (free resources, free pod slots, node-local exclusion
//                   bits) lives in LDS, or in an HBM slab when the cluster outgrows 160 KB.  The host folds
//                   consecutive pods of one class without hints into a run (class, k).  A run walks the nodes in
//                   cyclic order, T at a time: every fitting node takes one pod (block-wide rank = LDS prefix of
//                   the per-wave ballot counts) and the walk stops as soon as the k pods are placed — like the
//                   reference, a pod that fits nearby never looks at the rest of the cluster.  Only when the whole
//                   cluster was walked and pods are left do rounds 2.. run, in the closed form of casim_pack.h's
//                   a2 (after t more rounds node j holds min(c'_j, t) more pods; bisection on
//                   S(t) = sum_j min(c'_j, t) with block-wide sums), one walk per round naming the node of every pod.
//
// SimilarPodsScheduling (similar_pods.go:38-98) only memoises "a pod with this spec found no node"; the
// snapshot only fills up during the pass, so a class that failed once fails again whether or not the
// reference consults the memo (pods without a controller are re-tried and fail again): node_out is
// identical, the memo here (one LDS bit per class) just skips the re-scan.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <type_traits>
#include <unordered_map>
#include <string>
#include <utility>
#include <vector>

#include "casim_kernels.h"

namespace casim {

struct SchedArgs {
    int32_t N, C, n_runs, break_on_failure, last_index, cap, memo_classes;
    const int32_t* run_class;   // [n_runs]
    const int32_t* run_count;   // [n_runs] pods of the run (1 when hinted)
    const int32_t* run_hint;    // [n_runs] hinted node or -1
    const int32_t* run_first;   // [n_runs] index of the run's first pod in the caller's sequence
    const uint8_t* acceptable;  // [N] or null
    const uint64_t* fbits;      // [C][cap / 64] from sched_static_kernel
    int32_t* node_out;          // [P], pre-filled with -1
    int32_t* out;               // [8]: lastIndex, pods scheduled, runs processed, candidates processed, ext pods
    char* gstate;               // HBM slab (global variant) or null
    // ---- removal simulation: n_cand > 0 turns every candidate's runs into one transaction ----
    int32_t n_cand, persist, max_removable;
    const uint8_t* cand_atomic;   // [K] or null: removals that do not count toward max_removable
    const int32_t* cand_node;     // [K] node whose removal is simulated
    const int32_t* cand_run_off;  // [K+1] runs of candidate k
    const int32_t* cand_pod_off;  // [K+1] pods of candidate k in node_out
    uint8_t* removable_out;       // [K] 1 removable / 0 no place (pre-filled with 2 = not evaluated)
    char* committed;              // HBM: last committed sfree / sexcl / sslots (same layout as the working copy)
    // pods that a committed removal moved onto a later candidate are listed again by that candidate ("ext" pods)
    int32_t P, ext_cap;           // pods in the caller's flat list; room for ext pods (node_out has P + ext_cap entries)
    const int32_t* pod_class;     // [P] class of every listed pod
    const uint8_t* pod_sticky;    // [P] or null: the host must look at this pod again before it moves a second time
    int32_t* ext_ref;             // [ext_cap] out: flat index of the pod
    int32_t* ext_cand;            // [ext_cap] out: candidate whose simulation lists it again
    int32_t* log_ref;             // [P + ext_cap] committed moves in commit order: pod,
    int32_t* log_dest;            //               destination
    // ---- domain rules (casim_domain_rules): PodTopologySpread, anti-affinity on non-hostname keys ----
    int32_t n_rules;
    const int32_t* node_domain;   // [n_keys][N]
    const int32_t* rule_key; const int32_t* rule_kind; const int32_t* rule_max_skew; const int32_t* rule_min_domains;
    const int32_t* rule_self; const int32_t* rule_elig_row;
    const uint8_t* rule_ghost;    // [n_rules] or null: the removal candidate's ghost leaves this rule's domains (nodeTaintsPolicy: Honor)
    const int64_t* rule_off;      // [n_rules + 1]
    int32_t* rule_cnt;            // working counters (copied from count_init before every pass)
    int32_t* rule_dom_nodes;      // working copy: eligible nodes still carrying each domain (> 0 <=> the domain exists)
    int32_t* rule_contrib;        // working copy [n_rules][N]: what the pods on a node add to each rule (transactions only)
    const uint64_t* rule_elig;    // [rows][cap / 64]
    const int32_t* class_rule_off; const int32_t* inc_off; const int32_t* inc_rule;
    // SimilarPodsScheduling, exact form (only consulted for classes with spread rules): per run the (controller, class)
    // pair and the controller, -1 = pod without controller
    const int32_t* run_pair; const int32_t* run_ctrl;
    // removals_lean_kernel: the candidate and run columns above interleaved (16 bytes per record: one load per lane, one pointer each)
    const int32_t* lean_cand;     // [K + 1][4] node, first run, first pod, atomic (entry K: the end markers)
    const int32_t* lean_run;      // [n_runs][4] class, count, hint, first pod
    uint16_t* glog_dest;          // removals_lean_kernel<., true, true>: the log of committed moves in HBM (kLeanLogBuckets parts of log_cap / kLeanLogBuckets
    uint32_t* glog_ref;           // entries each): destination, pod (32 bits: calls of more than 65 536 pods), class
    uint8_t* glog_cls;
    int32_t lean_bulk_min;        // runs of at least this many unhinted pods of one class are placed a word of nodes at a time (schedule_run)
    int64_t* prof;                // [8] s_memtime ticks per phase of thread 0 (CASIM_PACK_PROF builds + CASIM_PACK_PROF_DUMP) or null
    int32_t* pair_memo;           // [n_pairs] 1 = cached as unschedulable (zeroed before every pass)
    int32_t* ctrl_count;          // [n_ctrl] specs cached for the controller (at most 10, similar_pods.go:52)
};
constexpr int kMaxRulesPerClass = 8;
constexpr int kMaxPodsPerOwnerRef = 10;

CS_GLOBAL void copy_i32_kernel(int32_t* dst, const int32_t* src, int64_t n) {
    const int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid();
    if (i < n) dst[i] = src[i];
}

CS_GLOBAL void fill_i32_kernel(int32_t* p, int64_t n, int32_t v) {
    const int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid();
    if (i < n) p[i] = v;
}

// grid = (cap / 64, C), block = 64
CS_GLOBAL void sched_static_kernel(DevTables t, uint64_t* CS_RESTRICT fbits, int S) {
    const int c = cs::bid_y();
    const int m = cs::bid() * 64 + cs::lane();
    bool ok = false;
    if (m < t.NG) ok = !(t.pflags[c] & CASIM_PEG_UNSUPPORTED) && static_filters_pass(t, c, m);
    const uint64_t b = cs::ballot(ok);
    if (cs::lane() == 0) fbits[(int64_t)c * S + cs::bid()] = b;
}

// Workgroup-wide collectives of K_sched: every value they return is identical in all waves of the block.
// Double-buffered LDS cells: a cell is rewritten two collectives later, i.e. after a barrier that every wave
// reaches only once it has read the previous content.
struct BlockCtl {
    uint64_t* red;    // [2][16]
    uint32_t* tab;    // [2][16]
    uint32_t* slot;   // [2]
    int W, wave, lane;
    int ph_red, ph_tab, ph_slot;

    // per-wave counts -> (block total, sum over the waves before mine)
    CS_DEVICE void count_prefix(uint32_t wave_count, uint32_t& total, uint32_t& before) {
        if (W == 1) { total = wave_count; before = 0; return; }
        uint32_t* tb = tab + ph_tab * 16;
        if (lane == 0) tb[wave] = wave_count;
        cs::sync();
        const uint32_t v = lane < W ? tb[lane] : 0u;
        total = cs::wave_sum_u32(v);
        before = cs::wave_sum_u32(lane < wave ? v : 0u);
        ph_tab ^= 1;
    }
    CS_DEVICE uint64_t sum(uint64_t lane_value) {
        const uint64_t w = cs::wave_sum_u64(lane_value);
        if (W == 1) return w;
        uint64_t* rb = red + ph_red * 16;
        if (lane == 0) rb[wave] = w;
        cs::sync();
        const uint64_t r = cs::wave_sum_u64(lane < W ? rb[lane] : 0ull);
        ph_red ^= 1;
        return r;
    }
    CS_DEVICE uint32_t max(uint32_t lane_value) {
        const uint32_t w = cs::wave_max_u32(lane_value);
        if (W == 1) return w;
        uint64_t* rb = red + ph_red * 16;
        if (lane == 0) rb[wave] = w;
        cs::sync();
        const uint32_t r = cs::wave_max_u32(lane < W ? (uint32_t)rb[lane] : 0u);
        ph_red ^= 1;
        return r;
    }
    // value of THE thread with `mine` (exactly one thread of the block has it): one barrier
    CS_DEVICE uint32_t pick(bool mine, uint32_t v) {
        uint32_t* s = slot + ph_slot;
        if (mine) *s = v;
        cs::sync();
        const uint32_t r = *s;
        ph_slot ^= 1;
        return r;
    }
};

// LDS control block: collective cells, the memo bits and (transactions only) the alive words + their rank prefix
CS_HOST_DEVICE int64_t casim_sched_ctrl_bytes(int memo_classes, int alive_words) {
    return 2 * 16 * 8 + 2 * 16 * 4 + 16 + ((4ll * ((memo_classes + 31) / 32) + 7) & ~7ll) + 20ll * alive_words + (alive_words & 1 ? 4 : 0);
}
inline int64_t casim_sched_state_bytes(int R, int Wx, int64_t cap) {
    return cap * (8ll * R + 8ll * Wx + 12ll) + 2ll * 8ll * (cap / 64);
}

// One workgroup of T = 64..512 threads; node m is owned by thread (m % T), chunk (m / T).
// kTxn (removal transactions) and kRules (domain rules) are compile-time: the plain TrySchedulePods instantiation
// does not carry their ~40 pointers — as one kernel, the uniform state overflowed the SGPR file and the hot loop
// was dominated by v_writelane / v_readlane spill traffic (r01l ISA: 1800 of them).
// RMAX_ = 2 for the common cpu + memory batch, else CASIM_KMAX_RES (see MemStore).
template <bool kLds, bool kTxn, bool kRules, int RMAX_>
// (launch bound 512: two waves per SIMD at most, so a wave may hold 256 vector registers — the instantiations with up to 8 resource
// lanes needed 128 + 40-52 spilled ones and ran with 80-132 bytes of scratch per lane under a bound of 1024, which no sweep ever favoured)
CS_GLOBAL CS_LAUNCH_BOUNDS(512, 1) void sched_kernel(DevTables t, SchedArgs a) {
    const int32_t n_rules = kRules ? a.n_rules : 0;
    using Store = MemStore<kLds, RMAX_>;
    const int tid = cs::tid(), lane = cs::lane(), wave = tid >> 6;
    const int T = cs::nthreads();
    const int R = t.R, Wx = t.Wx, N = a.N;
    const int Q = a.cap / T;  // chunks
    char* smem = cs::dyn_smem();
    BlockCtl bc;
    bc.red = (uint64_t*)smem;
    bc.tab = (uint32_t*)(bc.red + 32);
    bc.slot = bc.tab + 32;
    bc.W = T >> 6; bc.wave = wave; bc.lane = lane; bc.ph_red = bc.ph_tab = bc.ph_slot = 0;
    uint32_t* memo = bc.slot + 4;  // [ceil(memo_classes / 32)] class found no node
    // transactions: which nodes are still in the snapshot list, and how many live nodes precede each 64-node word
    // (lastIndex is a POSITION in the reference's node list, which shrinks when a removal is committed)
    constexpr bool txn = kTxn;
    const int nw = txn ? a.cap >> 6 : 0;
    uint64_t* alive = (uint64_t*)((char*)memo + ((4ll * ((a.memo_classes + 31) / 32) + 7) & ~7ll));  // [nw]
    uint64_t* arrivedb = alive + nw;                                                                    // [nw] node took pods of a committed removal
    uint32_t* wpre = (uint32_t*)(arrivedb + nw);                                                        // [nw]
    Store st;
    st.R = R; st.Wx = Wx; st.cap = a.cap;
    char* base = kLds ? smem + casim_sched_ctrl_bytes(a.memo_classes, nw) : a.gstate;
    st.sfree = (int64_t*)base;
    st.sexcl = (uint64_t*)(st.sfree + (int64_t)R * st.cap);
    uint64_t* scanb = st.sexcl + (int64_t)Wx * st.cap;  // [cap / 64] acceptable && !Spec.Unschedulable
    uint64_t* accb = scanb + (a.cap >> 6);               // [cap / 64] acceptable
    st.sslots = (int32_t*)(accb + (a.cap >> 6));
    st.snpods = st.sslots + st.cap;
    st.sctmp = st.snpods + st.cap;
    // committed copy (transactions): what Revert() restores
    int64_t* cfree = (int64_t*)a.committed;
    uint64_t* cexcl = (uint64_t*)(cfree + (int64_t)R * st.cap);
    int32_t* cslots = (int32_t*)(cexcl + (int64_t)Wx * st.cap);

    // ---- prologue: node state = what the running pods of each node hold (NodeInfo.Requested, types.go) ----
    for (int q = 0; q < Q; ++q) {
        const int m = q * T + tid;
        const bool live = m < N;
        for (int r = 0; r < R; ++r) st.sfree[(int64_t)r * st.cap + m] = live ? t.alloc[(int64_t)m * R + r] - t.init_req[(int64_t)m * R + r] : 0;
        for (int w = 0; w < Wx; ++w) st.sexcl[(int64_t)w * st.cap + m] = live ? t.init_excl[(int64_t)m * Wx + w] : 0ull;
        st.sslots[m] = live ? t.allowed[m] - t.init_pods[m] : 0;
        st.snpods[m] = 0;
        st.sctmp[m] = 0;
        const bool acc = live && (a.acceptable == nullptr || a.acceptable[m] != 0);
        const uint64_t ab = cs::ballot(acc);
        const uint64_t sb = cs::ballot(acc && !(t.gflags[live ? m : 0] & CASIM_NG_UNSCHEDULABLE));
        if (lane == 0) { accb[m >> 6] = ab; scanb[m >> 6] = sb; }
        if (txn) {
            for (int r = 0; r < R; ++r) cfree[(int64_t)r * st.cap + m] = st.sfree[(int64_t)r * st.cap + m];
            for (int w = 0; w < Wx; ++w) cexcl[(int64_t)w * st.cap + m] = st.sexcl[(int64_t)w * st.cap + m];
            cslots[m] = st.sslots[m];
            const uint64_t lb = cs::ballot(live);
            if (lane == 0) { arrivedb[m >> 6] = 0ull; alive[m >> 6] = lb; wpre[m >> 6] = (uint32_t)((m >> 6) << 6) < (uint32_t)N ? (uint32_t)((m >> 6) << 6) : (uint32_t)N; }
        }
    }
    for (int i = tid; i < (a.memo_classes + 31) / 32; i += T) memo[i] = 0u;
    cs::sync();

    CASIM_PROF_DECL;   // phases: 0 records / loop, 1 hint, 2 minima + origin, 3 walk (round 1), 4 lastIndex pick, 5 rounds 2.., 6 run end, 7 transactions
    int32_t last_index = a.last_index < -1 ? -1 : a.last_index;   // (a negative origin would index out of the Go slice)
    int32_t scheduled = 0;
    int32_t runs_done = 0;
    int32_t n_alive = N;       // len(nodeInfosList)
    int32_t removed = 0, cand_done = 0;
    int32_t log_n = 0, ext_n = 0;   // committed moves so far / ext pods listed so far
    bool any_dead = false;
    bool stop = false;
    // position of node m in the current node list / node at position p
    auto rank_of = [&](int m) -> int32_t {
        if (!any_dead) return m;
        return (int32_t)wpre[m >> 6] + cs::popc64(alive[m >> 6] & cs::low_mask(m & 63));
    };
    auto node_at = [&](int32_t pos) -> int32_t {
        if (!any_dead) return pos;
        bool mine = false; uint32_t w_mine = 0;
        for (int w = tid; w < nw; w += T) {
            const int32_t lo = (int32_t)wpre[w], cntw = cs::popc64(alive[w]);
            if (pos >= lo && pos < lo + cntw) { mine = true; w_mine = (uint32_t)w; }
        }
        const uint32_t w = bc.pick(mine, w_mine);
        // index of the (pos - wpre[w])-th live node of the word: binary search on popcounts
        uint64_t x = alive[w];
        uint32_t r = (uint32_t)(pos - (int32_t)wpre[w]);
        int idx = 0;
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {
            const uint32_t cl = (uint32_t)cs::popc64(x & ((1ull << sh) - 1ull));
            if (r >= cl) { r -= cl; x >>= sh; idx += sh; }
        }
        return (int32_t)(w << 6) + idx;
    };

    int32_t my_cand = 0, my_rlo = 0, my_rhi = 0, my_plo = 0, my_phi = 0;   // candidate records kc & ~63 .. of the removal loop
    // the loaded chunk of run records: runs [cstart, cend) of part cpart, record i in lane i - cstart
    int cstart = 0, cend = 0, cpart = -1;
    int32_t my_class = 0, my_count = 0, my_hint = -1, my_first = 0, my_pair = -1, my_ctrl = -1;
    uint32_t my_flags = 0;
    // per-class facts of the chunk's runs, fetched with the run records (one lane per run): the class's self-exclusion verdict and its
    // slices of the rule tables.  As per-run loads behind the class id they were three dependent HBM round trips in front of every
    // run — a third of the removal loop's time, whose runs are one or two pods long (profiles/r05d_sched_phase_profile.txt)
    uint32_t my_selfx = 0;
    int32_t my_klo = 0, my_khi = 0, my_ilo = 0, my_ihi = 0;   // (rules of the class: [klo, khi); increments: [ilo, ihi))
    int64_t my_req[RMAX_];
    double my_rq[RMAX_];
#pragma unroll
    for (int r = 0; r < RMAX_; ++r) { my_req[r] = 0; my_rq[r] = 0.0; }

    const int n_tx = txn ? a.n_cand : 1;
    for (int kc = 0; kc < n_tx; ++kc) {
    int run_lo = 0, run_hi = a.n_runs, Y = -1, e_lo = 0, e_hi = 0;
    bool break_on_failure = a.break_on_failure != 0, failed = false;
    if (txn) {
        // ---- SimulateNodeRemoval (cluster.go:131-172) of candidate kc, planner order (planner.go:300-330) ----
        if (a.max_removable > 0 && removed >= a.max_removable) break;
        if ((kc & 63) == 0) {   // candidate records ride in the lanes too: 64 per round trip
            const int kk = kc + lane;
            const bool have = kk < a.n_cand;
            my_cand = have ? a.cand_node[kk] : 0;
            my_rlo = have ? a.cand_run_off[kk] : 0; my_rhi = have ? a.cand_run_off[kk + 1] : 0;
            my_plo = have ? a.cand_pod_off[kk] : 0; my_phi = have ? a.cand_pod_off[kk + 1] : 0;
        }
        Y = (int)cs::bcast_u32((uint32_t)my_cand, kc & 63);
        if ((arrivedb[Y >> 6] >> (Y & 63)) & 1ull) {
            // Earlier committed removals moved pods onto this node: GetPodsToMove now lists them after the node's own
            // pods, in arrival order == commit order (NodeInfo.AddPod appends).  A sticky pod (PDB, drain rule) or a
            // full ext table hands the rest of the loop back to the caller.
            if (a.ext_cap <= 0) break;
            uint32_t base_n = 0;
            uint64_t bad = 0;
            for (int j0 = 0; j0 < log_n; j0 += T) {
                const int j = j0 + tid;
                const bool hit = j < log_n && a.log_dest[j] == Y;
                const int ref = hit ? a.log_ref[j] : 0;
                const uint64_t b = cs::ballot(hit);
                uint32_t tot, before;
                bc.count_prefix((uint32_t)cs::popc64(b), tot, before);
                const uint32_t pos = (uint32_t)ext_n + base_n + before + (uint32_t)cs::mbcnt(b);
                if (hit) {
                    if (pos < (uint32_t)a.ext_cap) { a.ext_ref[pos] = ref; a.ext_cand[pos] = kc; }
                    if (a.pod_sticky && a.pod_sticky[ref]) bad = 1;
                }
                base_n += tot;
            }
            if (bc.sum(bad) > 0 || (uint32_t)ext_n + base_n > (uint32_t)a.ext_cap) break;
            e_lo = ext_n; e_hi = ext_n + (int32_t)base_n; ext_n = e_hi;
        }
        cand_done = kc + 1;
        run_lo = (int)cs::bcast_u32((uint32_t)my_rlo, kc & 63); run_hi = (int)cs::bcast_u32((uint32_t)my_rhi, kc & 63);
        break_on_failure = true;
        if (!((alive[Y >> 6] >> (Y & 63)) & 1ull)) {   // NoNodeInfo (:139-147)
            if (tid == 0) a.removable_out[kc] = 0;
            continue;
        }
        // Fork; the candidate turns into a pod-less tainted ghost that keeps its list position (:243-265).
        // (every path into this point ends with a barrier: nobody still reads the words rewritten here)
        if (tid == 0) { accb[Y >> 6] &= ~(1ull << (Y & 63)); scanb[Y >> 6] &= ~(1ull << (Y & 63)); }
        // ... pod-less: whatever its pods added to the domain counters goes with them (the ghost itself stays a domain)
        for (int r = tid; r < n_rules; r += T) {
            const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + Y], v = cs::load_relaxed_i32(a.rule_contrib + (int64_t)r * N + Y);
            if (d >= 0 && v != 0) cs::atomic_add_i32(a.rule_cnt + a.rule_off[r] + d, -v);
            // ... and tainted (ToBeDeletedByClusterAutoscaler:NoSchedule, cluster.go:240-252): for a spread rule that honours node
            // taints and whose class does not tolerate this one, the ghost is no domain member while it is simulated
            if (a.rule_ghost && a.rule_ghost[r] && d >= 0) {
                const int row = a.rule_elig_row[r];
                const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (Y >> 6)] >> (Y & 63)) & 1ull);
                if (el) cs::atomic_add_i32(a.rule_dom_nodes + a.rule_off[r] + d, -1);
            }
        }
        cs::sync();
    }
    for (int part = 0; part < (kTxn ? 2 : 1); ++part) {   // the caller's runs, then (removal loop) one run per ext pod
    const int part_lo = part == 0 ? run_lo : e_lo, part_hi = part == 0 ? run_hi : e_hi;
    {
        for (int k = part_lo; k < part_hi && !stop && !failed; ++k) {
            if (cpart != part || k < cstart || k >= cend) {
                // 64 run records at a time, one per lane, each with its class record: one round trip to HBM per 64 runs —
                // across candidate boundaries too (a candidate of the removal loop has one or two runs)
                cpart = part; cstart = k;
                const int lim = part == 0 ? a.n_runs : e_hi;
                cend = cstart + 64 < lim ? cstart + 64 : lim;
                const int kk = cstart + lane;
                const bool have = kk < cend;
                my_class = !have ? 0 : part == 0 ? a.run_class[kk] : a.pod_class[a.ext_ref[kk]];
                my_count = !have ? 0 : part == 0 ? a.run_count[kk] : 1;
                my_hint = !have ? -1 : part == 0 ? a.run_hint[kk] : -1;   // an ext pod's hint is the node it sits on: the candidate
                my_first = !have ? 0 : part == 0 ? a.run_first[kk] : a.P + kk;
                my_pair = (kRules && have && part == 0 && a.run_pair) ? a.run_pair[kk] : -1;
                my_ctrl = (kRules && have && part == 0 && a.run_ctrl) ? a.run_ctrl[kk] : -1;
#pragma unroll
                for (int r = 0; r < RMAX_; ++r) {
                    my_req[r] = (have && r < R) ? t.req[(int64_t)my_class * R + r] : 0;
                    my_rq[r] = my_req[r] > 0 ? 1.0 / (double)my_req[r] : 0.0;
                }
                my_flags = have ? t.pflags[my_class] : 0u;
                my_selfx = (my_flags & CASIM_PEG_SELF_EXCL_NODE) != 0 ? 1u : 0u;
                if (have) for (int w = 0; w < Wx; ++w) my_selfx |= (t.xblock[(int64_t)my_class * Wx + w] & t.xmark[(int64_t)my_class * Wx + w]) != 0 ? 1u : 0u;
                my_klo = (have && n_rules > 0) ? a.class_rule_off[my_class] : 0; my_khi = (have && n_rules > 0) ? a.class_rule_off[my_class + 1] : 0;
                my_ilo = (have && n_rules > 0) ? a.inc_off[my_class] : 0; my_ihi = (have && n_rules > 0) ? a.inc_off[my_class + 1] : 0;
            }
            const int j = k - cstart;
            const int c = (int)cs::bcast_u32((uint32_t)my_class, j);
            const int32_t cnt = (int32_t)cs::bcast_u32((uint32_t)my_count, j);
            const int32_t hint = (int32_t)cs::bcast_u32((uint32_t)my_hint, j);
            const int32_t first = (int32_t)cs::bcast_u32((uint32_t)my_first, j);
            runs_done++;
            if (cnt <= 0) continue;
            typename Store::Peg pv;
#pragma unroll
            for (int r = 0; r < RMAX_; ++r) {
                pv.req[r] = r < R ? (int64_t)cs::bcast_u64((uint64_t)my_req[r], j) : 0;
                pv.rq[r] = r < R ? cs::bits_double(cs::bcast_u64(cs::double_bits(my_rq[r]), j)) : 0.0;
            }
            pv.xblock = t.xblock + (int64_t)c * Wx;
            pv.xmark = t.xmark + (int64_t)c * Wx;
            const bool selfx = cs::bcast_u32(my_selfx, j) != 0;
            const uint64_t* fb = a.fbits + (int64_t)c * (a.cap >> 6);
            int32_t placed = 0;
            // domain rules of the class; a class that feeds one of its own counters is walked pod by pod
            const int r_lo = (int)cs::bcast_u32((uint32_t)my_klo, j), r_hi = (int)cs::bcast_u32((uint32_t)my_khi, j);
            const int i_lo = (int)cs::bcast_u32((uint32_t)my_ilo, j), i_hi = (int)cs::bcast_u32((uint32_t)my_ihi, j);
            bool self_aff = false, monotone = true;   // monotone: a node that rejected the class once rejects it for good
            // (affinity rules are not monotone either: a node starts passing once a matching pod lands in its domain)
            bool has_aff = false, aff_self = false, aff_first = false;
            for (int r = r_lo; r < r_hi; ++r) {
                self_aff |= a.rule_self[r] != 0; monotone &= a.rule_kind[r] == 1;
                if (a.rule_kind[r] == 2) { has_aff = true; aff_self = a.rule_self[r] != 0; }
            }
            const int32_t pair = (int32_t)cs::bcast_u32((uint32_t)my_pair, j), ctrl = (int32_t)cs::bcast_u32((uint32_t)my_ctrl, j);
            int32_t minv[kMaxRulesPerClass];
            for (int ri = 0; ri < kMaxRulesPerClass; ++ri) minv[ri] = 0;
            // global minimum of every spread rule (minMatchNum, filtering.go:54-68): block-wide over the rule's domains
            auto refresh_minima = [&]() {
                if (has_aff) {
                    // len(affinityCounts) == 0 (filtering.go:402): no pod matching the class's affinity terms in any domain of
                    // any of their keys — together with the self match it lets the first pod of the series through
                    uint32_t some = 0;
                    for (int r = r_lo; r < r_hi; ++r) {
                        if (a.rule_kind[r] != 2) continue;
                        const int64_t lo = a.rule_off[r];
                        const int32_t D = (int32_t)(a.rule_off[r + 1] - lo);
                        for (int32_t d = tid; d < D; d += T) if (cs::load_relaxed_i32(a.rule_cnt + lo + d) > 0) some = 1;
                    }
                    aff_first = aff_self && bc.max(some) == 0;
                }
                for (int r = r_lo; r < r_hi; ++r) {
                    if (a.rule_kind[r] != 0) continue;
                    const int64_t lo = a.rule_off[r];
                    const int32_t D = (int32_t)(a.rule_off[r + 1] - lo);
                    uint32_t best = 0, nd = 0;   // best = INT32_MAX - min
                    for (int32_t d = tid; d < D; d += T)
                        if (cs::load_relaxed_i32(a.rule_dom_nodes + lo + d) > 0) {
                            const uint32_t inv = 0x7fffffffu - (uint32_t)cs::load_relaxed_i32(a.rule_cnt + lo + d);
                            best = inv > best ? inv : best; nd++;
                        }
                    const uint32_t bmax = bc.max(best);
                    const uint64_t ndom = bc.sum(nd);
                    minv[r - r_lo] = ndom < (uint64_t)a.rule_min_domains[r] ? 0 : (int32_t)(0x7fffffffu - bmax);
                }
            };
            auto rule_ok = [&](int m) -> bool {
                bool aff_missing = false;   // some affinity term without a matching pod in the node's domain
                for (int r = r_lo; r < r_hi; ++r) {
                    const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + m];
                    if (a.rule_kind[r] == 0) {
                        if (d < 0) return false;   // ErrReasonNodeLabelNotMatch
                        const int64_t skew = (int64_t)cs::load_relaxed_i32(a.rule_cnt + a.rule_off[r] + d) + a.rule_self[r] - minv[r - r_lo];
                        if (skew > a.rule_max_skew[r]) return false;
                    } else if (a.rule_kind[r] == 2) {
                        if (d < 0) return false;   // satisfyPodAffinity: all topology labels must exist on the node
                        if (cs::load_relaxed_i32(a.rule_cnt + a.rule_off[r] + d) <= 0) aff_missing = true;
                    } else if (d >= 0 && cs::load_relaxed_i32(a.rule_cnt + a.rule_off[r] + d) > 0) return false;
                }
                return !aff_missing || aff_first;
            };
            // NodeInfo.AddPod on node m + what the new pods mean for the rules of every class
            auto commit_pods = [&](int m, uint32_t x) {
                st.commit(0, m, x, pv);
                for (int ii = i_lo; ii < i_hi; ++ii) {
                    const int r = a.inc_rule[ii];
                    const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + m];
                    const int row = a.rule_elig_row[r];
                    const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (m >> 6)] >> (m & 63)) & 1ull);
                    if (d >= 0 && el) {
                        cs::atomic_add_i32(a.rule_cnt + a.rule_off[r] + d, (int32_t)x);
                        if (txn) cs::atomic_add_i32(a.rule_contrib + (int64_t)r * N + m, (int32_t)x);
                    }
                }
            };

            CASIM_PROF(0);
            // ---- tryScheduleUsingHints (:86-110): RunFiltersOnNode on the hinted node, no lastIndex update ----
            if (hint >= 0 && hint < N) {
                if (((fb[hint >> 6] & accb[hint >> 6]) >> (hint & 63)) & 1ull) {
                    const bool owner = tid == hint % T;
                    uint32_t ch = 0;
                    if (r_hi > r_lo) refresh_minima();
                    if (owner && rule_ok(hint)) ch = st.capacity(0, hint, pv, 1u, false);
                    if (bc.pick(owner, ch) > 0) {
                        if (owner) { commit_pods(hint, 1u); a.node_out[first] = hint; }
                        placed = 1;
                        if (i_hi > i_lo) cs::sync();   // the counters it fed are read by the next walk
                    }
                }
            }

            CASIM_PROF(1);
            // ---- trySchedule (:114-135): memo, then `cnt - placed` consecutive cyclic first-fits in closed form ----
            // IsSimilarUnschedulable (:114-118).  Monotone classes: one bit per class is equivalent to the reference's
            // per-controller cache (a re-try gives the same answer); classes with spread rules: the exact cache.
            const bool memo_hit = (monotone && c < a.memo_classes && ((memo[c >> 5] >> (c & 31)) & 1u)) ||
                                  (pair >= 0 && cs::load_relaxed_i32(a.pair_memo + pair) != 0);
            bool class_failed = memo_hit;
            while (placed < cnt && !class_failed) {
            const uint32_t keff = self_aff ? 1u : (uint32_t)(cnt - placed);
            const int32_t placed_before = placed;
            if (r_hi > r_lo) refresh_minima();
            {
                // The cyclic order starts at m0 = (lastIndex + 1) % N.  Pieces of T nodes in that order: chunk q0 from
                // m0 on, the following chunks (wrapping), and last the part of chunk q0 below m0.
                // (lastIndex + 1) % len(list): no division on the usual path (lastIndex is a position of the current list)
                uint32_t u0 = (uint32_t)last_index + 1u;
                if (u0 >= (uint32_t)n_alive) u0 %= (uint32_t)n_alive;   // rare: the list shrank, or the caller's index is stale
                const int32_t p0 = (int32_t)u0;
                const int32_t m0 = node_at(p0);
                const int q0 = m0 / T;
                const bool wrap_piece = (m0 % T) != 0;
                const int P = Q + (wrap_piece ? 1 : 0);
                auto piece_node = [&](int p, int& m, bool& valid) {
                    int q = p < Q ? q0 + p : q0;
                    if (q >= Q) q -= Q;
                    m = q * T + tid;
                    valid = m < N && (p != 0 || m >= m0) && (p != Q || m < m0);
                };
                // Round 1, fused with the capacity pass: walking the pieces in cyclic order, every fitting node takes one
                // pod until the run is exhausted — then the walk stops early (the reference would not look further
                // either); c' = c - 1 is what later rounds may still use.
                uint32_t cum = 0;
                int32_t pod_base = first + placed;
                uint32_t last_owner_val = 0;
                bool last_mine = false;
                CASIM_PROF(2);
                for (int p = 0; p < P; ++p) {
                    int m; bool valid;
                    piece_node(p, m, valid);
                    uint32_t cj = 0;
                    const bool elig = valid && (((fb[m >> 6] & scanb[m >> 6]) >> lane) & 1ull);
                    CASIM_PROF(8);    // piece setup + static word
                    if (elig && (r_hi == r_lo || rule_ok(m))) cj = st.capacity(0, m, pv, keff, selfx);
                    const bool fit = cj > 0;
                    const uint64_t b = cs::ballot(fit);
                    CASIM_PROF(9);    // capacity
                    uint32_t tot_p, before;
                    bc.count_prefix((uint32_t)cs::popc64(b), tot_p, before);
                    CASIM_PROF(10);   // block prefix (barrier)
                    const uint32_t rank = cum + before + (uint32_t)cs::mbcnt(b);
                    const bool gets = fit && rank < keff;
                    if (gets) { a.node_out[pod_base + (int32_t)rank] = m; commit_pods(m, 1u); }
                    if (valid) st.set_c(0, m, gets ? cj - 1u : 0u);
                    if (tot_p > 0) {  // every thread re-evaluates: only the owner in the LAST placing piece stays flagged
                        const uint32_t upto = cum + tot_p < keff ? cum + tot_p : keff;
                        last_mine = gets && rank == upto - 1u;
                        if (last_mine) last_owner_val = (uint32_t)rank_of(m);
                    }
                    cum += tot_p;
                    CASIM_PROF(11);   // placement: node_out, commit, c'
                    if (cum >= keff) break;
                }
                CASIM_PROF(3);
                if (cum > 0) {
                    // MarkMatch (plugin_runner.go:138): lastIndex = node of the last pod placed so far
                    last_index = (int32_t)bc.pick(last_mine, last_owner_val);
                    placed += (int32_t)(cum < keff ? cum : keff);
                }
                CASIM_PROF(4);
                if (cum > 0 && cum < keff) {
                    // ---- the whole cluster was walked and pods are left: rounds 2.. over c' in closed form ----
                    const uint32_t rem = keff - cum;
                    pod_base += (int32_t)cum;
                    uint64_t lane_sum = 0;
                    uint32_t lane_max = 0;
                    for (int q = 0; q < Q; ++q) {
                        const int m = q * T + tid;
                        const uint32_t cj = m < N ? st.get_c(0, m) : 0u;
                        lane_sum += cj;
                        lane_max = cj > lane_max ? cj : lane_max;
                    }
                    const uint64_t tot = bc.sum(lane_sum);
                    if (tot > 0) {
                        const uint32_t cmax = bc.max(lane_max);
                        uint32_t Tn, Rr;
                        int32_t got;
                        if (tot <= rem) { Tn = cmax; Rr = 0; got = (int32_t)tot; }
                        else {
                            // largest Tn with S(Tn) = sum_j min(c'_j, Tn) <= rem; S(0) = 0 <= rem < S(cmax) = tot
                            uint32_t lo = 0, hi = cmax; uint64_t slo = 0;
                            while (hi - lo > 1) {
                                const uint32_t mid = lo + ((hi - lo) >> 1);
                                uint64_t ls = 0;
                                for (int q = 0; q < Q; ++q) {
                                    const int m = q * T + tid;
                                    const uint32_t cj = m < N ? st.get_c(0, m) : 0u;
                                    ls += cj < mid ? cj : mid;
                                }
                                const uint64_t sm = bc.sum(ls);
                                if (sm <= rem) { lo = mid; slo = sm; } else hi = mid;
                            }
                            Tn = lo; Rr = rem - (uint32_t)slo; got = (int32_t)rem;
                        }
                        const uint32_t Tf = Rr > 0 ? Tn + 1 : Tn;
                        // one walk per round: the nodes with c' >= tr, in cyclic order, take the round's pods (pod index =
                        // pods of earlier rounds + rank); the last walk also commits
                        last_mine = false;
                        for (uint32_t tr = 1; tr <= Tf; ++tr) {
                            const bool partial = Rr > 0 && tr == Tf;
                            const bool lastr = tr == Tf;
                            uint32_t cumr = 0;
                            for (int p = 0; p < P; ++p) {
                                int m; bool valid;
                                piece_node(p, m, valid);
                                const uint32_t cj = valid ? st.get_c(0, m) : 0u;
                                const bool cand = cj >= tr;
                                const uint64_t b = cs::ballot(cand);
                                uint32_t tot_p, before;
                                bc.count_prefix((uint32_t)cs::popc64(b), tot_p, before);
                                const uint32_t rank = cumr + before + (uint32_t)cs::mbcnt(b);
                                const bool gets = cand && (!partial || rank < Rr);
                                if (gets) a.node_out[pod_base + (int32_t)rank] = m;
                                if (lastr) {
                                    if (tot_p > 0 && (!partial || cumr < Rr)) {  // this piece takes pods of the last round
                                        uint32_t upto = cumr + tot_p;
                                        if (partial && upto > Rr) upto = Rr;
                                        last_mine = gets && rank == upto - 1u;
                                        if (last_mine) last_owner_val = (uint32_t)rank_of(m);
                                    }
                                    const uint32_t x = (cj < Tn ? cj : Tn) + ((partial && gets) ? 1u : 0u);
                                    if (x > 0) commit_pods(m, x);
                                }
                                cumr += tot_p;
                            }
                            pod_base += (int32_t)(partial ? Rr : cumr);
                        }
                        last_index = (int32_t)bc.pick(last_mine, last_owner_val);
                        placed += got;
                    }
                }
            }
            CASIM_PROF(5);
            if (placed - placed_before < (int32_t)keff) class_failed = true;   // nothing changed since: the next one fails too
            else if (i_hi > i_lo) cs::sync();                                  // counters fed by this walk are read by the next
            }  // pods of the run
            scheduled += placed;
            if (placed < cnt) {
                // SetUnschedulable (:127, similar_pods.go:80-97: at most 10 cached specs per controller, and only when the
                // scan actually ran); breakOnFailure (:79-81)
                if (tid == 0) {
                    if (monotone && c < a.memo_classes) memo[c >> 5] |= 1u << (c & 31);
                    if (pair >= 0 && !memo_hit && cs::load_relaxed_i32(a.pair_memo + pair) == 0 &&
                        cs::load_relaxed_i32(a.ctrl_count + ctrl) < kMaxPodsPerOwnerRef) {
                        cs::atomic_add_i32(a.pair_memo + pair, 1);
                        cs::atomic_add_i32(a.ctrl_count + ctrl, 1);
                    }
                }
                cs::sync();
                if (break_on_failure) { if (txn) failed = true; else stop = true; }
            }
        }
    }
    }  // parts
    CASIM_PROF(6);
    if (txn) {
        // every pod found a place <=> the node is removable (findPlaceFor :219-224)
        const bool ok = !failed;
        const int p_lo = (int)cs::bcast_u32((uint32_t)my_plo, kc & 63), p_hi = (int)cs::bcast_u32((uint32_t)my_phi, kc & 63);
        const int n_own = p_hi - p_lo, n_listed = n_own + (e_hi - e_lo);
        // i-th listed pod of this candidate -> its slot in node_out / its flat pod index
        auto slot_of = [&](int i) -> int { return i < n_own ? p_lo + i : a.P + e_lo + (i - n_own); };
        auto pod_of = [&](int i) -> int { return i < n_own ? p_lo + i : a.ext_ref[e_lo + (i - n_own)]; };
        cs::sync();
        if (ok && a.persist) {
            // Commit (withForkedSnapshot :174-188): the touched nodes become the new committed state, the ghost leaves
            // the list (:230) and the destination set (planner.go:318)
            for (int i = tid; i < n_listed; i += T) {
                const int m = a.node_out[slot_of(i)];
                for (int r = 0; r < R; ++r) cfree[(int64_t)r * st.cap + m] = st.sfree[(int64_t)r * st.cap + m];
                for (int w = 0; w < Wx; ++w) cexcl[(int64_t)w * st.cap + m] = st.sexcl[(int64_t)w * st.cap + m];
                cslots[m] = st.sslots[m];
                cs::lds_or_u64(arrivedb + (m >> 6), 1ull << (m & 63));
                a.log_ref[log_n + i] = pod_of(i);
                a.log_dest[log_n + i] = m;
            }
            log_n += n_listed;
            if (tid == 0) alive[Y >> 6] &= ~(1ull << (Y & 63));
            for (int r = tid; r < n_rules; r += T) {   // RemoveNodeInfo: one eligible node less in its domains
                const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + Y];
                const int row = a.rule_elig_row[r];
                const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (Y >> 6)] >> (Y & 63)) & 1ull);
                if (d >= 0 && el && !(a.rule_ghost && a.rule_ghost[r])) cs::atomic_add_i32(a.rule_dom_nodes + a.rule_off[r] + d, -1);   // (a ghost that left already stays out)
            }
            for (int w = (Y >> 6) + 1 + tid; w < nw; w += T) wpre[w] -= 1u;   // one live node less in front of these words
            n_alive--; any_dead = true;
        } else {
            // Revert: touched nodes get their committed state back, the candidate its pods and its place
            for (int r = tid; r < n_rules; r += T) {   // the candidate gets its pods back, and its place in the domains it left
                const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + Y], v = cs::load_relaxed_i32(a.rule_contrib + (int64_t)r * N + Y);
                if (d >= 0 && v != 0) cs::atomic_add_i32(a.rule_cnt + a.rule_off[r] + d, v);
                if (a.rule_ghost && a.rule_ghost[r] && d >= 0) {
                    const int row = a.rule_elig_row[r];
                    const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (Y >> 6)] >> (Y & 63)) & 1ull);
                    if (el) cs::atomic_add_i32(a.rule_dom_nodes + a.rule_off[r] + d, 1);
                }
            }
            for (int i = tid; i < n_listed; i += T) {
                const int m = a.node_out[slot_of(i)];
                if (m < 0) continue;
                if (n_rules > 0) {   // and the destinations lose what the reverted placements fed
                    const int ci = a.pod_class[pod_of(i)];
                    for (int ii = a.inc_off[ci]; ii < a.inc_off[ci + 1]; ++ii) {
                        const int r = a.inc_rule[ii];
                        const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + m];
                        const int row = a.rule_elig_row[r];
                        const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (m >> 6)] >> (m & 63)) & 1ull);
                        if (d >= 0 && el) {
                            cs::atomic_add_i32(a.rule_cnt + a.rule_off[r] + d, -1);
                            cs::atomic_add_i32(a.rule_contrib + (int64_t)r * N + m, -1);
                        }
                    }
                }
                for (int r = 0; r < R; ++r) st.sfree[(int64_t)r * st.cap + m] = cfree[(int64_t)r * st.cap + m];
                for (int w = 0; w < Wx; ++w) st.sexcl[(int64_t)w * st.cap + m] = cexcl[(int64_t)w * st.cap + m];
                st.sslots[m] = cslots[m];
            }
            if (tid == 0) {
                const bool acc = a.acceptable == nullptr || a.acceptable[Y] != 0;
                if (acc) {
                    accb[Y >> 6] |= 1ull << (Y & 63);
                    if (!(t.gflags[Y] & CASIM_NG_UNSCHEDULABLE)) scanb[Y >> 6] |= 1ull << (Y & 63);
                }
            }
        }
        if (ok && !(a.cand_atomic && a.cand_atomic[kc])) removed++;   // len(removableList) - atomicScaleDownNodesCount
        if (tid == 0) a.removable_out[kc] = ok ? 1 : 0;
        cs::sync();
    }
    CASIM_PROF(7);
    }  // transactions
#if defined(CASIM_PACK_PROF) && !defined(CASIM_HOST_EMU)
    if (a.prof && tid == 0) for (int i = 0; i < 12; ++i) a.prof[i] = (int64_t)prof_acc[i];
#endif
    if (tid == 0) {
        a.out[0] = last_index;
        a.out[1] = scheduled;
        a.out[2] = runs_done;
        a.out[3] = cand_done;
        a.out[4] = ext_n;
    }
}


// ---- K_removals: the removal loop as ONE wave over per-class fit masks (round 5) ------------------------------------
//
// The removal loop (SimulateNodeRemoval per candidate, planner order; cluster.go:131-260) is a chain: lastIndex is a POSITION in a node list that
// shrinks with every committed removal and it is never reverted (plugin_runner.go:138), so candidate k starts where candidate k - 1 stopped
// (DESIGN.md 17e).  What can shrink is the cost of a link of that chain.  sched_kernel above pays ~950 instructions and five or six drains per
// candidate for the generality of its walk (blocks of T nodes, capacities of every node of a piece, block-wide prefix sums, closed-form rounds).
// A removal candidate moves one or two pods, each to the first node after lastIndex that takes it.  So, for clusters without domain rules and
// node-local exclusion words:
//
//   fit[c][w]   one bit per (class, node): the static Filters pass AND one more pod of the class fits the node NOW (fitsRequest on the node's
//               current free amounts and pod slots).  Kept exact incrementally: a placement on node m can only clear bit m (lane c = class c
//               re-evaluates the node against its own request: one compare per lane), a revert recomputes bit m from the static word.
//   successor   "first passing node at or after position p" = first set bit of fit[c] & scan in cyclic order: the 64 lanes look at 64 words at
//               once, ballot + two find-first-set; no capacities, no prefix sums, no barrier — the block IS one wave.
//
// Node state (free amounts, pod slots) stays int64 in LDS, the placements of the running transaction sit in an LDS ring (revert / commit read them
// back without waiting for stores).  Everything else — ghost, list positions after removals, pods listed again by a later candidate ("ext"), sticky
// pods, atomic groups, max_removable, hints, persist on / off — follows sched_kernel<.., kTxn = true, ..> statement by statement; results are
// identical (tests/test_removal_lean_emu.py runs every removal case through both kernels).  Not eligible (host side, SchedulerT::init): domain
// rules, exclusion words, more than 64 classes, more than 4 lanes, state beyond the LDS budget -> sched_kernel as before.
constexpr int kLeanLogBuckets = 8;   // removals_lean_kernel<., true>: the log of committed moves in this many parts, by destination & 7
constexpr int kLeanTxnCap = 256;   // pods of one transaction (destination, pod, class) kept in LDS; longer ones fall back on HBM (after a real wait)
// log_cap: committed moves the call can make (pods + ext capacity), rounded up to 256
CS_HOST_DEVICE int64_t casim_lean_removal_bytes(int R, int C, int64_t cap, int64_t log_cap) {
    const int64_t S = cap >> 6;
    return 8 * (int64_t)C * S + 4 * 8 * S + 4 * ((S + 1) & ~1ll) + 3 * 4 * kLeanTxnCap + 8 * (int64_t)R * cap + 4 * cap + ((5 * log_cap + 7) & ~7ll);
}

// grid (S, C), block 64: static word AND "one pod of the class fits the node as the snapshot stands"
CS_GLOBAL void lean_fit0_kernel(DevTables t, const uint64_t* CS_RESTRICT fbits, uint64_t* CS_RESTRICT fit0, int S) {
    const int c = cs::bid_y(), w = cs::bid(), lane = cs::lane();
    const int m = w * 64 + lane;
    bool ok = false;
    if (m < t.NG && ((fbits[(int64_t)c * S + w] >> lane) & 1ull)) {
        ok = t.allowed[m] - t.init_pods[m] > 0;
        for (int r = 0; r < t.R; ++r) {
            const int64_t q = t.req[(int64_t)c * t.R + r];
            if (q > 0 && t.alloc[(int64_t)m * t.R + r] - t.init_req[(int64_t)m * t.R + r] < q) ok = false;
        }
    }
    const uint64_t b = cs::ballot(ok);
    if (lane == 0) fit0[(int64_t)c * S + w] = b;
}

// BULK_: runs of unhinted pods of one class are placed a word of nodes at a time (schedule_run).  The host picks the instantiation by the call's own
// runs (lean_bulk_): a call of short runs keeps the pod-by-pod kernel as it was — the second code path costs it registers (3.55 -> 4.09 ms on the
// bench's 5000-node row when both lived in one kernel).
// GLOG_ (with BULK_): the log lives in HBM instead of LDS — calls whose live moves outgrow what LDS has left next to the node state, or with more
// than 65 536 pods (32-bit pod indices there).  Same walk, same squeeze; the wave orders its own global traffic with a workgroup fence where the
// LDS form has a wave barrier.  Sized for the worst case in every part: it never gives up.
template <int RMAX_, bool BULK_, bool GLOG_ = false>
CS_GLOBAL CS_LAUNCH_BOUNDS(64, 1) void removals_lean_kernel(DevTables t, SchedArgs a, const uint64_t* CS_RESTRICT fit0, int log_cap) {
    const int lane = cs::lane();
    const int R = RMAX_ == 2 ? 2 : t.R;   // (n_res >= 2 always: the two-lane instantiation knows its lane count, every `r < R` folds away)
    const int N = a.N, cap = a.cap, S = cap >> 6, C = a.C;
    char* smem = cs::dyn_smem();
    uint64_t* fit = (uint64_t*)smem;                 // [C][S]
    uint64_t* accb = fit + (int64_t)C * S;           // [S] acceptable (the hinted node's test)
    uint64_t* scanb = accb + S;                      // [S] acceptable && !Spec.Unschedulable (the scan's)
    uint64_t* alive = scanb + S;                     // [S] node still in the snapshot's list
    uint64_t* arrived = alive + S;                   // [S] node took pods of a committed removal
    uint32_t* wpre = (uint32_t*)(arrived + S);       // [S] live nodes in front of the word
    int32_t* txn_node = (int32_t*)(wpre + ((S + 1) & ~1));   // [kLeanTxnCap] the running transaction: destination of its i-th listed pod,
    int32_t* txn_ref = txn_node + kLeanTxnCap;               //               the pod's flat index,
    int32_t* txn_cls = txn_ref + kLeanTxnCap;                //               its class
    int64_t* sfree = (int64_t*)(txn_cls + kLeanTxnCap);      // [R][cap]
    int32_t* sslots = (int32_t*)(sfree + (int64_t)R * cap);  // [cap]
    // committed moves in commit order (what a later candidate that received pods lists again): destination, pod, class
    typedef typename std::conditional<GLOG_, uint32_t, uint16_t>::type ref_t;
    uint16_t* llog_dest = GLOG_ ? a.glog_dest : (uint16_t*)(sslots + cap);                       // [log_cap] (8-byte aligned: cap is a multiple of 64)
    ref_t* llog_ref = GLOG_ ? (ref_t*)a.glog_ref : (ref_t*)((uint16_t*)(sslots + cap) + log_cap);   // [log_cap]
    uint8_t* llog_cls = GLOG_ ? a.glog_cls : (uint8_t*)((uint16_t*)(sslots + cap) + 2 * (int64_t)log_cap);   // [log_cap]
    auto log_order = [&]() { if (GLOG_) cs::wave_sync(); else cs::lds_order(); };   // lanes read what other lanes wrote to the log
    // BULK_: the log in kLeanLogBuckets parts of log_cap / kLeanLogBuckets entries, an entry in the part of its destination & 7 — a candidate that
    // received pods walks ONE part (what it is after sits there in commit order); with 40-100 pods per candidate and every node a candidate
    // (BenchmarkRunOnceScaleDown) the walk of one undivided log was most of a candidate's time.  Lane b holds part b's fill.
    constexpr int B = BULK_ ? kLeanLogBuckets : 1;
    const int cap_b = log_cap / B;
    int32_t my_log_n = 0;

    // ---- prologue: node state = what the running pods of each node hold ----
    for (int m = lane; m < cap; m += 64) {
        const bool live = m < N;
        for (int r = 0; r < R; ++r) sfree[(int64_t)r * cap + m] = live ? t.alloc[(int64_t)m * R + r] - t.init_req[(int64_t)m * R + r] : 0;
        sslots[m] = live ? t.allowed[m] - t.init_pods[m] : 0;
        const bool acc = live && (a.acceptable == nullptr || a.acceptable[m] != 0);
        const uint64_t ab = cs::ballot(acc);
        const uint64_t sb = cs::ballot(acc && !(t.gflags[live ? m : 0] & CASIM_NG_UNSCHEDULABLE));
        const uint64_t lb = cs::ballot(live);
        if (lane == 0) {
            const int w = m >> 6;
            accb[w] = ab; scanb[w] = sb; alive[w] = lb; arrived[w] = 0ull;
            wpre[w] = (uint32_t)(w << 6) < (uint32_t)N ? (uint32_t)(w << 6) : (uint32_t)N;
        }
    }
    for (int i = lane; i < C * S; i += 64) fit[i] = fit0[i];
    // lane c holds class c's request: a node is re-evaluated against every class in one step
    int64_t my_q[RMAX_];
#pragma unroll
    for (int r = 0; r < RMAX_; ++r) my_q[r] = (lane < C && r < R) ? t.req[(int64_t)lane * R + r] : 0;
    cs::sync();

    int32_t last_index = a.last_index < -1 ? -1 : a.last_index;
    int32_t scheduled = 0, runs_done = 0, n_alive = N, removed = 0, cand_done = 0, log_n = 0, ext_n = 0;

    // First node at or behind list POSITION u0, in cyclic order, that passes for class c (RunFiltersUntilPassingNode, plugin_runner.go:54-143);
    // its own position comes back in pos_out (MarkMatch :138), -1 = no node passes.  Positions count the nodes still in the list: lane w
    // holds word w's live bits and how many live nodes precede it, so "position >= u0" is all of a word, none of it, or — in the one word
    // that holds position u0 — the bits from the u0-th live node on.  One round trip to LDS per 64 words; everything else in registers.
    auto find = [&](int c, uint32_t u0, int32_t& pos_out) -> int32_t {
        const uint64_t* row = fit + (int64_t)c * S;
        for (int pass = 0; pass < 2; ++pass) {   // at or behind u0, then (wrapped) in front of it
            for (int base = pass == 0 ? (int)((u0 >> 6) & ~63u) : 0; base < S; base += 64) {   // (a node's position never exceeds its index)
                // (every lane loads — lanes past the last word re-read it and drop what they got: four loads in flight, ONE wait; as guarded
                // loads each sat in a branch of its own behind the previous one's wait)
                const int w = base + lane;
                const bool in = w < S;
                const int wc = in ? w : S - 1;
                uint64_t al = alive[wc], x = row[wc] & scanb[wc];
                uint32_t lo = wpre[wc];
                if (!in) { al = 0ull; x = 0ull; lo = 0xffffffffu; }
                const uint32_t hi = lo + (uint32_t)cs::popc64(al);
                const bool bound = in && lo < u0 && hi > u0;   // the word that holds position u0
                uint64_t keep = pass == 0 ? ((in && lo >= u0) ? ~0ull : 0ull) : ((in && hi <= u0) ? ~0ull : 0ull);
                const uint64_t bb = cs::ballot(bound);
                if (bb != 0ull) {
                    const int jb = cs::ffs64(bb);
                    const uint64_t xa = cs::bcast_u64(al, jb);
                    const uint32_t r = u0 - cs::bcast_u32(lo, jb);
                    // the live node with exactly r live nodes below it
                    const int bit = cs::ffs64(cs::ballot(((xa >> lane) & 1ull) && (uint32_t)cs::mbcnt(xa) == r));
                    if (lane == jb) keep = pass == 0 ? ~cs::low_mask(bit) : cs::low_mask(bit);
                }
                const uint64_t y = x & keep;
                const uint64_t b = cs::ballot(y != 0ull);
                if (b != 0ull) {
                    const int j = cs::ffs64(b);
                    const int bit = cs::ffs64(cs::bcast_u64(y, j));
                    pos_out = (int32_t)cs::bcast_u32(lo, j) + cs::popc64(cs::bcast_u64(al, j) & cs::low_mask(bit));
                    return (int32_t)((base + j) << 6) + bit;
                }
                if (pass == 1 && bb != 0ull) break;   // (nothing in front of u0 lies behind its word)
            }
        }
        return -1;
    };
    // NodeInfo.AddPod of one pod of class c on node m (dir = +1) or its revert (dir = -1); every class's bit of the node follows
    auto move_pod = [&](int c, int m, int dir) {
        // (all lanes' reads go out together, lanes past R re-read lane 0 and are never looked at: my_q is zero there.  Guarded by r < R
        // each read sat behind the previous one's wait)
        int64_t f[RMAX_];
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) f[r] = sfree[(int64_t)(r < R ? r : 0) * cap + m];
        const int32_t sl = sslots[m] - dir;
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) {
            const int64_t q = (int64_t)cs::bcast_u64((uint64_t)my_q[r], c);
            f[r] -= dir > 0 ? q : -q;
        }
        cs::lds_order();   // (everybody has read the node before its record changes)
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RMAX_; ++r) if (r < R) sfree[(int64_t)r * cap + m] = f[r];
            sslots[m] = sl;
        }
        bool fits = sl > 0;
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) if (r < R && my_q[r] > 0 && f[r] < my_q[r]) fits = false;
        if (lane < C) {
            uint64_t* word = fit + (int64_t)lane * S + (m >> 6);
            const uint64_t bit = 1ull << (m & 63);
            if (dir > 0) { if (!fits) cs::lds_and_u64(word, ~bit); }   // (a fuller node never starts to fit)
            else {
                const bool stat = (a.fbits[(int64_t)lane * S + (m >> 6)] >> (m & 63)) & 1ull;
                if (fits && stat) cs::lds_or_u64(word, bit); else cs::lds_and_u64(word, ~bit);
            }
        }
        cs::lds_order();   // (the masks are read by all lanes in the next search)
    };

    // Entries of the log whose destination has left the list are dead: a node that goes takes every pod it had received along (they were listed
    // again and their moves logged anew).  Squeezing them out keeps the order of the rest — what a later listing walks.  Done when a commit
    // does not fit.
    auto part_n = [&](int b) -> int32_t { return B == 1 ? log_n : (int32_t)cs::bcast_u32((uint32_t)my_log_n, b); };
    auto squeeze_log = [&]() {
        for (int b = 0; b < B; ++b) {
            const int32_t nb = part_n(b);
            uint16_t* ld = llog_dest + (int64_t)b * cap_b; ref_t* lr = llog_ref + (int64_t)b * cap_b; uint8_t* lc = llog_cls + (int64_t)b * cap_b;
            int32_t keep_n = 0;
            for (int j0 = 0; j0 < nb; j0 += 64) {
                const int jj = j0 + lane;
                const bool in = jj < nb;
                const uint32_t d = in ? ld[jj] : 0u, rf = in ? lr[jj] : 0u, cl = in ? lc[jj] : 0u;
                const bool live = in && ((alive[d >> 6] >> (d & 63)) & 1ull) != 0ull;
                const uint64_t kb = cs::ballot(live);
                log_order();   // (every lane holds its entry before any slot is overwritten: slots only move down)
                if (live) { const int o = keep_n + cs::mbcnt(kb); ld[o] = (uint16_t)d; lr[o] = (ref_t)rf; lc[o] = (uint8_t)cl; }
                keep_n += cs::popc64(kb);
                log_order();
            }
            if (B == 1) log_n = keep_n; else if (lane == b) my_log_n = keep_n;
        }
    };

    int32_t my_cand = 0, my_rlo = 0, my_rhi = 0, my_plo = 0, my_phi = 0, my_atomic = 0;   // candidate records kc & ~63 .., one per lane
    int cstart = 0, cend = 0;                                               // run records [cstart, cend), one per lane
    int32_t my_class = 0, my_count = 0, my_hint = -1, my_first = 0;

    for (int kc = 0; kc < a.n_cand; ++kc) {
        // ---- SimulateNodeRemoval (cluster.go:131-172) of candidate kc, planner order (planner.go:300-330) ----
        if (a.max_removable > 0 && removed >= a.max_removable) break;
        if ((kc & 63) == 0) {
            const int kk = kc + lane < a.n_cand ? kc + lane : a.n_cand - 1;   // (every lane loads: lanes past the end re-read the last record)
            const cs::Words<4> r0 = cs::load4((const uint32_t*)a.lean_cand + 4 * (int64_t)kk), r1 = cs::load4((const uint32_t*)a.lean_cand + 4 * (int64_t)(kk + 1));
            my_cand = (int32_t)r0.w[0]; my_rlo = (int32_t)r0.w[1]; my_plo = (int32_t)r0.w[2]; my_atomic = (int32_t)r0.w[3];
            my_rhi = (int32_t)r1.w[1]; my_phi = (int32_t)r1.w[2];
        }
        const int Y = (int)cs::bcast_u32((uint32_t)my_cand, kc & 63);
        const int p_lo = (int)cs::bcast_u32((uint32_t)my_plo, kc & 63), p_hi = (int)cs::bcast_u32((uint32_t)my_phi, kc & 63);
        const int n_own = p_hi - p_lo;
        const uint64_t ybit = 1ull << (Y & 63);
        const bool has_arrivals = (arrived[Y >> 6] & ybit) != 0ull, is_alive = (alive[Y >> 6] & ybit) != 0ull;
        int e_lo = 0, e_hi = 0;
        if (has_arrivals) {
            // Pods that earlier committed removals moved onto this node are listed after its own, in commit order (see sched_kernel): the
            // log, four entries per lane and step; a step without a hit costs one compare round.  Each hit goes to the ext tables and, with
            // its class, into the transaction's ring right behind the node's own pods.
            if (a.ext_cap <= 0) break;
            const int lb = B == 1 ? 0 : (Y & (B - 1));
            const int32_t ln = part_n(lb);
            const uint16_t* ld = llog_dest + (int64_t)lb * cap_b; const ref_t* lr = llog_ref + (int64_t)lb * cap_b; const uint8_t* lc = llog_cls + (int64_t)lb * cap_b;
            uint32_t found = 0;
            bool bad = false;
            for (int j0 = 0; j0 < ln; j0 += 256) {
                const int jj = j0 + lane * 4;
                const uint64_t d4 = *(const uint64_t*)(ld + jj);
                bool h[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = jj + k < ln && (int)((d4 >> (16 * k)) & 0xffffull) == Y;
                if (cs::ballot(h[0] || h[1] || h[2] || h[3]) == 0ull) continue;
                uint32_t before = 0, total = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) { const uint64_t bk = cs::ballot(h[k]); before += (uint32_t)cs::mbcnt(bk); total += (uint32_t)cs::popc64(bk); }
                uint32_t idx = found + before;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!h[k]) continue;
                    const int ref = (int)lr[jj + k], cls = (int)lc[jj + k];
                    const uint32_t pos = (uint32_t)ext_n + idx;
                    if (pos < (uint32_t)a.ext_cap) { a.ext_ref[pos] = ref; a.ext_cand[pos] = kc; }
                    if ((uint32_t)n_own + idx < (uint32_t)kLeanTxnCap) { txn_ref[n_own + idx] = ref; txn_cls[n_own + idx] = cls; }
                    if (a.pod_sticky && a.pod_sticky[ref]) bad = true;
                    idx++;
                }
                found += total;
            }
            if (cs::ballot(bad) != 0ull || (uint32_t)ext_n + found > (uint32_t)a.ext_cap) break;
            e_lo = ext_n; e_hi = ext_n + (int32_t)found; ext_n = e_hi;
        }
        cand_done = kc + 1;
        if (!is_alive) {   // NoNodeInfo (:139-147)
            if (lane == 0) a.removable_out[kc] = 0;
            continue;
        }
        const int run_lo = (int)cs::bcast_u32((uint32_t)my_rlo, kc & 63), run_hi = (int)cs::bcast_u32((uint32_t)my_rhi, kc & 63);
        const int n_listed = n_own + (e_hi - e_lo);
        if (n_listed > kLeanTxnCap) cs::sync();   // (the tail of this transaction goes through HBM: ext_ref, node_out)
        auto slot_of = [&](int i) -> int { return i < n_own ? p_lo + i : a.P + e_lo + (i - n_own); };
        // Fork; the candidate turns into a pod-less tainted ghost that keeps its list position (:243-265)
        if (lane == 0) { cs::lds_and_u64(accb + (Y >> 6), ~ybit); cs::lds_and_u64(scanb + (Y >> 6), ~ybit); }
        cs::lds_order();

        bool failed = false;
        int n_done = 0;   // listed pods placed so far (they are tried in listing order; the first miss ends the transaction)
        // one pod of class c (hinted or not), the n_done-th listed pod of the transaction: tryScheduleUsingHints, then trySchedule (:86-135)
        auto schedule_pod = [&](int c, int32_t hint, int slot, int ref) {
            int32_t m = -1;
            if (hint >= 0 && hint < N && (((fit[(int64_t)c * S + (hint >> 6)] & accb[hint >> 6]) >> (hint & 63)) & 1ull)) m = hint;   // (no lastIndex update)
            else {
                uint32_t u0 = (uint32_t)last_index + 1u;
                if (u0 >= (uint32_t)n_alive) u0 %= (uint32_t)n_alive;
                int32_t pos = 0;
                m = find(c, u0, pos);
                if (m >= 0) last_index = pos;   // MarkMatch (plugin_runner.go:138)
            }
            if (m < 0) { failed = true; return; }      // breakOnFailure (:79-81)
            if (lane == 0) {
                a.node_out[slot] = m;
                if (n_done < kLeanTxnCap) { txn_node[n_done] = m; txn_ref[n_done] = ref; txn_cls[n_done] = c; }
            }
            move_pod(c, m, +1);
            scheduled++; n_done++;
        };
        // `cnt` unhinted pods of class c in a row (a node's replicas of one controller; BenchmarkRunOnceScaleDown moves nothing else): pod by pod each
        // of them is a search + a node update, two LDS round trips.  But the search of pod i + 1 starts behind the node pod i took (MarkMatch), and a
        // placement only changes the bits of ITS node: until the walk comes round to where it started, the nodes the pods take are simply the class's
        // passing nodes in cyclic list order from lastIndex + 1, one pod each.  So a run is placed a WORD of nodes at a time: the passing nodes of the
        // word (lanes = its nodes) take the next pods in lane order (rank = mbcnt), their records are updated side by side and every class's bit
        // of those nodes follows (one ballot per class).  Coming round (more pods than passing nodes) starts the next walk with the masks as they
        // are then — exactly where the pod-by-pod loop would be.  ref_base < 0: arrived pods (their refs and classes sit in the ring already).
        auto schedule_run = [&](int c, int32_t cnt, int slot_base, int ref_base) {
            if (!BULK_) return;
            const uint64_t* row = fit + (int64_t)c * S;
            int64_t q[RMAX_];
#pragma unroll
            for (int r = 0; r < RMAX_; ++r) q[r] = (int64_t)cs::bcast_u64((uint64_t)my_q[r], c);
            int32_t left = cnt, done = 0;
            while (left > 0) {
                uint32_t u0 = (uint32_t)last_index + 1u;
                if (u0 >= (uint32_t)n_alive) u0 %= (uint32_t)n_alive;
                const int32_t left_before = left;
                for (int pass = 0; pass < 2 && left > 0; ++pass) {   // at or behind u0, then (wrapped) in front of it: as in find()
                    for (int base = pass == 0 ? (int)((u0 >> 6) & ~63u) : 0; base < S && left > 0; base += 64) {
                        const int w0 = base + lane;
                        const bool in = w0 < S;
                        const int wc = in ? w0 : S - 1;
                        uint64_t al = alive[wc], x = row[wc] & scanb[wc];
                        uint32_t lo = wpre[wc];
                        if (!in) { al = 0ull; x = 0ull; lo = 0xffffffffu; }
                        const uint32_t hi = lo + (uint32_t)cs::popc64(al);
                        const bool bound = in && lo < u0 && hi > u0;
                        uint64_t keep = pass == 0 ? ((in && lo >= u0) ? ~0ull : 0ull) : ((in && hi <= u0) ? ~0ull : 0ull);
                        const uint64_t bb = cs::ballot(bound);
                        if (bb != 0ull) {
                            const int jb = cs::ffs64(bb);
                            const uint64_t xa = cs::bcast_u64(al, jb);
                            const uint32_t rr = u0 - cs::bcast_u32(lo, jb);
                            const int bit = cs::ffs64(cs::ballot(((xa >> lane) & 1ull) && (uint32_t)cs::mbcnt(xa) == rr));
                            if (lane == jb) keep = pass == 0 ? ~cs::low_mask(bit) : cs::low_mask(bit);
                        }
                        const uint64_t y = x & keep;
                        uint64_t b = cs::ballot(y != 0ull);
                        while (b != 0ull && left > 0) {   // the words of this block that hold passing nodes, in list order
                            const int j = cs::ffs64(b);
                            b &= b - 1ull;
                            const uint64_t yw = cs::bcast_u64(y, j);
                            const int w = base + j;
                            const int m = (w << 6) + lane;                       // lanes = the word's nodes
                            const int rank = cs::mbcnt(yw);
                            const int avail = cs::popc64(yw);
                            const int take = avail < left ? avail : left;
                            const bool sel = ((yw >> lane) & 1ull) != 0ull && rank < take;
                            if (sel) {
                                const int i = done + rank, ti = n_done + rank;
                                a.node_out[slot_base + i] = m;
                                if (ti < kLeanTxnCap) {
                                    txn_node[ti] = m;
                                    if (ref_base >= 0) { txn_ref[ti] = ref_base + i; txn_cls[ti] = c; }
                                }
                            }
                            int64_t f[RMAX_];
#pragma unroll
                            for (int r = 0; r < RMAX_; ++r) f[r] = sfree[(int64_t)(r < R ? r : 0) * cap + m] - (sel ? q[r] : 0);
                            const int32_t sl = sslots[m] - (sel ? 1 : 0);
                            if (sel) {
#pragma unroll
                                for (int r = 0; r < RMAX_; ++r) if (r < R) sfree[(int64_t)r * cap + m] = f[r];
                                sslots[m] = sl;
                            }
                            for (int cc = 0; cc < C; ++cc) {   // a fuller node never starts to fit: bits only fall
                                bool fits = sl > 0;
#pragma unroll
                                for (int r = 0; r < RMAX_; ++r) {
                                    const int64_t qq = (int64_t)cs::bcast_u64((uint64_t)my_q[r], cc);
                                    if (r < R && qq > 0 && f[r] < qq) fits = false;
                                }
                                const uint64_t nb = cs::ballot(sel && !fits);
                                if (nb != 0ull && lane == 0) cs::lds_and_u64(fit + (int64_t)cc * S + w, ~nb);
                            }
                            // MarkMatch of the run's last pod so far: the position of the highest node taken
                            const int bit_last = cs::fls64(cs::ballot(sel));
                            last_index = (int32_t)cs::bcast_u32(lo, j) + cs::popc64(cs::bcast_u64(al, j) & cs::low_mask(bit_last));
                            left -= take; done += take; n_done += take; scheduled += take;
                            cs::lds_order();
                        }
                        if (pass == 1 && bb != 0ull) break;
                    }
                }
                if (left == left_before) { failed = true; return; }   // nothing passes any more: breakOnFailure (:79-81)
            }
        };
        // ---- the candidate's own pods: runs of one class ----
        for (int k = run_lo; k < run_hi && !failed; ++k) {
            if (k < cstart || k >= cend) {
                cstart = k;
                cend = cstart + 64 < a.n_runs ? cstart + 64 : a.n_runs;
                const int kk = cstart + lane < cend ? cstart + lane : cend - 1;
                const cs::Words<4> rr = cs::load4((const uint32_t*)a.lean_run + 4 * (int64_t)kk);
                my_class = (int32_t)rr.w[0]; my_count = (int32_t)rr.w[1]; my_hint = (int32_t)rr.w[2]; my_first = (int32_t)rr.w[3];
            }
            const int j = k - cstart;
            const int c = (int)cs::bcast_u32((uint32_t)my_class, j);
            const int32_t cnt = (int32_t)cs::bcast_u32((uint32_t)my_count, j);
            const int32_t hint = (int32_t)cs::bcast_u32((uint32_t)my_hint, j);
            const int32_t first = (int32_t)cs::bcast_u32((uint32_t)my_first, j);
            runs_done++;
            if (BULK_ && hint < 0 && cnt >= a.lean_bulk_min) schedule_run(c, cnt, first, first);
            else for (int32_t i = 0; i < cnt && !failed; ++i) schedule_pod(c, hint, first + i, first + i);
        }
        // ---- then the pods that arrived, in the order they were listed ----
        for (int e0 = e_lo; e0 < e_hi && !failed; e0 += 64) {
            const int ee = e0 + lane, ti = n_own + (ee - e_lo);
            int32_t my_r = 0, my_c = 0;
            if (ee < e_hi) {
                my_r = ti < kLeanTxnCap ? txn_ref[ti] : a.ext_ref[ee];
                my_c = ti < kLeanTxnCap ? txn_cls[ti] : a.pod_class[my_r];
            }
            const int lim = e_hi - e0 < 64 ? e_hi - e0 : 64;
            if (!BULK_) {
                for (int u = 0; u < lim && !failed; ++u) {
                    runs_done++;
                    schedule_pod((int)cs::bcast_u32((uint32_t)my_c, u), -1, a.P + e0 + u, (int)cs::bcast_u32((uint32_t)my_r, u));
                }
                continue;
            }
            for (int u = 0; u < lim && !failed;) {
                // arrived pods of one class in a row are a run like any other
                const int c = (int)cs::bcast_u32((uint32_t)my_c, u);
                const uint64_t same = cs::ballot(ee < e_hi && my_c == c) >> u;   // bit 0 = lane u
                int len = cs::ffs64(~same);
                if (len < 0 || len > lim - u) len = lim - u;
                runs_done += len;
                if (len >= a.lean_bulk_min) schedule_run(c, len, a.P + e0 + u, -1);
                else for (int i = 0; i < len && !failed; ++i) schedule_pod(c, -1, a.P + e0 + u + i, (int)cs::bcast_u32((uint32_t)my_r, u + i));
                u += len;
            }
        }
        // ---- every pod found a place <=> the node is removable (findPlaceFor :219-224) ----
        const bool ok = !failed;
        if (n_listed > kLeanTxnCap) cs::sync();
        auto placed_at = [&](int i) -> int32_t { return i < kLeanTxnCap ? txn_node[i] : a.node_out[slot_of(i)]; };
        auto ref_at = [&](int i) -> int32_t { return i < n_own ? p_lo + i : (i < kLeanTxnCap ? txn_ref[i] : a.ext_ref[e_lo + (i - n_own)]); };
        auto class_at = [&](int i) -> int32_t { return i < kLeanTxnCap ? txn_cls[i] : a.pod_class[ref_at(i)]; };
        if (ok && a.persist) {
            // The log was sized to what LDS holds when the worst case (every pod of the call and every ext slot committed) does not fit: a
            // commit that would run past it — also after squeezing — ends the kernel with out[5] = 1 and the host runs the call again
            // (fetch_removals: with the log in HBM, or through K_sched).
            if (B == 1) {
                if (log_n + n_listed > log_cap) {
                    squeeze_log();
                    if (log_n + n_listed > log_cap) { if (lane == 0) a.out[5] = 1; return; }
                }
                // Commit (withForkedSnapshot :174-188): the ghost leaves the list (:230) and the destination set (planner.go:318)
                for (int i = lane; i < n_listed; i += 64) {
                    const int m = placed_at(i);
                    cs::lds_or_u64(arrived + (m >> 6), 1ull << (m & 63));
                    llog_dest[log_n + i] = (uint16_t)m; llog_ref[log_n + i] = (ref_t)ref_at(i); llog_cls[log_n + i] = (uint8_t)class_at(i);
                }
                log_n += n_listed;
            } else {
                // what each part of the log is about to receive (lane b counts part b), then — room permitting — the entries, part by part in
                // listing order
                int32_t my_add = 0;
                for (int i0 = 0; i0 < n_listed; i0 += 64) {
                    const bool v = i0 + lane < n_listed;
                    const int m = v ? placed_at(i0 + lane) : 0;
                    for (int b = 0; b < B; ++b) { const uint64_t mb = cs::ballot(v && (m & (B - 1)) == b); if (lane == b) my_add += cs::popc64(mb); }
                }
                if (cs::ballot(lane < B && my_log_n + my_add > cap_b) != 0ull) {
                    squeeze_log();
                    if (cs::ballot(lane < B && my_log_n + my_add > cap_b) != 0ull) { if (lane == 0) a.out[5] = 1; return; }
                }
                for (int i0 = 0; i0 < n_listed; i0 += 64) {
                    const int i = i0 + lane;
                    const bool v = i < n_listed;
                    const int m = v ? placed_at(i) : 0;
                    const int32_t rf = v ? ref_at(i) : 0, cl = v ? class_at(i) : 0;
                    if (v) cs::lds_or_u64(arrived + (m >> 6), 1ull << (m & 63));
                    for (int b = 0; b < B; ++b) {
                        const bool mine = v && (m & (B - 1)) == b;
                        const uint64_t mb = cs::ballot(mine);
                        if (mb == 0ull) continue;
                        const int32_t at = (int32_t)cs::bcast_u32((uint32_t)my_log_n, b);
                        if (mine) { const int64_t o = (int64_t)b * cap_b + at + cs::mbcnt(mb); llog_dest[o] = (uint16_t)m; llog_ref[o] = (ref_t)rf; llog_cls[o] = (uint8_t)cl; }
                        if (lane == b) my_log_n += cs::popc64(mb);
                    }
                }
            }
            if (lane == 0) cs::lds_and_u64(alive + (Y >> 6), ~ybit);
            for (int w = (Y >> 6) + 1 + lane; w < S; w += 64) cs::lds_sub_u32(wpre + w, 1u);
            n_alive--;
        } else {
            // Revert: every destination gets its pod's amounts back, the candidate its place among the destinations
            for (int i0 = 0; i0 < n_done; i0 += 64) {
                const int ii = i0 + lane;
                const int32_t mi = ii < n_done ? placed_at(ii) : -1;
                const int32_t ci = ii < n_done ? class_at(ii) : 0;
                const int lim = n_done - i0 < 64 ? n_done - i0 : 64;
                for (int u = 0; u < lim; ++u) move_pod((int)cs::bcast_u32((uint32_t)ci, u), (int32_t)cs::bcast_u32((uint32_t)mi, u), -1);
            }
            if (lane == 0) {
                const bool acc = a.acceptable == nullptr || a.acceptable[Y] != 0;
                if (acc) {
                    cs::lds_or_u64(accb + (Y >> 6), ybit);
                    if (!(t.gflags[Y] & CASIM_NG_UNSCHEDULABLE)) cs::lds_or_u64(scanb + (Y >> 6), ybit);
                }
            }
        }
        cs::lds_order();
        if (GLOG_) cs::wave_sync();   // (the entries just logged are in HBM: the next candidate's listing reads them through other lanes)
        if (ok && cs::bcast_u32((uint32_t)my_atomic, kc & 63) == 0u) removed++;   // len(removableList) - atomicScaleDownNodesCount
        if (lane == 0) a.removable_out[kc] = ok ? 1 : 0;
    }
    if (lane == 0) {
        a.out[0] = last_index;
        a.out[1] = scheduled;
        a.out[2] = runs_done;
        a.out[3] = cand_done;
        a.out[4] = ext_n;
        a.out[5] = 0;
    }
}


// ---- host side: one TrySchedulePods call ------------------------------------------------------------
// which kernel the calling thread's last removal simulation ran as (casim_last_removals_info): [0] 1 = removals_lean_kernel, [1] threads of the
// block, [2] node state in LDS, [3] runs
inline int32_t* last_removals_info() { static thread_local int32_t info[4] = {0, 0, 0, 0}; return info; }

template <class BK>
class SchedulerT {
public:
    explicit SchedulerT(BK& bk) : bk_(bk) {}
    ~SchedulerT() { for (void* p : allocs_) bk_.free(p); }
    SchedulerT(const SchedulerT&) = delete;
    SchedulerT& operator=(const SchedulerT&) = delete;

    // returns CASIM_OK, CASIM_NG_UNSUPPORTED (> 0) or an error code (< 0)
    // removal simulation: the pods of all candidates form the sequence, every candidate is one transaction
    int32_t init_removals(const casim_pegs* p, const casim_groups* g, const casim_removal_candidates* rc) {
        if (!rc) return fail(CASIM_ERR_INVALID, "null candidates");
        if (rc->n_candidates < 0) return fail(CASIM_ERR_INVALID, "negative size");
        if (rc->n_candidates > 0 && (!rc->cand_node || !rc->pod_offsets)) return fail(CASIM_ERR_INVALID, "candidate table has null columns");
        const int K = rc->n_candidates;
        if (K > 0 && rc->pod_offsets[0] != 0) return fail(CASIM_ERR_INVALID, "pod_offsets[0] != 0");
        for (int k = 0; k < K; ++k) {
            if (rc->pod_offsets[k + 1] < rc->pod_offsets[k]) return fail(CASIM_ERR_INVALID, "pod_offsets not monotone");
            if (!g || rc->cand_node[k] < 0 || rc->cand_node[k] >= g->n_groups) return fail(CASIM_ERR_INVALID, "candidate node out of range");
        }
        casim_pod_sequence q; memset(&q, 0, sizeof q);
        q.n_pods = K > 0 ? rc->pod_offsets[K] : 0;
        q.pod_class = rc->pod_class; q.hint_node = rc->hint_node; q.node_acceptable = rc->destination;
        q.break_on_failure = 1; q.last_index = rc->last_index; q.rules = rc->rules;
        return init(p, g, &q, rc);
    }

    int32_t init(const casim_pegs* p, const casim_groups* g, const casim_pod_sequence* q, const casim_removal_candidates* cand = nullptr) {
        if (!p || !g || !q) return fail(CASIM_ERR_INVALID, "null table");
        if (p->n_pegs < 0 || g->n_groups < 0 || q->n_pods < 0) return fail(CASIM_ERR_INVALID, "negative size");
        if (p->n_res < 2 || p->n_res > CASIM_KMAX_RES) return fail(CASIM_ERR_INVALID, "n_res must be in [2, 8]");
        if (p->w_taint < 0 || p->w_label < 0 || p->w_excl < 0 || p->w_zone < 0) return fail(CASIM_ERR_INVALID, "negative mask width");
        C_ = p->n_pegs; N_ = g->n_groups; P_ = q->n_pods;
        const size_t C = (size_t)C_, N = (size_t)N_, P = (size_t)P_;
        if (P > 0 && !q->pod_class) return fail(CASIM_ERR_INVALID, "pod_class is null");
        if (C > 0 && (!p->req || !p->flags)) return fail(CASIM_ERR_INVALID, "class table has null columns");
        if (N > 0 && (!g->alloc || !g->init_req || !g->allowed_pods || !g->init_pods || !g->flags))
            return fail(CASIM_ERR_INVALID, "node table has null columns");
        if (C > 0 && ((p->w_taint && !p->tol_mask) || (p->w_label && !p->sel_mask) || (p->w_excl && (!p->excl_block || !p->excl_mark)) ||
                      (p->w_zone && (!p->zone_block || !p->zone_mark))))
            return fail(CASIM_ERR_INVALID, "class mask column missing");
        if (N > 0 && ((p->w_taint && !g->taint_mask) || (p->w_label && !g->label_mask) || (p->w_excl && !g->init_excl)))
            return fail(CASIM_ERR_INVALID, "node mask column missing");
        if (N > 0x3fffffc0ull) return fail(CASIM_ERR_INVALID, "too many nodes");

        // ---- runs: consecutive pods of one class without a hint; predicates outside the subset -> delegate ----
        std::vector<int32_t> rc, rn, rh, rf, cro, rp, rk;      // class, count, hint, first pod, ..., (controller, class) pair, controller
        std::vector<int32_t> lc, lr;                           // the lean removal kernel's interleaved candidate / run records (live until the uploads have left)
        std::map<std::pair<int32_t, int32_t>, int32_t> pair_id;  // (similar_key, class) -> dense id
        std::map<int32_t, int32_t> ctrl_id;
        std::vector<uint8_t> used(C, 0);
        K_ = cand ? cand->n_candidates : 0;
        int next_cand = 0;
        for (size_t i = 0; i < P; ++i) {
            bool boundary = false;  // a run never spans two candidates
            while (cand && next_cand < K_ && (size_t)cand->pod_offsets[next_cand] <= i) { cro.push_back((int32_t)rc.size()); next_cand++; boundary = true; }
            const int32_t c = q->pod_class[i];
            if (c < 0 || c >= C_) return fail(CASIM_ERR_INVALID, "pod_class out of range");
            int32_t h = q->hint_node ? q->hint_node[i] : -1;
            if (h >= N_) h = -1;  // the hinted node left the cluster (:94-97)
            if (h < 0) h = -1;
            if (!used[(size_t)c]) {
                used[(size_t)c] = 1;
                if (p->flags[c] & (CASIM_PEG_UNSUPPORTED | CASIM_PEG_SELF_EXCL_ZONE)) return CASIM_NG_UNSUPPORTED;
                for (int w = 0; w < p->w_zone; ++w)
                    if (p->zone_block[(size_t)c * p->w_zone + w] | p->zone_mark[(size_t)c * p->w_zone + w]) return CASIM_NG_UNSUPPORTED;
            }
            int32_t pr = -1, ck = -1;
            if (q->similar_key && q->similar_key[i] >= 0) {
                auto ci = ctrl_id.find(q->similar_key[i]);
                if (ci == ctrl_id.end()) ci = ctrl_id.emplace(q->similar_key[i], (int32_t)ctrl_id.size()).first;
                ck = ci->second;
                auto pi = pair_id.find({ck, c});
                if (pi == pair_id.end()) pi = pair_id.emplace(std::make_pair(ck, c), (int32_t)pair_id.size()).first;
                pr = pi->second;
            }
            if (!boundary && h < 0 && !rc.empty() && rc.back() == c && rh.back() < 0 && rp.back() == pr && rn.back() < 0x7fffffff) rn.back()++;
            else { rc.push_back(c); rn.push_back(1); rh.push_back(h); rf.push_back((int32_t)i); rp.push_back(pr); rk.push_back(ck); }
        }
        n_runs_ = (int32_t)rc.size();
        while (cand && (int)cro.size() <= K_) cro.push_back(n_runs_);  // candidates without pods + the end marker
        if (N_ == 0 || (P_ == 0 && K_ == 0)) { trivial_ = true; last_index_ = q->last_index; return CASIM_OK; }

        memset(&dt_, 0, sizeof dt_); memset(&a_, 0, sizeof a_);
        dt_.G = C_; dt_.NG = N_; dt_.R = p->n_res; dt_.Wt = p->w_taint; dt_.Wl = p->w_label; dt_.Wx = p->w_excl; dt_.Wz = 0;
        const int R = dt_.R;
        // ---- ONE host-to-device copy for everything this call uploads (class / node tables unless resident, run tables, candidate and
        // pod columns, domain rules): the columns are packed into the backend's pinned staging buffer and land in one device slab, as
        // ProblemT does it — three dozen separate copies out of pageable memory were 0.2-0.3 ms of a 2.5 ms TrySchedulePods call
        // (profiles/r07e_bench.json: call 2.50 ms, kernels 2.21).  A column that does not fit the bound falls back on its own copy.
        {
            const size_t masks = (size_t)(dt_.Wt + dt_.Wl + dt_.Wx);
            size_t bound = 64 * 64 + 4096;
            if (!resident_) bound += C * (8 * (size_t)R + 8 + 8 * (masks + (size_t)dt_.Wx)) + N * (16 * (size_t)R + 16 + 8 * masks);
            bound += 4 * (rc.size() + rn.size() + rh.size() + rf.size() + rp.size() + rk.size() + cro.size()) + N;
            if (cand) bound += 16 * ((size_t)K_ + 1) + 8 * P + 16 * ((size_t)K_ + 2) + 16 * (rc.size() + 2);
            if (q->rules && q->rules->n_rules > 0 && q->rules->rule_offset) {
                const casim_domain_rules* r0 = q->rules;
                const size_t NR = (size_t)r0->n_rules, tot = (size_t)r0->rule_offset[NR];
                bound += 4 * (size_t)r0->n_keys * N + 64 * (NR + 1) + 8 * tot + (cand ? 4 * NR * N : 0) + 8 * (size_t)r0->n_elig_rows * ((N + 1023) / 64 + 16) +
                         8 * (C + 1) + 4 * (size_t)(r0->inc_off ? r0->inc_off[C] : 0);
            }
            if (getenv("CASIM_TEST_SMALL_UPLOAD_BOUND")) bound = 512;   // (tests: most columns take the fallback copy of their own)
            begin_uploads(bound);
        }
        if (resident_) {   // a resident cluster (ClusterT below): the tables are in HBM already, nothing to upload
            dt_.req = resident_->req; dt_.pflags = resident_->pflags; dt_.tol = resident_->tol; dt_.sel = resident_->sel;
            dt_.xblock = resident_->xblock; dt_.xmark = resident_->xmark; dt_.alloc = resident_->alloc; dt_.init_req = resident_->init_req;
            dt_.allowed = resident_->allowed; dt_.init_pods = resident_->init_pods; dt_.gflags = resident_->gflags;
            dt_.taint = resident_->taint; dt_.label = resident_->label; dt_.init_excl = resident_->init_excl;
        } else {
        dt_.req = up(p->req, C * R); dt_.pflags = up(p->flags, C);
        dt_.tol = up(p->tol_mask, C * dt_.Wt); dt_.sel = up(p->sel_mask, C * dt_.Wl);
        dt_.xblock = up(p->excl_block, C * dt_.Wx); dt_.xmark = up(p->excl_mark, C * dt_.Wx);
        dt_.alloc = up(g->alloc, N * R); dt_.init_req = up(g->init_req, N * R);
        dt_.allowed = up(g->allowed_pods, N); dt_.init_pods = up(g->init_pods, N); dt_.gflags = up(g->flags, N);
        dt_.taint = up(g->taint_mask, N * dt_.Wt); dt_.label = up(g->label_mask, N * dt_.Wl);
        dt_.init_excl = up(g->init_excl, N * dt_.Wx);
        }

        // one workgroup: 64..512 threads, node m -> thread m % T, chunk m / T
        // (r01w sweep: TrySchedulePods on 5000+ nodes is ~7 % faster with 512 threads — 2.31 vs 2.48 ms, 8.9 vs 9.5 ms —, the
        // removal loop does not care)
        int max_threads = cand ? kDefaultThreadsRemovals : (N_ >= 4096 ? 2 * kDefaultThreads : kDefaultThreads);
        if (const char* e = getenv("CASIM_SCHED_THREADS")) { const int v = atoi(e); if (v >= 64) max_threads = (v > 512 ? 512 : v) / 64 * 64; }   // (the kernel's launch bound)
        // the removal loop as one wave over per-class fit masks (removals_lean_kernel): no domain rules, no node-local exclusion state, a lane
        // per class, the node state in LDS.  CASIM_NO_LEAN_REMOVALS=1: sched_kernel as before (A/B, tests run both)
        lean_ = false;
        if (cand && K_ > 0 && !(q->rules && q->rules->n_rules > 0) && dt_.Wx == 0 && C_ <= 64 && R <= 4 &&
            !(getenv("CASIM_NO_LEAN_REMOVALS") && atoi(getenv("CASIM_NO_LEAN_REMOVALS")) != 0)) {
            bool plain = true;
            for (size_t c = 0; c < C; ++c) if (used[c] && (p->flags[c] & CASIM_PEG_SELF_EXCL_NODE)) plain = false;
            // (the log of committed moves sits in LDS as 16-bit node and pod indices)
            const int64_t worst = ((int64_t)P_ + (cand->ext_capacity > 0 ? cand->ext_capacity : 0) + 255) & ~255ll;
            const int64_t E0_worst_ = cand->ext_capacity > 0 ? cand->ext_capacity : 0;
            const int64_t capN = round_up64_((int64_t)N_);
            const int64_t fixed = casim_lean_removal_bytes(R, C_, capN, 0);
            const int64_t room = (((int64_t)bk_.lds_budget() - fixed - 8) / 5) & ~255ll;   // entries LDS has left for the log
            const int forced_cap = getenv("CASIM_LEAN_LOG_CAP") ? atoi(getenv("CASIM_LEAN_LOG_CAP")) : 0;   // tests: a small log, so that squeezing and giving up happen on small cases
            // which instantiation: the one that places runs a word of nodes at a time when the call has such runs of its own (pods that
            // arrive later travel in the runs they left in).  CASIM_LEAN_BULK_MIN: 0 = pod by pod always (A/B, tests run both)
            lean_bulk_min_ = 4;
            if (const char* ev = getenv("CASIM_LEAN_BULK_MIN")) { const int v = atoi(ev); lean_bulk_min_ = v <= 0 ? 0x7fffffff : (v < 2 ? 2 : v); }
            lean_bulk_ = false;
            const bool no_optimism = getenv("CASIM_NO_OPTIMISTIC_LOG") && atoi(getenv("CASIM_NO_OPTIMISTIC_LOG")) != 0;
            if (!no_optimism) for (size_t i = 0; i < rn.size(); ++i) if (rh[i] < 0 && rn[i] >= lean_bulk_min_) { lean_bulk_ = true; break; }
            lean_optimistic_ = false;
            if (lean_bulk_) {
                // its log comes in kLeanLogBuckets parts (by destination): each part gets an equal share of what LDS has left, never more than
                // the worst case.  A part can fill up before the whole would have: this instantiation may always give up (the same kernel with its
                // log in HBM then answers — switch_to_glog_ — or, where that is not possible, K_sched).
                int64_t part = worst < room / kLeanLogBuckets ? worst : (room / kLeanLogBuckets) & ~255ll;
                if (forced_cap >= 64 && forced_cap / kLeanLogBuckets < part) part = forced_cap / kLeanLogBuckets < 64 ? 64 : (forced_cap / kLeanLogBuckets) & ~63ll;
                if (part >= 64 && (part * kLeanLogBuckets * 2 >= (int64_t)P_ || forced_cap >= 64)) {
                    lean_log_cap_ = (int32_t)(part * kLeanLogBuckets); lean_optimistic_ = true;
                    lean_smem_ = (size_t)casim_lean_removal_bytes(R, C_, capN, lean_log_cap_);
                } else lean_bulk_ = false;
            }
            if (!lean_bulk_) {
                lean_log_cap_ = (int32_t)worst;
                lean_smem_ = (size_t)casim_lean_removal_bytes(R, C_, capN, lean_log_cap_);
                // The worst case — every pod of the call and every ext slot a committed move — rarely happens: a removal that fails commits
                // nothing, and ext_capacity is a bound the caller picks generously.  When it does not fit, the log gets what LDS has left (room for at
                // least half of the call's pods, or the attempt is not worth a launch; moves onto nodes that were removed since are squeezed out
                // when it fills up); the kernel gives up at the commit that still does not fit and fetch_removals / confirm_kernel run the call
                // again with the log in HBM (switch_to_glog_) or through K_sched.  Results are those of whichever kernel finished.
                if (forced_cap >= 256 && (forced_cap & ~255) < lean_log_cap_) {
                    lean_log_cap_ = forced_cap & ~255; lean_optimistic_ = true;
                    lean_smem_ = (size_t)casim_lean_removal_bytes(R, C_, capN, lean_log_cap_);
                }
                if (lean_smem_ > bk_.lds_budget() && !no_optimism) {
                    if (room >= 256 && room * 2 >= (int64_t)P_) {
                        lean_log_cap_ = (int32_t)room;
                        lean_smem_ = (size_t)casim_lean_removal_bytes(R, C_, capN, lean_log_cap_);
                        lean_optimistic_ = true;
                    }
                }
            }
            // The log in HBM (removals_lean_kernel<., true, true>): when the node state fits LDS but the log does not — not even the optimistic one —
            // or the call lists more than 65 536 pods.  Every part is sized for the worst case (it never gives up); 7 bytes per entry.
            lean_glog_ = false;
            {
                const bool force = getenv("CASIM_LEAN_HBM_LOG") && atoi(getenv("CASIM_LEAN_HBM_LOG")) != 0;   // tests: every eligible call
                const bool off = getenv("CASIM_LEAN_HBM_LOG") && atoi(getenv("CASIM_LEAN_HBM_LOG")) == 0;
                const bool lds_log_ok = P_ <= 65536 && lean_smem_ <= bk_.lds_budget();
                const int64_t glog_bytes = 7 * worst * kLeanLogBuckets;
                glog_possible_ = !off && fixed <= (int64_t)bk_.lds_budget() && glog_bytes <= (256ll << 20) && (int64_t)P_ + E0_worst_ <= 0x7fffffffll;
                glog_worst_ = worst; glog_fixed_ = fixed;
                if (glog_possible_ && (force || !lds_log_ok)) {
                    lean_glog_ = true; lean_bulk_ = true; lean_optimistic_ = false;
                    lean_log_cap_ = (int32_t)(worst * kLeanLogBuckets);
                    lean_smem_ = (size_t)fixed;
                }
            }
            lean_ = plain && N_ <= 65536 && (lean_glog_ || P_ <= 65536) && lean_smem_ <= bk_.lds_budget();
            if (!lean_) { lean_optimistic_ = false; lean_glog_ = false; }
        }
        if (lean_) max_threads = 64;   // (cap_ = the node count rounded up to whole words)
        threads_ = (int)(round_up64_((int64_t)N_) < max_threads ? round_up64_((int64_t)N_) : max_threads);
        cap_ = (int32_t)(((int64_t)N_ + threads_ - 1) / threads_ * threads_);
        S_ = cap_ >> 6;
        a_.N = N_; a_.C = C_; a_.n_runs = n_runs_; a_.break_on_failure = q->break_on_failure ? 1 : 0;
        // lastIndex of a list that shrank since the last loop: any value is a valid cyclic origin
        a_.last_index = q->last_index; a_.cap = cap_;
        a_.memo_classes = C_ < 65536 ? C_ : 65536;
        a_.run_class = up(rc.data(), rc.size()); a_.run_count = up(rn.data(), rn.size());
        a_.run_hint = up(rh.data(), rh.size()); a_.run_first = up(rf.data(), rf.size());
        if (!pair_id.empty()) {
            a_.run_pair = up(rp.data(), rp.size()); a_.run_ctrl = up(rk.data(), rk.size());
            n_pairs_ = pair_id.size(); n_ctrl_ = ctrl_id.size();
            a_.pair_memo = (int32_t*)dalloc(4 * n_pairs_); a_.ctrl_count = (int32_t*)dalloc(4 * n_ctrl_);
        }
        a_.acceptable = q->node_acceptable ? up(q->node_acceptable, N) : nullptr;
        d_fbits_ = (uint64_t*)dalloc(8 * C * (size_t)S_);
        a_.fbits = d_fbits_;
        if (lean_) d_fit0_ = (uint64_t*)dalloc(8 * C * (size_t)S_);
        if (lean_ && lean_glog_ && !alloc_glog_()) { lean_ = false; lean_glog_ = false; }
        if (getenv("CASIM_PACK_PROF_DUMP")) { a_.prof = (int64_t*)dalloc(96); bk_.zero(a_.prof, 96); }
        a_.node_out = (int32_t*)dalloc(4 * (P + (size_t)(cand && cand->ext_capacity > 0 ? cand->ext_capacity : 0)));
        a_.out = (int32_t*)dalloc(32);
        if (K_ > 0) {
            a_.memo_classes = 0;  // breakOnFailure ends a simulation at the first miss: the memo is never consulted
            a_.n_cand = K_; a_.persist = cand->persist ? 1 : 0; a_.max_removable = cand->max_removable > 0 ? cand->max_removable : 0;
            a_.lean_bulk_min = lean_bulk_min_;

            if (lean_) {   // the kernel's candidate / run records (SchedArgs::lean_cand, lean_run)
                lc.assign(4 * ((size_t)K_ + 1), 0); lr.assign(4 * (rc.size() > 0 ? rc.size() : 1), 0);
                for (int k = 0; k <= K_; ++k) {
                    lc[4 * (size_t)k + 0] = k < K_ ? cand->cand_node[k] : 0;
                    lc[4 * (size_t)k + 1] = cro[(size_t)k];
                    lc[4 * (size_t)k + 2] = cand->pod_offsets[k];
                    lc[4 * (size_t)k + 3] = (k < K_ && cand->cand_atomic && cand->cand_atomic[k]) ? 1 : 0;
                }
                for (size_t i = 0; i < rc.size(); ++i) { lr[4 * i] = rc[i]; lr[4 * i + 1] = rn[i]; lr[4 * i + 2] = rh[i]; lr[4 * i + 3] = rf[i]; }
                a_.lean_cand = up(lc.data(), lc.size()); a_.lean_run = up(lr.data(), lr.size());
                if (!a_.lean_cand || !a_.lean_run) lean_ = false;
            }
            a_.cand_node = up(cand->cand_node, (size_t)K_);
            a_.cand_atomic = cand->cand_atomic ? up(cand->cand_atomic, (size_t)K_) : nullptr;
            a_.cand_run_off = up(cro.data(), cro.size());
            a_.cand_pod_off = up(cand->pod_offsets, (size_t)K_ + 1);
            a_.removable_out = (uint8_t*)dalloc((size_t)K_);
            E_ = cand->ext_capacity > 0 ? cand->ext_capacity : 0;
            a_.P = P_; a_.ext_cap = E_;
            a_.pod_class = up(q->pod_class, P);
            a_.pod_sticky = cand->pod_sticky ? up(cand->pod_sticky, P) : nullptr;
            a_.ext_ref = (int32_t*)dalloc(4 * (size_t)E_); a_.ext_cand = (int32_t*)dalloc(4 * (size_t)E_);
            a_.log_ref = (int32_t*)dalloc(4 * (P + (size_t)E_)); a_.log_dest = (int32_t*)dalloc(4 * (P + (size_t)E_));
            a_.committed = (char*)dalloc((size_t)cap_ * (8u * (size_t)R + 8u * (size_t)dt_.Wx + 4u));
        }
        // ---- domain rules ----
        const casim_domain_rules* dr = q->rules;
        if (dr && dr->n_rules > 0) {
            if (dr->n_nodes != N_ || dr->n_classes != C_) return fail(CASIM_ERR_INVALID, "domain rules were built for other tables");
            // nodeTaintsPolicy: Honor + removal simulation: the ghost's ToBeDeleted taint takes it out of the domains of such rules
            // for the length of its transaction (rule_ghost_leaves, written by the encoder); tables from an encoder that does not
            // provide the column keep being delegated
            if (K_ > 0 && dr->n_taint_policy_rules > 0 && !dr->rule_ghost_leaves) return CASIM_NG_UNSUPPORTED;
            for (int c = 0; c < C_; ++c)
                if (dr->class_rule_off[c + 1] - dr->class_rule_off[c] > kMaxRulesPerClass) return CASIM_NG_UNSUPPORTED;
            const size_t NR = (size_t)dr->n_rules, tot = (size_t)dr->rule_offset[NR];
            a_.n_rules = dr->n_rules;
            a_.node_domain = up(dr->node_domain, (size_t)dr->n_keys * N);
            a_.rule_key = up(dr->rule_key, NR); a_.rule_kind = up(dr->rule_kind, NR); a_.rule_max_skew = up(dr->rule_max_skew, NR);
            a_.rule_min_domains = up(dr->rule_min_domains, NR); a_.rule_self = up(dr->rule_self, NR); a_.rule_elig_row = up(dr->rule_elig_row, NR);
            a_.rule_off = up(dr->rule_offset, NR + 1);
            a_.rule_ghost = (K_ > 0 && dr->rule_ghost_leaves && dr->n_taint_policy_rules > 0) ? up(dr->rule_ghost_leaves, NR) : nullptr;
            d_rule_init_ = up(dr->count_init, tot); rule_total_ = (int64_t)tot;
            a_.rule_cnt = (int32_t*)dalloc(4 * tot);
            d_dom_init_ = up(dr->domain_nodes, tot);
            a_.rule_dom_nodes = (int32_t*)dalloc(4 * tot);
            if (K_ > 0) {
                contrib_total_ = (int64_t)NR * (int64_t)N_;
                d_contrib_init_ = up(dr->node_contrib, (size_t)contrib_total_);
                a_.rule_contrib = (int32_t*)dalloc(4 * (size_t)contrib_total_);
            }
            // eligibility rows are addressed by 64-node words of the padded node range
            if (dr->n_elig_rows > 0) {
                const size_t w_in = (N + 63) / 64;
                std::vector<uint64_t> rows((size_t)dr->n_elig_rows * (size_t)S_, 0ull);
                for (int r = 0; r < dr->n_elig_rows; ++r) for (size_t w = 0; w < w_in; ++w) rows[(size_t)r * (size_t)S_ + w] = dr->elig_bits[(size_t)r * w_in + w];
                a_.rule_elig = up(rows.data(), rows.size());
                if (direct_uploads_ > 0) bk_.sync();   // (`rows` is a local; a packed upload has copied it already)
            }
            a_.class_rule_off = up(dr->class_rule_off, C + 1); a_.inc_off = up(dr->inc_off, C + 1);
            a_.inc_rule = up(dr->inc_rule, (size_t)dr->inc_off[C]);
        }
        const int alive_words = K_ > 0 ? S_ : 0;
        const int64_t ctrl = casim_sched_ctrl_bytes(a_.memo_classes, alive_words), bytes = casim_sched_state_bytes(R, dt_.Wx, cap_);
        lds_ = ctrl + bytes <= (int64_t)bk_.lds_budget();
        smem_ = (size_t)(lds_ ? ctrl + bytes : ctrl);
        if (!lds_) a_.gstate = (char*)dalloc((size_t)bytes);
        end_uploads();
        // the run tables above are locals: uploads that went out on their own must have left them.  Packed ones were copied into the
        // staging buffer when up() returned; nothing else uses that buffer before this call's results have been waited for
        if (direct_uploads_ > 0) bk_.sync();
        if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
        ready_ = true;
        return CASIM_OK;
    }

    int32_t run() {
        if (trivial_) return CASIM_OK;
        if (!ready_) return fail(CASIM_ERR_INVALID, "scheduler not initialised");
        if (P_ + E_ > 0) bk_.launch(fill_i32_kernel, (P_ + E_ + 255) / 256, 1, 256, (size_t)0, a_.node_out, (int64_t)(P_ + E_), (int32_t)-1);
        if (K_ > 0) bk_.fill8(a_.removable_out, 2, (size_t)K_);
        if (n_pairs_ > 0) { bk_.zero(a_.pair_memo, 4 * n_pairs_); bk_.zero(a_.ctrl_count, 4 * n_ctrl_); }
        if (rule_total_ > 0) {
            bk_.launch(copy_i32_kernel, (int)((rule_total_ + 255) / 256), 1, 256, (size_t)0, a_.rule_cnt, d_rule_init_, rule_total_);
            bk_.launch(copy_i32_kernel, (int)((rule_total_ + 255) / 256), 1, 256, (size_t)0, a_.rule_dom_nodes, d_dom_init_, rule_total_);
            if (contrib_total_ > 0)
                bk_.launch(copy_i32_kernel, (int)((contrib_total_ + 255) / 256), 1, 256, (size_t)0, a_.rule_contrib, d_contrib_init_, contrib_total_);
        }
        if (C_ > 0) bk_.launch(sched_static_kernel, S_, C_, 64, (size_t)0, dt_, d_fbits_, S_);
        if (K_ > 0) { int32_t* li = last_removals_info(); li[0] = lean_ ? 1 : 0; li[1] = lean_ ? 64 : threads_; li[2] = (lean_ || lds_) ? 1 : 0; li[3] = n_runs_; }
        if (lean_) {
            bk_.launch(lean_fit0_kernel, S_, C_, 64, (size_t)0, dt_, (const uint64_t*)d_fbits_, d_fit0_, S_);
            if (lean_glog_) {
                if (dt_.R <= 2) bk_.launch(removals_lean_kernel<2, true, true>, 1, 1, 64, lean_smem_, dt_, a_, (const uint64_t*)d_fit0_, lean_log_cap_);
                else bk_.launch(removals_lean_kernel<4, true, true>, 1, 1, 64, lean_smem_, dt_, a_, (const uint64_t*)d_fit0_, lean_log_cap_);
            } else if (lean_bulk_) {
                if (dt_.R <= 2) bk_.launch(removals_lean_kernel<2, true>, 1, 1, 64, lean_smem_, dt_, a_, (const uint64_t*)d_fit0_, lean_log_cap_);
                else bk_.launch(removals_lean_kernel<4, true>, 1, 1, 64, lean_smem_, dt_, a_, (const uint64_t*)d_fit0_, lean_log_cap_);
            } else {
                if (dt_.R <= 2) bk_.launch(removals_lean_kernel<2, false>, 1, 1, 64, lean_smem_, dt_, a_, (const uint64_t*)d_fit0_, lean_log_cap_);
                else bk_.launch(removals_lean_kernel<4, false>, 1, 1, 64, lean_smem_, dt_, a_, (const uint64_t*)d_fit0_, lean_log_cap_);
            }
            return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
        }
        const bool tx = K_ > 0, ru = a_.n_rules > 0;
#define CASIM_SCHED_LAUNCH(L, X, Y) do { if (dt_.R <= 2) bk_.launch(sched_kernel<L, X, Y, 2>, 1, 1, threads_, smem_, dt_, a_); \
                                         else bk_.launch(sched_kernel<L, X, Y, CASIM_KMAX_RES>, 1, 1, threads_, smem_, dt_, a_); } while (0)
        if (lds_) { if (tx) { if (ru) CASIM_SCHED_LAUNCH(true, true, true); else CASIM_SCHED_LAUNCH(true, true, false); }
                    else    { if (ru) CASIM_SCHED_LAUNCH(true, false, true); else CASIM_SCHED_LAUNCH(true, false, false); } }
        else      { if (tx) { if (ru) CASIM_SCHED_LAUNCH(false, true, true); else CASIM_SCHED_LAUNCH(false, true, false); }
                    else    { if (ru) CASIM_SCHED_LAUNCH(false, false, true); else CASIM_SCHED_LAUNCH(false, false, false); } }
#undef CASIM_SCHED_LAUNCH
        if (a_.prof) {   // profiling builds: where thread 0 spent its time
            int64_t h[12]; bk_.d2h(h, a_.prof, 96); bk_.sync();
            fprintf(stderr, "[sched prof] ticks: records %lld hint %lld minima+origin %lld walk-rest %lld pick %lld rounds %lld runend %lld txn %lld | piece: setup %lld capacity %lld prefix %lld place %lld (runs %d)\n",
                    (long long)h[0], (long long)h[1], (long long)h[2], (long long)h[3], (long long)h[4], (long long)h[5], (long long)h[6], (long long)h[7],
                    (long long)h[8], (long long)h[9], (long long)h[10], (long long)h[11], n_runs_);
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    int32_t fetch(int32_t* node_out, int32_t* last_index_out, int32_t* n_scheduled_out) {
        if (trivial_) {
            for (int i = 0; i < P_; ++i) if (node_out) node_out[i] = -1;
            if (last_index_out) *last_index_out = last_index_;
            if (n_scheduled_out) *n_scheduled_out = 0;
            return CASIM_OK;
        }
        // through the pinned fetch staging buffer: a copy into the caller's pageable array waits for the stream by itself and moves at a
        // fraction of the link's rate
        const size_t nb = node_out ? 4 * (size_t)P_ : 0;
        char* st = (char*)bk_.stage(1, 64 + nb);
        int32_t o4[4] = {0, 0, 0, 0};
        int32_t* o = st ? (int32_t*)st : o4;
        if (st) { if (nb) bk_.d2h(st + 64, a_.node_out, nb); }
        else if (nb) bk_.d2h(node_out, a_.node_out, nb);
        bk_.d2h(o, a_.out, 16);
        bk_.sync();
        if (st && nb) memcpy(node_out, st + 64, nb);
        if (last_index_out) *last_index_out = o[0];
        if (n_scheduled_out) *n_scheduled_out = o[1];
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    // removal simulation results: removable[K] (1 / 0 / 2 = not evaluated), candidates with a final answer
    int32_t fetch_removals(casim_removal_results* out) {
        if (!out) return fail(CASIM_ERR_INVALID, "null results");
        out->n_ext = 0; out->n_processed = 0; out->last_index = last_index_;
        if (trivial_) {
            for (int k = 0; k < K_; ++k) if (out->removable) out->removable[k] = 2;
            for (int i = 0; i < P_; ++i) if (out->node_out) out->node_out[i] = -1;
            return CASIM_OK;
        }
        // everything in ONE round trip through the pinned fetch staging buffer (the ext tables up to their capacity: how many entries
        // are valid comes with the same copies)
        const size_t nb = out->node_out && P_ > 0 ? 4 * (size_t)P_ : 0, rb = out->removable && K_ > 0 ? ((size_t)K_ + 15) & ~(size_t)15 : 0;
        const size_t eb = 4 * (size_t)E_;
        char* st = (char*)bk_.stage(1, 64 + nb + rb + 3 * eb + 64);
        if (!st) return fail(CASIM_ERR_NOMEM, "no staging buffer");
        int32_t* o = (int32_t*)st;
        char* s_node = st + 64; char* s_rem = s_node + nb; char* s_ec = s_rem + rb; char* s_ep = s_ec + eb; char* s_en = s_ep + eb;
        bk_.d2h(o, a_.out, 32);
        if (nb) bk_.d2h(s_node, a_.node_out, nb);
        if (rb) bk_.d2h(s_rem, a_.removable_out, (size_t)K_);
        if (E_ > 0) {
            if (out->ext_candidate) bk_.d2h(s_ec, a_.ext_cand, eb);
            if (out->ext_pod) bk_.d2h(s_ep, a_.ext_ref, eb);
            if (out->ext_node) bk_.d2h(s_en, a_.node_out + P_, eb);
        }
        bk_.sync();
        if (lean_ && lean_optimistic_ && bk_.ok() && o[5] == 1) {   // the LDS log overflowed: the same kernel with its log in HBM answers, or the general loop (run() re-initialises every output)
            lean_optimistic_ = false; lean_gave_up_ = true;
            if (!switch_to_glog_()) lean_ = false;
            const int32_t rc = run();
            return rc == CASIM_OK ? fetch_removals(out) : rc;
        }
        if (nb) memcpy(out->node_out, s_node, nb);
        if (rb) memcpy(out->removable, s_rem, (size_t)K_);
        const int ne = o[4] < E_ ? o[4] : E_;
        if (ne > 0) {
            if (out->ext_candidate) memcpy(out->ext_candidate, s_ec, 4 * (size_t)ne);
            if (out->ext_pod) memcpy(out->ext_pod, s_ep, 4 * (size_t)ne);
            if (out->ext_node) memcpy(out->ext_node, s_en, 4 * (size_t)ne);
        }
        out->n_ext = ne; out->last_index = o[0]; out->n_processed = o[3];
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    const std::string& error() const { return err_; }
    void use_resident(const DevTables* t) { resident_ = t; }
    const int32_t* node_out_dev() const { return trivial_ ? nullptr : a_.node_out; }
    int runs() const { return n_runs_; }
    int threads() const { return threads_; }
    bool in_lds() const { return lds_; }
    bool lean() const { return lean_; }   // the removal loop runs as removals_lean_kernel
    bool lean_gave_up() const { return lean_gave_up_; }
    // After a first run(): did the one-wave kernel with a log smaller than the worst case finish?  If not, later run()s take K_sched (callers that
    // run() repeatedly without fetching: the timing entry points).  Synchronises only when there is something to ask.
    int32_t confirm_kernel() {
        if (trivial_ || !lean_ || !lean_optimistic_) return CASIM_OK;
        int32_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bk_.d2h(o, a_.out, 32); bk_.sync();
        if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
        lean_optimistic_ = false;
        if (o[5] == 1) { lean_gave_up_ = true; if (!switch_to_glog_()) lean_ = false; }
        return CASIM_OK;
    }

    // Workgroup size cap.  Every wave runs the run's uniform instruction stream and meets the others at each
    // collective, so more waves only pay off while a run has to look at many nodes; a piece of 256 nodes also lets
    // a run that fits nearby stop after a quarter of the work of a 1024-node piece.  Sweep on MI355X
    // (profiles/r01h_sched_threads_sweep.txt): 64 / 128 / 256 / 512 / 1024 threads -> 20.9 / 16.1 / 14.2 / 15.5 / 20.6 ms
    // for 15000 nodes x 150000 pods, 41.2 / 41.6 / 42.1 / 44.9 / 53.3 ms for 3000 removals on 15000 nodes.
    static constexpr int kDefaultThreads = 256;
    static constexpr int kDefaultThreadsRemovals = 128;   // short runs (1-2 pods per candidate): fewer waves per barrier

private:
    static int64_t round_up64_(int64_t v) { return (v + 63) & ~63ll; }
    void begin_uploads(size_t bound) {
        up_host_ = (char*)bk_.stage(0, bound);
        up_dev_ = up_host_ ? (char*)dalloc(bound) : nullptr;
        up_cap_ = up_dev_ ? bound : 0; up_used_ = 0; direct_uploads_ = 0;
    }
    void end_uploads() {
        if (up_dev_ && up_used_ > 0) bk_.h2d(up_dev_, up_host_, up_used_);
        up_dev_ = up_host_ = nullptr; up_cap_ = 0;
    }
    template <class T>
    const T* up(const T* src, size_t n) {
        if (n == 0 || !src) return nullptr;
        const size_t bytes = sizeof(T) * n, at = (up_used_ + 15) & ~(size_t)15;
        if (up_dev_ && at + bytes <= up_cap_) {
            memcpy(up_host_ + at, src, bytes);
            up_used_ = at + bytes;
            return (const T*)(up_dev_ + at);
        }
        T* d = (T*)dalloc(bytes);   // outside a packed section (or a bound that was too small): its own copy
        if (d) { bk_.h2d(d, src, bytes); ++direct_uploads_; }
        return d;
    }
    void* dalloc(size_t bytes) {
        if (bytes == 0) bytes = 8;
        void* p = bk_.alloc(bytes);
        if (p) allocs_.push_back(p);
        return p;
    }
    int32_t fail(int32_t code, const char* msg) { err_ = msg ? msg : ""; return code; }

    BK& bk_;
    char* up_host_ = nullptr; char* up_dev_ = nullptr; size_t up_cap_ = 0, up_used_ = 0; int direct_uploads_ = 0;   // packed uploads (begin_uploads / up / end_uploads)
    DevTables dt_; SchedArgs a_;
    const DevTables* resident_ = nullptr;
    int C_ = 0, N_ = 0, P_ = 0, S_ = 0, K_ = 0, E_ = 0, threads_ = 64;
    int32_t cap_ = 0, n_runs_ = 0, last_index_ = 0;
    bool ready_ = false, trivial_ = false, lds_ = true, lean_ = false;
    size_t smem_ = 0, lean_smem_ = 0; int32_t lean_log_cap_ = 0;
    bool lean_gave_up_ = false, lean_bulk_ = false, lean_glog_ = false, glog_possible_ = false;
    int64_t glog_worst_ = 0, glog_fixed_ = 0;
    bool alloc_glog_() {
        a_.glog_dest = (uint16_t*)dalloc(2 * (size_t)lean_log_cap_); a_.glog_ref = (uint32_t*)dalloc(4 * (size_t)lean_log_cap_); a_.glog_cls = (uint8_t*)dalloc((size_t)lean_log_cap_);
        return a_.glog_dest && a_.glog_ref && a_.glog_cls;
    }
    // the LDS log gave up: the third instantiation (log in HBM, every part sized for the worst case) when it is possible at all
    bool switch_to_glog_() {
        if (!glog_possible_) return false;
        lean_log_cap_ = (int32_t)(glog_worst_ * kLeanLogBuckets);
        lean_smem_ = (size_t)glog_fixed_;
        if (!alloc_glog_()) return false;
        lean_glog_ = true; lean_bulk_ = true;
        return true;
    }
    int32_t lean_bulk_min_ = 4;
    bool lean_optimistic_ = false;   // the LDS log is smaller than the call's worst case: the kernel may give up (out[5]), K_sched then runs
    uint64_t* d_fbits_ = nullptr; uint64_t* d_fit0_ = nullptr;
    const int32_t* d_rule_init_ = nullptr; const int32_t* d_dom_init_ = nullptr; const int32_t* d_contrib_init_ = nullptr;
    int64_t rule_total_ = 0, contrib_total_ = 0;
    size_t n_pairs_ = 0, n_ctrl_ = 0;
    std::vector<void*> allocs_;
    std::string err_;
};


// ---- resident cluster: the snapshot's node table stays in HBM for a whole RunOnce iteration ----------------------------
// The reference forks / commits / reverts ONE ClusterSnapshot through a loop iteration (CA/core/static_autoscaler.go:391 ff):
// filter-out-schedulable adds the pods it placed to the snapshot (filter_out_schedulable.go:94-103 via SchedulePod), the
// scale-down planner then simulates removals on that state inside Fork / Revert (planner.go:286-336,
// CA/simulator/clustersnapshot/store/delta.go:292-323,442-463).  Here the node table (one record per node: allocatable,
// what its pods request, pod count, node-local exclusion bits, taint / label bits) is uploaded ONCE; every call forks from
// that committed image — the kernels build their working state from it at the start of every pass, so a Revert costs
// nothing —, and a Commit folds the call's placements into the image with one scatter kernel (commit_placements_kernel).
// Node-level deltas between calls (a node's pods changed, a node joined) replace single records (scatter_rows_kernel).
CS_GLOBAL void commit_placements_kernel(DevTables t, const int32_t* CS_RESTRICT node_out, const int32_t* CS_RESTRICT pod_class, int P) {
    const int i = cs::bid() * cs::nthreads() + cs::tid();
    if (i >= P) return;
    const int m = node_out[i];
    if (m < 0) return;
    const int c = pod_class[i];
    for (int r = 0; r < t.R; ++r) {
        const int64_t q = t.req[(int64_t)c * t.R + r];
        if (q) cs::atomic_add_i64((int64_t*)t.init_req + (int64_t)m * t.R + r, q);      // NodeInfo.AddPodInfo: requested += request (types.go:439-463)
    }
    cs::atomic_add_i32((int32_t*)t.init_pods + m, 1);
    for (int w = 0; w < t.Wx; ++w) {
        const uint64_t b = t.xmark[(int64_t)c * t.Wx + w];
        if (b) cs::atomic_or_u64((uint64_t*)t.init_excl + (int64_t)m * t.Wx + w, b);    // used host ports / anti-affinity occupancy
    }
}
// dst[idx[k]][0..width) = src[k][0..width) for one column (element size 4 or 8 bytes, as 4-byte words)
CS_GLOBAL void scatter_rows_kernel(uint32_t* CS_RESTRICT dst, const uint32_t* CS_RESTRICT src, const int32_t* CS_RESTRICT idx, int n, int words_per_row) {
    const int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid();
    if (i >= (int64_t)n * words_per_row) return;
    const int k = (int)(i / words_per_row), w = (int)(i % words_per_row);
    dst[(int64_t)idx[k] * words_per_row + w] = src[i];
}

template <class BK>
class ClusterT {
public:
    explicit ClusterT(BK& bk) : bk_(bk) {}
    ~ClusterT() { for (void* p : allocs_) bk_.free(p); }
    ClusterT(const ClusterT&) = delete;
    ClusterT& operator=(const ClusterT&) = delete;

    int32_t init(const casim_pegs* p, const casim_groups* g) {
        if (!p || !g) return fail(CASIM_ERR_INVALID, "null table");
        if (p->n_pegs < 0 || g->n_groups < 0 || p->n_res < 2 || p->n_res > CASIM_KMAX_RES) return fail(CASIM_ERR_INVALID, "bad table sizes");
        if (p->w_taint < 0 || p->w_label < 0 || p->w_excl < 0 || p->w_zone < 0) return fail(CASIM_ERR_INVALID, "negative mask width");
        const size_t C = (size_t)p->n_pegs, N = (size_t)g->n_groups, R = (size_t)p->n_res;
        if (C > 0 && (!p->req || !p->flags)) return fail(CASIM_ERR_INVALID, "class table has null columns");
        if (N > 0 && (!g->alloc || !g->init_req || !g->allowed_pods || !g->init_pods || !g->flags)) return fail(CASIM_ERR_INVALID, "node table has null columns");
        if (C > 0 && ((p->w_taint && !p->tol_mask) || (p->w_label && !p->sel_mask) || (p->w_excl && (!p->excl_block || !p->excl_mark)) || (p->w_zone && (!p->zone_block || !p->zone_mark))))
            return fail(CASIM_ERR_INVALID, "class mask column missing");
        if (N > 0 && ((p->w_taint && !g->taint_mask) || (p->w_label && !g->label_mask) || (p->w_excl && !g->init_excl))) return fail(CASIM_ERR_INVALID, "node mask column missing");
        // host copy of the class table (validation of later calls reads flags and zone words) and of the node table's shape
        hp_ = *p; hg_ = *g;
        keep(h_req_, p->req, C * R, hp_.req); keep(h_count_, p->count, C, hp_.count); keep(h_flags_, p->flags, C, hp_.flags);
        keep(h_tol_, p->tol_mask, C * p->w_taint, hp_.tol_mask); keep(h_sel_, p->sel_mask, C * p->w_label, hp_.sel_mask);
        keep(h_xb_, p->excl_block, C * p->w_excl, hp_.excl_block); keep(h_xm_, p->excl_mark, C * p->w_excl, hp_.excl_mark);
        keep(h_zb_, p->zone_block, C * p->w_zone, hp_.zone_block); keep(h_zm_, p->zone_mark, C * p->w_zone, hp_.zone_mark);
        hp_.fp_cpu = hp_.fp_mem = nullptr;
        keep(h_alloc_, g->alloc, N * R, hg_.alloc); keep(h_ireq_, g->init_req, N * R, hg_.init_req); keep(h_allowed_, g->allowed_pods, N, hg_.allowed_pods);
        keep(h_ipods_, g->init_pods, N, hg_.init_pods); keep(h_gflags_, g->flags, N, hg_.flags); keep(h_taint_, g->taint_mask, N * p->w_taint, hg_.taint_mask);
        keep(h_label_, g->label_mask, N * p->w_label, hg_.label_mask); keep(h_iexcl_, g->init_excl, N * p->w_excl, hg_.init_excl);
        hg_.init_zone = hg_.zone_valid = nullptr; hg_.max_nodes = hg_.existing_nodes = hg_.last_index = nullptr;
        hg_.cap_cpu = hg_.cap_mem = nullptr; hg_.waste_cpu = hg_.waste_mem = nullptr; hg_.peg_offsets = hg_.peg_index = nullptr;
        hg_.peg_lo = hg_.peg_hi = hg_.global_id = nullptr; hg_.n_sims = 0; hg_.sim_offsets = nullptr;
        // device image
        memset(&dt_, 0, sizeof dt_);
        dt_.G = p->n_pegs; dt_.NG = g->n_groups; dt_.R = p->n_res; dt_.Wt = p->w_taint; dt_.Wl = p->w_label; dt_.Wx = p->w_excl; dt_.Wz = 0;
        dt_.req = up(p->req, C * R); dt_.pflags = up(p->flags, C); dt_.tol = up(p->tol_mask, C * dt_.Wt); dt_.sel = up(p->sel_mask, C * dt_.Wl);
        dt_.xblock = up(p->excl_block, C * dt_.Wx); dt_.xmark = up(p->excl_mark, C * dt_.Wx);
        dt_.alloc = up(g->alloc, N * R); dt_.init_req = up(g->init_req, N * R); dt_.allowed = up(g->allowed_pods, N); dt_.init_pods = up(g->init_pods, N);
        dt_.gflags = up(g->flags, N); dt_.taint = up(g->taint_mask, N * dt_.Wt); dt_.label = up(g->label_mask, N * dt_.Wl); dt_.init_excl = up(g->init_excl, N * dt_.Wx);
        bk_.sync();
        if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
        ready_ = true; uploads_ = 1;
        return CASIM_OK;
    }

    // delta: the records of n nodes are replaced (rows->n_groups == n, same lanes / mask widths); only those rows travel
    int32_t update_nodes(int32_t n, const int32_t* idx, const casim_groups* rows) {
        if (!ready_) return fail(CASIM_ERR_INVALID, "cluster not initialised");
        if (n < 0 || (n > 0 && (!idx || !rows || rows->n_groups != n))) return fail(CASIM_ERR_INVALID, "bad delta");
        if (n == 0) return CASIM_OK;
        for (int k = 0; k < n; ++k) if (idx[k] < 0 || idx[k] >= dt_.NG) return fail(CASIM_ERR_INVALID, "node index out of range");
        if (!rows->alloc || !rows->init_req || !rows->allowed_pods || !rows->init_pods || !rows->flags || (dt_.Wt && !rows->taint_mask) ||
            (dt_.Wl && !rows->label_mask) || (dt_.Wx && !rows->init_excl)) return fail(CASIM_ERR_INVALID, "delta rows miss a column");
        const size_t mark = allocs_.size();   // the staging copies of this delta are released once the kernels are done
        const int32_t* d_idx = up(idx, (size_t)n);
        auto col = [&](const void* dst, const void* src, int bytes_per_row) {
            if (bytes_per_row <= 0) return;
            const uint32_t* d_src = (const uint32_t*)up((const char*)src, (size_t)n * (size_t)bytes_per_row);
            const int wpr = bytes_per_row / 4;
            bk_.launch(scatter_rows_kernel, (int)(((int64_t)n * wpr + 255) / 256), 1, 256, (size_t)0, (uint32_t*)dst, d_src, d_idx, n, wpr);
        };
        const int R = dt_.R;
        col(dt_.alloc, rows->alloc, 8 * R); col(dt_.init_req, rows->init_req, 8 * R); col(dt_.allowed, rows->allowed_pods, 4);
        col(dt_.init_pods, rows->init_pods, 4); col(dt_.gflags, rows->flags, 4); col(dt_.taint, rows->taint_mask, 8 * dt_.Wt);
        col(dt_.label, rows->label_mask, 8 * dt_.Wl); col(dt_.init_excl, rows->init_excl, 8 * dt_.Wx);
        bk_.sync();
        drop_temps(mark);
        delta_rows_ += n;
        if (!committed_.empty()) {   // the new records (and the caller's rule counters) describe these nodes from now on
            std::vector<uint8_t> gone((size_t)dt_.NG, 0);
            for (int k = 0; k < n; ++k) gone[(size_t)idx[k]] = 1;
            for (auto it = committed_.begin(); it != committed_.end();) {
                if (gone[(size_t)(uint32_t)(it->first & 0xffffffffull)]) { committed_total_ -= it->second; it = committed_.erase(it); } else ++it;
            }
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    // TrySchedulePods on the committed image; commit != 0 keeps the placements (SchedulePod's AddPod on the snapshot)
    int32_t try_schedule(const casim_pod_sequence* q, int commit, int32_t* node_out, int32_t* last_index_out, int32_t* n_scheduled_out) {
        if (!ready_) return fail(CASIM_ERR_INVALID, "cluster not initialised");
        if (!q) return fail(CASIM_ERR_INVALID, "null pod sequence");
        PatchedRules pr;
        casim_pod_sequence qq = *q;
        int32_t rc = with_committed(q->rules, pr);
        if (rc != CASIM_OK) return rc;
        qq.rules = pr.rules;
        SchedulerT<BK> s(bk_);
        s.use_resident(&dt_);
        rc = s.init(&hp_, &hg_, &qq);
        if (rc == CASIM_OK) rc = s.run();
        const size_t mark = allocs_.size();
        const bool folds = rc == CASIM_OK && commit && q->n_pods > 0 && s.node_out_dev();
        if (folds) {
            const int32_t* d_pc = up(q->pod_class, (size_t)q->n_pods);
            bk_.launch(commit_placements_kernel, (q->n_pods + 255) / 256, 1, 256, (size_t)0, dt_, s.node_out_dev(), d_pc, (int)q->n_pods);
            commits_++;
        }
        std::vector<int32_t> own;
        if (folds && !node_out) { own.resize((size_t)q->n_pods); node_out = own.data(); }   // (the placements are remembered below)
        if (rc == CASIM_OK) rc = s.fetch(node_out, last_index_out, n_scheduled_out);   // (synchronises: the commit kernel is done)
        else bk_.sync();
        drop_temps(mark);
        if (rc < 0) err_ = s.error();
        // what the committed pods mean for the domain rules of LATER calls: the caller's count_init / node_contrib come from the
        // encoder, which saw the snapshot before this commit (with_committed adds them back in)
        if (folds && rc == CASIM_OK)
            for (int32_t i = 0; i < q->n_pods; ++i) if (node_out[i] >= 0) { committed_[((uint64_t)(uint32_t)q->pod_class[i] << 32) | (uint32_t)node_out[i]] += 1; committed_total_++; }
        return rc;
    }
    // the planner's removal loop on the committed image (its own Fork / Revert: nothing persists)
    int32_t simulate_removals(const casim_removal_candidates* cand, casim_removal_results* out) {
        if (!ready_) return fail(CASIM_ERR_INVALID, "cluster not initialised");
        if (!cand) return fail(CASIM_ERR_INVALID, "null candidates");
        PatchedRules pr;
        casim_removal_candidates cc = *cand;
        int32_t rc = with_committed(cand->rules, pr);
        if (rc != CASIM_OK) return rc;
        cc.rules = pr.rules;
        SchedulerT<BK> s(bk_);
        s.use_resident(&dt_);
        rc = s.init_removals(&hp_, &hg_, &cc);
        if (rc == CASIM_OK) rc = s.run();
        if (rc == CASIM_OK) rc = s.fetch_removals(out);
        if (rc < 0) err_ = s.error();
        return rc;
    }
    int64_t committed_pods() const { return committed_total_; }
    // The caller's NEXT rules already contain every pod committed so far (its encoder re-read the snapshot after the commits —
    // a shim that rebuilds its domain rules from a fresh snapshot each loop but keeps the resident cluster): nothing is added to
    // count_init / node_contrib for them any more.  Commits made after this call are remembered again.
    void forget_commits() { committed_.clear(); committed_total_ = 0; }
    // the committed image of the mutable node columns (tests, and a shim that wants to cross-check its snapshot)
    int32_t fetch_nodes(int64_t* init_req_out, int32_t* init_pods_out, uint64_t* init_excl_out) {
        if (!ready_) return fail(CASIM_ERR_INVALID, "cluster not initialised");
        const size_t N = (size_t)dt_.NG;
        if (init_req_out) bk_.d2h(init_req_out, dt_.init_req, 8 * N * (size_t)dt_.R);
        if (init_pods_out) bk_.d2h(init_pods_out, dt_.init_pods, 4 * N);
        if (init_excl_out && dt_.Wx) bk_.d2h(init_excl_out, dt_.init_excl, 8 * N * (size_t)dt_.Wx);
        bk_.sync();
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }
    void stats(int64_t out[4]) const { out[0] = uploads_; out[1] = delta_rows_; out[2] = commits_; out[3] = dt_.NG; }
    const std::string& error() const { return err_; }

private:
    template <class T> void keep(std::vector<T>& v, const T* src, size_t n, const T*& field) { if (src && n) v.assign(src, src + n); else v.clear(); field = v.empty() ? nullptr : v.data(); }
    template <class T>
    const T* up(const T* src, size_t n) {
        if (n == 0 || !src) return nullptr;
        T* d = (T*)bk_.alloc(sizeof(T) * n);
        if (d) { allocs_.push_back(d); bk_.h2d(d, src, sizeof(T) * n); }
        return d;
    }
    void drop_temps(size_t mark) { while (allocs_.size() > mark) { bk_.free(allocs_.back()); allocs_.pop_back(); } }
    int32_t fail(int32_t code, const char* msg) { err_ = msg ? msg : ""; return code; }
    // Domain rules after commits.  The image in HBM carries what a committed pod does to its node (requests, pod count, exclusion
    // bits); what it does to the OTHER nodes of its topology domains — PodTopologySpread counts, zone anti-affinity, required pod
    // affinity — lives in the rules' counters, and those arrive per call from an encoder that read the snapshot BEFORE the commits
    // (the reference's one ClusterSnapshot sees the pods AddPod placed: CA/simulator/clustersnapshot/store/delta.go:292-323).  So
    // the cluster remembers (class, node) of every committed pod and, in front of every later call that brings rules, adds
    // them to the caller's count_init / node_contrib with the kernels' own rule (casim_sched.h, commit_pods: the node carries the
    // rule's key and is eligible for the rule).  A node whose record is replaced (update_nodes) forgets its pods: the new record and
    // the caller's counters already describe it.
    struct PatchedRules {
        casim_domain_rules copy;
        std::vector<int32_t> count, contrib;
        const casim_domain_rules* rules = nullptr;
    };
    int32_t with_committed(const casim_domain_rules* in, PatchedRules& pr) {
        pr.rules = in;
        if (!in || in->n_rules <= 0 || committed_.empty()) return CASIM_OK;
        if (in->n_nodes != dt_.NG) return fail(CASIM_ERR_INVALID, "domain rules describe another node table");
        if (!in->rule_offset || !in->rule_key || !in->rule_elig_row || !in->node_domain || !in->inc_off || !in->count_init)
            return fail(CASIM_ERR_INVALID, "domain rules miss a column");
        if (in->n_classes > 0 && in->inc_off[in->n_classes] > 0 && !in->inc_rule) return fail(CASIM_ERR_INVALID, "domain rules: inc_off lists rules but inc_rule is null");
        for (int32_t r = 0; r < in->n_rules; ++r)
            if (in->rule_elig_row[r] >= 0 && !in->elig_bits) return fail(CASIM_ERR_INVALID, "domain rules: a rule names an eligibility row but elig_bits is null");
        const int64_t N = in->n_nodes, total = in->rule_offset[in->n_rules];
        pr.count.assign(in->count_init, in->count_init + total);
        if (in->node_contrib) pr.contrib.assign(in->node_contrib, in->node_contrib + (int64_t)in->n_rules * N);
        const int64_t wpr = (N + 63) / 64;
        // one entry per (class, node) with the number of pods committed there: the cost of a call is O(distinct pairs x rules of the
        // class), however many pods the iteration has committed
        for (const auto& kv : committed_) {
            const int32_t cls = (int32_t)(kv.first >> 32), node = (int32_t)(uint32_t)(kv.first & 0xffffffffull), k = kv.second;
            if (cls < 0 || cls >= in->n_classes || node >= N) continue;   // (a class the rules do not know increments nothing)
            for (int32_t ii = in->inc_off[cls]; ii < in->inc_off[cls + 1]; ++ii) {
                const int32_t r = in->inc_rule[ii];
                if (r < 0 || r >= in->n_rules) return fail(CASIM_ERR_INVALID, "domain rules: inc_rule out of range");
                const int32_t d = in->node_domain[(int64_t)in->rule_key[r] * N + node];
                const int32_t row = in->rule_elig_row[r];
                const bool el = row < 0 || ((in->elig_bits[(int64_t)row * wpr + (node >> 6)] >> (node & 63)) & 1ull);
                if (d >= 0 && el) {
                    pr.count[(size_t)(in->rule_offset[r] + d)] += k;
                    if (!pr.contrib.empty()) pr.contrib[(size_t)((int64_t)r * N + node)] += k;
                }
            }
        }
        pr.copy = *in;
        pr.copy.count_init = pr.count.data();
        if (!pr.contrib.empty()) pr.copy.node_contrib = pr.contrib.data();
        pr.rules = &pr.copy;
        return CASIM_OK;
    }
    std::unordered_map<uint64_t, int32_t> committed_;   // (class << 32 | node) -> pods committed there since the node's record was last replaced
    int64_t committed_total_ = 0;
    BK& bk_;
    DevTables dt_;
    casim_pegs hp_; casim_groups hg_;
    std::vector<int64_t> h_req_, h_alloc_, h_ireq_; std::vector<int32_t> h_count_, h_allowed_, h_ipods_; std::vector<uint32_t> h_flags_, h_gflags_;
    std::vector<uint64_t> h_tol_, h_sel_, h_xb_, h_xm_, h_zb_, h_zm_, h_taint_, h_label_, h_iexcl_;
    std::vector<void*> allocs_;
    std::string err_;
    bool ready_ = false;
    int64_t uploads_ = 0, delta_rows_ = 0, commits_ = 0;
};

}  // namespace casim
