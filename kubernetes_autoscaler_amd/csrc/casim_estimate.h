// casim_estimate.h — K_est: BinpackingNodeEstimator.Estimate on the WHOLE snapshot (SURVEY §8 row f3,
// CA/estimator/binpacking_estimator.go:102-342; `CA/` = /root/reference/cluster-autoscaler/).
//
// The template-mode packer (casim_pack.h) knows one node group and treats the nodes already in the cluster as
// list positions only.  That is exact for every Filter that looks at one node — but PodTopologySpread and
// anti-affinity on non-hostname keys look at the node's topology DOMAIN, i.e. at the cluster, and the estimator's
// hostname-spread retry (:212-227) may even place a pod on a node that already exists.  Groups whose PEGs carry
// such rules run here instead: the node table holds the E cluster nodes with their real state followed by the
// not-yet-created clones of the template; one workgroup walks the pods one by one with the machinery of K_sched
// (casim_sched.h): node m owned by thread m % T, state in LDS or an HBM slab, block-wide first-fit by ballot +
// LDS prefix, domain rules as per-domain counters (casim_domain_rules).
//
//   a2  tryToScheduleOnExistingNodes (:163-186): per pod, first passing CREATED node in cyclic order from lastIndex + 1
//   a3  tryToScheduleOnNewNodes (:190-269): per pod, the newest node; when it fails on a hostname spread constraint,
//       any other node of the snapshot (:212-227); else the exits of SURVEY N2 and a new node under the limiter
//       (a class that has no rules, feeds no counter and does not exclude itself fills a node in one step)
// No fastpath (a PEG with spread constraints is never fast-pathed, :411-425; callers with fastpath on delegate).
#pragma once
#include "casim_sched.h"

namespace casim {

struct EstArgs {
    int32_t N, E, cap, n_pegs, max_nodes, last_index;
    const int32_t* peg_class;     // [n_pegs] processing order (DecreasingPodOrderer, host)
    const int32_t* peg_count;     // [n_pegs]
    const uint64_t* fbits;        // [C][cap / 64] static Filters (sched_static_kernel)
    const uint64_t* xports;       // [C][Wx] the host-port part of excl_block (NodePorts runs before PodTopologySpread)
    const uint8_t* host_spread;   // [C] the class has a spread constraint on kubernetes.io/hostname (:358-369)
    // domain rules, as in SchedArgs
    int32_t n_rules;
    const int32_t* node_domain; const int32_t* rule_key; const int32_t* rule_kind; const int32_t* rule_max_skew;
    const int32_t* rule_min_domains; const int32_t* rule_self; const int32_t* rule_elig_row; const int64_t* rule_off;
    int32_t* rule_cnt; int32_t* rule_dom_nodes; const int32_t* rule_contrib; const uint64_t* rule_elig;
    const int32_t* class_rule_off; const int32_t* inc_off; const int32_t* inc_rule;
    int32_t* placed;              // [n_pegs] out
    int32_t* out;                 // [8] node_count, pods_scheduled, nodes_added, limiter_nodes, last_index_out
    int64_t* sums;                // [2] sum of cpu / memory requests of the scheduled pods
    char* gstate;
};

template <bool kLds, int RMAX_>
CS_GLOBAL CS_LAUNCH_BOUNDS(1024, 1) void estimate_kernel(DevTables t, EstArgs a) {
    using Store = MemStore<kLds, RMAX_>;
    const int tid = cs::tid(), lane = cs::lane(), wave = tid >> 6;
    const int T = cs::nthreads();
    const int R = t.R, Wx = t.Wx, N = a.N, E = a.E;
    const int Q = a.cap / T;
    char* smem = cs::dyn_smem();
    BlockCtl bc;
    bc.red = (uint64_t*)smem;
    bc.tab = (uint32_t*)(bc.red + 32);
    bc.slot = bc.tab + 32;
    bc.W = T >> 6; bc.wave = wave; bc.lane = lane; bc.ph_red = bc.ph_tab = bc.ph_slot = 0;
    Store st;
    st.R = R; st.Wx = Wx; st.cap = a.cap;
    char* base = kLds ? smem + casim_sched_ctrl_bytes(0, 0) : a.gstate;
    st.sfree = (int64_t*)base;
    st.sexcl = (uint64_t*)(st.sfree + (int64_t)R * st.cap);
    uint64_t* schedb = st.sexcl + (int64_t)Wx * st.cap;   // [cap / 64] !Spec.Unschedulable
    st.sslots = (int32_t*)(schedb + (a.cap >> 6));
    st.snpods = st.sslots + st.cap;                        // pods placed by THIS Estimate (newNodesWithPods)
    st.sctmp = st.snpods + st.cap;

    for (int q = 0; q < Q; ++q) {
        const int m = q * T + tid;
        const bool live = m < N;
        for (int r = 0; r < R; ++r) st.sfree[(int64_t)r * st.cap + m] = live ? t.alloc[(int64_t)m * R + r] - t.init_req[(int64_t)m * R + r] : 0;
        for (int w = 0; w < Wx; ++w) st.sexcl[(int64_t)w * st.cap + m] = live ? t.init_excl[(int64_t)m * Wx + w] : 0ull;
        st.sslots[m] = live ? t.allowed[m] - t.init_pods[m] : 0;
        st.snpods[m] = 0;
        const uint64_t sb = cs::ballot(live && !(t.gflags[live ? m : 0] & CASIM_NG_UNSCHEDULABLE));
        if (lane == 0) schedb[m >> 6] = sb;
    }
    // the clones of the template are not in the snapshot yet: they are no domain members and their DaemonSet pods
    // are counted nowhere until the node is created (addNewNodeToSnapshot :326-342)
    auto node_joins = [&](int m, int sign) {
        for (int r = tid; r < a.n_rules; r += T) {
            const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + m];
            if (d < 0) continue;
            const int row = a.rule_elig_row[r];
            const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (m >> 6)] >> (m & 63)) & 1ull);
            if (!el) continue;
            cs::atomic_add_i32(a.rule_dom_nodes + a.rule_off[r] + d, sign);
            const int32_t v = a.rule_contrib[(int64_t)r * N + m];
            if (v != 0) cs::atomic_add_i32(a.rule_cnt + a.rule_off[r] + d, sign * v);
        }
    };
    for (int m = E; m < N; ++m) node_joins(m, -1);
    cs::sync();

    int32_t M = 0;                       // created nodes (estimationState.newNodeNameIndex)
    int32_t last_index = a.last_index;   // lastIndexOrderMapping.lastIndex
    int32_t granted = 0;                 // limiter.nodes
    bool more = true;                    // newNodesAvailable
    int32_t total_placed = 0;
    int64_t sum0 = 0, sum1 = 0;

    for (int k = 0; k < a.n_pegs; ++k) {
        const int c = a.peg_class[k];
        const int32_t cnt = a.peg_count[k];
        typename Store::Peg pv;
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) {
            pv.req[r] = r < R ? t.req[(int64_t)c * R + r] : 0;
            pv.rq[r] = pv.req[r] > 0 ? 1.0 / (double)pv.req[r] : 0.0;
        }
        pv.xblock = t.xblock + (int64_t)c * Wx;
        pv.xmark = t.xmark + (int64_t)c * Wx;
        const uint64_t* fb = a.fbits + (int64_t)c * (a.cap >> 6);
        const uint64_t* xp = a.xports + (int64_t)c * Wx;
        const bool host_spread = a.host_spread && a.host_spread[c] != 0;
        const int r_lo = a.n_rules > 0 ? a.class_rule_off[c] : 0, r_hi = a.n_rules > 0 ? a.class_rule_off[c + 1] : 0;
        const int i_lo = a.n_rules > 0 ? a.inc_off[c] : 0, i_hi = a.n_rules > 0 ? a.inc_off[c + 1] : 0;
        int32_t minv[kMaxRulesPerClass];
        for (int ri = 0; ri < kMaxRulesPerClass; ++ri) minv[ri] = 0;
        bool has_aff = false, aff_self = false, aff_first = false;
        for (int r = r_lo; r < r_hi; ++r) if (a.rule_kind[r] == 2) { has_aff = true; aff_self = a.rule_self[r] != 0; }
        auto refresh_minima = [&]() {   // minMatchNum (filtering.go:54-68) of every spread rule of the class
            if (has_aff) {   // len(affinityCounts) == 0 && the class matches its own terms: the first pod of the series may pass
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
                uint32_t best = 0, nd = 0;
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
        // PodTopologySpread + (after it) the non-hostname anti-affinity rules: 0 pass, 1 spread constraint not met
        // (ErrReasonConstraintsNotMatch), 2 anything else
        auto rules_verdict = [&](int m) -> int {
            bool aff_missing = false;
            for (int r = r_lo; r < r_hi; ++r) {   // (spread rules come first in a class's list: PodTopologySpread runs before InterPodAffinity)
                const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + m];
                if (a.rule_kind[r] == 0) {
                    if (d < 0) return 2;
                    const int64_t skew = (int64_t)cs::load_relaxed_i32(a.rule_cnt + a.rule_off[r] + d) + a.rule_self[r] - minv[r - r_lo];
                    if (skew > a.rule_max_skew[r]) return 1;
                } else if (a.rule_kind[r] == 2) {
                    if (d < 0) return 2;
                    if (cs::load_relaxed_i32(a.rule_cnt + a.rule_off[r] + d) <= 0) aff_missing = true;
                } else if (d >= 0 && cs::load_relaxed_i32(a.rule_cnt + a.rule_off[r] + d) > 0) return 2;
            }
            return (aff_missing && !aff_first) ? 2 : 0;
        };
        // RunFilterPlugins on node m in the default order (NodeUnschedulable / TaintToleration / NodeAffinity = static
        // bit, NodePorts, NodeResourcesFit, PodTopologySpread, InterPodAffinity): 0 pass, 1 failed on a spread
        // constraint with every earlier Filter passing, 2 failed otherwise
        auto filters_verdict = [&](int m) -> int {
            if (!((fb[m >> 6] >> (m & 63)) & 1ull)) return 2;
            for (int w = 0; w < Wx; ++w) if (st.sexcl[(int64_t)w * st.cap + m] & xp[w]) return 2;
            int64_t fr[RMAX_];
#pragma unroll
            for (int r = 0; r < RMAX_; ++r) fr[r] = r < R ? st.sfree[(int64_t)r * st.cap + m] : 0;
            if (capacity_lanes<int64_t, RMAX_>(fr, st.sslots[m], R, pv, 1u) == 0) return 2;
            const int rv = rules_verdict(m);
            if (rv != 0) return rv;
            for (int w = 0; w < Wx; ++w) if (st.sexcl[(int64_t)w * st.cap + m] & pv.xblock[w]) return 2;   // hostname anti-affinity
            return 0;
        };
        // SchedulePod's success path on node m, by its owner thread
        auto commit = [&](int m) {
            st.commit(0, m, 1u, pv);
            for (int ii = i_lo; ii < i_hi; ++ii) {
                const int r = a.inc_rule[ii];
                const int32_t d = a.node_domain[(int64_t)a.rule_key[r] * N + m];
                const int row = a.rule_elig_row[r];
                const bool el = row < 0 || ((a.rule_elig[(int64_t)row * (a.cap >> 6) + (m >> 6)] >> (m & 63)) & 1ull);
                if (d >= 0 && el) cs::atomic_add_i32(a.rule_cnt + a.rule_off[r] + d, 1);
            }
        };
        // SchedulePodOnAnyNodeMatching: first passing node in cyclic order from lastIndex + 1 among list positions
        // [lo_accept, E + M) except `skip`, unschedulable nodes skipped (plugin_runner.go:54-143); -1 = none
        auto walk = [&](int lo_accept, int skip) -> int32_t {
            const int32_t n = E + M;
            if (n <= 0) return -1;
            int32_t m0 = (int32_t)(((int64_t)last_index + 1) % n);
            if (m0 < 0) m0 += n;
            const int Qn = (n + T - 1) / T;
            const int q0 = m0 / T;
            const bool wrap_piece = (m0 % T) != 0;
            const int P = Qn + (wrap_piece ? 1 : 0);
            if (r_hi > r_lo) refresh_minima();
            for (int p = 0; p < P; ++p) {
                int q = p < Qn ? q0 + p : q0;
                if (q >= Qn) q -= Qn;
                const int m = q * T + tid;
                const bool valid = m < n && (p != 0 || m >= m0) && (p != Qn || m < m0);
                bool ok = valid && m >= lo_accept && m != skip && ((schedb[m >> 6] >> lane) & 1ull);
                if (ok) ok = filters_verdict(m) == 0;
                const uint64_t b = cs::ballot(ok);
                uint32_t tot, before;
                bc.count_prefix((uint32_t)cs::popc64(b), tot, before);
                if (tot > 0) {
                    const bool first = ok && before + (uint32_t)cs::mbcnt(b) == 0;
                    if (first) commit(m);
                    const int32_t found = (int32_t)bc.pick(first, (uint32_t)m);
                    last_index = found;   // MarkMatch: position == index, the list only grows at its end
                    return found;
                }
            }
            return -1;
        };
        // RunFiltersOnNode (:146-181) on one node, then SchedulePod: verdict as above, committed when 0
        auto try_node = [&](int m) -> int {
            if (r_hi > r_lo) refresh_minima();
            const bool owner = tid == m % T;
            uint32_t v = 0;
            if (owner) { v = (uint32_t)filters_verdict(m); if (v == 0) commit(m); }
            return (int)bc.pick(owner, v);
        };
        auto account = [&]() { total_placed++; sum0 = cs::wrap_madd_i64(sum0, 1, pv.req[0]); sum1 = cs::wrap_madd_i64(sum1, 1, pv.req[1]); };

        int32_t placed = 0;
        // ---- a2: tryToScheduleOnExistingNodes (:163-186): created nodes only; the first miss ends it ----
        while (placed < cnt && M > 0) {
            if (walk(E, -1) < 0) break;
            placed++; account();
            if (i_hi > i_lo) cs::sync();
        }
        // ---- a3: tryToScheduleOnNewNodes (:190-269) ----
        // A class without domain rules that neither feeds a counter nor excludes itself sees every node on its own: a
        // node takes as many of its pods as fit in one step (the reference would add them one by one to the same node).
        bool plain = r_hi == r_lo && i_hi == i_lo;
        for (int w = 0; w < Wx; ++w) plain = plain && (pv.xblock[w] & pv.xmark[w]) == 0;
        auto fill_node = [&](int m, uint32_t clamp) -> uint32_t {   // RunFiltersOnNode + SchedulePod, up to `clamp` times
            const bool owner = tid == m % T;
            uint32_t x = 0;
            if (owner && ((fb[m >> 6] >> (m & 63)) & 1ull)) {
                x = st.capacity(0, m, pv, clamp, false);
                if (x > 0) st.commit(0, m, x, pv);
            }
            return bc.pick(owner, x);
        };
        while (plain && placed < cnt && more) {
            const int last_node = M > 0 ? E + M - 1 : -1;
            uint32_t x = 0;
            if (last_node >= 0) {
                x = fill_node(last_node, (uint32_t)(cnt - placed));
                placed += (int32_t)x; total_placed += (int32_t)x; sum0 = cs::wrap_madd_i64(sum0, (int64_t)x, pv.req[0]); sum1 = cs::wrap_madd_i64(sum1, (int64_t)x, pv.req[1]);
                if (placed >= cnt) break;
                // the next pod does not fit the newest node; if that node is still empty a fresh one would not help (:234-236)
                if (x == 0 && bc.pick(tid == last_node % T, (uint32_t)st.snpods[last_node]) == 0) break;
            }
            if (a.max_nodes < 0 || (a.max_nodes > 0 && granted >= a.max_nodes)) { more = false; break; }
            granted++;
            if (E + M >= N) { more = false; break; }
            const int fresh = E + M;
            node_joins(fresh, +1);
            M++;
            cs::sync();
            x = fill_node(fresh, (uint32_t)(cnt - placed));
            if (x == 0) break;   // :257-263 the node stays, the PEG is abandoned
            placed += (int32_t)x; total_placed += (int32_t)x; sum0 = cs::wrap_madd_i64(sum0, (int64_t)x, pv.req[0]); sum1 = cs::wrap_madd_i64(sum1, (int64_t)x, pv.req[1]);
        }
        while (!plain && placed < cnt && more) {
            bool found = false;
            const int last_node = M > 0 ? E + M - 1 : -1;
            if (last_node >= 0) {
                const int v = try_node(last_node);
                if (v == 0) found = true;
                else if (v == 1 && host_spread) found = walk(0, last_node) >= 0;   // :212-227 any OTHER node of the snapshot
                if (found && i_hi > i_lo) cs::sync();
            }
            if (!found) {
                if (last_node >= 0) {   // an empty newest node that rejects the pod: a fresh one would too (:234-236)
                    const uint32_t np = bc.pick(tid == last_node % T, (uint32_t)st.snpods[last_node]);
                    if (np == 0) break;
                }
                // limiter.PermissionToAddNode (threshold_based_limiter.go:57-69)
                if (a.max_nodes < 0 || (a.max_nodes > 0 && granted >= a.max_nodes)) { more = false; break; }
                granted++;
                if (E + M >= N) { more = false; break; }   // node table exhausted (the host sized it from the limiter)
                const int fresh = E + M;
                node_joins(fresh, +1);
                M++;
                cs::sync();
                if (try_node(fresh) != 0) break;            // :257-263 the node stays, the PEG is abandoned
                if (i_hi > i_lo) cs::sync();
            }
            placed++; account();
        }
        if (tid == 0) a.placed[k] = placed;
    }

    // len(newNodesWithPods) (:160): every node — created or already in the cluster — that took a pod
    uint32_t with_pods = 0;
    for (int q = 0; q < Q; ++q) {
        const int m = q * T + tid;
        with_pods += (m < E + M && st.snpods[m] > 0) ? 1u : 0u;
    }
    const uint64_t nodes_with_pods = bc.sum(with_pods);
    if (tid == 0) {
        a.out[0] = (int32_t)nodes_with_pods;
        a.out[1] = total_placed;
        a.out[2] = M;
        a.out[3] = granted;
        a.out[4] = last_index;
        a.sums[0] = sum0; a.sums[1] = sum1;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
template <class BK>
class ClusterEstimatorT {
public:
    explicit ClusterEstimatorT(BK& bk) : bk_(bk) {}
    ~ClusterEstimatorT() { for (void* p : allocs_) bk_.free(p); }
    ClusterEstimatorT(const ClusterEstimatorT&) = delete;
    ClusterEstimatorT& operator=(const ClusterEstimatorT&) = delete;

    // CASIM_OK, CASIM_NG_UNSUPPORTED (> 0: delegate) or an error (< 0)
    int32_t init(const casim_pegs* p, const casim_groups* g, const casim_cluster_estimate* ce) {
        if (!p || !g || !ce) return fail(CASIM_ERR_INVALID, "null table");
        if (p->n_pegs < 0 || g->n_groups < 0) return fail(CASIM_ERR_INVALID, "negative size");
        if (p->n_res < 2 || p->n_res > CASIM_KMAX_RES) return fail(CASIM_ERR_INVALID, "n_res must be in [2, 8]");
        C_ = p->n_pegs; N_ = g->n_groups; E_ = ce->n_existing;
        if (E_ < 0 || E_ >= N_) return fail(CASIM_ERR_INVALID, "the node table needs the cluster nodes followed by at least one template clone");
        const size_t C = (size_t)C_, N = (size_t)N_;
        if (C > 0 && (!p->req || !p->flags || !p->count)) return fail(CASIM_ERR_INVALID, "class table has null columns");
        if (!g->alloc || !g->init_req || !g->allowed_pods || !g->init_pods || !g->flags) return fail(CASIM_ERR_INVALID, "node table has null columns");
        if (C > 0 && ((p->w_taint && !p->tol_mask) || (p->w_label && !p->sel_mask) || (p->w_excl && (!p->excl_block || !p->excl_mark))))
            return fail(CASIM_ERR_INVALID, "class mask column missing");
        if ((p->w_taint && !g->taint_mask) || (p->w_label && !g->label_mask) || (p->w_excl && !g->init_excl))
            return fail(CASIM_ERR_INVALID, "node mask column missing");
        for (size_t c = 0; c < C; ++c) {
            if (p->count[c] < 0) return fail(CASIM_ERR_INVALID, "negative PEG count");
            if (p->flags[c] & (CASIM_PEG_UNSUPPORTED | CASIM_PEG_SELF_EXCL_ZONE)) return CASIM_NG_UNSUPPORTED;
            for (int w = 0; w < p->w_zone; ++w)
                if (p->zone_block[c * (size_t)p->w_zone + w] | p->zone_mark[c * (size_t)p->w_zone + w]) return CASIM_NG_UNSUPPORTED;
        }
        const int R = p->n_res;
        // DecreasingPodOrderer (decreasing_pod_orderer.go:46-88): score against the template, ties keep input order
        order_.resize(C);
        std::vector<double> score(C, 0.0);
        const int64_t ca = g->alloc[(size_t)E_ * R + 0], ma = g->alloc[(size_t)E_ * R + 1];
        for (size_t c = 0; c < C; ++c) {
            order_[c] = (int32_t)c;
            if (p->count[c] <= 0) continue;   // Exemplar() == nil
            double s = 0.0;
            if (ca > 0) s += (double)p->req[c * R + 0] / (double)ca;
            if (ma > 0) s += (double)p->req[c * R + 1] / (double)ma;
            score[c] = s;
        }
        for (size_t i = 1; i < C; ++i) {   // stable insertion sort, descending
            const int32_t x = order_[i]; size_t j = i;
            while (j > 0 && score[(size_t)order_[j - 1]] < score[(size_t)x]) { order_[j] = order_[j - 1]; --j; }
            order_[j] = x;
        }
        std::vector<int32_t> cnt(C);
        for (size_t k = 0; k < C; ++k) cnt[k] = p->count[(size_t)order_[k]];

        memset(&dt_, 0, sizeof dt_); memset(&a_, 0, sizeof a_);
        dt_.G = C_; dt_.NG = N_; dt_.R = R; dt_.Wt = p->w_taint; dt_.Wl = p->w_label; dt_.Wx = p->w_excl; dt_.Wz = 0;
        // ONE host-to-device copy for every table of the call (packed into the backend's pinned staging buffer, as SchedulerT and
        // ProblemT do it); a column beyond the bound falls back on a copy of its own
        {
            const size_t masks = (size_t)(dt_.Wt + dt_.Wl + dt_.Wx);
            size_t bound = 64 * 64 + 4096 + C * (8 * (size_t)R + 24 + 8 * (masks + 2 * (size_t)dt_.Wx)) + N * (16 * (size_t)R + 16 + 8 * masks);
            if (ce->rules && ce->rules->n_rules > 0 && ce->rules->rule_offset) {
                const casim_domain_rules* r0 = ce->rules;
                const size_t NR = (size_t)r0->n_rules, tot = (size_t)r0->rule_offset[NR];
                bound += 4 * (size_t)r0->n_keys * N + 64 * (NR + 1) + 8 * tot + 4 * NR * N + 8 * (size_t)r0->n_elig_rows * ((N + 1023) / 64 + 16) + 8 * (C + 1) +
                         4 * (size_t)(r0->inc_off ? r0->inc_off[C] : 0);
            }
            if (getenv("CASIM_TEST_SMALL_UPLOAD_BOUND")) bound = 512;   // (tests: most columns take the fallback copy of their own)
            begin_uploads(bound);
        }
        dt_.req = up(p->req, C * R); dt_.pflags = up(p->flags, C);
        dt_.tol = up(p->tol_mask, C * dt_.Wt); dt_.sel = up(p->sel_mask, C * dt_.Wl);
        dt_.xblock = up(p->excl_block, C * dt_.Wx); dt_.xmark = up(p->excl_mark, C * dt_.Wx);
        dt_.alloc = up(g->alloc, N * R); dt_.init_req = up(g->init_req, N * R);
        dt_.allowed = up(g->allowed_pods, N); dt_.init_pods = up(g->init_pods, N); dt_.gflags = up(g->flags, N);
        dt_.taint = up(g->taint_mask, N * dt_.Wt); dt_.label = up(g->label_mask, N * dt_.Wl);
        dt_.init_excl = up(g->init_excl, N * dt_.Wx);

        int max_threads = 256;
        if (const char* e = getenv("CASIM_SCHED_THREADS")) { const int v = atoi(e); if (v >= 64 && v <= 1024) max_threads = v / 64 * 64; }
        const int64_t n64 = ((int64_t)N_ + 63) & ~63ll;
        threads_ = (int)(n64 < max_threads ? n64 : max_threads);
        cap_ = (int32_t)(((int64_t)N_ + threads_ - 1) / threads_ * threads_);
        S_ = cap_ >> 6;
        a_.N = N_; a_.E = E_; a_.cap = cap_; a_.n_pegs = C_; a_.max_nodes = ce->max_nodes; a_.last_index = ce->last_index;
        a_.peg_class = up(order_.data(), C); a_.peg_count = up(cnt.data(), C);
        d_fbits_ = (uint64_t*)dalloc(8 * C * (size_t)S_);
        a_.fbits = d_fbits_;
        a_.xports = ce->port_block ? up(ce->port_block, C * dt_.Wx) : dt_.xblock;
        a_.placed = (int32_t*)dalloc(4 * C); a_.out = (int32_t*)dalloc(32); a_.sums = (int64_t*)dalloc(16);

        const casim_domain_rules* dr = ce->rules;
        if (dr && dr->n_rules > 0) {
            if (dr->n_nodes != N_ || dr->n_classes != C_) return fail(CASIM_ERR_INVALID, "domain rules were built for other tables");
            for (int c = 0; c < C_; ++c)
                if (dr->class_rule_off[c + 1] - dr->class_rule_off[c] > kMaxRulesPerClass) return CASIM_NG_UNSUPPORTED;
            const size_t NR = (size_t)dr->n_rules, tot = (size_t)dr->rule_offset[NR];
            a_.n_rules = dr->n_rules;
            a_.node_domain = up(dr->node_domain, (size_t)dr->n_keys * N);
            a_.rule_key = up(dr->rule_key, NR); a_.rule_kind = up(dr->rule_kind, NR); a_.rule_max_skew = up(dr->rule_max_skew, NR);
            a_.rule_min_domains = up(dr->rule_min_domains, NR); a_.rule_self = up(dr->rule_self, NR); a_.rule_elig_row = up(dr->rule_elig_row, NR);
            a_.rule_off = up(dr->rule_offset, NR + 1);
            d_rule_init_ = up(dr->count_init, tot); d_dom_init_ = up(dr->domain_nodes, tot); rule_total_ = (int64_t)tot;
            a_.rule_cnt = (int32_t*)dalloc(4 * tot); a_.rule_dom_nodes = (int32_t*)dalloc(4 * tot);
            a_.rule_contrib = up(dr->node_contrib, NR * N);
            if (dr->n_elig_rows > 0) {
                const size_t w_in = (N + 63) / 64;
                std::vector<uint64_t> rows((size_t)dr->n_elig_rows * (size_t)S_, 0ull);
                for (int r = 0; r < dr->n_elig_rows; ++r) for (size_t w = 0; w < w_in; ++w) rows[(size_t)r * (size_t)S_ + w] = dr->elig_bits[(size_t)r * w_in + w];
                a_.rule_elig = up(rows.data(), rows.size());
                if (direct_uploads_ > 0) bk_.sync();   // (`rows` is a local; a packed upload has copied it already)
            }
            a_.class_rule_off = up(dr->class_rule_off, C + 1); a_.inc_off = up(dr->inc_off, C + 1);
            a_.inc_rule = up(dr->inc_rule, (size_t)dr->inc_off[C]);
            // isPodUsingHostNameTopologyKey (:358-369)
            std::vector<uint8_t> hs(C, 0);
            for (size_t r = 0; r < NR; ++r)
                if (dr->rule_kind[r] == 0 && dr->key_is_hostname && dr->key_is_hostname[dr->rule_key[r]]) hs[(size_t)dr->rule_class[r]] = 1;
            a_.host_spread = up(hs.data(), C);
        }
        const int64_t ctrl = casim_sched_ctrl_bytes(0, 0);
        const int64_t bytes = (int64_t)cap_ * (8ll * R + 8ll * dt_.Wx + 12ll) + 8ll * S_;
        lds_ = ctrl + bytes <= (int64_t)bk_.lds_budget();
        smem_ = (size_t)(lds_ ? ctrl + bytes : ctrl);
        if (!lds_) a_.gstate = (char*)dalloc((size_t)bytes);
        end_uploads();
        if (direct_uploads_ > 0) bk_.sync();   // (locals uploaded on their own must have left; packed ones were copied when up() returned)
        if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
        ready_ = true;
        return CASIM_OK;
    }

    int32_t run() {
        if (!ready_) return fail(CASIM_ERR_INVALID, "estimator not initialised");
        if (rule_total_ > 0) {
            bk_.launch(copy_i32_kernel, (int)((rule_total_ + 255) / 256), 1, 256, (size_t)0, a_.rule_cnt, d_rule_init_, rule_total_);
            bk_.launch(copy_i32_kernel, (int)((rule_total_ + 255) / 256), 1, 256, (size_t)0, a_.rule_dom_nodes, d_dom_init_, rule_total_);
        }
        if (C_ > 0) bk_.launch(sched_static_kernel, S_, C_, 64, (size_t)0, dt_, d_fbits_, S_);
        if (dt_.R <= 2) {
            if (lds_) bk_.launch(estimate_kernel<true, 2>, 1, 1, threads_, smem_, dt_, a_);
            else bk_.launch(estimate_kernel<false, 2>, 1, 1, threads_, smem_, dt_, a_);
        } else {
            if (lds_) bk_.launch(estimate_kernel<true, CASIM_KMAX_RES>, 1, 1, threads_, smem_, dt_, a_);
            else bk_.launch(estimate_kernel<false, CASIM_KMAX_RES>, 1, 1, threads_, smem_, dt_, a_);
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    int32_t fetch(casim_cluster_estimate_result* out) {
        if (!ready_ || !out) return fail(CASIM_ERR_INVALID, "nothing to fetch");
        // one round trip through the pinned fetch staging buffer
        char* st = (char*)bk_.stage(1, 64 + 4 * (size_t)(C_ > 0 ? C_ : 1));
        if (!st) return fail(CASIM_ERR_NOMEM, "no staging buffer");
        const int32_t* o = (const int32_t*)st;
        const int64_t* sums = (const int64_t*)(st + 32);
        const int32_t* placed = (const int32_t*)(st + 64);
        bk_.d2h(st, a_.out, 32); bk_.d2h(st + 32, a_.sums, 16);
        if (C_ > 0) bk_.d2h(st + 64, a_.placed, 4 * (size_t)C_);
        bk_.sync();
        out->node_count = o[0]; out->pods_scheduled = o[1]; out->nodes_added = o[2]; out->limiter_nodes = o[3]; out->last_index_out = o[4];
        out->status = CASIM_NG_OK; out->req_cpu_sum = sums[0]; out->req_mem_sum = sums[1];
        for (int k = 0; k < C_; ++k) {
            if (out->order) out->order[k] = order_[(size_t)k];
            if (out->placed) out->placed[k] = placed[(size_t)k];
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    const std::string& error() const { return err_; }
    bool in_lds() const { return lds_; }

private:
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
        T* d = (T*)dalloc(bytes);
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
    char* up_host_ = nullptr; char* up_dev_ = nullptr; size_t up_cap_ = 0, up_used_ = 0; int direct_uploads_ = 0;   // packed uploads
    DevTables dt_; EstArgs a_;
    int C_ = 0, N_ = 0, E_ = 0, S_ = 0, threads_ = 64;
    int32_t cap_ = 0;
    bool ready_ = false, lds_ = true;
    size_t smem_ = 0;
    uint64_t* d_fbits_ = nullptr;
    const int32_t* d_rule_init_ = nullptr; const int32_t* d_dom_init_ = nullptr;
    int64_t rule_total_ = 0;
    std::vector<int32_t> order_;
    std::vector<void*> allocs_;
    std::string err_;
};

}  // namespace casim
