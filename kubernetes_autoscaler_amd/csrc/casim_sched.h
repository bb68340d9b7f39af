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
//   K_sched         ONE wavefront; node m lives in lane (m & 63), slot (m >> 6), state (free resources,
//                   free pod slots, node-local exclusion bits) in LDS or an HBM slab.  The host folds
//                   consecutive pods of one class without hints into a run (class, k); a run is the
//                   closed form of k consecutive cyclic first-fits (the a2 form of casim_pack.h: after t
//                   full rounds node j holds min(c_j, t) pods; bisection on S(T) = sum_j min(c_j, T)),
//                   plus a per-round pass that names the node of every pod of the run.
//
// SimilarPodsScheduling (similar_pods.go:38-98) only memoises "a pod with this spec found no node"; the
// snapshot only fills up during the pass, so a class that failed once fails again whether or not the
// reference consults the memo (pods without a controller are re-tried and fail again): node_out is
// identical, the memo here (one LDS bit per class) just skips the re-scan.
#pragma once
#include <string.h>

#include <string>
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
    int32_t* out;               // [4]: lastIndex, pods scheduled, runs processed, 0
    char* gstate;               // HBM slab (global variant) or null
};

inline int64_t casim_sched_state_bytes(int R, int Wx, int64_t cap, int memo_classes) {
    return cap * (8ll * R + 8ll * Wx + 12ll) + 2ll * 8ll * (cap / 64) + 4ll * ((memo_classes + 31) / 32) + 64;
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

template <bool kLds>
CS_GLOBAL CS_LAUNCH_BOUNDS(64, 1) void sched_kernel(DevTables t, SchedArgs a) {
    using Store = MemStore<kLds>;
    const int lane = cs::lane();
    const int R = t.R, Wx = t.Wx, N = a.N;
    const int S = a.cap >> 6;
    Store st;
    st.R = R; st.Wx = Wx; st.cap = a.cap;
    char* base = kLds ? cs::dyn_smem() : a.gstate;
    st.sfree = (int64_t*)base;
    st.sexcl = (uint64_t*)(st.sfree + (int64_t)R * st.cap);
    uint64_t* scanb = st.sexcl + (int64_t)Wx * st.cap;  // [S] acceptable && !Spec.Unschedulable
    uint64_t* accb = scanb + S;                          // [S] acceptable
    st.sslots = (int32_t*)(accb + S);
    st.snpods = st.sslots + st.cap;
    st.sctmp = st.snpods + st.cap;
    uint32_t* memo = (uint32_t*)(st.sctmp + st.cap);     // [ceil(memo_classes / 32)] class found no node

    // ---- prologue: node state = what the running pods of each node hold (NodeInfo.Requested, types.go) ----
    for (int s = 0; s < S; ++s) {
        const int m = s * 64 + lane;
        const bool live = m < N;
        for (int r = 0; r < R; ++r) st.sfree[(int64_t)r * st.cap + m] = live ? t.alloc[(int64_t)m * R + r] - t.init_req[(int64_t)m * R + r] : 0;
        for (int w = 0; w < Wx; ++w) st.sexcl[(int64_t)w * st.cap + m] = live ? t.init_excl[(int64_t)m * Wx + w] : 0ull;
        st.sslots[m] = live ? t.allowed[m] - t.init_pods[m] : 0;
        st.snpods[m] = 0;
        st.sctmp[m] = 0;
        const bool acc = live && (a.acceptable == nullptr || a.acceptable[m] != 0);
        const uint64_t ab = cs::ballot(acc);
        const uint64_t sb = cs::ballot(acc && !(t.gflags[m < N ? m : 0] & CASIM_NG_UNSCHEDULABLE));
        if (lane == 0) { accb[s] = ab; scanb[s] = sb; }
    }
    for (int i = lane; i < (a.memo_classes + 31) / 32; i += 64) memo[i] = 0u;
    if (kLds) cs::sync();

    int32_t last_index = a.last_index;
    int32_t scheduled = 0;
    int32_t runs_done = 0;
    bool stop = false;

    for (int k0 = 0; k0 < a.n_runs && !stop; k0 += 64) {
        const int kk = k0 + lane;
        const bool have = kk < a.n_runs;
        const int32_t my_class = have ? a.run_class[kk] : 0;
        const int32_t my_count = have ? a.run_count[kk] : 0;
        const int32_t my_hint = have ? a.run_hint[kk] : -1;
        const int32_t my_first = have ? a.run_first[kk] : 0;
        const int nk = a.n_runs - k0 < 64 ? a.n_runs - k0 : 64;
        for (int j = 0; j < nk && !stop; ++j) {
            const int c = (int)cs::bcast_u32((uint32_t)my_class, j);
            const int32_t cnt = (int32_t)cs::bcast_u32((uint32_t)my_count, j);
            const int32_t hint = (int32_t)cs::bcast_u32((uint32_t)my_hint, j);
            const int32_t first = (int32_t)cs::bcast_u32((uint32_t)my_first, j);
            runs_done++;
            if (cnt <= 0) continue;
            typename Store::Peg pv;
#pragma unroll
            for (int r = 0; r < CASIM_KMAX_RES; ++r) {
                pv.req[r] = r < R ? t.req[(int64_t)c * R + r] : 0;
                pv.rq[r] = pv.req[r] > 0 ? 1.0 / (double)pv.req[r] : 0.0;
            }
            pv.xblock = t.xblock + (int64_t)c * Wx;
            pv.xmark = t.xmark + (int64_t)c * Wx;
            bool selfx = (t.pflags[c] & CASIM_PEG_SELF_EXCL_NODE) != 0;
            for (int w = 0; w < Wx; ++w) selfx |= (pv.xblock[w] & pv.xmark[w]) != 0;
            const uint64_t* fb = a.fbits + (int64_t)c * S;
            int32_t placed = 0;

            // ---- tryScheduleUsingHints (:86-110): RunFiltersOnNode on the hinted node, no lastIndex update ----
            if (hint >= 0 && hint < N) {
                const int hs = hint >> 6, owner = hint & 63;
                uint32_t ch = 0;
                if (((fb[hs] & accb[hs]) >> owner) & 1ull) {
                    if (lane == owner) ch = st.capacity(hs, hint, pv, 1u, false);
                    ch = cs::bcast_u32(ch, owner);
                }
                if (ch > 0) {
                    if (lane == owner) st.commit(hs, hint, 1u, pv);
                    if (lane == 0) a.node_out[first] = hint;
                    placed = 1;
                }
            }

            // ---- trySchedule (:114-135): memo, then the closed form of `cnt - placed` cyclic first-fits ----
            const uint32_t keff = (uint32_t)(cnt - placed);
            const bool memo_hit = c < a.memo_classes && ((memo[c >> 5] >> (c & 31)) & 1u);
            if (keff > 0 && !memo_hit) {
                const uint64_t cap1 = (uint64_t)keff + 1;
                auto wsum = [&](uint64_t v) -> uint64_t {
                    if (v > cap1) v = cap1;
                    return cap1 <= (1ull << 25) ? (uint64_t)cs::wave_sum_u32((uint32_t)v) : cs::wave_sum_u32_wide((uint32_t)v);
                };
                // pass A: capacities c_j of the schedulable, acceptable, statically passing nodes
                int32_t n1 = 0;
                for (int s = 0; s < S; ++s) {
                    const int m = s * 64 + lane;
                    const uint64_t elig = fb[s] & scanb[s];  // wave-uniform
                    uint32_t cj = 0;
                    if (elig) {
                        if ((elig >> lane) & 1ull) cj = st.capacity(s, m, pv, keff, selfx);
                        n1 += cs::popc64(cs::ballot(cj > 0));
                    }
                    st.set_c(s, m, cj);
                }
                if (n1 > 0) {
                    uint32_t T, Rr;
                    int32_t got;
                    if ((uint32_t)n1 > keff) { T = 0; Rr = keff; got = (int32_t)keff; }
                    else {
                        uint64_t lane_sum = 0;
                        uint32_t lane_max = 0;
                        for (int s = 0; s < S; ++s) {
                            const uint32_t cj = st.get_c(s, s * 64 + lane);
                            lane_sum += cj;
                            lane_max = cj > lane_max ? cj : lane_max;
                        }
                        const uint64_t tot = wsum(lane_sum);
                        const uint32_t cmax = cs::wave_max_u32(lane_max);
                        if (tot <= keff) { T = cmax; Rr = 0; got = (int32_t)tot; }
                        else {
                            uint32_t lo = 1, hi = cmax; uint64_t slo = (uint64_t)n1;  // S(lo) <= keff < S(hi)
                            while (hi - lo > 1) {
                                const uint32_t mid = lo + ((hi - lo) >> 1);
                                uint64_t ls = 0;
                                for (int s = 0; s < S; ++s) {
                                    const uint32_t cj = st.get_c(s, s * 64 + lane);
                                    ls += cj < mid ? cj : mid;
                                }
                                const uint64_t sm = wsum(ls);
                                if (sm <= keff) { lo = mid; slo = sm; } else hi = mid;
                            }
                            T = lo; Rr = keff - (uint32_t)slo; got = (int32_t)keff;
                        }
                    }
                    const uint32_t Tf = Rr > 0 ? T + 1 : T;
                    const int32_t m0 = (int32_t)(((int64_t)last_index + 1) % N);  // rotated order starts here
                    // one pass per round t: the nodes with c_j >= t, in rotated order, take the round's pods
                    // (pod index = pods of earlier rounds + rotated rank); the last round also commits
                    int32_t pod_base = first + placed;
                    int32_t new_last = last_index;
                    for (uint32_t tr = 1; tr <= Tf; ++tr) {
                        const bool partial = Rr > 0 && tr == Tf;
                        const bool lastr = tr == Tf;
                        int32_t A = 0, Tot = 0;
                        for (int s = 0; s < S; ++s) {
                            const uint64_t b = cs::ballot(st.get_c(s, s * 64 + lane) >= tr);
                            Tot += cs::popc64(b);
                            A += cs::popc64(b & cs::low_mask(m0 - s * 64));
                        }
                        const int32_t take = partial ? (int32_t)Rr : Tot;
                        int32_t basec = 0;
                        for (int s = 0; s < S; ++s) {
                            const int m = s * 64 + lane;
                            const uint32_t cj = st.get_c(s, m);
                            const bool cand = cj >= tr;
                            const uint64_t b = cs::ballot(cand);
                            if (b || lastr) {  // wave-uniform
                                const int32_t pex = basec + cs::mbcnt(b);
                                const int32_t rot = m >= m0 ? pex - A : (Tot - A) + pex;
                                const bool gets = cand && rot < take;
                                if (gets) a.node_out[pod_base + rot] = m;
                                if (lastr) {
                                    const uint64_t hit = cs::ballot(gets && rot == take - 1);
                                    if (hit) new_last = s * 64 + cs::ffs64(hit);
                                    const uint32_t x = (cj < T ? cj : T) + ((partial && gets) ? 1u : 0u);
                                    if (x > 0) st.commit(s, m, x, pv);
                                }
                                basec += cs::popc64(b);
                            }
                        }
                        pod_base += take;
                    }
                    last_index = new_last;
                    placed += got;
                }
            }
            scheduled += placed;
            if (placed < cnt) {
                // SetUnschedulable (:127); breakOnFailure (:79-81)
                if (c < a.memo_classes && lane == 0) memo[c >> 5] |= 1u << (c & 31);
                if (kLds) cs::sync();
                if (a.break_on_failure) stop = true;
            }
        }
    }
    if (lane == 0) {
        a.out[0] = last_index;
        a.out[1] = scheduled;
        a.out[2] = runs_done;
        a.out[3] = 0;
    }
}

// ---- host side: one TrySchedulePods call ------------------------------------------------------------
template <class BK>
class SchedulerT {
public:
    explicit SchedulerT(BK& bk) : bk_(bk) {}
    ~SchedulerT() { for (void* p : allocs_) bk_.free(p); }
    SchedulerT(const SchedulerT&) = delete;
    SchedulerT& operator=(const SchedulerT&) = delete;

    // returns CASIM_OK, CASIM_NG_UNSUPPORTED (> 0) or an error code (< 0)
    int32_t init(const casim_pegs* p, const casim_groups* g, const casim_pod_sequence* q) {
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
        std::vector<int32_t> rc, rn, rh, rf;
        std::vector<uint8_t> used(C, 0);
        for (size_t i = 0; i < P; ++i) {
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
            if (h < 0 && !rc.empty() && rc.back() == c && rh.back() < 0 && rn.back() < 0x7fffffff) rn.back()++;
            else { rc.push_back(c); rn.push_back(1); rh.push_back(h); rf.push_back((int32_t)i); }
        }
        n_runs_ = (int32_t)rc.size();
        if (N_ == 0 || P_ == 0) { trivial_ = true; last_index_ = q->last_index; return CASIM_OK; }

        memset(&dt_, 0, sizeof dt_); memset(&a_, 0, sizeof a_);
        dt_.G = C_; dt_.NG = N_; dt_.R = p->n_res; dt_.Wt = p->w_taint; dt_.Wl = p->w_label; dt_.Wx = p->w_excl; dt_.Wz = 0;
        const int R = dt_.R;
        dt_.req = up(p->req, C * R); dt_.pflags = up(p->flags, C);
        dt_.tol = up(p->tol_mask, C * dt_.Wt); dt_.sel = up(p->sel_mask, C * dt_.Wl);
        dt_.xblock = up(p->excl_block, C * dt_.Wx); dt_.xmark = up(p->excl_mark, C * dt_.Wx);
        dt_.alloc = up(g->alloc, N * R); dt_.init_req = up(g->init_req, N * R);
        dt_.allowed = up(g->allowed_pods, N); dt_.init_pods = up(g->init_pods, N); dt_.gflags = up(g->flags, N);
        dt_.taint = up(g->taint_mask, N * dt_.Wt); dt_.label = up(g->label_mask, N * dt_.Wl);
        dt_.init_excl = up(g->init_excl, N * dt_.Wx);

        cap_ = (int32_t)round_up64_((int64_t)N_);
        S_ = cap_ >> 6;
        a_.N = N_; a_.C = C_; a_.n_runs = n_runs_; a_.break_on_failure = q->break_on_failure ? 1 : 0;
        // lastIndex of a list that shrank since the last loop: any value is a valid cyclic origin
        a_.last_index = q->last_index; a_.cap = cap_;
        a_.memo_classes = C_ < 65536 ? C_ : 65536;
        a_.run_class = up(rc.data(), rc.size()); a_.run_count = up(rn.data(), rn.size());
        a_.run_hint = up(rh.data(), rh.size()); a_.run_first = up(rf.data(), rf.size());
        a_.acceptable = q->node_acceptable ? up(q->node_acceptable, N) : nullptr;
        d_fbits_ = (uint64_t*)dalloc(8 * C * (size_t)S_);
        a_.fbits = d_fbits_;
        a_.node_out = (int32_t*)dalloc(4 * P);
        a_.out = (int32_t*)dalloc(16);
        const int64_t bytes = casim_sched_state_bytes(R, dt_.Wx, cap_, a_.memo_classes);
        smem_ = (size_t)bytes;
        lds_ = bytes <= (int64_t)bk_.lds_budget();
        if (!lds_) a_.gstate = (char*)dalloc((size_t)bytes);
        bk_.sync();  // the run tables above are locals: the uploads must have left them
        if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
        ready_ = true;
        return CASIM_OK;
    }

    int32_t run() {
        if (trivial_) return CASIM_OK;
        if (!ready_) return fail(CASIM_ERR_INVALID, "scheduler not initialised");
        bk_.launch(fill_i32_kernel, (P_ + 255) / 256, 1, 256, (size_t)0, a_.node_out, (int64_t)P_, (int32_t)-1);
        if (C_ > 0) bk_.launch(sched_static_kernel, S_, C_, 64, (size_t)0, dt_, d_fbits_, S_);
        if (lds_) bk_.launch(sched_kernel<true>, 1, 1, 64, smem_, dt_, a_);
        else bk_.launch(sched_kernel<false>, 1, 1, 64, (size_t)0, dt_, a_);
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    int32_t fetch(int32_t* node_out, int32_t* last_index_out, int32_t* n_scheduled_out) {
        if (trivial_) {
            for (int i = 0; i < P_; ++i) if (node_out) node_out[i] = -1;
            if (last_index_out) *last_index_out = last_index_;
            if (n_scheduled_out) *n_scheduled_out = 0;
            return CASIM_OK;
        }
        int32_t o[4] = {0, 0, 0, 0};
        if (node_out) bk_.d2h(node_out, a_.node_out, 4 * (size_t)P_);
        bk_.d2h(o, a_.out, 16);
        bk_.sync();
        if (last_index_out) *last_index_out = o[0];
        if (n_scheduled_out) *n_scheduled_out = o[1];
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    const std::string& error() const { return err_; }
    int runs() const { return n_runs_; }
    bool in_lds() const { return lds_; }

private:
    static int64_t round_up64_(int64_t v) { return (v + 63) & ~63ll; }
    template <class T>
    const T* up(const T* src, size_t n) {
        if (n == 0 || !src) return nullptr;
        T* d = (T*)dalloc(sizeof(T) * n);
        if (d) bk_.h2d(d, src, sizeof(T) * n);
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
    DevTables dt_; SchedArgs a_;
    int C_ = 0, N_ = 0, P_ = 0, S_ = 0;
    int32_t cap_ = 0, n_runs_ = 0, last_index_ = 0;
    bool ready_ = false, trivial_ = false, lds_ = true;
    size_t smem_ = 0;
    uint64_t* d_fbits_ = nullptr;
    std::vector<void*> allocs_;
    std::string err_;
};

}  // namespace casim
