// casim_multi.h — one scale-up batch over SEVERAL devices of one node, driven by ONE host thread (the caller is the single
// RunOnce goroutine, CA/core/scaleup/orchestrator/orchestrator.go:1053): SURVEY 8e inside the library.
//
//   partition   the node groups of every simulation are block-partitioned over the devices (rotated by simulation index so
//               that 20 groups on 8 devices still balance over a batch); the PEG table is replicated; no pods x nodes data
//               ever crosses devices;
//   run         every device gets its own ProblemT on its own stream; all uploads and launches are enqueued before the
//               first wait, so the devices work concurrently;
//   exchange    the only step that needs one: the expander's choice.  Each device reduces its groups to one packed key per
//               simulation (option_kernel, cluster-wide group ids inside); the keys are min-reduced across devices by the
//               `reduce` hook — RCCL all-reduce(min) over xGMI in the product (casim_engine.hip), a host loop in the emulator
//               and when RCCL is switched off.  Chains that start with least-waste (float64 metric) take the key blocks
//               to the host and pick the lexicographic minimum there.
//   results     scattered back into the caller's arrays in the caller's group order (CSR offsets rebuilt).
#pragma once
#include <functional>
#include <memory>

#include "casim_pipeline.h"

namespace casim {

struct GroupRows {   // host copy of the rows of casim_groups that one device owns
    std::vector<int64_t> alloc, init_req, waste_cpu, waste_mem;
    std::vector<int32_t> allowed, init_pods, max_nodes, existing, last_index, peg_lo, peg_hi, global_id, sim_off, peg_offsets, peg_index;
    std::vector<uint32_t> flags;
    std::vector<uint64_t> taint, label, init_excl, init_zone, zone_valid;
    std::vector<double> cap_cpu, cap_mem;
    std::vector<int32_t> src;   // caller's index of every local group
    casim_groups view;
};

template <class BK>
class MultiProblemT {
public:
    // reduce(dev_keys[d] = device pointer to S int64 on device d, S): in-place min across devices (every device ends up
    // with the result); nullptr = reduce on the host
    typedef std::function<bool(const std::vector<int64_t*>&, int)> ReduceFn;
    explicit MultiProblemT(std::vector<BK*> bks) : bks_(std::move(bks)) {}

    int32_t run(const casim_pegs* p, const casim_groups* g, const casim_options* o, casim_results* out, int32_t* offsets_out,
                const casim_option_query* q, const ReduceFn& reduce) {
        if (!p || !g) return fail(CASIM_ERR_INVALID, "null table");
        const int D = (int)bks_.size(), NG = g->n_groups, R = p->n_res;
        if (D <= 0) return fail(CASIM_ERR_INVALID, "no device");
        if (NG < 0 || R < 2 || R > CASIM_KMAX_RES) return fail(CASIM_ERR_INVALID, "bad table sizes");
        if (o && o->node_pods) return fail(CASIM_ERR_INVALID, "casim_options.node_pods is not available on the multi-device path (per-node pod lists are not scattered across devices)");
        // the id a group carries inside expander keys: the caller's global_id, else group_id_base + its index in the caller's
        // table (include/casim.h) — the shards always carry explicit ids, so the rule is applied HERE (with a
        // non-zero base the shards used to carry the bare index and the winner lookup below never matched)
        const int64_t id_base = q ? (int64_t)q->group_id_base : 0;
        for (int i = 0; i < NG; ++i) {
            const int64_t gid = g->global_id ? (int64_t)g->global_id[i] : id_base + i;
            if (gid < 0 || gid >= (1ll << 20)) return fail(CASIM_ERR_INVALID, "group id inside expander keys must be in [0, 2^20)");
        }
        // ---- simulations + owner of every group -------------------------------------------------------
        std::vector<int32_t> so;
        if (g->n_sims > 0) {
            if (!g->sim_offsets || g->sim_offsets[0] != 0 || g->sim_offsets[g->n_sims] != NG) return fail(CASIM_ERR_INVALID, "sim_offsets must run from 0 to n_groups");
            so.assign(g->sim_offsets, g->sim_offsets + g->n_sims + 1);
        } else { so = {0, NG}; }
        const int S = (int)so.size() - 1;
        std::vector<int> owner((size_t)NG, 0);
        for (int s = 0; s < S; ++s) {
            const int a = so[(size_t)s], n = so[(size_t)s + 1] - a;
            if (n < 0) return fail(CASIM_ERR_INVALID, "sim_offsets not monotone");
            for (int d = 0; d < D; ++d) {
                const int r = (d + s) % D;   // block r of simulation s belongs to device d
                const int lo = (int)((int64_t)n * r / D), hi = (int)((int64_t)n * (r + 1) / D);
                for (int i = lo; i < hi; ++i) owner[(size_t)(a + i)] = d;
            }
        }
        // ---- per-device rows ---------------------------------------------------------------------------
        std::vector<GroupRows> rows((size_t)D);
        for (int d = 0; d < D; ++d) rows[(size_t)d].sim_off.push_back(0);
        for (int s = 0; s < S; ++s) {
            for (int i = so[(size_t)s]; i < so[(size_t)s + 1]; ++i) gather(rows[(size_t)owner[(size_t)i]], p, g, i, (int32_t)id_base);
            for (int d = 0; d < D; ++d) rows[(size_t)d].sim_off.push_back((int32_t)rows[(size_t)d].src.size());
        }
        // ---- upload + launch everywhere, then wait -------------------------------------------------------
        std::vector<std::unique_ptr<ProblemT<BK>>> probs;
        std::vector<int64_t*> dev_keys((size_t)D, nullptr);
        // every exit (errors included) leaves no device busy and returns the key blocks to their pools
        struct Guard {
            std::vector<BK*>& bks; std::vector<int64_t*>& keys;
            ~Guard() { for (size_t d = 0; d < bks.size(); ++d) { bks[d]->bind(); bks[d]->sync(); if (keys[d]) { bks[d]->free(keys[d]); keys[d] = nullptr; } } }
        } guard{bks_, dev_keys};
        for (int d = 0; d < D; ++d) {
            GroupRows& gr = rows[(size_t)d];
            finish_view(gr, p, g, S);
            bks_[(size_t)d]->bind();
            probs.emplace_back(new ProblemT<BK>(*bks_[(size_t)d]));
            const int32_t rc = probs.back()->init(p, &gr.view, o);
            if (rc != CASIM_OK) return fail(rc, probs.back()->error().c_str());
        }
        for (int d = 0; d < D; ++d) {
            bks_[(size_t)d]->bind();
            const int32_t rc = probs[(size_t)d]->run();
            if (rc != CASIM_OK) return fail(rc, probs[(size_t)d]->error().c_str());
        }
        // ---- expander: per device reduce -> cross-device min ---------------------------------------------
        const bool want_q = q && q->n_kinds >= 0;
        std::vector<std::vector<int64_t>> keyblocks((size_t)D);
        std::vector<int64_t> packed;
        if (want_q) {
            const bool integer_first = q->n_kinds >= 1 && (q->kinds[0] == CASIM_EXPANDER_LEAST_NODES || q->kinds[0] == CASIM_EXPANDER_MOST_PODS) && q->n_kinds == 1;
            std::vector<std::vector<uint8_t>> valid((size_t)D);
            for (int d = 0; d < D; ++d) {
                bks_[(size_t)d]->bind();
                casim_option_query lq; memset(&lq, 0, sizeof lq);
                lq.kinds = q->kinds; lq.n_kinds = q->n_kinds; lq.group_id_base = q->group_id_base; lq.per_sim = 1;
                if (q->valid) { for (int32_t i : rows[(size_t)d].src) valid[(size_t)d].push_back(q->valid[i]); lq.valid = valid[(size_t)d].data(); }
                dev_keys[(size_t)d] = (int64_t*)bks_[(size_t)d]->alloc(8 * (size_t)S);
                lq.dev_packed_out = dev_keys[(size_t)d];
                keyblocks[(size_t)d].assign(10 * (size_t)S, 0);
                if (!integer_first || !reduce) lq.key_out = keyblocks[(size_t)d].data();   // general chains settle on the host
                const int32_t rc = probs[(size_t)d]->best_option_query(&lq);
                if (rc != CASIM_OK) return fail(rc, probs[(size_t)d]->error().c_str());
            }
            packed.assign((size_t)S, 0x7fffffffffffffffll);
            std::vector<int64_t> win_gid((size_t)S, -1);
            if (integer_first && reduce) {
                if (!reduce(dev_keys, S)) return fail(CASIM_ERR_HIP, "cross-device reduce failed");
                bks_[0]->bind();
                bks_[0]->d2h(packed.data(), dev_keys[0], 8 * (size_t)S); bks_[0]->sync();
                for (int s = 0; s < S; ++s) win_gid[(size_t)s] = packed[(size_t)s] == 0x7fffffffffffffffll ? -1 : (packed[(size_t)s] & 0xfffff);
                if (q->key_out)   // the collective carries the packed key only: [0] packed, [1] the (single, integer) filter's metric, [9] the id
                    for (int s = 0; s < S; ++s) {
                        int64_t* kb = q->key_out + 10 * (size_t)s;
                        const bool none = win_gid[(size_t)s] < 0;
                        for (int f = 0; f < 10; ++f) kb[f] = 0x7fffffffffffffffll;
                        if (!none) { kb[0] = packed[(size_t)s]; kb[1] = (int64_t)((uint64_t)(packed[(size_t)s] >> 20) ^ 0x8000000000000000ull); kb[9] = win_gid[(size_t)s]; }
                    }
                reduced_by_ = 1;
            } else {
                for (int d = 0; d < D; ++d) { bks_[(size_t)d]->bind(); bks_[(size_t)d]->sync(); }
                // lexicographic minimum of (m_1 .. m_k, global id) over the devices' winners == the chain over the union
                for (int s = 0; s < S; ++s) {
                    const int64_t* best = nullptr;
                    for (int d = 0; d < D; ++d) {
                        const int64_t* kb = keyblocks[(size_t)d].data() + 10 * (size_t)s;
                        if (kb[9] == 0x7fffffffffffffffll) continue;
                        bool less = best == nullptr;
                        if (!less) {
                            int c = 0;
                            for (int f = 1; f <= q->n_kinds && c == 0; ++f) c = kb[f] < best[f] ? -1 : (kb[f] > best[f] ? 1 : 0);
                            if (c == 0) c = kb[9] < best[9] ? -1 : 1;
                            less = c < 0;
                        }
                        if (less) best = kb;
                    }
                    if (best) { packed[(size_t)s] = best[0]; win_gid[(size_t)s] = best[9]; if (q->key_out) memcpy(q->key_out + 10 * (size_t)s, best, 80); }
                    else if (q->key_out) for (int f = 0; f < 10; ++f) q->key_out[10 * (size_t)s + f] = 0x7fffffffffffffffll;
                }
                reduced_by_ = 0;
            }
            if (q->packed_out) memcpy(q->packed_out, packed.data(), 8 * (size_t)S);
            if (q->best_out || q->n_best_out) {
                // the caller's index of the winner: the group of simulation s whose cluster-wide id is win_gid
                for (int s = 0; s < S; ++s) {
                    int32_t idx = -1;
                    for (int i = so[(size_t)s]; i < so[(size_t)s + 1] && idx < 0; ++i) {
                        const int64_t gid = g->global_id ? g->global_id[i] : id_base + i;
                        if (win_gid[(size_t)s] >= 0 && (gid & 0xfffff) == (win_gid[(size_t)s] & 0xfffff)) idx = i;
                    }
                    if (q->best_out) q->best_out[s] = idx;
                    if (q->n_best_out) q->n_best_out[s] = idx >= 0 ? 1 : 0;   // survivors are not counted across devices
                }
            }
        }
        // ---- results back in the caller's order ------------------------------------------------------------
        if (out) {
            std::vector<int32_t> nnz_of((size_t)NG, 0);
            std::vector<std::vector<int32_t>> loff((size_t)D), lorder((size_t)D), lplaced((size_t)D);
            for (int d = 0; d < D; ++d) {
                bks_[(size_t)d]->bind();
                GroupRows& gr = rows[(size_t)d];
                const size_t n = gr.src.size();
                loff[(size_t)d].assign(n + 1, 0);
                int32_t nnz = 0;
                int32_t rc = probs[(size_t)d]->csr(&nnz, loff[(size_t)d].data());
                if (rc != CASIM_OK) return fail(rc, probs[(size_t)d]->error().c_str());
                std::vector<int32_t> a(n + 1), b(n + 1), c(n + 1), e(n + 1), f(n + 1), st(n + 1);
                std::vector<int64_t> cs(n + 1), ms(n + 1);
                lorder[(size_t)d].assign((size_t)nnz + 1, 0); lplaced[(size_t)d].assign((size_t)nnz + 1, 0);
                casim_results lr = {a.data(), b.data(), c.data(), e.data(), f.data(), st.data(), cs.data(), ms.data(), lorder[(size_t)d].data(), lplaced[(size_t)d].data()};
                rc = probs[(size_t)d]->fetch(&lr);
                if (rc != CASIM_OK) return fail(rc, probs[(size_t)d]->error().c_str());
                for (size_t k = 0; k < n; ++k) {
                    const int32_t i = gr.src[k];
                    if (out->node_count) out->node_count[i] = a[k];
                    if (out->pods_scheduled) out->pods_scheduled[i] = b[k];
                    if (out->nodes_added) out->nodes_added[i] = c[k];
                    if (out->limiter_nodes) out->limiter_nodes[i] = e[k];
                    if (out->last_index_out) out->last_index_out[i] = f[k];
                    if (out->status) out->status[i] = st[k];
                    if (out->req_cpu_sum) out->req_cpu_sum[i] = cs[k];
                    if (out->req_mem_sum) out->req_mem_sum[i] = ms[k];
                    nnz_of[(size_t)i] = loff[(size_t)d][k + 1] - loff[(size_t)d][k];
                }
            }
            std::vector<int32_t> goff((size_t)NG + 1, 0);
            for (int i = 0; i < NG; ++i) goff[(size_t)i + 1] = goff[(size_t)i] + nnz_of[(size_t)i];
            for (int d = 0; d < D; ++d) {
                const GroupRows& gr = rows[(size_t)d];
                for (size_t k = 0; k < gr.src.size(); ++k) {
                    const int32_t i = gr.src[k], la = loff[(size_t)d][k], n = loff[(size_t)d][k + 1] - la;
                    if (out->order) memcpy(out->order + goff[(size_t)i], lorder[(size_t)d].data() + la, 4 * (size_t)n);
                    if (out->placed) memcpy(out->placed + goff[(size_t)i], lplaced[(size_t)d].data() + la, 4 * (size_t)n);
                }
            }
            if (offsets_out) memcpy(offsets_out, goff.data(), 4 * ((size_t)NG + 1));
        }
        groups_per_device_.clear();
        for (int d = 0; d < D; ++d) groups_per_device_.push_back((int32_t)rows[(size_t)d].src.size());
        return CASIM_OK;
    }
    const std::string& error() const { return err_; }
    int reduced_by() const { return reduced_by_; }   // 1 = the cross-device hook (RCCL), 0 = host
    const std::vector<int32_t>& groups_per_device() const { return groups_per_device_; }

private:
    template <class T> static void row(std::vector<T>& dst, const T* src, int i, int w) { if (src && w > 0) dst.insert(dst.end(), src + (int64_t)i * w, src + (int64_t)(i + 1) * w); }
    static void gather(GroupRows& r, const casim_pegs* p, const casim_groups* g, int i, int32_t id_base) {
        const int R = p->n_res;
        row(r.alloc, g->alloc, i, R); row(r.init_req, g->init_req, i, R); row(r.allowed, g->allowed_pods, i, 1); row(r.init_pods, g->init_pods, i, 1);
        row(r.flags, g->flags, i, 1); row(r.taint, g->taint_mask, i, p->w_taint); row(r.label, g->label_mask, i, p->w_label);
        row(r.init_excl, g->init_excl, i, p->w_excl); row(r.init_zone, g->init_zone, i, p->w_zone); row(r.zone_valid, g->zone_valid, i, p->w_zone);
        row(r.max_nodes, g->max_nodes, i, 1); row(r.existing, g->existing_nodes, i, 1); row(r.last_index, g->last_index, i, 1);
        row(r.cap_cpu, g->cap_cpu, i, 1); row(r.cap_mem, g->cap_mem, i, 1); row(r.waste_cpu, g->waste_cpu, i, 1); row(r.waste_mem, g->waste_mem, i, 1);
        if (g->peg_offsets) {
            if (r.peg_offsets.empty()) r.peg_offsets.push_back(0);
            r.peg_index.insert(r.peg_index.end(), g->peg_index + g->peg_offsets[i], g->peg_index + g->peg_offsets[i + 1]);
            r.peg_offsets.push_back((int32_t)r.peg_index.size());
        } else {
            r.peg_lo.push_back(g->peg_lo ? g->peg_lo[i] : 0); r.peg_hi.push_back(g->peg_hi ? g->peg_hi[i] : p->n_pegs);
        }
        r.global_id.push_back(g->global_id ? g->global_id[i] : id_base + i);
        r.src.push_back(i);
    }
    static void finish_view(GroupRows& r, const casim_pegs* p, const casim_groups* g, int S) {
        casim_groups& v = r.view; memset(&v, 0, sizeof v);
        v.n_groups = (int32_t)r.src.size();
        static const int64_t z64[CASIM_KMAX_RES] = {0}; static const int32_t z32[2] = {0, 0}; static const uint32_t zu32[1] = {0}; static const uint64_t zu64[64] = {0};
        // an empty shard still needs non-null mandatory columns
        v.alloc = r.alloc.empty() ? z64 : r.alloc.data(); v.init_req = r.init_req.empty() ? z64 : r.init_req.data();
        v.allowed_pods = r.allowed.empty() ? z32 : r.allowed.data(); v.init_pods = r.init_pods.empty() ? z32 : r.init_pods.data();
        v.flags = r.flags.empty() ? zu32 : r.flags.data(); v.max_nodes = r.max_nodes.empty() ? z32 : r.max_nodes.data();
        v.existing_nodes = r.existing.empty() ? z32 : r.existing.data(); v.last_index = r.last_index.empty() ? z32 : r.last_index.data();
        v.taint_mask = r.taint.empty() ? zu64 : r.taint.data(); v.label_mask = r.label.empty() ? zu64 : r.label.data();
        v.init_excl = r.init_excl.empty() ? zu64 : r.init_excl.data(); v.init_zone = r.init_zone.empty() ? zu64 : r.init_zone.data();
        v.zone_valid = r.zone_valid.empty() ? zu64 : r.zone_valid.data();
        static const double zf64[1] = {0.0};
        v.cap_cpu = g->cap_cpu ? (r.cap_cpu.empty() ? zf64 : r.cap_cpu.data()) : nullptr; v.cap_mem = g->cap_mem ? (r.cap_mem.empty() ? zf64 : r.cap_mem.data()) : nullptr;
        v.waste_cpu = g->waste_cpu ? (r.waste_cpu.empty() ? z64 : r.waste_cpu.data()) : nullptr;
        v.waste_mem = g->waste_mem ? (r.waste_mem.empty() ? z64 : r.waste_mem.data()) : nullptr;
        if (g->peg_offsets) { if (r.peg_offsets.empty()) r.peg_offsets.push_back(0); v.peg_offsets = r.peg_offsets.data(); v.peg_index = r.peg_index.empty() ? z32 : r.peg_index.data(); }
        else { v.peg_lo = r.peg_lo.empty() ? z32 : r.peg_lo.data(); v.peg_hi = r.peg_hi.empty() ? z32 : r.peg_hi.data(); }
        v.global_id = r.global_id.empty() ? z32 : r.global_id.data();
        v.n_sims = S; v.sim_offsets = r.sim_off.data();
        (void)p;
    }
    int32_t fail(int32_t code, const char* msg) { err_ = msg ? msg : ""; return code; }

    std::vector<BK*> bks_;
    std::string err_;
    int reduced_by_ = 0;
    std::vector<int32_t> groups_per_device_;
};

}  // namespace casim
