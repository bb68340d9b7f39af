// casim_prefetch.cpp — the prefetch cache of the estimator shim, host C++ over the public C ABI.
//
// Reference shape: ComputeExpansionOption builds a fresh estimator per node group and calls Estimate(podGroups, nodeInfo, nodeGroup)
// ONCE per group (CA/core/scaleup/orchestrator/orchestrator.go:383-427; estimator.Estimator, CA/estimator/estimator.go:53-56).  A
// device wants the whole loop in one launch, so the shim's NodeGroupListProcessor wrapper — which sees every candidate group, every
// template and the pending pods before the loop (orchestrator.go:121-123) — runs ONE batch over all of them (fill) and every
// Estimate() becomes a lookup.  The cache is an accelerator, never the source of truth: a lookup hits only when the call asks
// EXACTLY the question the batch answered —
//   * the node group (caller's 64-bit key: nodeGroup.Id() + template generation),
//   * the PEG list, as a SET of keys (the orchestrator passes SchedulablePodGroups' result, the batch derived the same subset on the
//     device; a caller that filtered differently, or re-grouped the pods, misses).  Not as a sequence: the orchestrator's list comes
//     out of a Go map (equivalence.groupPodsBySchedulingProperties, CA/core/scaleup/equivalence/groups.go:62-104: `range` over
//     map[equivalenceGroupId]) — its order is random per loop, and the estimator sorts the PEGs by score before it uses them
//     (DecreasingPodOrderer); input order only breaks score ties, which the reference therefore breaks at random itself,
//   * the limiter's answer (max_nodes after StartEstimation), the snapshot's node count E and lastIndex (with casim_options.chain_last_index:
//     the lastIndex the group's predecessor in the batch left behind — the runner's value when the calls arrive in the batch's order) —
// and everything else is a miss with its reason, answered by the per-call path (one casim_estimate_batch with one group).
// Keys are opaque to this file: the Go side hashes what it has (pointer of the exemplar pod, group id string).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/casim.h"

namespace {
struct Entry {
    int32_t max_nodes = 0, existing = 0, last_index = 0;
    std::vector<uint64_t> peg_keys;   // the group's schedulable PEGs, table order
    std::vector<int32_t> by_key;      // positions into peg_keys, ascending key (set comparison, mapping to the caller's positions)
    std::vector<int32_t> order;       // processing order, as positions into peg_keys
    std::vector<int32_t> placed;
    casim_prefetch_result r;
};
}  // namespace

struct casim_prefetch {
    casim_ctx* ctx = nullptr;
    std::unordered_map<uint64_t, Entry> by_group;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // fills, groups cached, hits, miss: unknown group, PEG list, limits, (reserved x2)
    std::string err;
};

extern "C" {

casim_prefetch* casim_prefetch_create(casim_ctx* ctx) {
    if (!ctx) return nullptr;
    casim_prefetch* p = new (std::nothrow) casim_prefetch();
    if (p) p->ctx = ctx;
    return p;
}
void casim_prefetch_destroy(casim_prefetch* p) { delete p; }
void casim_prefetch_clear(casim_prefetch* p) { if (p) p->by_group.clear(); }
const char* casim_prefetch_error(const casim_prefetch* p) { return p ? p->err.c_str() : "null cache"; }

int32_t casim_prefetch_fill(casim_prefetch* p, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                            const uint64_t* group_key, const uint64_t* peg_key) {
    if (!p || !pegs || !groups || !group_key || !peg_key) return CASIM_ERR_INVALID;
    p->err.clear();
    p->by_group.clear();
    const int NG = groups->n_groups, G = pegs->n_pegs;
    if (NG <= 0) return CASIM_OK;
    // one call, one wait for the device (casim_estimate_batch_query keeps the upload, the kernels, the results and the offsets on one
    // round trip): the arrays are sized for the bound of the lists — explicit offsets, the candidate ranges, or every PEG for every group
    int64_t cap = 0;
    if (groups->peg_offsets) cap = groups->peg_offsets[NG];
    else if (groups->peg_lo && groups->peg_hi) for (int i = 0; i < NG; ++i) cap += (int64_t)groups->peg_hi[i] - groups->peg_lo[i];
    else cap = (int64_t)NG * G;
    if (cap < 0 || cap > 0x7fffffffll) { p->err = "PEG lists too long"; return CASIM_ERR_INVALID; }
    std::vector<int32_t> off((size_t)NG + 1, 0);
    std::vector<int32_t> a((size_t)NG * 6), order((size_t)cap + 1), placed((size_t)cap + 1);
    std::vector<int64_t> sums((size_t)NG * 2);
    casim_results r; memset(&r, 0, sizeof r);
    r.node_count = a.data(); r.pods_scheduled = a.data() + NG; r.nodes_added = a.data() + 2 * (size_t)NG; r.limiter_nodes = a.data() + 3 * (size_t)NG;
    r.last_index_out = a.data() + 4 * (size_t)NG; r.status = a.data() + 5 * (size_t)NG; r.req_cpu_sum = sums.data(); r.req_mem_sum = sums.data() + NG;
    r.order = order.data(); r.placed = placed.data();
    const int32_t rc = casim_estimate_batch_query(p->ctx, pegs, groups, opts, &r, off.data(), nullptr);
    if (rc != CASIM_OK) { p->err = casim_last_error(); return rc == CASIM_ERR_NO_DEVICE || casim_device_count() > 0 ? rc : CASIM_ERR_NO_DEVICE; }
    std::vector<int32_t> ids, pos((size_t)G, -1);
    for (int i = 0; i < NG; ++i) {
        Entry e;
        e.max_nodes = groups->max_nodes ? groups->max_nodes[i] : 0;
        e.existing = groups->existing_nodes ? groups->existing_nodes[i] : 0;
        e.last_index = groups->last_index ? groups->last_index[i] : 0;
        // casim_options.chain_last_index: group i ran with the lastIndex its predecessor (of the same simulation) left — THAT is the question the
        // batch answered for it, and what a lookup has to come with (the shim passes the runner's current lastIndex: a hit then proves that the
        // Estimate() calls so far arrived in the batch's order, and the shim moves the runner on to this group's last_index_out)
        if (opts && opts->chain_last_index && i > 0) {
            bool first_of_sim = false;
            if (groups->n_sims > 0 && groups->sim_offsets) for (int32_t s2 = 0; s2 < groups->n_sims && !first_of_sim; ++s2) first_of_sim = groups->sim_offsets[s2] == i;
            if (!first_of_sim) e.last_index = a[4 * (size_t)NG + (size_t)i - 1];
        }
        const int32_t lo = off[(size_t)i], n = off[(size_t)i + 1] - lo;
        // Estimate() receives the schedulable PEGs in the order of the caller's list: explicit lists keep theirs, device-derived
        // subsets come in table order (ascending PEG id)
        ids.assign(order.begin() + lo, order.begin() + lo + n);
        if (groups->peg_offsets && groups->peg_index) for (int k = 0; k < n; ++k) ids[(size_t)k] = groups->peg_index[groups->peg_offsets[i] + k];
        else std::sort(ids.begin(), ids.end());
        e.peg_keys.resize((size_t)n); e.order.resize((size_t)n); e.placed.assign(placed.begin() + lo, placed.begin() + lo + n);
        for (int k = 0; k < n; ++k) { e.peg_keys[(size_t)k] = peg_key[ids[(size_t)k]]; pos[(size_t)ids[(size_t)k]] = k; }
        for (int k = 0; k < n; ++k) e.order[(size_t)k] = pos[(size_t)order[(size_t)(lo + k)]];
        e.by_key.resize((size_t)n);
        for (int k = 0; k < n; ++k) e.by_key[(size_t)k] = k;
        std::sort(e.by_key.begin(), e.by_key.end(), [&](int32_t x, int32_t y) { return e.peg_keys[(size_t)x] < e.peg_keys[(size_t)y]; });
        memset(&e.r, 0, sizeof e.r);
        e.r.node_count = a[(size_t)i]; e.r.pods_scheduled = a[(size_t)NG + i]; e.r.nodes_added = a[2 * (size_t)NG + i]; e.r.limiter_nodes = a[3 * (size_t)NG + i];
        e.r.last_index_out = a[4 * (size_t)NG + i]; e.r.status = a[5 * (size_t)NG + i]; e.r.req_cpu_sum = sums[(size_t)i]; e.r.req_mem_sum = sums[(size_t)NG + i];
        e.r.n_pegs = n;
        p->by_group[group_key[i]] = std::move(e);   // (a key given twice keeps the later group: the caller's keys are its business)
    }
    p->stats[0]++; p->stats[1] += NG;
    return CASIM_OK;
}

int32_t casim_prefetch_lookup(casim_prefetch* p, uint64_t group_key, const uint64_t* peg_keys, int32_t n_pegs, int32_t max_nodes,
                              int32_t existing_nodes, int32_t last_index, casim_prefetch_result* out, int32_t* order_out, int32_t* placed_out) {
    if (!p || !out || n_pegs < 0 || (n_pegs > 0 && !peg_keys)) return CASIM_ERR_INVALID;
    memset(out, 0, sizeof *out);
    auto it = p->by_group.find(group_key);
    if (it == p->by_group.end()) { out->miss_reason = CASIM_PREFETCH_MISS_GROUP; p->stats[3]++; return CASIM_PREFETCH_MISS; }
    const Entry& e = it->second;
    // the caller's list against the group's subset as SETS; to_caller[i] = where the batch's i-th PEG stands in the caller's list
    std::vector<int32_t> mine((size_t)n_pegs), to_caller((size_t)n_pegs);
    bool same = (size_t)n_pegs == e.peg_keys.size();
    if (same) {
        for (int32_t k = 0; k < n_pegs; ++k) mine[(size_t)k] = k;
        std::sort(mine.begin(), mine.end(), [&](int32_t x, int32_t y) { return peg_keys[x] < peg_keys[y]; });
        for (int32_t k = 0; k < n_pegs && same; ++k) {
            same = peg_keys[mine[(size_t)k]] == e.peg_keys[(size_t)e.by_key[(size_t)k]] && (k == 0 || peg_keys[mine[(size_t)k]] != peg_keys[mine[(size_t)k - 1]]);
            to_caller[(size_t)e.by_key[(size_t)k]] = mine[(size_t)k];
        }
    }
    if (!same) { out->miss_reason = CASIM_PREFETCH_MISS_PEGS; p->stats[4]++; return CASIM_PREFETCH_MISS; }
    if (max_nodes != e.max_nodes || existing_nodes != e.existing) { out->miss_reason = CASIM_PREFETCH_MISS_LIMITS; p->stats[5]++; return CASIM_PREFETCH_MISS; }
    if (last_index != e.last_index) {   // the limiter's answers agree: the call only comes with another lastIndex (a chained batch whose order was left)
        out->miss_reason = CASIM_PREFETCH_MISS_LAST_INDEX; p->stats[5]++; p->stats[6]++; return CASIM_PREFETCH_MISS;
    }
    *out = e.r;
    if (order_out) for (int32_t k = 0; k < n_pegs; ++k) order_out[k] = to_caller[(size_t)e.order[(size_t)k]];
    if (placed_out) memcpy(placed_out, e.placed.data(), sizeof(int32_t) * (size_t)n_pegs);
    p->stats[2]++;
    return CASIM_OK;
}

int32_t casim_prefetch_stats(const casim_prefetch* p, int64_t out[8]) {
    if (!p || !out) return CASIM_ERR_INVALID;
    memcpy(out, p->stats, sizeof p->stats);
    return CASIM_OK;
}

}  // extern "C"
