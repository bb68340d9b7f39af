// casim_incr_bench: full encode vs incremental re-encode of a cluster in per-node mode (casim_enc_begin_update / _group_reset /
// _refinalize).  Plain C++ over include/casim.h, host only (no device needed).
//   casim_incr_bench [nodes=15000] [pods_per_node=10] [classes=128] [churn_percent=1] [repeat=5]
// The cluster: every node carries hostname / zone / pool labels and `pods_per_node` running pods of ~1500 controllers (one spec record
// per running pod, as a shim without a spec cache would produce); `classes` pending-pod classes with tolerations, node selectors, a
// quarter of them with a zone spread constraint (domain rules) and a few with hostname anti-affinity (node bits).
// An iteration later `churn_percent` of the nodes changed: one pod left, one pod of a known controller arrived.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

#include "../include/casim.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint32_t rng_state = 2463534242u;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5; return rng_state; }

static int32_t running_spec(casim_encoder* e, int controller) {
    int64_t req[CASIM_MAX_RES] = {100 + 50 * (controller % 5), (int64_t)(128 + 64 * (controller % 4)) << 20};
    const int32_t s = casim_enc_add_pod_spec(e, "default", req);
    const std::string app = "app-" + std::to_string(controller);
    casim_enc_pod_add_label(e, s, "app", app.c_str());
    casim_enc_pod_add_label(e, s, "tier", (controller % 3) ? "backend" : "frontend");
    return s;
}
static void describe_node(casim_encoder* e, int32_t g, int n, const std::vector<int32_t>& specs) {
    char host[32]; snprintf(host, sizeof host, "node-%05d", n);
    casim_enc_group_add_label(e, g, "kubernetes.io/hostname", host);
    casim_enc_group_add_label(e, g, "topology.kubernetes.io/zone", (n % 3) == 0 ? "zone-a" : ((n % 3) == 1 ? "zone-b" : "zone-c"));
    casim_enc_group_add_label(e, g, "pool", (n % 4) ? "general" : "highmem");
    if (n % 16 == 0) casim_enc_group_add_taint(e, g, "dedicated", "infra", "NoSchedule");
    for (int32_t s : specs) casim_enc_group_add_preloaded_pod(e, g, s);
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 15000, P = argc > 2 ? atoi(argv[2]) : 10, C = argc > 3 ? atoi(argv[3]) : 128;
    const double churn = argc > 4 ? atof(argv[4]) : 1.0;
    const int rep = argc > 5 ? atoi(argv[5]) : 5;
    casim_encoder_options o = {}; o.n_res = 2; o.explicit_self_exclusion = 1;
    const int64_t alloc[CASIM_MAX_RES] = {16000, (int64_t)64 << 30};
    std::vector<double> t_calls, t_fin, t_upd_calls, t_refin, t_rows, t_bulk_calls, t_bulk_fin;
    int32_t n_changed = 0, rc_inc = 0, rules = 0;
    for (int r = 0; r < rep; ++r) {
        rng_state = 2463534242u;
        casim_encoder* e = casim_enc_create(&o);
        // ---- iteration 1: the whole cluster through the encoder ----
        const double t0 = now_ms();
        for (int c = 0; c < C; ++c) {
            int64_t req[CASIM_MAX_RES] = {250 + 250 * (c % 8), (int64_t)(512 + 256 * (c % 5)) << 20};
            const int32_t s = casim_enc_add_pod_spec(e, "default", req);
            const std::string app = "pending-" + std::to_string(c);
            casim_enc_pod_add_label(e, s, "app", app.c_str());
            casim_enc_pod_add_toleration(e, s, "dedicated", "Equal", (c % 7) ? "batch" : "infra", "NoSchedule");
            if (c % 3 == 0) casim_enc_pod_add_node_selector(e, s, "pool", (c % 2) ? "general" : "highmem");
            if (c % 4 == 0) { const int32_t ci = casim_enc_pod_add_spread_constraint(e, s, 1 + c % 3, "topology.kubernetes.io/zone", 0);
                              const char* v[1] = {app.c_str()}; casim_enc_spread_add_requirement(e, s, ci, "app", "In", v, 1); }
            if (c % 16 == 1) { const int32_t t = casim_enc_pod_add_anti_affinity_term(e, s, "kubernetes.io/hostname", nullptr, 0);
                               const char* v[1] = {app.c_str()}; casim_enc_term_add_requirement(e, s, t, "app", "In", v, 1); }
            casim_enc_add_peg(e, s, 1 + (int32_t)(rnd() % 40));
        }
        std::vector<std::vector<int32_t>> pods_of((size_t)N);
        for (int n = 0; n < N; ++n) {
            const int32_t g = casim_enc_add_group(e, "", alloc, 110, 16000, (int64_t)64 << 30, 0);
            for (int k = 0; k < P; ++k) pods_of[(size_t)n].push_back(running_spec(e, (int)(rnd() % 1500)));
            describe_node(e, g, n, pods_of[(size_t)n]);
        }
        const double t1 = now_ms();
        if (casim_enc_finalize(e) != CASIM_OK) { fprintf(stderr, "finalize failed\n"); return 1; }
        const double t2 = now_ms();
        casim_domain_rules dr; casim_enc_domain_rules(e, &dr); rules = dr.n_rules;
        // ---- iteration 2: churn ----
        const int K = std::max(1, (int)(N * churn / 100.0));
        std::vector<int32_t> changed;
        const double t3 = now_ms();
        casim_enc_begin_update(e);
        for (int k = 0; k < K; ++k) {
            const int n = (int)(rnd() % (uint32_t)N);
            std::vector<int32_t>& ps = pods_of[(size_t)n];
            if (!ps.empty()) ps.erase(ps.begin() + (rnd() % ps.size()));        // a pod finished
            ps.push_back(running_spec(e, (int)(rnd() % 1500)));                 // a pod of a known controller was bound here
            casim_enc_group_reset(e, n, alloc, 110, 16000, (int64_t)64 << 30, 0);
            describe_node(e, n, n, ps);
        }
        for (int c = 0; c < C; c += 3) casim_enc_set_peg_count(e, c, 1 + (int32_t)(rnd() % 40));
        const double t4 = now_ms();
        changed.resize((size_t)N);
        rc_inc = casim_enc_refinalize(e, changed.data(), N, &n_changed);
        const double t5 = now_ms();
        casim_groups rows;
        if (rc_inc == CASIM_OK) casim_enc_group_rows(e, changed.data(), n_changed, &rows);
        const double t6 = now_ms();
        t_calls.push_back(t1 - t0); t_fin.push_back(t2 - t1); t_upd_calls.push_back(t4 - t3); t_refin.push_back(t5 - t4); t_rows.push_back(t6 - t5);
        casim_enc_destroy(e);
        // ---- the same cluster with the running pods handed over in ONE call (casim_enc_add_running_pods): what a shim that keeps its pods in
        // flat arrays pays; the string table holds every distinct namespace / label key / label value once ----
        {
            rng_state = 2463534242u;
            casim_encoder* b = casim_enc_create(&o);
            const double u0 = now_ms();
            for (int c = 0; c < C; ++c) {
                int64_t req[CASIM_MAX_RES] = {250 + 250 * (c % 8), (int64_t)(512 + 256 * (c % 5)) << 20};
                const int32_t s = casim_enc_add_pod_spec(b, "default", req);
                const std::string app = "pending-" + std::to_string(c);
                casim_enc_pod_add_label(b, s, "app", app.c_str());
                casim_enc_pod_add_toleration(b, s, "dedicated", "Equal", (c % 7) ? "batch" : "infra", "NoSchedule");
                if (c % 3 == 0) casim_enc_pod_add_node_selector(b, s, "pool", (c % 2) ? "general" : "highmem");
                if (c % 4 == 0) { const int32_t ci = casim_enc_pod_add_spread_constraint(b, s, 1 + c % 3, "topology.kubernetes.io/zone", 0);
                                  const char* v[1] = {app.c_str()}; casim_enc_spread_add_requirement(b, s, ci, "app", "In", v, 1); }
                if (c % 16 == 1) { const int32_t t = casim_enc_pod_add_anti_affinity_term(b, s, "kubernetes.io/hostname", nullptr, 0);
                                   const char* v[1] = {app.c_str()}; casim_enc_term_add_requirement(b, s, t, "app", "In", v, 1); }
                casim_enc_add_peg(b, s, 1 + (int32_t)(rnd() % 40));
            }
            // string table: "default", "app", "tier", "backend", "frontend", app-0 .. app-1499
            std::vector<std::string> tab = {"default", "app", "tier", "backend", "frontend"};
            for (int k = 0; k < 1500; ++k) tab.push_back("app-" + std::to_string(k));
            std::vector<const char*> strs; for (auto& x : tab) strs.push_back(x.c_str());
            std::vector<int32_t> grp, nsx, loff = {0}, lk, lv; std::vector<int64_t> rq;
            for (int n = 0; n < N; ++n) {
                const int32_t g = casim_enc_add_group(b, "", alloc, 110, 16000, (int64_t)64 << 30, 0);
                for (int k = 0; k < P; ++k) {
                    const int ctl = (int)(rnd() % 1500);
                    grp.push_back(g); nsx.push_back(0);
                    rq.push_back(100 + 50 * (ctl % 5)); rq.push_back((int64_t)(128 + 64 * (ctl % 4)) << 20);
                    lk.push_back(1); lv.push_back(5 + ctl); lk.push_back(2); lv.push_back((ctl % 3) ? 3 : 4);
                    loff.push_back((int32_t)lk.size());
                }
                describe_node(b, g, n, {});
            }
            const int32_t first = casim_enc_add_running_pods(b, (int32_t)grp.size(), grp.data(), nsx.data(), rq.data(), loff.data(), lk.data(), lv.data(), strs.data(), (int32_t)strs.size());
            const double u1 = now_ms();
            if (first < 0 || casim_enc_finalize(b) != CASIM_OK) { fprintf(stderr, "bulk encode failed\n"); return 1; }
            const double u2 = now_ms();
            t_bulk_calls.push_back(u1 - u0); t_bulk_fin.push_back(u2 - u1);
            casim_enc_destroy(b);
        }
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("{\"what\": \"incremental re-encode, per-node mode\", \"nodes\": %d, \"running_pods\": %d, \"classes\": %d, \"domain_rules\": %d, \"churn_percent\": %.2f, "
           "\"full\": {\"encoder_calls_ms\": %.3f, \"finalize_ms\": %.3f, \"encode_ms\": %.3f}, "
           "\"full_with_bulk_running_pods\": {\"encoder_calls_ms\": %.3f, \"finalize_ms\": %.3f, \"encode_ms\": %.3f}, "
           "\"incremental\": {\"status\": %d, \"nodes_re_described\": %d, \"encoder_calls_ms\": %.3f, \"refinalize_ms\": %.3f, \"group_rows_ms\": %.3f, \"encode_ms\": %.3f}}\n",
           N, N * P, C, rules, churn, med(t_calls), med(t_fin), med(t_calls) + med(t_fin), med(t_bulk_calls), med(t_bulk_fin), med(t_bulk_calls) + med(t_bulk_fin), rc_inc, n_changed, med(t_upd_calls), med(t_refin), med(t_rows),
           med(t_upd_calls) + med(t_refin) + med(t_rows));
    return 0;
}
