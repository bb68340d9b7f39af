// casim_group_bench: times equivalence.BuildPodGroups behind the C ABI (casim_enc_group_pods) at cluster scale.
// Plain C++ over include/casim.h, host only (no device needed).
//   casim_group_bench [n_pods=150000] [controllers=1500] [specs_per_controller=3] [repeat=5]
// Two shapes: "interned" = the shim registered one spec record per distinct spec (it keys them by its own pointer / resourceVersion
// cache) and "one-per-pod" = every pod carries a spec record of its own and the library finds the equal ones by content.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

#include "../include/casim.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int32_t add_spec(casim_encoder* e, int controller, int variant) {
    int64_t req[CASIM_MAX_RES] = {100 + 50 * (variant % 4), (int64_t)(256 + 64 * (variant % 3)) << 20};
    const int32_t s = casim_enc_add_pod_spec(e, "default", req);
    std::string app = "app-" + std::to_string(controller), rev = "rev-" + std::to_string(variant);
    casim_enc_pod_add_label(e, s, "app", app.c_str());
    casim_enc_pod_add_label(e, s, "pod-template-hash", rev.c_str());
    casim_enc_pod_add_toleration(e, s, "dedicated", "Equal", (controller % 7) ? "batch" : "infra", "NoSchedule");
    if (controller % 3 == 0) casim_enc_pod_add_node_selector(e, s, "pool", (controller % 2) ? "general" : "highmem");
    return s;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 150000, C = argc > 2 ? atoi(argv[2]) : 1500, V = argc > 3 ? atoi(argv[3]) : 3, rep = argc > 4 ? atoi(argv[4]) : 5;
    std::vector<std::string> uid_s(C);
    for (int c = 0; c < C; ++c) { char b[64]; snprintf(b, sizeof b, "6f1c2d3e-%04x-4a5b-8c9d-%012x", c & 0xffff, c * 2654435761u); uid_s[c] = b; }
    std::vector<const char*> uid(n);
    std::vector<int> ctl(n), var(n);
    uint32_t x = 12345;
    for (int i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; ctl[i] = (x >> 8) % C; var[i] = (x >> 20) % V; uid[i] = uid_s[ctl[i]].c_str(); }
    casim_encoder_options o = {}; o.n_res = 2;
    for (int shape = 0; shape < 2; ++shape) {
        std::vector<double> t_spec, t_group;
        int32_t ng = 0;
        for (int r = 0; r < rep; ++r) {
            casim_encoder* e = casim_enc_create(&o);
            std::vector<int32_t> spec(n), group(n);
            const double t0 = now_ms();
            if (shape == 0) {
                std::vector<int32_t> cache((size_t)C * V, -1);
                for (int i = 0; i < n; ++i) { int32_t& s = cache[(size_t)ctl[i] * V + var[i]]; if (s < 0) s = add_spec(e, ctl[i], var[i]); spec[i] = s; }
            } else for (int i = 0; i < n; ++i) spec[i] = add_spec(e, ctl[i], var[i]);
            const double t1 = now_ms();
            if (casim_enc_group_pods(e, n, spec.data(), uid.data(), nullptr, group.data(), &ng) != CASIM_OK) { fprintf(stderr, "group_pods failed\n"); return 1; }
            const double t2 = now_ms();
            t_spec.push_back(t1 - t0); t_group.push_back(t2 - t1);
            casim_enc_destroy(e);
        }
        std::sort(t_spec.begin(), t_spec.end()); std::sort(t_group.begin(), t_group.end());
        printf("{\"what\": \"casim_enc_group_pods\", \"shape\": \"%s\", \"pods\": %d, \"controllers\": %d, \"specs_per_controller\": %d, \"groups\": %d, "
               "\"register_specs_ms\": %.3f, \"group_pods_ms\": %.3f, \"ns_per_pod\": %.1f}\n",
               shape == 0 ? "interned spec records" : "one spec record per pod", n, C, V, ng, t_spec[rep / 2], t_group[rep / 2], t_group[rep / 2] * 1e6 / n);
    }
    return 0;
}
