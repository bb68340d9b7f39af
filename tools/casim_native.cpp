// casim_native — native end-to-end harness for libcasim.so: plain C++ against include/casim.h, no Python, no torch.
//
//   casim_native TRACE [--device N] [--dump FILE] [--repeat K]
//
// TRACE is a call trace written by kubernetes_autoscaler_amd/trace.py: every casim_enc_* call a host (the cgo shim of
// INTEGRATION.md) makes for one loop iteration, followed by one '@' directive naming the engine entry point.  The harness
// parses the trace into typed argument blocks FIRST, then times
//     encode    (all casim_enc_* calls + casim_enc_finalize)
//     the engine call(s): casim_estimate_batch_timed (tables -> HBM, kernels, results -> host, phase by phase),
//                         casim_try_schedule_pods, casim_simulate_node_removals (whole call, enter -> return)
// and prints one JSON object.  --dump writes the raw results (int32 arrays) for the parity tests.
// This is what a Go shim would see: the Python mirror spends most of a call in ctypes and object walking.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../include/casim.h"

namespace {

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

// argument kinds of one trace line: i = int32, l = int64, d = double, s = string (may be NULL), S = string array,
// A = int64 array, a = int32 array (arrays: count followed by the items)
struct Sig { int op; const char* args; };
enum Op {
    CREATE, ADD_GROUP, GROUP_LABEL, GROUP_TAINT, GROUP_FP_CAP, GROUP_LIMITS, GROUP_PRELOADED, GROUP_SET_PEGS, ADD_POD_SPEC, POD_LABEL,
    POD_TOLERATION, POD_NODE_SELECTOR, POD_NODE_AFF_REQ, POD_NODE_AFF_TERM, NODE_TERM_REQ, POD_HOST_PORT, POD_AA_TERM, TERM_REQ,
    POD_AFF_TERM, AFF_TERM_REQ, POD_SPREAD, SPREAD_REQ, SPREAD_TAINTS, SPREAD_AFFINITY, ADD_NAMESPACE, NAMESPACE_LABEL, TERM_NS_SELECTOR, TERM_NS_REQ, POD_FP_REQ,
    POD_UNSUPPORTED, POD_SPEC_EXTRA, ADD_PEG, ADD_RESOURCE_PEGS, ADD_EXISTING_POD, FINALIZE
};
const std::map<std::string, Sig> kSigs = {
    {"casim_enc_create", {CREATE, "iii"}},
    {"casim_enc_add_group", {ADD_GROUP, "sAilli"}},
    {"casim_enc_group_add_label", {GROUP_LABEL, "iss"}},
    {"casim_enc_group_add_taint", {GROUP_TAINT, "isss"}},
    {"casim_enc_group_set_fastpath_capacity", {GROUP_FP_CAP, "idd"}},
    {"casim_enc_group_set_limits", {GROUP_LIMITS, "iiii"}},
    {"casim_enc_group_add_preloaded_pod", {GROUP_PRELOADED, "ii"}},
    {"casim_enc_group_set_pegs", {GROUP_SET_PEGS, "iai"}},
    {"casim_enc_add_pod_spec", {ADD_POD_SPEC, "sA"}},
    {"casim_enc_pod_add_label", {POD_LABEL, "iss"}},
    {"casim_enc_pod_add_toleration", {POD_TOLERATION, "issss"}},
    {"casim_enc_pod_add_node_selector", {POD_NODE_SELECTOR, "iss"}},
    {"casim_enc_pod_add_node_affinity_req", {POD_NODE_AFF_REQ, "issSi"}},
    {"casim_enc_pod_add_node_affinity_term", {POD_NODE_AFF_TERM, "i"}},
    {"casim_enc_node_term_add_requirement", {NODE_TERM_REQ, "iiissSi"}},
    {"casim_enc_pod_add_host_port", {POD_HOST_PORT, "issi"}},
    {"casim_enc_pod_add_anti_affinity_term", {POD_AA_TERM, "isSi"}},
    {"casim_enc_term_add_requirement", {TERM_REQ, "iissSi"}},
    {"casim_enc_pod_add_affinity_term", {POD_AFF_TERM, "isSi"}},
    {"casim_enc_aff_term_add_requirement", {AFF_TERM_REQ, "iissSi"}},
    {"casim_enc_pod_add_spread_constraint", {POD_SPREAD, "iisi"}},
    {"casim_enc_spread_add_requirement", {SPREAD_REQ, "iissSi"}},
    {"casim_enc_spread_set_taints_policy", {SPREAD_TAINTS, "iii"}},
    {"casim_enc_spread_set_affinity_policy", {SPREAD_AFFINITY, "iii"}},
    {"casim_enc_add_namespace", {ADD_NAMESPACE, "s"}},
    {"casim_enc_namespace_add_label", {NAMESPACE_LABEL, "sss"}},
    {"casim_enc_term_set_namespace_selector", {TERM_NS_SELECTOR, "ii"}},
    {"casim_enc_term_add_namespace_requirement", {TERM_NS_REQ, "iissSi"}},
    {"casim_enc_pod_set_fastpath_requests", {POD_FP_REQ, "idd"}},
    {"casim_enc_pod_mark_unsupported", {POD_UNSUPPORTED, "is"}},
    {"casim_enc_pod_set_spec_extra", {POD_SPEC_EXTRA, "is"}},
    {"casim_enc_add_peg", {ADD_PEG, "ii"}},
    {"casim_enc_add_resource_pegs", {ADD_RESOURCE_PEGS, "siAaa"}},
    {"casim_enc_add_existing_pod", {ADD_EXISTING_POD, "iSSi"}},
    {"casim_enc_finalize", {FINALIZE, ""}},
};

struct Call {
    int op = 0;
    std::vector<int64_t> I;
    std::vector<double> D;
    std::vector<std::string> S;           // storage
    std::vector<uint8_t> S_null;
    std::vector<std::vector<std::string>> SA_store;
    std::vector<std::vector<const char*>> SA;
    std::vector<std::vector<int64_t>> A64;
    std::vector<std::vector<int32_t>> A32;
    const char* s(size_t k) const { return S_null[k] ? nullptr : S[k].c_str(); }
};

std::string unesc(const std::string& t) {
    std::string o; o.reserve(t.size());
    for (size_t i = 0; i < t.size(); ++i) {
        if (t[i] == '\\' && i + 1 < t.size()) { ++i; o.push_back(t[i] == 't' ? '\t' : (t[i] == 'n' ? '\n' : t[i])); }
        else o.push_back(t[i]);
    }
    return o;
}
std::vector<std::string> split_tabs(const std::string& line) {
    std::vector<std::string> out; size_t a = 0;
    for (;;) { const size_t b = line.find('\t', a); if (b == std::string::npos) { out.push_back(line.substr(a)); break; } out.push_back(line.substr(a, b - a)); a = b + 1; }
    return out;
}

struct Directive { std::string name; std::vector<int64_t> v; };

bool parse_trace(const char* path, std::vector<Call>& calls, Directive& dir, std::string& err) {
    std::ifstream f(path);
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::string line; size_t ln = 0;
    while (std::getline(f, line)) {
        ++ln;
        if (line.empty()) continue;
        std::vector<std::string> t = split_tabs(line);
        if (t[0][0] == '@') {
            dir.name = t[0].substr(1);
            for (size_t i = 1; i < t.size(); ++i) dir.v.push_back(strtoll(t[i].c_str(), nullptr, 10));
            continue;
        }
        auto it = kSigs.find(t[0]);
        if (it == kSigs.end()) { err = "line " + std::to_string(ln) + ": unknown call " + t[0]; return false; }
        calls.emplace_back();
        Call& c = calls.back(); c.op = it->second.op;
        size_t k = 1;
        auto need = [&](size_t n) { if (k + n > t.size()) { err = "line " + std::to_string(ln) + ": short argument list for " + t[0]; return false; } return true; };
        for (const char* a = it->second.args; *a; ++a) {
            if (*a == 'i' || *a == 'l') { if (!need(1)) return false; c.I.push_back(strtoll(t[k++].c_str(), nullptr, 10)); }
            else if (*a == 'd') { if (!need(1)) return false; c.D.push_back(strtod(t[k++].c_str(), nullptr)); }
            else if (*a == 's') { if (!need(1)) return false; c.S_null.push_back(t[k] == "~"); c.S.push_back(unesc(t[k++])); }
            else {
                if (!need(1)) return false;
                const size_t n = (size_t)strtoll(t[k++].c_str(), nullptr, 10);
                if (!need(n)) return false;
                if (*a == 'S') { c.SA_store.emplace_back(); for (size_t j = 0; j < n; ++j) c.SA_store.back().push_back(unesc(t[k++])); }
                else if (*a == 'A') { c.A64.emplace_back(); for (size_t j = 0; j < n; ++j) c.A64.back().push_back(strtoll(t[k++].c_str(), nullptr, 10)); }
                else { c.A32.emplace_back(); for (size_t j = 0; j < n; ++j) c.A32.back().push_back((int32_t)strtoll(t[k++].c_str(), nullptr, 10)); }
            }
        }
    }
    for (Call& c : calls) {   // pointer arrays after the storage stopped moving
        for (auto& v : c.SA_store) { c.SA.emplace_back(); for (auto& s : v) c.SA.back().push_back(s.c_str()); if (c.SA.back().empty()) c.SA.back().push_back(nullptr); }
        for (auto& v : c.A64) if (v.size() < CASIM_MAX_RES) v.resize(CASIM_MAX_RES, 0);   // lane vectors are read as MAX_RES slots by some hosts
    }
    return true;
}

// replay of the encoder calls; returns the first failing status (< 0) or 0
int32_t replay(const std::vector<Call>& calls, size_t n_calls, casim_encoder*& e) {
    for (size_t ci = 0; ci < n_calls; ++ci) {
        const Call& c = calls[ci];
        int32_t rc = 0;
        const auto& I = c.I;
        switch (c.op) {
        case CREATE: {
            casim_encoder_options o; memset(&o, 0, sizeof o);
            o.n_res = (int32_t)I[0]; o.enable_taint_comparison_ops = (int32_t)I[1]; o.explicit_self_exclusion = (int32_t)I[2];
            e = casim_enc_create(&o); rc = e ? 0 : -1; break;
        }
        case ADD_GROUP: rc = casim_enc_add_group(e, c.s(0), c.A64[0].data(), (int32_t)I[0], I[1], I[2], (int32_t)I[3]); break;
        case GROUP_LABEL: rc = casim_enc_group_add_label(e, (int32_t)I[0], c.s(0), c.s(1)); break;
        case GROUP_TAINT: rc = casim_enc_group_add_taint(e, (int32_t)I[0], c.s(0), c.s(1), c.s(2)); break;
        case GROUP_FP_CAP: rc = casim_enc_group_set_fastpath_capacity(e, (int32_t)I[0], c.D[0], c.D[1]); break;
        case GROUP_LIMITS: rc = casim_enc_group_set_limits(e, (int32_t)I[0], (int32_t)I[1], (int32_t)I[2], (int32_t)I[3]); break;
        case GROUP_PRELOADED: rc = casim_enc_group_add_preloaded_pod(e, (int32_t)I[0], (int32_t)I[1]); break;
        case GROUP_SET_PEGS: rc = casim_enc_group_set_pegs(e, (int32_t)I[0], c.A32[0].data(), (int32_t)I[1]); break;
        case ADD_POD_SPEC: rc = casim_enc_add_pod_spec(e, c.s(0), c.A64[0].data()); break;
        case POD_LABEL: rc = casim_enc_pod_add_label(e, (int32_t)I[0], c.s(0), c.s(1)); break;
        case POD_TOLERATION: rc = casim_enc_pod_add_toleration(e, (int32_t)I[0], c.s(0), c.s(1), c.s(2), c.s(3)); break;
        case POD_NODE_SELECTOR: rc = casim_enc_pod_add_node_selector(e, (int32_t)I[0], c.s(0), c.s(1)); break;
        case POD_NODE_AFF_REQ: rc = casim_enc_pod_add_node_affinity_req(e, (int32_t)I[0], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[1]); break;
        case POD_NODE_AFF_TERM: rc = casim_enc_pod_add_node_affinity_term(e, (int32_t)I[0]); break;
        case NODE_TERM_REQ: rc = casim_enc_node_term_add_requirement(e, (int32_t)I[0], (int32_t)I[1], (int32_t)I[2], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[3]); break;
        case POD_HOST_PORT: rc = casim_enc_pod_add_host_port(e, (int32_t)I[0], c.s(0), c.s(1), (int32_t)I[1]); break;
        case POD_AA_TERM: rc = casim_enc_pod_add_anti_affinity_term(e, (int32_t)I[0], c.s(0), c.SA[0].data(), (int32_t)I[1]); break;
        case TERM_REQ: rc = casim_enc_term_add_requirement(e, (int32_t)I[0], (int32_t)I[1], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[2]); break;
        case POD_AFF_TERM: rc = casim_enc_pod_add_affinity_term(e, (int32_t)I[0], c.s(0), c.SA[0].data(), (int32_t)I[1]); break;
        case AFF_TERM_REQ: rc = casim_enc_aff_term_add_requirement(e, (int32_t)I[0], (int32_t)I[1], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[2]); break;
        case POD_SPREAD: rc = casim_enc_pod_add_spread_constraint(e, (int32_t)I[0], (int32_t)I[1], c.s(0), (int32_t)I[2]); break;
        case SPREAD_REQ: rc = casim_enc_spread_add_requirement(e, (int32_t)I[0], (int32_t)I[1], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[2]); break;
        case SPREAD_TAINTS: rc = casim_enc_spread_set_taints_policy(e, (int32_t)I[0], (int32_t)I[1], (int32_t)I[2]); break;
        case SPREAD_AFFINITY: rc = casim_enc_spread_set_affinity_policy(e, (int32_t)I[0], (int32_t)I[1], (int32_t)I[2]); break;
        case ADD_NAMESPACE: rc = casim_enc_add_namespace(e, c.s(0)); break;
        case NAMESPACE_LABEL: rc = casim_enc_namespace_add_label(e, c.s(0), c.s(1), c.s(2)); break;
        case TERM_NS_SELECTOR: rc = casim_enc_term_set_namespace_selector(e, (int32_t)I[0], (int32_t)I[1]); break;
        case TERM_NS_REQ: rc = casim_enc_term_add_namespace_requirement(e, (int32_t)I[0], (int32_t)I[1], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[2]); break;
        case POD_FP_REQ: rc = casim_enc_pod_set_fastpath_requests(e, (int32_t)I[0], c.D[0], c.D[1]); break;
        case POD_UNSUPPORTED: rc = casim_enc_pod_mark_unsupported(e, (int32_t)I[0], c.s(0)); break;
        case POD_SPEC_EXTRA: rc = casim_enc_pod_set_spec_extra(e, (int32_t)I[0], c.s(0)); break;
        case ADD_PEG: rc = casim_enc_add_peg(e, (int32_t)I[0], (int32_t)I[1]); break;
        case ADD_RESOURCE_PEGS: rc = casim_enc_add_resource_pegs(e, c.s(0), (int32_t)I[0], c.A64[0].data(), c.A32[0].data(), nullptr); break;
        case ADD_EXISTING_POD: rc = casim_enc_add_existing_pod(e, (int32_t)I[0], c.SA[0].data(), c.SA[1].data(), (int32_t)I[1]); break;
        case FINALIZE: rc = casim_enc_finalize(e); break;
        default: rc = -1;
        }
        if (rc < 0) { fprintf(stderr, "casim_native: encoder call (op %d) failed with %d: %s\n", c.op, rc, casim_last_error()); return rc; }
    }
    return 0;
}

// cursor over a directive's integer list: scalar fields, then (count, items...) arrays
struct Cur {
    const std::vector<int64_t>& v; size_t k = 0;
    explicit Cur(const std::vector<int64_t>& vv) : v(vv) {}
    int64_t one() { return k < v.size() ? v[k++] : 0; }
    template <class T> std::vector<T> arr() { const size_t n = (size_t)one(); std::vector<T> o; o.reserve(n); for (size_t i = 0; i < n && k < v.size(); ++i) o.push_back((T)v[k++]); return o; }
};

// FNV-1a over every column of the finalized tables, in header order: the parity handle of the native replay
uint64_t fnv(uint64_t h, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; } return h; }
uint64_t tables_hash(const casim_pegs& p, const casim_groups& g) {
    uint64_t h = 0xcbf29ce484222325ull;
    const size_t G = (size_t)p.n_pegs, NG = (size_t)g.n_groups, R = (size_t)p.n_res;
    auto col = [&](const void* ptr, size_t bytes) { if (ptr && bytes) h = fnv(h, ptr, bytes); };
    col(p.req, G * R * 8); col(p.count, G * 4); col(p.flags, G * 4); col(p.tol_mask, G * p.w_taint * 8); col(p.sel_mask, G * p.w_label * 8);
    col(p.excl_block, G * p.w_excl * 8); col(p.excl_mark, G * p.w_excl * 8); col(p.zone_block, G * p.w_zone * 8); col(p.zone_mark, G * p.w_zone * 8);
    col(g.alloc, NG * R * 8); col(g.init_req, NG * R * 8); col(g.allowed_pods, NG * 4); col(g.init_pods, NG * 4); col(g.flags, NG * 4);
    col(g.taint_mask, NG * p.w_taint * 8); col(g.label_mask, NG * p.w_label * 8); col(g.init_excl, NG * p.w_excl * 8);
    col(g.init_zone, NG * p.w_zone * 8); col(g.zone_valid, NG * p.w_zone * 8); col(g.max_nodes, NG * 4); col(g.existing_nodes, NG * 4); col(g.last_index, NG * 4);
    if (g.peg_offsets) { col(g.peg_offsets, (NG + 1) * 4); col(g.peg_index, (size_t)g.peg_offsets[NG] * 4); }
    return h;
}

void dump_i32(FILE* f, const char* tag, const std::vector<int32_t>& a) {
    const int32_t n = (int32_t)a.size(); char name[16]; memset(name, 0, sizeof name); strncpy(name, tag, 15);
    fwrite(name, 1, 16, f); fwrite(&n, 4, 1, f); if (n) fwrite(a.data(), 4, (size_t)n, f);
}
template <class T> double median(std::vector<T> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return (double)v[v.size() / 2]; }

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: casim_native TRACE [--device N] [--dump FILE] [--repeat K]\n"); return 2; }
    int device = 0, repeat = 3; const char* dump = nullptr;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
    }
    std::vector<Call> calls; Directive dir; std::string err;
    const auto tp = clk::now();
    if (!parse_trace(argv[1], calls, dir, err)) { fprintf(stderr, "casim_native: %s\n", err.c_str()); return 2; }
    const double parse_ms = ms_since(tp);

    // ---- encode (repeat times; the last encoder is kept) -----------------------------------------------
    casim_encoder* enc = nullptr;
    std::vector<double> enc_ms, fin_ms;
    for (int r = 0; r < repeat; ++r) {
        if (enc) { casim_enc_destroy(enc); enc = nullptr; }
        // all calls but the trailing finalize are "walk the objects"; finalize builds dictionaries + tables
        size_t nfin = calls.size();
        while (nfin > 0 && calls[nfin - 1].op == FINALIZE) --nfin;
        const auto t1 = clk::now();
        if (replay(calls, nfin, enc) < 0 || !enc) return 1;
        enc_ms.push_back(ms_since(t1));
        const auto t2 = clk::now();
        if (casim_enc_finalize(enc) < 0) { fprintf(stderr, "casim_native: finalize failed: %s\n", casim_last_error()); return 1; }
        fin_ms.push_back(ms_since(t2));
    }
    casim_pegs pegs; casim_groups groups;
    if (casim_enc_tables(enc, &pegs, &groups) < 0) { fprintf(stderr, "casim_native: casim_enc_tables failed\n"); return 1; }
    casim_domain_rules rules; memset(&rules, 0, sizeof rules);
    (void)casim_enc_domain_rules(enc, &rules);
    int32_t dict[4] = {0, 0, 0, 0}; (void)casim_enc_dict_sizes(enc, dict);

    printf("{\"trace\": \"%s\", \"enc_calls\": %zu, \"parse_ms\": %.3f, \"encode_calls_ms\": %.3f, \"finalize_ms\": %.3f, \"encode_ms\": %.3f, "
           "\"pegs\": %d, \"groups\": %d, \"dict\": [%d, %d, %d, %d], \"domain_rules\": %d, \"tables_fnv\": \"%016llx\"",
           argv[1], calls.size(), parse_ms, median(enc_ms), median(fin_ms), median(enc_ms) + median(fin_ms), pegs.n_pegs, groups.n_groups,
           dict[0], dict[1], dict[2], dict[3], rules.n_rules, (unsigned long long)tables_hash(pegs, groups));

    FILE* df = dump ? fopen(dump, "wb") : nullptr;
    int exit_code = 0;
    if (!dir.name.empty()) {
        casim_ctx* ctx = casim_ctx_create(device, nullptr);
        if (!ctx) { printf(", \"engine_error\": \"%s\"}\n", casim_last_error()); casim_enc_destroy(enc); if (df) fclose(df); return 3; }
        Cur c(dir.v);
        if (dir.name == "estimate") {
            const int fastpath = (int)c.one(); const int iters = (int)c.one();
            std::vector<int32_t> kinds = c.arr<int32_t>();
            // one simulation over every group, schedulable subsets as the trace says (explicit lists or device-side)
            casim_groups g = groups; const int32_t so[2] = {0, groups.n_groups}; g.n_sims = 1; g.sim_offsets = so;
            const int NG = groups.n_groups;
            const int64_t nnz_cap = groups.peg_offsets ? groups.peg_offsets[NG] : (int64_t)pegs.n_pegs * NG;
            std::vector<int32_t> node_count(NG + 1), pods(NG + 1), added(NG + 1), lim(NG + 1), li(NG + 1), st(NG + 1), order((size_t)nnz_cap + 1), placed((size_t)nnz_cap + 1);
            std::vector<int64_t> cpu(NG + 1), mem(NG + 1);
            casim_results res = {node_count.data(), pods.data(), added.data(), lim.data(), li.data(), st.data(), cpu.data(), mem.data(), order.data(), placed.data()};
            casim_options opt; memset(&opt, 0, sizeof opt); opt.fastpath = fastpath;
            int32_t best = -1, nbest = 0; int64_t packed = 0;
            casim_option_query q; memset(&q, 0, sizeof q);
            q.kinds = kinds.data(); q.n_kinds = (int32_t)kinds.size(); q.per_sim = 1; q.best_out = &best; q.n_best_out = &nbest; q.packed_out = &packed;
            double ph[8], acc[8] = {0}; std::vector<double> walls;
            int32_t rc = casim_estimate_batch_timed(ctx, &pegs, &g, &opt, &res, kinds.empty() ? nullptr : &q, ph);   // first call: code objects, LDS opt-ins
            for (int i = 0; i < iters && rc == 0; ++i) {
                rc = casim_estimate_batch_timed(ctx, &pegs, &g, &opt, &res, kinds.empty() ? nullptr : &q, ph);
                for (int k = 0; k < 8; ++k) acc[k] += ph[k];
            }
            for (int i = 0; i < iters && rc == 0; ++i) {   // the plain (undrained) call
                const auto t0 = clk::now();
                rc = casim_estimate_batch(ctx, &pegs, &g, &opt, &res);
                walls.push_back(ms_since(t0));
            }
            if (rc != 0) { printf(", \"engine_error\": \"%d %s\"", rc, casim_last_error()); exit_code = 4; }
            const double n = iters > 0 ? iters : 1;
            printf(", \"entry\": \"casim_estimate_batch\", \"iters\": %d, \"upload_ms\": %.4f, \"feasibility_csr_ms\": %.4f, \"order_ms\": %.4f, \"pack_ms\": %.4f, "
                   "\"expander_ms\": %.4f, \"fetch_ms\": %.4f, \"timed_wall_ms\": %.4f, \"wall_ms\": %.4f, \"best_group\": %d, \"n_best\": %d",
                   iters, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, acc[6] / n, median(walls), best, nbest);
            if (df) {
                // CSR offsets actually used
                std::vector<int32_t> off(NG + 1, 0);
                if (groups.peg_offsets) off.assign(groups.peg_offsets, groups.peg_offsets + NG + 1);
                else { casim_problem* p = casim_problem_create(ctx, &pegs, &g, &opt); if (p) { casim_problem_run(p); int32_t nnz = 0; casim_problem_csr(p, &nnz, off.data()); casim_problem_destroy(p); } }
                node_count.resize(NG); pods.resize(NG); added.resize(NG); lim.resize(NG); li.resize(NG); st.resize(NG);
                order.resize((size_t)off[NG]); placed.resize((size_t)off[NG]);
                dump_i32(df, "offsets", off); dump_i32(df, "node_count", node_count); dump_i32(df, "pods", pods); dump_i32(df, "nodes_added", added);
                dump_i32(df, "limiter", lim); dump_i32(df, "last_index", li); dump_i32(df, "status", st); dump_i32(df, "order", order); dump_i32(df, "placed", placed);
                dump_i32(df, "best", std::vector<int32_t>{best, nbest});
            }
        } else if (dir.name == "try_schedule") {
            const int iters = (int)c.one();
            casim_pod_sequence seq; memset(&seq, 0, sizeof seq);
            seq.break_on_failure = (int32_t)c.one(); seq.last_index = (int32_t)c.one(); const int use_rules = (int)c.one();
            std::vector<int32_t> pc = c.arr<int32_t>(), hint = c.arr<int32_t>(); std::vector<uint8_t> acc = c.arr<uint8_t>(); std::vector<int32_t> sim = c.arr<int32_t>();
            seq.n_pods = (int32_t)pc.size(); seq.pod_class = pc.data(); seq.hint_node = hint.empty() ? nullptr : hint.data();
            seq.node_acceptable = acc.empty() ? nullptr : acc.data(); seq.similar_key = sim.empty() ? nullptr : sim.data();
            seq.rules = (use_rules && rules.n_rules > 0) ? &rules : nullptr;
            std::vector<int32_t> node_out(pc.size() + 1, -1); int32_t li = 0, ns = 0; std::vector<double> walls;
            int32_t rc = casim_try_schedule_pods(ctx, &pegs, &groups, &seq, node_out.data(), &li, &ns);
            for (int i = 0; i < iters && rc >= 0; ++i) { const auto t0 = clk::now(); rc = casim_try_schedule_pods(ctx, &pegs, &groups, &seq, node_out.data(), &li, &ns); walls.push_back(ms_since(t0)); }
            float kms = 0; if (rc == 0 && iters > 0) (void)casim_time_try_schedule_pods(ctx, &pegs, &groups, &seq, iters, &kms);
            if (rc < 0) { printf(", \"engine_error\": \"%d %s\"", rc, casim_last_error()); exit_code = 4; }
            printf(", \"entry\": \"casim_try_schedule_pods\", \"status\": %d, \"pods\": %zu, \"scheduled\": %d, \"last_index\": %d, \"wall_ms\": %.4f, \"kernels_ms\": %.4f",
                   rc, pc.size(), ns, li, median(walls), iters > 0 ? kms / iters : 0.0);
            if (df) { node_out.resize(pc.size()); dump_i32(df, "node_out", node_out); dump_i32(df, "tail", std::vector<int32_t>{rc, li, ns}); }
        } else if (dir.name == "removals") {
            const int iters = (int)c.one();
            casim_removal_candidates cd; memset(&cd, 0, sizeof cd);
            cd.persist = (int32_t)c.one(); cd.max_removable = (int32_t)c.one(); cd.last_index = (int32_t)c.one(); const int use_rules = (int)c.one();
            std::vector<int32_t> cn = c.arr<int32_t>(), po = c.arr<int32_t>(), pc = c.arr<int32_t>(), hint = c.arr<int32_t>(); std::vector<uint8_t> dest = c.arr<uint8_t>();
            cd.n_candidates = (int32_t)cn.size(); cd.cand_node = cn.data(); cd.pod_offsets = po.data(); cd.pod_class = pc.data();
            cd.hint_node = hint.empty() ? nullptr : hint.data(); cd.destination = dest.empty() ? nullptr : dest.data();
            cd.ext_capacity = (int32_t)(2 * pc.size() + 64); cd.rules = (use_rules && rules.n_rules > 0) ? &rules : nullptr;
            std::vector<uint8_t> removable(cn.size() + 1, 2); std::vector<int32_t> node_out(pc.size() + 1, -1), ec((size_t)cd.ext_capacity + 1), ep((size_t)cd.ext_capacity + 1), en((size_t)cd.ext_capacity + 1);
            casim_removal_results rr; memset(&rr, 0, sizeof rr);
            rr.removable = removable.data(); rr.node_out = node_out.data(); rr.ext_candidate = ec.data(); rr.ext_pod = ep.data(); rr.ext_node = en.data();
            std::vector<double> walls;
            int32_t rc = casim_simulate_node_removals(ctx, &pegs, &groups, &cd, &rr);
            for (int i = 0; i < iters && rc >= 0; ++i) { const auto t0 = clk::now(); rc = casim_simulate_node_removals(ctx, &pegs, &groups, &cd, &rr); walls.push_back(ms_since(t0)); }
            float kms = 0; if (rc == 0 && iters > 0) (void)casim_time_node_removals(ctx, &pegs, &groups, &cd, iters, &kms);
            int nrem = 0; for (size_t i = 0; i < cn.size(); ++i) nrem += removable[i] == 1;
            if (rc < 0) { printf(", \"engine_error\": \"%d %s\"", rc, casim_last_error()); exit_code = 4; }
            printf(", \"entry\": \"casim_simulate_node_removals\", \"status\": %d, \"candidates\": %zu, \"pods\": %zu, \"removable\": %d, \"n_ext\": %d, \"wall_ms\": %.4f, \"kernels_ms\": %.4f",
                   rc, cn.size(), pc.size(), nrem, rr.n_ext, median(walls), iters > 0 ? kms / iters : 0.0);
            if (df) {
                std::vector<int32_t> rem(cn.size()); for (size_t i = 0; i < cn.size(); ++i) rem[i] = removable[i];
                node_out.resize(pc.size()); ec.resize((size_t)rr.n_ext); ep.resize((size_t)rr.n_ext); en.resize((size_t)rr.n_ext);
                dump_i32(df, "removable", rem); dump_i32(df, "node_out", node_out); dump_i32(df, "ext_candidate", ec); dump_i32(df, "ext_pod", ep); dump_i32(df, "ext_node", en);
                dump_i32(df, "tail", std::vector<int32_t>{rc, rr.last_index, rr.n_processed});
            }
        } else {
            printf(", \"engine_error\": \"unknown directive %s\"", dir.name.c_str()); exit_code = 2;
        }
        casim_ctx_destroy(ctx);
    }
    printf("}\n");
    if (df) fclose(df);
    casim_enc_destroy(enc);
    return exit_code;
}
