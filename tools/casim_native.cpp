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
// A = int64 array, a = int32 array, D = double array (arrays: count followed by the items)
struct Sig { int op; const char* args; };
enum Op {
    CREATE, ADD_GROUP, GROUP_LABEL, GROUP_TAINT, GROUP_FP_CAP, GROUP_LIMITS, GROUP_PRELOADED, GROUP_SET_PEGS, ADD_POD_SPEC, POD_LABEL,
    POD_TOLERATION, POD_NODE_SELECTOR, POD_NODE_AFF_REQ, POD_NODE_AFF_TERM, NODE_TERM_REQ, POD_HOST_PORT, POD_AA_TERM, TERM_REQ,
    POD_AFF_TERM, AFF_TERM_REQ, POD_SPREAD, SPREAD_REQ, SPREAD_TAINTS, SPREAD_AFFINITY, ADD_NAMESPACE, NAMESPACE_LABEL, TERM_NS_SELECTOR, TERM_NS_REQ, AFF_TERM_NS_SELECTOR, AFF_TERM_NS_REQ, POD_FP_REQ,
    POD_UNSUPPORTED, POD_SPEC_EXTRA, ADD_PEG, ADD_RESOURCE_PEGS, ADD_EXISTING_POD, FINALIZE, ENC_LANE, POD_SET_REQUEST, GROUP_SET_ALLOCATABLE, LANE_COUNT, LANE_NAME, ADD_PODS, ADD_RUNNING_PODS
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
    {"casim_enc_aff_term_set_namespace_selector", {AFF_TERM_NS_SELECTOR, "ii"}},
    {"casim_enc_aff_term_add_namespace_requirement", {AFF_TERM_NS_REQ, "iissSi"}},
    {"casim_enc_pod_set_fastpath_requests", {POD_FP_REQ, "idd"}},
    {"casim_enc_pod_mark_unsupported", {POD_UNSUPPORTED, "is"}},
    {"casim_enc_pod_set_spec_extra", {POD_SPEC_EXTRA, "is"}},
    {"casim_enc_add_peg", {ADD_PEG, "ii"}},
    {"casim_enc_add_resource_pegs", {ADD_RESOURCE_PEGS, "siAaa"}},
    {"casim_enc_add_existing_pod", {ADD_EXISTING_POD, "iSSi"}},
    {"casim_enc_finalize", {FINALIZE, ""}},
    {"casim_enc_lane", {ENC_LANE, "s"}},
    {"casim_enc_pod_set_request", {POD_SET_REQUEST, "isl"}},
    {"casim_enc_group_set_allocatable", {GROUP_SET_ALLOCATABLE, "isl"}},
    {"casim_enc_lane_count", {LANE_COUNT, ""}},
    {"casim_enc_lane_name", {LANE_NAME, "i"}},
    // casim_pod_columns (ABI 11): n_pods, strings, ns, req, fastpath_req, peg_count, label_off / key / val, tol_off / key / op / value / effect, sel_off / key / val
    {"casim_enc_add_pods", {ADD_PODS, "iSaADaaaaaaaaaaaa"}},
    // n_pods, group, ns, req, label_off, label_key, label_val, strings
    {"casim_enc_add_running_pods", {ADD_RUNNING_PODS, "iaaAaaaS"}},
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
    std::vector<std::vector<double>> AD;
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
                else if (*a == 'D') { c.AD.emplace_back(); for (size_t j = 0; j < n; ++j) c.AD.back().push_back(strtod(t[k++].c_str(), nullptr)); }
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
        case AFF_TERM_NS_SELECTOR: rc = casim_enc_aff_term_set_namespace_selector(e, (int32_t)I[0], (int32_t)I[1]); break;
        case AFF_TERM_NS_REQ: rc = casim_enc_aff_term_add_namespace_requirement(e, (int32_t)I[0], (int32_t)I[1], c.s(0), c.s(1), c.SA[0].data(), (int32_t)I[2]); break;
        case POD_FP_REQ: rc = casim_enc_pod_set_fastpath_requests(e, (int32_t)I[0], c.D[0], c.D[1]); break;
        case POD_UNSUPPORTED: rc = casim_enc_pod_mark_unsupported(e, (int32_t)I[0], c.s(0)); break;
        case POD_SPEC_EXTRA: rc = casim_enc_pod_set_spec_extra(e, (int32_t)I[0], c.s(0)); break;
        case ADD_PEG: rc = casim_enc_add_peg(e, (int32_t)I[0], (int32_t)I[1]); break;
        case ADD_RESOURCE_PEGS: rc = casim_enc_add_resource_pegs(e, c.s(0), (int32_t)I[0], c.A64[0].data(), c.A32[0].data(), nullptr); break;
        case ADD_EXISTING_POD: rc = casim_enc_add_existing_pod(e, (int32_t)I[0], c.SA[0].data(), c.SA[1].data(), (int32_t)I[1]); break;
        case FINALIZE: rc = casim_enc_finalize(e); break;
        // resources by name (ABI 9): CASIM_ENC_DELEGATED (1) is an answer, not an error; a lane lookup that finds none is the caller's to handle
        case ENC_LANE: (void)casim_enc_lane(e, c.s(0)); rc = 0; break;
        case POD_SET_REQUEST: rc = casim_enc_pod_set_request(e, (int32_t)I[0], c.s(0), I[1]); break;
        case GROUP_SET_ALLOCATABLE: rc = casim_enc_group_set_allocatable(e, (int32_t)I[0], c.s(0), I[1]); break;
        case LANE_COUNT: (void)casim_enc_lane_count(e); rc = 0; break;
        case LANE_NAME: (void)casim_enc_lane_name(e, (int32_t)I[0]); rc = 0; break;
        case ADD_RUNNING_PODS: {
            auto col = [&](size_t k) -> const int32_t* { return c.A32[k].empty() ? nullptr : c.A32[k].data(); };
            const int32_t first = casim_enc_add_running_pods(e, (int32_t)I[0], col(0), col(1), c.A64[0].data(), col(2), col(3), col(4), c.SA[0].data(), (int32_t)c.SA_store[0].size());
            rc = first < 0 ? first : 0; break;
        }
        case ADD_PODS: {
            casim_pod_columns pc; memset(&pc, 0, sizeof pc);
            auto col = [&](size_t k) -> const int32_t* { return c.A32[k].empty() ? nullptr : c.A32[k].data(); };
            pc.n_pods = (int32_t)I[0]; pc.n_strings = (int32_t)c.SA_store[0].size(); pc.strings = c.SA[0].data();
            pc.ns = col(0); pc.req = c.A64[0].data(); pc.fastpath_req = c.AD[0].empty() ? nullptr : c.AD[0].data(); pc.peg_count = col(1);
            pc.label_off = col(2); pc.label_key = col(3); pc.label_val = col(4);
            pc.tol_off = col(5); pc.tol_key = col(6); pc.tol_op = col(7); pc.tol_value = col(8); pc.tol_effect = col(9);
            pc.sel_off = col(10); pc.sel_key = col(11); pc.sel_val = col(12);
            rc = casim_enc_add_pods(e, &pc, nullptr); break;
        }
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
    col(p.zone_polarity, (size_t)p.w_zone * 8);
    col(p.excl_polarity, (size_t)p.w_excl * 8);
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


// ---- the estimator shim's call sequence, replayed in plain C++ ----------------------------------------------------
// What integration/go/gpubinpacking/{prefetch,estimator}.go do over cgo, call for call: the NodeGroupListProcessor wrapper fills ONE
// batch (casim_prefetch_fill); then, group by group in the orchestrator's order, Estimate() = casim_prefetch_lookup with the PEG list
// the orchestrator passes (SchedulablePodGroups' result), and on a miss the per-call path (one casim_estimate_batch with one group
// record and that list).  Checked here: every hit equals the per-call answer field by field, and each miss path reports its reason.
struct OneGroup { int32_t v[6]; int64_t sums[2]; std::vector<int32_t> order, placed; };
casim_groups group_view(const casim_pegs& p, const casim_groups& g, int i) {
    casim_groups w = g;
    const int64_t R = p.n_res;
    auto at = [&](auto* ptr, int64_t off) { return ptr ? ptr + off : ptr; };
    w.n_groups = 1;
    w.alloc = at(g.alloc, i * R); w.init_req = at(g.init_req, i * R); w.allowed_pods = at(g.allowed_pods, i); w.init_pods = at(g.init_pods, i); w.flags = at(g.flags, i);
    w.taint_mask = at(g.taint_mask, (int64_t)i * p.w_taint); w.label_mask = at(g.label_mask, (int64_t)i * p.w_label); w.init_excl = at(g.init_excl, (int64_t)i * p.w_excl);
    w.init_zone = at(g.init_zone, (int64_t)i * p.w_zone); w.zone_valid = at(g.zone_valid, (int64_t)i * p.w_zone);
    w.max_nodes = at(g.max_nodes, i); w.existing_nodes = at(g.existing_nodes, i); w.last_index = at(g.last_index, i);
    w.cap_cpu = at(g.cap_cpu, i); w.cap_mem = at(g.cap_mem, i); w.waste_cpu = at(g.waste_cpu, i); w.waste_mem = at(g.waste_mem, i);
    w.peg_lo = w.peg_hi = w.global_id = nullptr; w.n_sims = 0; w.sim_offsets = nullptr;
    return w;
}
// per-call mode: ONE Estimate() = one casim_estimate_batch with one group record and the caller's PEG list; order as positions in that list
// (enc != NULL: the group's row through casim_enc_group_rows, the call the Go shim makes — estimateOnLoopTables in
// integration/go/gpubinpacking/prefetch.go; NULL: pointer offsets into the caller's own tables)
int32_t per_call(casim_ctx* ctx, const casim_pegs& p, const casim_groups& g, int i, const std::vector<int32_t>& list, int32_t max_nodes, const casim_options& opt, OneGroup& out,
                 casim_encoder* enc = nullptr, const int32_t* last_index = nullptr /* the runner's lastIndex for this call, or the table's entry */) {
    casim_groups w = group_view(p, g, i);
    if (enc) {
        const int32_t row = i;
        memset(&w, 0, sizeof w);
        if (casim_enc_group_rows(enc, &row, 1, &w) != CASIM_OK) return CASIM_ERR_INVALID;
    }
    const int32_t off[2] = {0, (int32_t)list.size()};
    const int32_t mn[1] = {max_nodes};
    w.peg_offsets = off; w.peg_index = list.data(); w.max_nodes = mn;
    if (last_index) w.last_index = last_index;
    const size_t n = list.size();
    out.order.assign(n + 1, 0); out.placed.assign(n + 1, 0);
    casim_results r; memset(&r, 0, sizeof r);
    r.node_count = &out.v[0]; r.pods_scheduled = &out.v[1]; r.nodes_added = &out.v[2]; r.limiter_nodes = &out.v[3]; r.last_index_out = &out.v[4]; r.status = &out.v[5];
    r.req_cpu_sum = &out.sums[0]; r.req_mem_sum = &out.sums[1]; r.order = out.order.data(); r.placed = out.placed.data();
    const int32_t rc = casim_estimate_batch(ctx, &p, &w, &opt, &r);
    out.order.resize(n); out.placed.resize(n);
    std::vector<int32_t> pos((size_t)p.n_pegs, -1);
    for (size_t k = 0; k < n; ++k) pos[(size_t)list[k]] = (int32_t)k;
    for (size_t k = 0; k < n; ++k) out.order[k] = pos[(size_t)out.order[k]];
    return rc;
}
// returns the number of failed checks; prints one JSON member
int shim_replay(casim_ctx* ctx, const casim_pegs& pegs, const casim_groups& groups, int fastpath, casim_encoder* enc) {
    const int NG = groups.n_groups, G = pegs.n_pegs;
    casim_options opt; memset(&opt, 0, sizeof opt); opt.fastpath = fastpath;
    int bad = 0;
    auto fail = [&](const char* what, int i) { if (bad < 8) fprintf(stderr, "casim_native --shim: %s (group %d)\n", what, i); ++bad; };
    // the orchestrator's SchedulablePodGroups lists (here: the device's own subsets, ascending PEG id = the order of the caller's list)
    std::vector<std::vector<int32_t>> lists((size_t)NG);
    {
        casim_problem* pr = casim_problem_create(ctx, &pegs, &groups, &opt);
        if (!pr) { printf(", \"shim\": {\"error\": \"%s\"}", casim_last_error()); return 1; }
        casim_problem_run(pr);
        int32_t nnz = 0; std::vector<int32_t> off((size_t)NG + 1);
        casim_problem_csr(pr, &nnz, off.data());
        std::vector<int32_t> a((size_t)NG * 6 + 1), order((size_t)nnz + 1), placed((size_t)nnz + 1); std::vector<int64_t> sums((size_t)NG * 2 + 1);
        casim_results r; memset(&r, 0, sizeof r);
        r.node_count = a.data(); r.pods_scheduled = a.data() + NG; r.nodes_added = a.data() + 2 * NG; r.limiter_nodes = a.data() + 3 * NG; r.last_index_out = a.data() + 4 * NG;
        r.status = a.data() + 5 * NG; r.req_cpu_sum = sums.data(); r.req_mem_sum = sums.data() + NG; r.order = order.data(); r.placed = placed.data();
        casim_problem_fetch(pr, &r);
        casim_problem_destroy(pr);
        for (int i = 0; i < NG; ++i) {
            lists[(size_t)i].assign(order.begin() + off[(size_t)i], order.begin() + off[(size_t)i + 1]);
            if (groups.peg_offsets) lists[(size_t)i].assign(groups.peg_index + groups.peg_offsets[i], groups.peg_index + groups.peg_offsets[i + 1]);
            else std::sort(lists[(size_t)i].begin(), lists[(size_t)i].end());
        }
    }
    // keys: opaque 64-bit identities (the Go side hashes the exemplar pod pointer / the group id + template generation)
    std::vector<uint64_t> gkey((size_t)NG), pkey((size_t)G);
    for (int i = 0; i < NG; ++i) gkey[(size_t)i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    for (int g = 0; g < G; ++g) pkey[(size_t)g] = 0xC2B2AE3D27D4EB4Full * (uint64_t)(g + 1);
    auto keys_of = [&](const std::vector<int32_t>& l) { std::vector<uint64_t> k(l.size()); for (size_t j = 0; j < l.size(); ++j) k[j] = pkey[(size_t)l[j]]; return k; };
    casim_prefetch* pf = casim_prefetch_create(ctx);
    const auto tf = clk::now();
    int32_t rc = casim_prefetch_fill(pf, &pegs, &groups, &opt, gkey.data(), pkey.data());   // ---- NodeGroupListProcessor.Process
    const double fill_ms = ms_since(tf);
    if (rc != 0) { printf(", \"shim\": {\"error\": \"fill %d %s\"}", rc, casim_prefetch_error(pf)); casim_prefetch_destroy(pf); return 1; }
    int hits = 0, equal = 0, rows_calls = 0; double lookup_ms = 0, percall_ms = 0, rows_ms = 0;
    for (int i = 0; i < NG; ++i) {                                                         // ---- ComputeExpansionOption, group by group
        const std::vector<int32_t>& l = lists[(size_t)i];
        std::vector<uint64_t> k = keys_of(l);
        std::vector<int32_t> order(l.size() + 1), placed(l.size() + 1);
        casim_prefetch_result r;
        const auto t0 = clk::now();
        rc = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, order.data(), placed.data());
        lookup_ms += ms_since(t0);
        if (rc != CASIM_OK) { fail("expected a hit", i); continue; }
        ++hits;
        OneGroup pc;
        const auto t1 = clk::now();
        if (per_call(ctx, pegs, groups, i, l, groups.max_nodes[i], opt, pc) != 0) { fail("per-call estimate failed", i); continue; }
        percall_ms += ms_since(t1);
        {   // the same Estimate() the way the Go shim serves a miss: row through casim_enc_group_rows, no encoding (3 calls, the median)
            OneGroup pr; std::vector<double> t;
            for (int rep = 0; rep < 3; ++rep) {
                const auto t2 = clk::now();
                if (per_call(ctx, pegs, groups, i, l, groups.max_nodes[i], opt, pr, enc) != 0) { fail("per-call through casim_enc_group_rows failed", i); break; }
                t.push_back(ms_since(t2));
            }
            if (!t.empty()) { rows_ms += median(t); ++rows_calls; }
            if (pr.order != pc.order || pr.placed != pc.placed || memcmp(pr.v, pc.v, sizeof pc.v) != 0) fail("per-call through casim_enc_group_rows differs", i);
        }
        order.resize(l.size()); placed.resize(l.size());
        const bool same = r.node_count == pc.v[0] && r.pods_scheduled == pc.v[1] && r.nodes_added == pc.v[2] && r.limiter_nodes == pc.v[3] && r.last_index_out == pc.v[4] &&
                          r.status == pc.v[5] && r.req_cpu_sum == pc.sums[0] && r.req_mem_sum == pc.sums[1] && (r.status != 0 || (order == pc.order && placed == pc.placed));
        if (same) ++equal; else fail("hit differs from the per-call answer", i);
    }
    // ---- the miss paths, on the first group with at least two schedulable PEGs
    int miss_checked = 0;
    for (int i = 0; i < NG && miss_checked == 0; ++i) {
        const std::vector<int32_t>& l = lists[(size_t)i];
        if (l.size() < 2) continue;
        miss_checked = 1;
        casim_prefetch_result r; OneGroup pc;
        std::vector<int32_t> sub(l.begin(), l.end() - 1);                                  // the orchestrator passed another subset
        std::vector<uint64_t> k = keys_of(sub);
        rc = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, nullptr, nullptr);
        if (rc != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_PEGS) fail("another PEG list must miss", i);
        if (per_call(ctx, pegs, groups, i, sub, groups.max_nodes[i], opt, pc) != 0) fail("per-call after a PEG-list miss", i);
        {   // the same set in another order (the orchestrator's list comes out of a Go map): a hit, positions follow the caller's list
            std::vector<int32_t> swapped(l); std::swap(swapped[0], swapped[l.size() - 1]);
            k = keys_of(swapped);
            std::vector<int32_t> o1(l.size()), o2(l.size());
            rc = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, o2.data(), nullptr);
            std::vector<uint64_t> k1 = keys_of(l);
            const int32_t rc1 = casim_prefetch_lookup(pf, gkey[(size_t)i], k1.data(), (int32_t)k1.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, o1.data(), nullptr);
            bool ok = rc == CASIM_OK && rc1 == CASIM_OK;
            for (size_t j = 0; j < l.size() && ok; ++j) ok = swapped[(size_t)o2[j]] == l[(size_t)o1[j]];
            if (!ok) fail("a reordered PEG list must hit with positions in the caller's order", i);
        }
        std::vector<uint64_t> dup = keys_of(l); dup[0] = dup[1];                             // not a set: one PEG twice
        rc = casim_prefetch_lookup(pf, gkey[(size_t)i], dup.data(), (int32_t)dup.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, nullptr, nullptr);
        if (rc != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_PEGS) fail("a list with a PEG twice must miss", i);
        k = keys_of(l);
        const int32_t other = groups.max_nodes[i] > 1 ? groups.max_nodes[i] - 1 : groups.max_nodes[i] + 1;   // the limiter answered differently
        rc = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), other, groups.existing_nodes[i], groups.last_index[i], &r, nullptr, nullptr);
        if (rc != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_LIMITS) fail("another max_nodes must miss", i);
        if (per_call(ctx, pegs, groups, i, l, other, opt, pc) != 0) fail("per-call after a limits miss", i);
        rc = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i] + 1, &r, nullptr, nullptr);
        if (rc != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_LAST_INDEX) fail("another lastIndex must miss (on lastIndex alone)", i);
        rc = casim_prefetch_lookup(pf, 0xDEADBEEFull, k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, nullptr, nullptr);
        if (rc != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_GROUP) fail("an unknown group must miss", i);
        casim_prefetch_clear(pf);                                                          // the next loop iteration
        rc = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], groups.last_index[i], &r, nullptr, nullptr);
        if (rc != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_GROUP) fail("a cleared cache must miss", i);
    }
    // ---- the CHAINED loop (round 5: prefetch.go fills with casim_options.chain_last_index; estimator.go looks up with the runner's lastIndex as of
    // the call and moves it on at every hit): in the batch's order every call hits and equals the per-call Estimate from the runner's value
    int chain_hits = 0, chain_equal = 0;
    {
        casim_options copt = opt; copt.chain_last_index = 1;
        rc = casim_prefetch_fill(pf, &pegs, &groups, &copt, gkey.data(), pkey.data());
        if (rc != 0) fail("chained fill", 0);
        int32_t runner = NG > 0 && groups.last_index ? groups.last_index[0] : 0;
        for (int i = 0; i < NG && rc == 0; ++i) {
            const std::vector<int32_t>& l = lists[(size_t)i];
            std::vector<uint64_t> k = keys_of(l);
            std::vector<int32_t> order(l.size() + 1), placed(l.size() + 1);
            casim_prefetch_result r;
            const int32_t lr = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], runner, &r, order.data(), placed.data());
            if (lr != CASIM_OK) { fail("chained batch: expected a hit with the runner's lastIndex", i); break; }
            ++chain_hits;
            OneGroup pc;
            if (per_call(ctx, pegs, groups, i, l, groups.max_nodes[i], opt, pc, enc, &runner) != 0) { fail("per-call estimate (chained)", i); break; }
            order.resize(l.size()); placed.resize(l.size());
            const bool same = r.node_count == pc.v[0] && r.pods_scheduled == pc.v[1] && r.nodes_added == pc.v[2] && r.limiter_nodes == pc.v[3] && r.last_index_out == pc.v[4] &&
                              r.status == pc.v[5] && (r.status != 0 || (order == pc.order && placed == pc.placed));
            if (same) ++chain_equal; else fail("chained hit differs from the per-call answer with the runner's lastIndex", i);
            if (r.status == 0) runner = r.last_index_out;     // (a delegated group hands its input on: the reference path would move the real runner)
        }
    }
    // ---- the chain LEFT and re-chained (round 6: prefetch.go rechain).  Group `mid - 1` runs on the reference path and moves the runner somewhere the
    // batch did not expect: the lookup of group `mid` misses on lastIndex ALONE; the rest of the loop — rows mid .. NG - 1 of the loop's group table
    // through casim_enc_group_rows, the first one's last_index overridden — is filled again as one chained batch from the runner's value, and every
    // call from `mid` on hits and equals the per-call Estimate from the runner's lastIndex
    int rechain_hits = 0, rechain_equal = 0, rechain_groups = 0;
    if (enc && NG >= 3) {
        casim_options copt = opt; copt.chain_last_index = 1;
        rc = casim_prefetch_fill(pf, &pegs, &groups, &copt, gkey.data(), pkey.data());
        if (rc != 0) fail("chained fill before the re-chain", 0);
        const int mid = NG / 2;
        casim_prefetch_result r;
        // the calls before `mid` arrive in order and hit; then the runner ends one node further than the batch assumed
        int32_t runner = groups.last_index ? groups.last_index[0] : 0;
        for (int i = 0; i < mid; ++i) {
            std::vector<uint64_t> k = keys_of(lists[(size_t)i]);
            if (casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], runner, &r, nullptr, nullptr) != CASIM_OK) { fail("chained batch before the re-chain: expected a hit", i); break; }
            if (r.status == 0) runner = r.last_index_out;
        }
        runner += 1;
        {
            std::vector<uint64_t> k = keys_of(lists[(size_t)mid]);
            const int32_t lr = casim_prefetch_lookup(pf, gkey[(size_t)mid], k.data(), (int32_t)k.size(), groups.max_nodes[mid], groups.existing_nodes[mid], runner, &r, nullptr, nullptr);
            if (lr != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_LAST_INDEX) fail("a left chain must miss on lastIndex alone", mid);
        }
        const int n = NG - mid;
        std::vector<int32_t> rows((size_t)n), li((size_t)n, runner);
        for (int i = 0; i < n; ++i) rows[(size_t)i] = mid + i;
        casim_groups rest; memset(&rest, 0, sizeof rest);
        if (casim_enc_group_rows(enc, rows.data(), n, &rest) != CASIM_OK) fail("casim_enc_group_rows of the rest of the loop", mid);
        else {
            rest.last_index = li.data();
            rc = casim_prefetch_fill(pf, &pegs, &rest, &copt, gkey.data() + mid, pkey.data());
            if (rc != 0) fail("re-chained fill", mid);
            for (int i = mid; i < NG && rc == 0; ++i) {
                const std::vector<int32_t>& l = lists[(size_t)i];
                std::vector<uint64_t> k = keys_of(l);
                std::vector<int32_t> order(l.size() + 1), placed(l.size() + 1);
                const int32_t lr = casim_prefetch_lookup(pf, gkey[(size_t)i], k.data(), (int32_t)k.size(), groups.max_nodes[i], groups.existing_nodes[i], runner, &r, order.data(), placed.data());
                ++rechain_groups;
                if (lr != CASIM_OK) { fail("re-chained batch: expected a hit with the runner's lastIndex", i); break; }
                ++rechain_hits;
                OneGroup pc;
                if (per_call(ctx, pegs, groups, i, l, groups.max_nodes[i], opt, pc, enc, &runner) != 0) { fail("per-call estimate (re-chained)", i); break; }
                order.resize(l.size()); placed.resize(l.size());
                const bool same = r.node_count == pc.v[0] && r.pods_scheduled == pc.v[1] && r.nodes_added == pc.v[2] && r.limiter_nodes == pc.v[3] && r.last_index_out == pc.v[4] &&
                                  r.status == pc.v[5] && (r.status != 0 || (order == pc.order && placed == pc.placed));
                if (same) ++rechain_equal; else fail("re-chained hit differs from the per-call answer with the runner's lastIndex", i);
                if (r.status == 0) runner = r.last_index_out;
            }
            // a group answered before the re-chain is gone from the cache
            std::vector<uint64_t> k0 = keys_of(lists[0]);
            const int32_t lr = casim_prefetch_lookup(pf, gkey[0], k0.data(), (int32_t)k0.size(), groups.max_nodes[0], groups.existing_nodes[0], groups.last_index ? groups.last_index[0] : 0, &r, nullptr, nullptr);
            if (lr != CASIM_PREFETCH_MISS || r.miss_reason != CASIM_PREFETCH_MISS_GROUP) fail("a group before the re-chained rest must miss on the group", 0);
        }
    }
    int64_t st[8]; casim_prefetch_stats(pf, st);
    casim_prefetch_destroy(pf);
    printf(", \"shim_chained\": {\"hits\": %d, \"hits_equal_to_per_call\": %d}", chain_hits, chain_equal);
    printf(", \"shim_rechained\": {\"groups\": %d, \"hits\": %d, \"hits_equal_to_per_call\": %d}", rechain_groups, rechain_hits, rechain_equal);
    printf(", \"shim\": {\"groups\": %d, \"hits\": %d, \"hits_equal_to_per_call\": %d, \"miss_paths_checked\": %d, \"failed_checks\": %d, \"fill_ms\": %.4f, "
           "\"lookups_ms\": %.4f, \"per_call_ms_total\": %.4f, \"per_call_on_loop_tables_ms\": %.4f, \"stats\": [%lld, %lld, %lld, %lld, %lld, %lld]}",
           NG, hits, equal, miss_checked, bad, fill_ms, lookup_ms, percall_ms, rows_calls ? rows_ms / rows_calls : 0.0, (long long)st[0], (long long)st[1], (long long)st[2], (long long)st[3], (long long)st[4], (long long)st[5]);
    return bad;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: casim_native TRACE [--device N] [--dump FILE] [--repeat K] [--shim]\n"); return 2; }
    int device = 0, repeat = 3; const char* dump = nullptr; bool shim = false;
    for (int i = 2; i < argc; ++i) {
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dump") && i + 1 < argc) dump = argv[++i];
        else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--shim")) shim = true;
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
            if (shim && shim_replay(ctx, pegs, groups, fastpath, enc) != 0) exit_code = 5;
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
