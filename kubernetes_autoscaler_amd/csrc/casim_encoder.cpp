// casim_encoder.cpp — host encoder (SURVEY Appendix C "H1"): dictionary-encodes the string side of
// the scheduler Filter plugins once per scale-up loop so that the device only sees integers
// and bitmasks.  Pure host C++, no HIP.  C ABI in include/casim.h (casim_enc_*).
//
// What is evaluated HERE, once per (PEG, dictionary entry), instead of per pod x node in Go:
//   Toleration.ToleratesTaint          V/api/core/v1/toleration.go:52-114
//   labels.Requirement.Matches         V/apimachinery/pkg/labels/selector.go:247-292
//   RequiredNodeAffinity.Match         V/component-helpers/scheduling/corev1/nodeaffinity/nodeaffinity.go:323-333
//   HostPortInfo.CheckConflict         V/kube-scheduler/framework/types.go:602-640
//   AffinityTerm.Matches               V/kube-scheduler/framework/types.go:390-395
//   shouldUseFastPath / labelSelectorMatches  CA/estimator/binpacking_estimator.go:399-425,444-450
#include <errno.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/casim.h"

// Loops of casim_enc_finalize that touch every node / every running pod's spec record run on up to four threads (like the host loops of
// the engine, casim_pipeline.h par_for): f(lo, hi, t) over [0, n) cut into T slices.  CASIM_HOST_THREADS = 1 switches it off,
// CASIM_HOST_GRAIN sets the smallest slice worth a thread (tests: threads on small inputs).
namespace {
inline int enc_threads(size_t n, size_t grain) {
    int T = 4;
    if (const char* ev = getenv("CASIM_HOST_THREADS")) { const int v = atoi(ev); T = v < 1 ? 1 : (v > 4 ? 4 : v); }
    if (const char* ev = getenv("CASIM_HOST_GRAIN")) { const long v = atol(ev); if (v > 0) grain = (size_t)v; }
    if (grain > 0 && n / grain < (size_t)T) T = (int)(n / grain);
    return T < 1 ? 1 : T;
}
template <class F>
inline void enc_par_for(size_t n, int T, F f) {
    if (T <= 1) { f((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&f, n, T, t] { f(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T, t); });
    f((size_t)0, n / (size_t)T, 0);
    for (auto& x : th) x.join();
}
}  // namespace

// CASIM_ENC_TIMING=1: casim_enc_finalize prints the milliseconds of its stages to stderr (tools/casim_incr_bench)
namespace {
struct EncStageTimer {
    bool on; std::chrono::steady_clock::time_point t0; std::string line;
    EncStageTimer() : on(getenv("CASIM_ENC_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        char b[96]; snprintf(b, sizeof b, " %s %.2f", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        line += b; t0 = t1;
    }
    ~EncStageTimer() { if (on) fprintf(stderr, "[enc finalize ms]%s\n", line.c_str()); }
};
}  // namespace

namespace {

const char* kHostname = "kubernetes.io/hostname";
const char* kUnschedulableTaint = "node.kubernetes.io/unschedulable";

enum ReqOp { kIn, kNotIn, kExists, kDoesNotExist, kGt, kLt, kBadOp, kNodeTerms };
enum TolOp { kTolEqual, kTolExists, kTolLt, kTolGt, kTolBad };

std::string S(const char* s) { return s ? std::string(s) : std::string(); }

struct NodeTerm;
// op == kNodeTerms: a whole nodeSelectorTerms list (ORed) folded into ONE dictionary entry; key / values unused
struct Requirement { std::string key; ReqOp op; std::vector<std::string> values; std::vector<NodeTerm> terms; };
struct NodeTerm { std::vector<Requirement> exprs, fields; };   // v1.NodeSelectorTerm: matchExpressions AND matchFields
struct Toleration { std::string key; TolOp op; std::string value, effect; };
struct Taint { std::string key, value, effect; bool operator<(const Taint& o) const { return std::tie(key, value, effect) < std::tie(o.key, o.value, o.effect); } };
struct Port { std::string ip, proto; int32_t port; bool operator<(const Port& o) const { return std::tie(ip, proto, port) < std::tie(o.ip, o.proto, o.port); } };
// namespaceSelector: has_ns_sel = the field is set (an EMPTY selector selects every namespace, labels.Everything); finalize
// resolves it into `namespaces` / `all_ns` the way InterPodAffinity.PreFilter does (mergeAffinityTermNamespacesIfNotEmpty,
// V/kubernetes/pkg/scheduler/framework/plugins/interpodaffinity/plugin.go:144-157)
struct Term { std::string topology_key; std::vector<std::string> namespaces; std::vector<Requirement> selector;
              bool has_ns_sel = false, auto_ns = false, all_ns = false, ns_resolved = false; std::vector<Requirement> ns_sel; };
// A label set: a handful of (key, value) pairs kept sorted by key in one vector — the subset of std::map the encoder uses
// (operator[], find, count, end).  One record per pod at cluster scale: a tree node per label was a third of the encode calls' time.
struct Labels {
    typedef std::pair<std::string, std::string> Item;
    typedef std::vector<Item>::const_iterator const_iterator;
    std::vector<Item> v;
    const_iterator begin() const { return v.begin(); }
    const_iterator end() const { return v.end(); }
    const_iterator find(const std::string& k) const {
        for (auto it = v.begin(); it != v.end(); ++it) if (it->first == k) return it;   // (linear: label sets are tiny)
        return v.end();
    }
    size_t count(const std::string& k) const { return find(k) != v.end() ? 1 : 0; }
    std::string& operator[](const std::string& k) {
        auto it = v.begin();
        while (it != v.end() && it->first < k) ++it;
        if (it == v.end() || it->first != k) it = v.insert(it, Item(k, std::string()));
        return it->second;
    }
};

struct Spread { int32_t max_skew; std::string key; int32_t min_domains; std::vector<Requirement> selector; bool taints_honor = false; bool affinity_honor = true; };
struct PodSpec {
    std::vector<Spread> spread;   // DoNotSchedule topologySpreadConstraints
    std::string ns;
    int64_t req[CASIM_MAX_RES];
    Labels labels;
    std::vector<int32_t> tolerations;   // ids into casim_encoder::tol_dict (tolerations are interned when they arrive: casim_enc_pod_add_toleration)
    std::vector<std::pair<std::string, std::string>> node_selector;
    std::vector<Requirement> node_affinity;   // ONE required term (casim_enc_pod_add_node_affinity_req)
    std::vector<NodeTerm> node_terms;         // nodeSelectorTerms, ORed (casim_enc_pod_add_node_affinity_term); exclusive with the above
    bool has_node_terms = false;
    std::vector<Port> ports;
    std::vector<Term> anti;
    std::vector<Term> aff;        // REQUIRED pod affinity terms (casim_enc_pod_add_affinity_term): a pod "matches" when it matches ALL
    double fp_cpu = 0, fp_mem = 0;
    bool unsupported = false;
    std::string why;
    std::string extra;            // casim_enc_pod_set_spec_extra: digest of the sanitized PodSpec fields the encoder does not model (grouping only)
};
struct Group {
    std::string name;
    int64_t alloc[CASIM_MAX_RES];
    // Allocatable[name] handed over (casim_enc_group_set_allocatable) while no pod had asked for the name yet: kept aside, written into the
    // name's lane by finalize IF a request has opened one by then.  Allocatable never opens a lane (real nodes list hugepages-1Gi: 0,
    // hugepages-2Mi: 0, attachable-volumes-*: columns no pending pod reads, which used to widen every table past the register packer's four lanes)
    std::vector<std::pair<std::string, int64_t>> named_alloc;
    int32_t allowed = 0;
    int64_t cap_cpu = 0, cap_mem = 0;
    double fp_cap_cpu = 0, fp_cap_mem = 0;
    bool unschedulable = false;
    Labels labels;
    std::vector<Taint> taints;
    int32_t max_nodes = 0, existing = 0, last_index = 0;
    std::vector<int32_t> preloaded;
    bool has_pegs = false;
    std::vector<int32_t> pegs;
};
struct Peg { int32_t spec; int32_t count; };
struct ExistingPod { int32_t spec; Labels node_labels; };

bool parse_int64(const std::string& s, int64_t* out) {  // strconv.ParseInt(s, 10, 64)
    if (s.empty()) return false;
    size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (i == s.size()) return false;
    for (size_t j = i; j < s.size(); ++j) if (s[j] < '0' || s[j] > '9') return false;
    errno = 0;
    long long v = strtoll(s.c_str(), nullptr, 10);
    if (errno == ERANGE) return false;
    *out = v;
    return true;
}
ReqOp parse_req_op(const char* op) {
    const std::string o = S(op);
    if (o == "In") return kIn;
    if (o == "NotIn") return kNotIn;
    if (o == "Exists") return kExists;
    if (o == "DoesNotExist") return kDoesNotExist;
    if (o == "Gt") return kGt;
    if (o == "Lt") return kLt;
    return kBadOp;
}
TolOp parse_tol_op(const char* op) {
    const std::string o = S(op);
    if (o.empty() || o == "Equal") return kTolEqual;  // empty operator means Equal
    if (o == "Exists") return kTolExists;
    if (o == "Lt") return kTolLt;
    if (o == "Gt") return kTolGt;
    return kTolBad;
}
bool requirement_matches(const Requirement& r, const Labels& ls) {
    auto it = ls.find(r.key);
    const bool exists = it != ls.end();
    switch (r.op) {
    case kIn: if (!exists) return false; for (auto& v : r.values) if (v == it->second) return true; return false;
    case kNotIn: if (!exists) return true; for (auto& v : r.values) if (v == it->second) return false; return true;
    case kExists: return exists;
    case kDoesNotExist: return !exists;
    case kGt: case kLt: {
        if (!exists) return false;
        int64_t lv, rv;
        if (!parse_int64(it->second, &lv)) return false;
        if (r.values.size() != 1 || !parse_int64(r.values[0], &rv)) return false;
        return (r.op == kGt && lv > rv) || (r.op == kLt && lv < rv);
    }
    default: return false;
    }
}
bool selector_matches(const std::vector<Requirement>& sel, const Labels& ls) {
    for (auto& r : sel) if (!requirement_matches(r, ls)) return false;
    return true;
}
// labels.NewRequirement rules 1, 2, 4, 5 (V/apimachinery/pkg/labels/selector.go:183-215); a requirement that does not parse
// is a parse error of its whole term (nodeaffinity.go:210-246)
bool requirement_parses(const Requirement& r) {
    int64_t v;
    switch (r.op) {
    case kIn: case kNotIn: return !r.values.empty();
    case kExists: case kDoesNotExist: return r.values.empty();
    case kGt: case kLt: return r.values.size() == 1 && parse_int64(r.values[0], &v);
    default: return false;
    }
}
// LazyErrorNodeSelector.Match (nodeaffinity.go:85-107) over nodeSelectorTerm.match (:187-198): terms ORed, empty terms dropped
// (:60-64), terms with a parse error never match, matchFields (In / NotIn with one value, :260-291) see metadata.name only and
// are skipped for a node without a name.
bool node_terms_match(const std::vector<NodeTerm>& terms, const Labels& ls, const std::string& node_name) {
    for (auto& t : terms) {
        if (t.exprs.empty() && t.fields.empty()) continue;
        bool ok = true;
        for (auto& r : t.exprs) ok = ok && requirement_parses(r);
        for (auto& r : t.fields) ok = ok && (r.op == kIn || r.op == kNotIn) && r.values.size() == 1;
        if (!ok) continue;
        if (!t.exprs.empty() && !selector_matches(t.exprs, ls)) continue;
        if (!t.fields.empty() && !node_name.empty()) {
            for (auto& r : t.fields) {
                const std::string have = r.key == "metadata.name" ? node_name : std::string();
                ok = ok && (r.op == kIn ? have == r.values[0] : have != r.values[0]);
            }
            if (!ok) continue;
        }
        return true;
    }
    return false;
}
bool term_matches(const Term& t, const PodSpec& target) {
    bool ns_ok = t.all_ns;
    for (auto& n : t.namespaces) if (n == target.ns) { ns_ok = true; break; }
    return ns_ok && selector_matches(t.selector, target.labels);
}
bool tolerates(const Toleration& t, const Taint& tn, bool cmp_ops) {
    if (!t.effect.empty() && t.effect != tn.effect) return false;
    if (!t.key.empty() && t.key != tn.key) return false;
    switch (t.op) {
    case kTolEqual: return t.value == tn.value;
    case kTolExists: return true;
    case kTolLt: case kTolGt: {
        if (!cmp_ops) return false;
        int64_t tv, nv;
        if (t.value.empty() || t.value[0] == '+' || !parse_int64(t.value, &tv)) return false;  // IsDecimalInteger
        if (tn.value.empty() || tn.value[0] == '+' || !parse_int64(tn.value, &nv)) return false;
        return t.op == kTolLt ? nv < tv : nv > tv;
    }
    default: return false;
    }
}
Port sanitize(Port p) {
    if (p.ip.empty()) p.ip = "0.0.0.0";
    if (p.proto.empty()) p.proto = "TCP";
    return p;
}
bool ports_conflict(const Port& want, const Port& used) {  // CheckConflict(want) against one used entry
    if (want.port <= 0 || used.port <= 0) return false;
    if (want.proto != used.proto || want.port != used.port) return false;
    return want.ip == "0.0.0.0" || used.ip == "0.0.0.0" || want.ip == used.ip;
}
std::string req_signature(const Requirement& r) {
    std::string s = std::to_string((int)r.op) + "\x1f" + r.key;
    for (auto& v : r.values) { s += "\x1f"; s += v; }
    for (auto& t : r.terms) {
        s += "\x1d";
        for (auto& x : t.exprs) { s += "\x1e"; s += req_signature(x); }
        s += "\x1c";
        for (auto& x : t.fields) { s += "\x1e"; s += req_signature(x); }
    }
    return s;
}

struct BitAlloc {
    int n = 0;
    int next() { return n++; }
    int words() const { return (n + 63) / 64; }
};
inline void set_bit(std::vector<uint64_t>& m, size_t row, int W, int bit) { m[row * (size_t)W + (size_t)(bit >> 6)] |= 1ull << (bit & 63); }

}  // namespace

struct casim_encoder {
    casim_encoder_options opt;
    // Named resource lanes (ABI 9, casim_enc_lane): lane_names[i] is the scalar / extended resource of lane first_named + i.  The flat-array entry
    // points keep opt.n_res as their STRIDE; the tables carry out_res = max(opt.n_res, first_named + names) lanes.  Frozen by finalize.
    std::vector<std::string> lane_names;
    bool touch_ephemeral = false;   // a two-lane encoder that was handed ephemeral-storage by name: lane 2 joins the tables
    int out_res = 0;       // lanes of the flat tables, fixed by the (last full) finalize; 0 before
    int first_named() const { return opt.n_res > 3 ? opt.n_res : 3; }
    int lanes_now() const {
        int n = lane_names.empty() ? opt.n_res : first_named() + (int)lane_names.size();
        if (touch_ephemeral && n < 3) n = 3;
        return n > opt.n_res ? n : opt.n_res;
    }
    // Distinct tolerations (key, operator, value, effect), in order of arrival; pod specs hold ids.  Clusters repeat a handful of tolerations
    // over thousands of pods: a hit stores four bytes instead of three strings, and finalize asks "does it tolerate taint T" once per
    // distinct toleration instead of once per pod (peg_table stage: C3 1.33 -> 0.2 ms).
    std::vector<Toleration> tol_dict;
    std::unordered_map<std::string, int32_t> tol_id;
    std::string tol_sig;   // scratch
    int32_t intern_toleration(const char* key, TolOp op, const char* value, const char* effect) {
        const char* k = key ? key : ""; const char* v = value ? value : ""; const char* f = effect ? effect : "";
        const uint32_t lens[2] = {(uint32_t)strlen(k), (uint32_t)strlen(v)};   // (lengths in: no separator a string could contain)
        tol_sig.assign(1, (char)('0' + (int)op));
        tol_sig.append((const char*)lens, sizeof lens);
        tol_sig.append(k, lens[0]); tol_sig.append(v, lens[1]); tol_sig.append(f);
        auto it = tol_id.find(tol_sig);
        if (it != tol_id.end()) return it->second;
        const int32_t id = (int32_t)tol_dict.size();
        tol_dict.push_back(Toleration{std::string(k), op, std::string(v), std::string(f)});
        tol_id.emplace(tol_sig, id);
        return id;
    }
    std::vector<int32_t> term_specs;   // specs with (anti-)affinity terms, in the order they got their first one (finalize looks at these only: at cluster scale the
                                       // other 150 000 spec records stay cold)
    std::deque<PodSpec> specs;   // (a deque: at cluster scale there is one spec per running pod, and growing a vector moved every one of them ~2.5 times)
    std::deque<Group> groups;    // (same: one record per node in the per-node entry points)
    std::vector<Peg> pegs;
    std::vector<ExistingPod> existing;
    std::map<std::string, Labels> namespaces;   // the namespace lister: name -> labels (only read by namespaceSelector terms)
    bool finalized = false;
    // flat tables
    int Wt = 0, Wl = 0, Wx = 0, Wz = 0;
    std::vector<int64_t> req, alloc, init_req, waste_cpu, waste_mem;
    std::vector<int32_t> count, allowed, init_pods, max_nodes, existing_nodes, last_index, peg_off, peg_idx;
    std::vector<uint32_t> pflags, gflags;
    std::vector<uint64_t> tol, sel, xblock, xmark, zblock, zmark, zpol, xpol, taint, label, init_excl, init_zone, zone_valid, xports;
    std::vector<double> fp_cpu, fp_mem, cap_cpu, cap_mem;
    int dict[4] = {0, 0, 0, 0};
    // domain rules (per-node mode)
    struct {
        int32_t n_keys = 0, n_rules = 0, n_rows = 0;
        std::vector<int32_t> node_domain, key_domains, r_class, r_key, r_kind, r_skew, r_mind, r_self, r_row, count_init, class_off, inc_off, inc_rule;
        std::vector<int32_t> dom_nodes, node_contrib;
        std::vector<uint8_t> key_host, r_ghost;
        std::vector<int64_t> r_off;
        std::vector<uint8_t> exists;
        std::vector<uint64_t> elig;
        int32_t n_taint_rules = 0;
    } dr;
    // ---- what an incremental re-encode needs from the last full finalize (casim_enc_refinalize, per-node mode) ----
    struct FinalRule { int cls, key, kind, row; const Spread* sc; };
    struct {
        bool valid = false;
        size_t n_specs = 0, NG = 0, G = 0;
        std::map<Taint, int> taint_id;
        std::vector<Requirement> lreqs;
        std::map<Port, int> port_bit;
        std::map<int32_t, int> pre_occ_bit;
        std::vector<uint8_t> running;                         // [specs] preloaded on some node at the last full finalize (or proved inert since)
        bool hostname_inert = false;                          // hostname anti-affinity terms exist and no node carried kubernetes.io/hostname: no node bits for them
        std::vector<size_t> with_terms;                       // PEGs whose spec carries hostname anti-affinity terms
        std::vector<std::string> keys;                        // topology keys of the domain rules
        std::vector<std::map<std::string, int>> val_id;       // per key: label value -> domain id
        std::vector<FinalRule> rules;
        std::vector<std::array<int, 4>> class_rows;           // per class: eligibility row per (affinity honoured, taints honoured) combination, -1 = none
        std::vector<uint8_t> dirty;                           // [groups] reset since the last (re)finalize
        std::vector<int32_t> dirty_list;
    } fs;
    bool updating = false;
    std::vector<int64_t> rows_alloc, rows_init_req, rows_waste_cpu, rows_waste_mem;   // casim_enc_group_rows: compact copies
    std::vector<int32_t> rows_allowed, rows_init_pods, rows_max_nodes, rows_existing, rows_last_index;
    std::vector<uint32_t> rows_gflags;
    std::vector<uint64_t> rows_taint, rows_label, rows_init_excl, rows_init_zone, rows_zone_valid;
    std::vector<double> rows_cap_cpu, rows_cap_mem;
};

extern "C" {

casim_encoder* casim_enc_create(const casim_encoder_options* opts) {
    if (!opts || opts->n_res < 2 || opts->n_res > CASIM_MAX_RES) return nullptr;
    casim_encoder* e = new (std::nothrow) casim_encoder();
    if (e) e->opt = *opts;
    return e;
}
void casim_enc_destroy(casim_encoder* e) { delete e; }

// After finalize the encoder is read-only — except inside an update session (casim_enc_begin_update .. casim_enc_refinalize), where
// NEW pod specs may be described and the groups that were reset (casim_enc_group_reset) refilled.  Everything else stays frozen:
// the specs of the last finalize feed dictionaries that an update does not rebuild.
#define ENC_OPEN(e) if (!(e) || ((e)->finalized && !(e)->updating)) return CASIM_ERR_INVALID
#define ENC_CHECK(e) if (!(e) || (e)->finalized) return CASIM_ERR_INVALID
#define POD_CHECK(e, p) ENC_OPEN(e); if ((p) < 0 || (size_t)(p) >= (e)->specs.size() || ((e)->updating && (size_t)(p) < (e)->fs.n_specs)) return CASIM_ERR_INVALID
#define GRP_CHECK(e, g) ENC_OPEN(e); if ((g) < 0 || (size_t)(g) >= (e)->groups.size() || ((e)->updating && !(e)->fs.dirty[(size_t)(g)])) return CASIM_ERR_INVALID

int32_t casim_enc_add_group(casim_encoder* e, const char* template_name, const int64_t* alloc, int32_t allowed_pods,
                            int64_t capacity_cpu_milli, int64_t capacity_mem_bytes, int32_t unschedulable) {
    ENC_CHECK(e);
    if (!alloc) return CASIM_ERR_INVALID;
    Group g;
    g.name = S(template_name);
    for (int r = 0; r < CASIM_MAX_RES; ++r) g.alloc[r] = r < e->opt.n_res ? alloc[r] : 0;
    g.allowed = allowed_pods; g.cap_cpu = capacity_cpu_milli; g.cap_mem = capacity_mem_bytes; g.unschedulable = unschedulable != 0;
    // Capacity.Cpu().AsApproximateFloat64() for a milli quantity = float64(milli) * 10^-3 (quantity.go:468-483)
    g.fp_cap_cpu = (double)capacity_cpu_milli * 1e-3; g.fp_cap_mem = (double)capacity_mem_bytes;
    e->groups.push_back(g);
    return (int32_t)e->groups.size() - 1;
}
int32_t casim_enc_group_set_fastpath_capacity(casim_encoder* e, int32_t group, double cpu, double mem) {
    GRP_CHECK(e, group); e->groups[group].fp_cap_cpu = cpu; e->groups[group].fp_cap_mem = mem; return CASIM_OK;
}
int32_t casim_enc_group_add_label(casim_encoder* e, int32_t group, const char* key, const char* value) {
    GRP_CHECK(e, group); e->groups[group].labels[S(key)] = S(value); return CASIM_OK;
}
int32_t casim_enc_group_add_taint(casim_encoder* e, int32_t group, const char* key, const char* value, const char* effect) {
    GRP_CHECK(e, group); e->groups[group].taints.push_back(Taint{S(key), S(value), S(effect)}); return CASIM_OK;
}
int32_t casim_enc_group_set_limits(casim_encoder* e, int32_t group, int32_t max_nodes, int32_t existing_nodes, int32_t last_index) {
    GRP_CHECK(e, group);
    if (existing_nodes < 0 || last_index < 0) return CASIM_ERR_INVALID;
    e->groups[group].max_nodes = max_nodes; e->groups[group].existing = existing_nodes; e->groups[group].last_index = last_index;
    return CASIM_OK;
}
int32_t casim_enc_group_add_preloaded_pod(casim_encoder* e, int32_t group, int32_t pod_spec) {
    GRP_CHECK(e, group);
    if (pod_spec < 0 || (size_t)pod_spec >= e->specs.size()) return CASIM_ERR_INVALID;
    e->groups[group].preloaded.push_back(pod_spec); return CASIM_OK;
}
int32_t casim_enc_add_running_pods(casim_encoder* e, int32_t n_pods, const int32_t* group, const int32_t* ns, const int64_t* req,
                                   const int32_t* label_off, const int32_t* label_key, const int32_t* label_val,
                                   const char* const* strings, int32_t n_strings) {
    ENC_OPEN(e);
    if (e->updating || n_pods < 0 || n_strings < 0) return CASIM_ERR_INVALID;
    if (n_pods == 0) return (int32_t)e->specs.size();
    if (!ns || !req || !label_off || !strings || (label_off[n_pods] > 0 && (!label_key || !label_val))) return CASIM_ERR_INVALID;
    // validate everything first: nothing is added on an error
    if (label_off[0] != 0) return CASIM_ERR_INVALID;
    const int32_t NGr = (int32_t)e->groups.size();
    for (int32_t i = 0; i < n_pods; ++i) {
        if (ns[i] < 0 || ns[i] >= n_strings || label_off[i + 1] < label_off[i]) return CASIM_ERR_INVALID;
        if (group && (group[i] < -1 || group[i] >= NGr)) return CASIM_ERR_INVALID;
    }
    for (int32_t k = 0; k < label_off[n_pods]; ++k)
        if (label_key[k] < 0 || label_key[k] >= n_strings || label_val[k] < 0 || label_val[k] >= n_strings) return CASIM_ERR_INVALID;
    std::vector<std::string> strs((size_t)n_strings);
    for (int32_t k = 0; k < n_strings; ++k) strs[(size_t)k] = S(strings[k]);
    const int32_t first = (int32_t)e->specs.size();
    const int R = e->opt.n_res;
    for (int32_t i = 0; i < n_pods; ++i) {
        e->specs.emplace_back();
        PodSpec& p = e->specs.back();
        p.ns = strs[(size_t)ns[i]];
        for (int r = 0; r < CASIM_MAX_RES; ++r) p.req[r] = r < R ? req[(size_t)i * (size_t)R + (size_t)r] : 0;
        p.labels.v.reserve((size_t)(label_off[i + 1] - label_off[i]));
        for (int32_t k = label_off[i]; k < label_off[i + 1]; ++k) p.labels[strs[(size_t)label_key[k]]] = strs[(size_t)label_val[k]];
        if (group && group[i] >= 0) e->groups[(size_t)group[i]].preloaded.push_back(first + i);
    }
    return first;
}
int32_t casim_enc_group_set_pegs(casim_encoder* e, int32_t group, const int32_t* pegs, int32_t n) {
    GRP_CHECK(e, group);
    if (n < 0 || (n > 0 && !pegs)) return CASIM_ERR_INVALID;
    e->groups[group].has_pegs = true;
    e->groups[group].pegs.assign(pegs, pegs + n);
    return CASIM_OK;
}
int32_t casim_enc_add_pod_spec(casim_encoder* e, const char* namespace_, const int64_t* req) {
    ENC_OPEN(e);
    if (!req) return CASIM_ERR_INVALID;
    PodSpec p;
    p.ns = S(namespace_);
    for (int r = 0; r < CASIM_MAX_RES; ++r) p.req[r] = r < e->opt.n_res ? req[r] : 0;
    e->specs.push_back(std::move(p));
    return (int32_t)e->specs.size() - 1;
}
// ---- named resources (ABI 9) ------------------------------------------------------------------------------------------------
// framework.Resource (V/kubernetes/pkg/scheduler/framework/types.go:989-998): MilliCPU, Memory, EphemeralStorage are fields of their own,
// everything else lives in ScalarResources BY NAME — fitsRequest walks the pod's map (noderesources/fit.go:731-763), AddPodInfo adds name by
// name (types.go:444-448).  Lane numbers are an encoding detail: they stay behind the ABI, so that no binding can drop or mis-order one
// (VERDICT r4 missing #2: the Go shim copied three lanes and a pod asking for nvidia.com/gpu was estimated without it).
static int lane_of(casim_encoder* e, const std::string& name, bool may_add) {
    if (name == "cpu") return CASIM_RES_CPU;
    if (name == "memory") return CASIM_RES_MEM;
    if (name == "ephemeral-storage") {   // always lane 2; a two-lane encoder grows to three when a non-zero value arrives (touch_ephemeral)
        if (e->opt.n_res > CASIM_RES_EPHEMERAL || e->touch_ephemeral || may_add) return CASIM_RES_EPHEMERAL;
        return CASIM_ERR_NO_LANE;
    }
    for (size_t i = 0; i < e->lane_names.size(); ++i) if (e->lane_names[i] == name) return e->first_named() + (int)i;
    if (!may_add || e->first_named() + (int)e->lane_names.size() >= CASIM_MAX_RES) return CASIM_ERR_NO_LANE;
    e->lane_names.push_back(name);
    return e->first_named() + (int)e->lane_names.size() - 1;
}
int32_t casim_enc_lane(casim_encoder* e, const char* resource_name) {
    if (!e || !resource_name || !*resource_name) return CASIM_ERR_INVALID;
    const std::string name = S(resource_name);
    if (name == "pods") return CASIM_ERR_INVALID;   // (the pod count is allowed_pods, not a lane)
    // new names only while the lane count is still open: before the first finalize (an update session re-encodes rows of FIXED width)
    return lane_of(e, name, /*may_add=*/!e->finalized);
}
int32_t casim_enc_pod_set_request(casim_encoder* e, int32_t pod, const char* resource_name, int64_t value) {
    POD_CHECK(e, pod);
    if (!resource_name || !*resource_name || value < 0) return CASIM_ERR_INVALID;
    const std::string name = S(resource_name);
    if (name == "pods") return CASIM_ERR_INVALID;
    // a request of ZERO opens no lane: fitsRequest skips zero quantities (fit.go:733) and adding zero to a node's sums changes nothing
    // (types.go:444-448) — the name only matters once some pod asks for a non-zero amount
    const int lane = lane_of(e, name, /*may_add=*/!e->finalized && value != 0);
    PodSpec& p = e->specs[pod];
    if (lane < 0) {   // every lane is taken: the pod leaves the encoded subset (fail closed), unless it asks for nothing
        if (value == 0) return CASIM_OK;
        p.unsupported = true;
        if (p.why.empty()) p.why = "resource " + name + ": no lane left (CASIM_MAX_RES)";
        return CASIM_ENC_DELEGATED;
    }
    p.req[lane] = value;
    if (lane == CASIM_RES_EPHEMERAL && e->opt.n_res < 3 && value != 0) e->touch_ephemeral = true;
    return CASIM_OK;
}
int32_t casim_enc_group_set_allocatable(casim_encoder* e, int32_t group, const char* resource_name, int64_t value) {
    GRP_CHECK(e, group);
    if (!resource_name || !*resource_name) return CASIM_ERR_INVALID;
    const std::string name = S(resource_name);
    Group& g = e->groups[group];
    if (name == "pods") { g.allowed = (int32_t)value; return CASIM_OK; }
    // Allocatable never opens a lane: only a pod's non-zero request does (a column no pod reads is not a column).  A name without a lane
    // is kept aside until finalize, by when every pod of the loop has been seen — the order of pods and groups does not matter.
    const int lane = lane_of(e, name, /*may_add=*/false);
    for (auto it = g.named_alloc.begin(); it != g.named_alloc.end(); ++it) if (it->first == name) { g.named_alloc.erase(it); break; }
    if (lane >= 0) { g.alloc[lane] = value; return CASIM_OK; }
    if (e->finalized) return CASIM_ENC_DELEGATED;   // (fixed table width: the name has no column, and every pod that asks for it is delegated by casim_enc_pod_set_request)
    g.named_alloc.emplace_back(name, value);
    return CASIM_OK;
}
// finalize: names a request has opened a lane for since the group was described
static void resolve_named_allocatable(casim_encoder* e) {
    for (auto& g : e->groups) {
        if (g.named_alloc.empty()) continue;
        size_t keep = 0;
        for (auto& nv : g.named_alloc) {
            const int lane = lane_of(e, nv.first, /*may_add=*/false);
            if (lane >= 0) g.alloc[lane] = nv.second; else g.named_alloc[keep++] = nv;
        }
        g.named_alloc.resize(keep);
    }
}
int32_t casim_enc_lane_count(const casim_encoder* e) { return e ? (e->finalized ? e->out_res : e->lanes_now()) : CASIM_ERR_INVALID; }
const char* casim_enc_lane_name(const casim_encoder* e, int32_t lane) {
    if (!e || lane < 0 || lane >= CASIM_MAX_RES) return nullptr;
    if (lane == CASIM_RES_CPU) return "cpu";
    if (lane == CASIM_RES_MEM) return "memory";
    if (lane == CASIM_RES_EPHEMERAL && e->first_named() == 3) return "ephemeral-storage";
    const int i = lane - e->first_named();
    return i >= 0 && (size_t)i < e->lane_names.size() ? e->lane_names[(size_t)i].c_str() : nullptr;
}
int32_t casim_enc_pod_add_label(casim_encoder* e, int32_t pod, const char* key, const char* value) {
    POD_CHECK(e, pod); e->specs[pod].labels[S(key)] = S(value); return CASIM_OK;
}
int32_t casim_enc_pod_add_toleration(casim_encoder* e, int32_t pod, const char* key, const char* op, const char* value, const char* effect) {
    POD_CHECK(e, pod); e->specs[pod].tolerations.push_back(e->intern_toleration(key, parse_tol_op(op), value, effect)); return CASIM_OK;
}
int32_t casim_enc_pod_add_node_selector(casim_encoder* e, int32_t pod, const char* key, const char* value) {
    POD_CHECK(e, pod); e->specs[pod].node_selector.push_back({S(key), S(value)}); return CASIM_OK;
}
static Requirement make_req(const char* key, const char* op, const char* const* values, int32_t n) {
    Requirement r; r.key = S(key); r.op = parse_req_op(op);
    for (int i = 0; i < n; ++i) r.values.push_back(S(values[i]));
    return r;
}
int32_t casim_enc_pod_add_node_affinity_req(casim_encoder* e, int32_t pod, const char* key, const char* op, const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (n_values < 0 || (n_values > 0 && !values)) return CASIM_ERR_INVALID;
    if (e->specs[pod].has_node_terms) return CASIM_ERR_INVALID;   // one NodeSelector per pod: either API, not both
    e->specs[pod].node_affinity.push_back(make_req(key, op, values, n_values)); return CASIM_OK;
}
int32_t casim_enc_pod_add_node_affinity_term(casim_encoder* e, int32_t pod) {
    POD_CHECK(e, pod);
    if (!e->specs[pod].node_affinity.empty()) return CASIM_ERR_INVALID;
    e->specs[pod].node_terms.emplace_back();
    e->specs[pod].has_node_terms = true;
    return (int32_t)e->specs[pod].node_terms.size() - 1;
}
int32_t casim_enc_node_term_add_requirement(casim_encoder* e, int32_t pod, int32_t term, int32_t is_field, const char* key, const char* op,
                                            const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].node_terms.size()) return CASIM_ERR_INVALID;
    if (n_values < 0 || (n_values > 0 && !values)) return CASIM_ERR_INVALID;
    NodeTerm& t = e->specs[pod].node_terms[(size_t)term];
    (is_field ? t.fields : t.exprs).push_back(make_req(key, op, values, n_values)); return CASIM_OK;
}
int32_t casim_enc_add_namespace(casim_encoder* e, const char* name) {
    ENC_CHECK(e); e->namespaces[S(name)]; return CASIM_OK;
}
int32_t casim_enc_namespace_add_label(casim_encoder* e, const char* name, const char* key, const char* value) {
    ENC_CHECK(e); e->namespaces[S(name)][S(key)] = S(value); return CASIM_OK;
}
int32_t casim_enc_term_set_namespace_selector(casim_encoder* e, int32_t pod, int32_t term) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].anti.size()) return CASIM_ERR_INVALID;
    Term& t = e->specs[pod].anti[(size_t)term];
    // the pod's own namespace only stands in when there are neither namespaces nor a selector (types.go:439-447)
    if (t.auto_ns) { t.namespaces.clear(); t.auto_ns = false; }
    t.has_ns_sel = true; return CASIM_OK;
}
int32_t casim_enc_term_add_namespace_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key, const char* op,
                                                 const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].anti.size() || !e->specs[pod].anti[(size_t)term].has_ns_sel) return CASIM_ERR_INVALID;
    if (n_values < 0 || (n_values > 0 && !values)) return CASIM_ERR_INVALID;
    e->specs[pod].anti[(size_t)term].ns_sel.push_back(make_req(key, op, values, n_values)); return CASIM_OK;
}
// the same for a required AFFINITY term: the term is the incoming pod's, PreFilter replaces a non-empty selector by the namespaces the
// lister returns for it (interpodaffinity/plugin.go:144-157) — exactly what finalize does with the namespaces it was given
int32_t casim_enc_aff_term_set_namespace_selector(casim_encoder* e, int32_t pod, int32_t term) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].aff.size()) return CASIM_ERR_INVALID;
    Term& t = e->specs[pod].aff[(size_t)term];
    if (t.auto_ns) { t.namespaces.clear(); t.auto_ns = false; }
    t.has_ns_sel = true; return CASIM_OK;
}
int32_t casim_enc_aff_term_add_namespace_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key, const char* op,
                                                     const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].aff.size() || !e->specs[pod].aff[(size_t)term].has_ns_sel) return CASIM_ERR_INVALID;
    if (n_values < 0 || (n_values > 0 && !values)) return CASIM_ERR_INVALID;
    e->specs[pod].aff[(size_t)term].ns_sel.push_back(make_req(key, op, values, n_values)); return CASIM_OK;
}
int32_t casim_enc_pod_add_host_port(casim_encoder* e, int32_t pod, const char* ip, const char* protocol, int32_t port) {
    POD_CHECK(e, pod); e->specs[pod].ports.push_back(Port{S(ip), S(protocol), port}); return CASIM_OK;
}
int32_t casim_enc_pod_add_anti_affinity_term(casim_encoder* e, int32_t pod, const char* topology_key, const char* const* namespaces, int32_t n_namespaces) {
    POD_CHECK(e, pod);
    if (n_namespaces < 0 || (n_namespaces > 0 && !namespaces)) return CASIM_ERR_INVALID;
    Term t; t.topology_key = S(topology_key);
    if (n_namespaces == 0) { t.namespaces.push_back(e->specs[pod].ns); t.auto_ns = true; }  // getNamespacesFromPodAffinityTerm
    for (int i = 0; i < n_namespaces; ++i) t.namespaces.push_back(S(namespaces[i]));
    if (e->specs[pod].anti.empty() && e->specs[pod].aff.empty()) e->term_specs.push_back(pod);
    e->specs[pod].anti.push_back(t);
    return (int32_t)e->specs[pod].anti.size() - 1;
}
int32_t casim_enc_term_add_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key, const char* op, const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].anti.size()) return CASIM_ERR_INVALID;
    if (n_values < 0 || (n_values > 0 && !values)) return CASIM_ERR_INVALID;
    e->specs[pod].anti[term].selector.push_back(make_req(key, op, values, n_values)); return CASIM_OK;
}
int32_t casim_enc_pod_add_affinity_term(casim_encoder* e, int32_t pod, const char* topology_key, const char* const* namespaces, int32_t n_namespaces) {
    POD_CHECK(e, pod);
    if (n_namespaces < 0 || (n_namespaces > 0 && !namespaces)) return CASIM_ERR_INVALID;
    Term t; t.topology_key = S(topology_key);
    if (n_namespaces == 0) { t.namespaces.push_back(e->specs[pod].ns); t.auto_ns = true; }  // getNamespacesFromPodAffinityTerm
    for (int i = 0; i < n_namespaces; ++i) t.namespaces.push_back(S(namespaces[i]));
    if (e->specs[pod].anti.empty() && e->specs[pod].aff.empty()) e->term_specs.push_back(pod);
    e->specs[pod].aff.push_back(t);
    return (int32_t)e->specs[pod].aff.size() - 1;
}
int32_t casim_enc_aff_term_add_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key, const char* op, const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (term < 0 || (size_t)term >= e->specs[pod].aff.size()) return CASIM_ERR_INVALID;
    if (n_values < 0 || (n_values > 0 && !values)) return CASIM_ERR_INVALID;
    e->specs[pod].aff[term].selector.push_back(make_req(key, op, values, n_values)); return CASIM_OK;
}
int32_t casim_enc_pod_add_spread_constraint(casim_encoder* e, int32_t pod, int32_t max_skew, const char* topology_key, int32_t min_domains) {
    POD_CHECK(e, pod);
    if (max_skew <= 0 || !topology_key || !*topology_key) return CASIM_ERR_INVALID;
    Spread sc; sc.max_skew = max_skew; sc.key = S(topology_key); sc.min_domains = min_domains > 0 ? min_domains : 1;  // nil => 1 (common.go:107)
    e->specs[pod].spread.push_back(sc);
    return (int32_t)e->specs[pod].spread.size() - 1;
}
int32_t casim_enc_spread_add_requirement(casim_encoder* e, int32_t pod, int32_t constraint, const char* key, const char* op,
                                         const char* const* values, int32_t n_values) {
    POD_CHECK(e, pod);
    if (constraint < 0 || (size_t)constraint >= e->specs[pod].spread.size()) return CASIM_ERR_INVALID;
    Requirement r; r.key = S(key); r.op = parse_req_op(op);
    if (r.op == kBadOp) return CASIM_ERR_INVALID;
    for (int i = 0; i < n_values; ++i) r.values.push_back(S(values[i]));
    e->specs[pod].spread[(size_t)constraint].selector.push_back(r);
    return CASIM_OK;
}
int32_t casim_enc_spread_set_taints_policy(casim_encoder* e, int32_t pod, int32_t constraint, int32_t honor) {
    POD_CHECK(e, pod);
    if (constraint < 0 || (size_t)constraint >= e->specs[pod].spread.size()) return CASIM_ERR_INVALID;
    e->specs[pod].spread[(size_t)constraint].taints_honor = honor != 0;
    return CASIM_OK;
}
int32_t casim_enc_spread_set_affinity_policy(casim_encoder* e, int32_t pod, int32_t constraint, int32_t honor) {
    POD_CHECK(e, pod);
    if (constraint < 0 || (size_t)constraint >= e->specs[pod].spread.size()) return CASIM_ERR_INVALID;
    e->specs[pod].spread[(size_t)constraint].affinity_honor = honor != 0;
    return CASIM_OK;
}
int32_t casim_enc_pod_set_fastpath_requests(casim_encoder* e, int32_t pod, double cpu, double mem) {
    POD_CHECK(e, pod); e->specs[pod].fp_cpu = cpu; e->specs[pod].fp_mem = mem; return CASIM_OK;
}
int32_t casim_enc_pod_mark_unsupported(casim_encoder* e, int32_t pod, const char* why) {
    POD_CHECK(e, pod); e->specs[pod].unsupported = true; e->specs[pod].why = S(why); return CASIM_OK;
}
int32_t casim_enc_add_peg(casim_encoder* e, int32_t pod_spec, int32_t count) {
    POD_CHECK(e, pod_spec);
    if (count < 0) return CASIM_ERR_INVALID;
    e->pegs.push_back(Peg{pod_spec, count});
    return (int32_t)e->pegs.size() - 1;
}
int32_t casim_enc_add_resource_pegs(casim_encoder* e, const char* namespace_, int32_t n, const int64_t* req, const int32_t* count,
                                    int32_t* ids_out) {
    ENC_CHECK(e);
    if (n < 0 || (n > 0 && (!req || !count))) return CASIM_ERR_INVALID;
    const int R = e->opt.n_res;
    const int32_t first = (int32_t)e->pegs.size();
    for (int32_t i = 0; i < n; ++i) {  // (no exact reserve(): repeated bulk calls would re-copy every spec each time)
        if (count[i] < 0) return CASIM_ERR_INVALID;
        PodSpec p;
        p.ns = S(namespace_);
        for (int r = 0; r < CASIM_MAX_RES; ++r) p.req[r] = r < R ? req[(size_t)i * (size_t)R + (size_t)r] : 0;
        // Containers[0].Resources.Requests as float64 (binpacking_estimator.go:451-458): milli * 10^-3, bytes
        p.fp_cpu = (double)p.req[0] * 1e-3; p.fp_mem = (double)p.req[1];
        e->specs.push_back(p);
        e->pegs.push_back(Peg{(int32_t)e->specs.size() - 1, count[i]});
        if (ids_out) ids_out[i] = first + i;
    }
    return first;
}
// ABI 11: the pods of a loop in one crossing (casim.h: casim_pod_columns).  Record for record what the per-pod calls build, in their order:
// add_pod_spec, labels, tolerations, nodeSelector pairs, fastpath requests, add_peg.
int32_t casim_enc_add_pods(casim_encoder* e, const casim_pod_columns* c, int32_t* peg_ids_out) {
    ENC_CHECK(e);
    if (!c || c->n_pods < 0 || c->n_strings < 0 || (c->n_strings > 0 && !c->strings)) return CASIM_ERR_INVALID;
    const int32_t n = c->n_pods, NSTR = c->n_strings;
    if (n == 0) return (int32_t)e->specs.size();
    if (!c->ns || !c->req) return CASIM_ERR_INVALID;
    // ---- validate everything first: nothing is added on an error
    auto str_ok = [&](int32_t k, bool may_be_null) { return (k >= 0 && k < NSTR) || (may_be_null && k == -1); };
    auto offsets_ok = [&](const int32_t* off) {
        if (!off) return true;
        if (off[0] != 0) return false;
        for (int32_t i = 0; i < n; ++i) if (off[i + 1] < off[i]) return false;
        return true;
    };
    if (!offsets_ok(c->label_off) || !offsets_ok(c->tol_off) || !offsets_ok(c->sel_off)) return CASIM_ERR_INVALID;
    for (int32_t i = 0; i < n; ++i) {
        if (!str_ok(c->ns[i], true)) return CASIM_ERR_INVALID;
        if (c->peg_count && c->peg_count[i] < -1) return CASIM_ERR_INVALID;
    }
    const int32_t NL = c->label_off ? c->label_off[n] : 0, NT = c->tol_off ? c->tol_off[n] : 0, NSEL = c->sel_off ? c->sel_off[n] : 0;
    if ((NL > 0 && (!c->label_key || !c->label_val)) || (NT > 0 && (!c->tol_key || !c->tol_op || !c->tol_value || !c->tol_effect)) ||
        (NSEL > 0 && (!c->sel_key || !c->sel_val))) return CASIM_ERR_INVALID;
    for (int32_t k = 0; k < NL; ++k) if (!str_ok(c->label_key[k], true) || !str_ok(c->label_val[k], true)) return CASIM_ERR_INVALID;
    for (int32_t k = 0; k < NT; ++k)
        if (!str_ok(c->tol_key[k], true) || !str_ok(c->tol_op[k], true) || !str_ok(c->tol_value[k], true) || !str_ok(c->tol_effect[k], true)) return CASIM_ERR_INVALID;
    for (int32_t k = 0; k < NSEL; ++k) if (!str_ok(c->sel_key[k], true) || !str_ok(c->sel_val[k], true)) return CASIM_ERR_INVALID;
    // ---- the string table once; slot NSTR is the empty string every -1 reads
    std::vector<std::string> strs((size_t)NSTR + 1);
    for (int32_t k = 0; k < NSTR; ++k) strs[(size_t)k] = S(c->strings[k]);
    auto str = [&](int32_t k) -> const std::string& { return strs[(size_t)(k < 0 ? NSTR : k)]; };
    // tolerations: equal index quadruples are equal tolerations — found again without building a signature (a miss goes through the
    // encoder-wide dictionary, where the same toleration may already sit under other indices or from the per-pod calls)
    struct Quad { int32_t k, o, v, f; bool operator==(const Quad& x) const { return k == x.k && o == x.o && v == x.v && f == x.f; } };
    struct QuadHash { size_t operator()(const Quad& q) const {
        uint64_t h = ((uint64_t)(uint32_t)q.k << 32 | (uint32_t)q.v) * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)(uint32_t)q.o << 32 | (uint32_t)q.f) * 0xC2B2AE3D27D4EB4Full; return (size_t)(h ^ (h >> 29)); } };
    std::unordered_map<Quad, int32_t, QuadHash> tol_of_quad;
    const int32_t first = (int32_t)e->specs.size();
    const int R = e->opt.n_res;
    for (int32_t i = 0; i < n; ++i) {
        e->specs.emplace_back();
        PodSpec& p = e->specs.back();
        p.ns = str(c->ns[i]);
        for (int r = 0; r < CASIM_MAX_RES; ++r) p.req[r] = r < R ? c->req[(size_t)i * (size_t)R + (size_t)r] : 0;
        if (c->label_off) {
            p.labels.v.reserve((size_t)(c->label_off[i + 1] - c->label_off[i]));
            for (int32_t k = c->label_off[i]; k < c->label_off[i + 1]; ++k) p.labels[str(c->label_key[k])] = str(c->label_val[k]);
        }
        if (c->tol_off) {
            p.tolerations.reserve((size_t)(c->tol_off[i + 1] - c->tol_off[i]));
            for (int32_t k = c->tol_off[i]; k < c->tol_off[i + 1]; ++k) {
                const Quad q{c->tol_key[k], c->tol_op[k], c->tol_value[k], c->tol_effect[k]};
                auto it = tol_of_quad.find(q);
                if (it == tol_of_quad.end())
                    it = tol_of_quad.emplace(q, e->intern_toleration(str(q.k).c_str(), parse_tol_op(q.o < 0 ? nullptr : str(q.o).c_str()), str(q.v).c_str(), str(q.f).c_str())).first;
                p.tolerations.push_back(it->second);
            }
        }
        if (c->sel_off)
            for (int32_t k = c->sel_off[i]; k < c->sel_off[i + 1]; ++k) p.node_selector.push_back({str(c->sel_key[k]), str(c->sel_val[k])});
        if (c->fastpath_req) { p.fp_cpu = c->fastpath_req[2 * (size_t)i]; p.fp_mem = c->fastpath_req[2 * (size_t)i + 1]; }
        else { p.fp_cpu = (double)p.req[0] * 1e-3; p.fp_mem = (double)p.req[1]; }
        int32_t peg = -1;
        if (c->peg_count && c->peg_count[i] >= 0) { peg = (int32_t)e->pegs.size(); e->pegs.push_back(Peg{first + i, c->peg_count[i]}); }
        if (peg_ids_out) peg_ids_out[i] = peg;
    }
    return first;
}
int32_t casim_enc_add_existing_pod(casim_encoder* e, int32_t pod_spec, const char* const* keys, const char* const* values, int32_t n) {
    POD_CHECK(e, pod_spec);
    if (n < 0 || (n > 0 && (!keys || !values))) return CASIM_ERR_INVALID;
    ExistingPod x; x.spec = pod_spec;
    for (int i = 0; i < n; ++i) x.node_labels[S(keys[i])] = S(values[i]);
    e->existing.push_back(x);
    return CASIM_OK;
}

int32_t casim_enc_finalize(casim_encoder* e) {
    ENC_OPEN(e);   // (also the full fallback of an update session: casim_enc_refinalize said CASIM_ENC_NEEDS_FULL)
    e->fs.valid = false;
    EncStageTimer stage;
    resolve_named_allocatable(e);
    const int R = e->out_res = e->lanes_now();   // (positional lanes + the named ones: casim_enc_lane)
    const size_t G = e->pegs.size(), NG = e->groups.size(), NS = e->specs.size();
    const bool cmp_ops = e->opt.enable_taint_comparison_ops != 0;
    for (auto& g : e->groups)
        if (g.has_pegs) for (int32_t pg : g.pegs) if (pg < 0 || (size_t)pg >= G) return CASIM_ERR_INVALID;

    // ---- (0) namespaceSelector of anti-affinity terms -> explicit namespace sets ----------------------------
    // The plugin resolves the INCOMING pod's selectors through the namespace lister (plugin.go:144-157) and evaluates the
    // selectors of pods already on nodes against the incoming pod's namespace labels, an unlisted namespace counting as
    // unlabelled (plugin.go:161-169).  The two only differ for a namespace the lister does not know; the conflict bits
    // below are symmetric in who arrives first, so that corner is delegated (every PEG flagged) instead of guessed.
    {
        // (only specs with terms are visited — e->term_specs; "every namespace listed" is only asked when a selector exists)
        bool any_sel = false;
        for (int32_t sp : e->term_specs)
            for (auto& t : e->specs[(size_t)sp].anti) if (t.has_ns_sel && !t.ns_sel.empty()) any_sel = true;
        for (int32_t sp : e->term_specs)
            for (auto& t : e->specs[(size_t)sp].anti) {
                if (!t.has_ns_sel || t.ns_resolved) continue;
                t.ns_resolved = true;   // (a second finalize of an update session must not append the namespaces again)
                if (t.ns_sel.empty()) { t.all_ns = true; continue; }
                for (auto& kv : e->namespaces) if (selector_matches(t.ns_sel, kv.second)) t.namespaces.push_back(kv.first);
            }
        if (any_sel) {
            bool all_known = true;
            for (auto& p : e->specs) if (!e->namespaces.count(p.ns)) { all_known = false; break; }
            if (!all_known)
                for (auto& p : e->specs) { p.unsupported = true; p.why = "namespaceSelector next to a pod whose namespace is not listed"; }
        }
        // required AFFINITY terms are only ever the incoming pod's (existing pods' affinity does not constrain it): no symmetry to keep,
        // an unlisted namespace simply is not selected by a non-empty selector
        for (int32_t sp : e->term_specs)
            for (auto& t : e->specs[(size_t)sp].aff) {
                if (!t.has_ns_sel || t.ns_resolved) continue;
                t.ns_resolved = true;
                if (t.ns_sel.empty()) { t.all_ns = true; continue; }
                for (auto& kv : e->namespaces) if (selector_matches(t.ns_sel, kv.second)) t.namespaces.push_back(kv.first);
            }
    }

    stage.mark("namespaces");
    // ---- dictionaries ------------------------------------------------------------------
    // (1) taints that reject scheduling (NoSchedule / NoExecute)
    std::map<Taint, int> taint_id;
    for (auto& g : e->groups)
        for (auto& t : g.taints)
            if ((t.effect == "NoSchedule" || t.effect == "NoExecute") && !taint_id.count(t)) { const int id = (int)taint_id.size(); taint_id[t] = id; }
    e->Wt = ((int)taint_id.size() + 63) / 64;
    stage.mark("taints");
    // (2) label requirements used by some PEG spec (nodeSelector pair == In{value})
    std::unordered_map<std::string, int> lreq_id;   // (ids are handed out in order of discovery; the map only finds them again)
    std::vector<Requirement> lreqs;
    std::vector<std::vector<int>> spec_lreqs(NS);
    std::vector<bool> spec_used(NS, false);
    for (auto& pg : e->pegs) spec_used[(size_t)pg.spec] = true;
    std::string lreq_sig;
    for (size_t s = 0; s < NS; ++s) {
        if (!spec_used[s]) continue;
        PodSpec& p = e->specs[s];
        // the spec's requirements in the order nodeAffinity's one term, nodeSelector pairs, the ORed term list; `make` builds the
        // dictionary entry only when the signature is new (a nodeSelector pair is In{value}: its signature is written without one)
        auto visit = [&](const std::string& key, const std::string& sig, auto make) {
            // every simulated node gets its own hostname label (node_info_utils.go:130): not a template property
            // (per-node consumers pass real nodes: there the label is an ordinary one)
            if (key == kHostname && !e->opt.explicit_self_exclusion) { p.unsupported = true; p.why = "node selector on kubernetes.io/hostname"; return; }
            auto it = lreq_id.find(sig);
            int id;
            if (it == lreq_id.end()) { id = (int)lreqs.size(); lreq_id.emplace(sig, id); lreqs.push_back(make()); } else id = it->second;
            spec_lreqs[s].push_back(id);
        };
        for (auto& r : p.node_affinity) visit(r.key, req_signature(r), [&] { return r; });
        for (auto& kv : p.node_selector) {
            lreq_sig.assign(1, (char)('0' + (int)kIn)); lreq_sig.push_back('\x1f'); lreq_sig.append(kv.first); lreq_sig.push_back('\x1f'); lreq_sig.append(kv.second);
            visit(kv.first, lreq_sig, [&] { Requirement r; r.key = kv.first; r.op = kIn; r.values = {kv.second}; return r; });
        }
        if (p.has_node_terms) {
            // the ORed term list is ONE dictionary entry: a node's bit = LazyErrorNodeSelector.Match of that node, evaluated
            // below with the node's labels and name.  Template mode: the nodes of an estimate get fresh names and hostname
            // labels (node_info_utils.go:93-137), so a term reading either is not a template property.
            bool per_node = false;
            for (auto& t : p.node_terms) { if (!t.fields.empty()) per_node = true; for (auto& x : t.exprs) if (x.key == kHostname) per_node = true; }
            if (per_node && !e->opt.explicit_self_exclusion) { p.unsupported = true; p.why = "node affinity term on metadata.name / kubernetes.io/hostname"; }
            else { Requirement r; r.op = kNodeTerms; r.terms = p.node_terms; visit(r.key, req_signature(r), [&] { return r; }); }
        }
    }
    e->Wl = ((int)lreqs.size() + 63) / 64;

    stage.mark("label_reqs");
    // ---- content classes of the running pods ----------------------------------------------------
    // At cluster scale a shim without a spec cache hands over one spec record per RUNNING pod (150 000 records of ~1 500 shapes).  What
    // finalize asks of a running pod — does a PEG's term / a rule's selector match it, does it hold a host port, does it carry terms
    // of its own — looks at its namespace, labels, anti-affinity terms and host ports only.  So the distinct running specs are put
    // into classes of equal (namespace, labels) once (hash, then equality against the class representative: exact); a spec with
    // anti-affinity terms or host ports is a class of its own ("special").  The one pass that touches every (cold) spec record runs on up to
    // four threads, each interning its slice; the slices' classes are merged in spec order, so class ids do not depend on the threads.
    struct RunningClasses {
        std::vector<uint8_t> seen;        // [specs] preloaded on some group
        std::vector<int32_t> specs;       // distinct running specs, ascending
        std::vector<int32_t> cls;         // [specs] class of a running spec, -1 otherwise
        std::vector<int32_t> rep;         // [classes] representative spec (the first of the class in spec order)
        std::vector<uint8_t> is_special;  // [classes] carries anti-affinity terms or host ports: a class of its own
        std::vector<int32_t> special;     // the special classes, ascending
    } rc;
    {
        rc.seen.assign(NS, 0);
        for (auto& g : e->groups) for (int32_t s : g.preloaded) rc.seen[(size_t)s] = 1;
        for (size_t s = 0; s < NS; ++s) if (rc.seen[s]) rc.specs.push_back((int32_t)s);
        rc.cls.assign(NS, -1);
        const size_t n = rc.specs.size();
        const int T = enc_threads(n, 8192);
        struct Local { std::vector<int32_t> rep; std::vector<uint64_t> hash; std::vector<int32_t> cls; };   // cls: per spec of the slice
        std::vector<Local> loc((size_t)T);
        // (eight bytes per multiply; the length goes in, so "ab" + "c" != "a" + "bc")
        auto mix = [](uint64_t h, const std::string& x) {
            const char* d = x.data();
            size_t len = x.size();
            h = (h ^ (uint64_t)len) * 0x9E3779B97F4A7C15ull;
            while (len >= 8) { uint64_t w; memcpy(&w, d, 8); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; d += 8; len -= 8; }
            if (len) { uint64_t w = 0; memcpy(&w, d, len); h = (h ^ w) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; }
            return h;
        };
        auto same = [&](const PodSpec& a, const PodSpec& b) { return a.ns == b.ns && a.labels.v == b.labels.v; };
        auto work = [&](int t) {
            Local& L = loc[(size_t)t];
            const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
            std::vector<int32_t> table(4096, 0);
            size_t filled = 0;
            L.cls.resize(hi - lo);
            for (size_t k = lo; k < hi; ++k) {
                // (the records are cold and each is two dependent misses — the record, then its label block: fetched 48 / 24 specs ahead)
                if (k + 48 < hi) { const PodSpec& f = e->specs[(size_t)rc.specs[k + 48]]; __builtin_prefetch(&f.ns); __builtin_prefetch(&f.labels); __builtin_prefetch(&f.anti); }
                if (k + 24 < hi) { const PodSpec& f = e->specs[(size_t)rc.specs[k + 24]]; if (!f.labels.v.empty()) __builtin_prefetch(f.labels.v.data()); }
                const PodSpec& q = e->specs[(size_t)rc.specs[k]];
                if (!q.anti.empty() || !q.ports.empty()) { L.cls[k - lo] = (int32_t)L.rep.size(); L.rep.push_back(rc.specs[k]); L.hash.push_back(0); continue; }
                uint64_t h = mix(1469598103934665603ull, q.ns);
                for (auto& kv : q.labels.v) { h = mix(h, kv.first); h = mix(h, kv.second); }
                h |= 1ull;   // (0 marks the special ones)
                // open addressing on the hash (a slot = local class + 1): the few thousand shapes of a slice stay in L1 / L2
                size_t at = (size_t)(h >> 17) & (table.size() - 1);
                int32_t c = -1;
                for (;;) {
                    const int32_t x = table[at] - 1;
                    if (x < 0) break;
                    if (L.hash[(size_t)x] == h && same(e->specs[(size_t)L.rep[(size_t)x]], q)) { c = x; break; }
                    at = (at + 1) & (table.size() - 1);
                }
                if (c < 0) {
                    c = (int32_t)L.rep.size(); L.rep.push_back(rc.specs[k]); L.hash.push_back(h); table[at] = c + 1;
                    if (++filled * 2 > table.size()) {   // keep it at most half full
                        std::vector<int32_t> bigger(table.size() * 2, 0);
                        for (size_t x = 0; x < L.rep.size(); ++x) {
                            if (L.hash[x] == 0) continue;
                            size_t b = (size_t)(L.hash[x] >> 17) & (bigger.size() - 1);
                            while (bigger[b]) b = (b + 1) & (bigger.size() - 1);
                            bigger[b] = (int32_t)x + 1;
                        }
                        table.swap(bigger);
                    }
                }
                L.cls[k - lo] = c;
            }
        };
        enc_par_for(n, T, [&](size_t, size_t, int t) { work(t); });   // (work(t) takes the same slice bounds)
        // merge: slices in order, local classes in order of first appearance == ascending representative
        std::unordered_map<uint64_t, std::vector<int32_t>> by_hash;
        for (int t = 0; t < T; ++t) {
            Local& L = loc[(size_t)t];
            std::vector<int32_t> global(L.rep.size(), -1);
            for (size_t x = 0; x < L.rep.size(); ++x) {
                const PodSpec& q = e->specs[(size_t)L.rep[x]];
                int32_t c = -1;
                if (L.hash[x] != 0) {
                    auto& cands = by_hash[L.hash[x]];
                    for (int32_t y : cands) if (same(e->specs[(size_t)rc.rep[(size_t)y]], q)) { c = y; break; }
                    if (c < 0) { c = (int32_t)rc.rep.size(); rc.rep.push_back(L.rep[x]); rc.is_special.push_back(0); cands.push_back(c); }
                } else { c = (int32_t)rc.rep.size(); rc.rep.push_back(L.rep[x]); rc.is_special.push_back(1); rc.special.push_back(c); }
                global[x] = c;
            }
            const size_t lo = n * (size_t)t / (size_t)T;
            for (size_t k = 0; k < L.cls.size(); ++k) rc.cls[(size_t)rc.specs[lo + k]] = global[(size_t)L.cls[k]];
        }
    }
    stage.mark("running_classes");
    // (3) node-local exclusion bits: host ports and hostname anti-affinity across DIFFERENT units.
    // Units: every PEG, plus every spec preloaded on some template.
    BitAlloc xbits;
    std::vector<uint32_t> pflags(G, 0);
    // ports
    std::map<Port, int> port_bit;                       // sanitized used port -> bit (only when shared)
    {
        std::map<Port, std::set<int>> users;            // port -> units using it (unit = PEG id, or -1-spec for preloaded)
        for (size_t i = 0; i < G; ++i)
            for (auto& p : e->specs[(size_t)e->pegs[i].spec].ports) if (p.port > 0) users[sanitize(p)].insert((int)i);
        for (int32_t c : rc.special) {   // (running pods with host ports are content classes of their own)
            const int32_t s = rc.rep[(size_t)c];
            for (auto& p : e->specs[(size_t)s].ports) if (p.port > 0) users[sanitize(p)].insert(-1 - s);
        }
        // a used port needs a bit when some OTHER unit wants a conflicting port
        for (auto& a : users) {
            bool shared = false;
            for (auto& b : users) {
                if (!ports_conflict(b.first, a.first)) continue;
                for (int ub : b.second) for (int ua : a.second) if (ua != ub) shared = true;
                if (shared) break;
            }
            // explicit_self_exclusion (per-pod consumers such as casim_try_schedule_pods): every used port
            // gets a node bit, so that "this class already sits on the node" is remembered by the node
            if (shared || e->opt.explicit_self_exclusion) port_bit[a.first] = xbits.next();
        }
    }
    // hostname anti-affinity between different units
    std::vector<int> peg_occ_bit(G, -1);                 // occupancy bit of a PEG (set when someone else conflicts with it)
    std::map<int32_t, int> pre_occ_bit;                  // preloaded spec -> bit
    std::vector<std::vector<int>> peg_blockers(G);       // bits that block PEG i at node level
    auto host_conflict = [&](const PodSpec& a, const PodSpec& b) {
        for (auto& t : a.anti) if (t.topology_key == kHostname && term_matches(t, b)) return true;
        for (auto& t : b.anti) if (t.topology_key == kHostname && term_matches(t, a)) return true;
        return false;
    };
    // Per-node mode with NO node carrying kubernetes.io/hostname (the reference's BuildTestNode never sets it: BenchmarkRunFiltersUntilPassingNode's
    // 5 001 nodes, most of its unit tests): a required anti-affinity term counts and blocks through the topology PAIRS of the nodes' labels
    // (V/kubernetes/pkg/scheduler/framework/plugins/interpodaffinity/filtering.go: updateWithAntiAffinityTerms skips a node without the key,
    // satisfyExistingPodsAntiAffinity walks the node's labels, satisfyPodAntiAffinity asks for the term's key on the node) — without the label on
    // any node the hostname terms are inert as a whole, and no node bit is needed for them.  Some nodes with, some without: delegated (below).
    bool hostname_inert = e->opt.explicit_self_exclusion && NG > 0;
    if (hostname_inert) for (auto& g : e->groups) if (g.labels.count(kHostname)) { hostname_inert = false; break; }
    bool any_hostname_terms = false;
    {
        std::vector<size_t> with_terms;  // PEGs whose spec has hostname terms
        for (size_t i = 0; i < G; ++i)
            for (auto& t : e->specs[(size_t)e->pegs[i].spec].anti) if (t.topology_key == kHostname) { with_terms.push_back(i); break; }
        for (int32_t c : rc.special) for (auto& t : e->specs[(size_t)rc.rep[(size_t)c]].anti) if (t.topology_key == kHostname) any_hostname_terms = true;
        if (!with_terms.empty()) any_hostname_terms = true;
        if (hostname_inert) with_terms.clear();
        e->fs.with_terms = with_terms;
        // Which PEGs can a term match at all?  A selector with an In requirement (matchLabels is one) only matches pods that CARRY one of
        // its (key, value) pairs, so the candidates of such a term come from an index pair -> PEGs (ascending) instead of a walk over every
        // PEG (C4: 200 PEGs with terms x 400 PEGs x a string compare = 1.7 of finalize's 2.2 ms).  A term without an In requirement keeps
        // the full walk.  Candidates are visited in ascending PEG order, as before: the bits are handed out in order of discovery.
        std::unordered_map<std::string, std::vector<int32_t>> pegs_of_pair;
        std::string pair_sig;
        auto pair_key = [&](const std::string& k, const std::string& v) -> const std::string& {
            const uint32_t len = (uint32_t)k.size();
            pair_sig.assign((const char*)&len, sizeof len); pair_sig.append(k); pair_sig.append(v);
            return pair_sig;
        };
        if (!with_terms.empty())
            for (size_t j = 0; j < G; ++j)
                for (auto& kv : e->specs[(size_t)e->pegs[j].spec].labels) pegs_of_pair[pair_key(kv.first, kv.second)].push_back((int32_t)j);
        std::vector<int32_t> cand;
        for (size_t i : with_terms) {
            const PodSpec& a = e->specs[(size_t)e->pegs[i].spec];
            bool indexed = true;
            cand.clear();
            for (auto& t : a.anti) {
                if (t.topology_key != kHostname) continue;
                const Requirement* in = nullptr;
                for (auto& r : t.selector) if (r.op == kIn) { in = &r; break; }
                if (!in) { indexed = false; break; }
                for (auto& v : in->values) {
                    auto it = pegs_of_pair.find(pair_key(in->key, v));
                    if (it != pegs_of_pair.end()) cand.insert(cand.end(), it->second.begin(), it->second.end());
                }
            }
            if (indexed) { std::sort(cand.begin(), cand.end()); cand.erase(std::unique(cand.begin(), cand.end()), cand.end()); }
            const size_t n_visit = indexed ? cand.size() : G;
            for (size_t q = 0; q < n_visit; ++q) {
                const size_t j = indexed ? (size_t)cand[q] : q;
                const PodSpec& b = e->specs[(size_t)e->pegs[j].spec];
                bool hit = false;
                for (auto& t : a.anti) if (t.topology_key == kHostname && term_matches(t, b)) { hit = true; break; }
                if (!hit) continue;
                if (i == j) {
                    pflags[i] |= CASIM_PEG_SELF_EXCL_NODE;
                    if (e->opt.explicit_self_exclusion) {  // self-conflict as a real node bit (see above)
                        if (peg_occ_bit[i] < 0) peg_occ_bit[i] = xbits.next();
                        peg_blockers[i].push_back(peg_occ_bit[i]);
                    }
                    continue;
                }
                if (peg_occ_bit[i] < 0) peg_occ_bit[i] = xbits.next();
                if (peg_occ_bit[j] < 0) peg_occ_bit[j] = xbits.next();
                peg_blockers[i].push_back(peg_occ_bit[j]);
                peg_blockers[j].push_back(peg_occ_bit[i]);
            }
        }
        // distinct specs preloaded on some template, ascending (a flag per spec: at cluster scale every running pod has its own)
        const std::vector<int32_t>& pre_specs = rc.specs;
        e->fs.running = rc.seen;
        // A running pod without terms of its own conflicts with a PEG only through the PEG's terms, which look at its namespace and
        // labels: one verdict per (content class, PEG with terms), not one per running pod (1.2 M term evaluations over 150 000 cold
        // spec records were 35 ms of a full finalize).  Every conflicting SPEC still gets its own bit, in ascending spec order.
        std::vector<int8_t> verdict(with_terms.empty() ? 0 : rc.rep.size() * with_terms.size(), (int8_t)-1);
        for (int32_t s : pre_specs) {
            if (hostname_inert) break;   // (what follows finds hostname conflicts between running pods and PEGs)
            const int32_t c = rc.cls[(size_t)s];
            auto hit = [&](size_t i) {
                if (!pre_occ_bit.count(s)) pre_occ_bit[s] = xbits.next();
                peg_blockers[i].push_back(pre_occ_bit[s]);
            };
            if (rc.is_special[(size_t)c]) {   // (s is the class: it may carry hostname terms — then it can conflict with a term-less PEG)
                bool s_has_terms = false;
                for (auto& t : e->specs[(size_t)s].anti) if (t.topology_key == kHostname) s_has_terms = true;
                auto visit = [&](size_t i) { if (host_conflict(e->specs[(size_t)s], e->specs[(size_t)e->pegs[i].spec])) hit(i); };
                if (s_has_terms) for (size_t i = 0; i < G; ++i) visit(i);
                else for (size_t i : with_terms) visit(i);
                continue;
            }
            for (size_t w = 0; w < with_terms.size(); ++w) {
                int8_t& v = verdict[(size_t)c * with_terms.size() + w];
                if (v < 0) v = host_conflict(e->specs[(size_t)rc.rep[(size_t)c]], e->specs[(size_t)e->pegs[with_terms[w]].spec]) ? 1 : 0;
                if (v) hit(with_terms[w]);
            }
        }
    }
    e->Wx = xbits.words();
    // Hostname terms assume that every node carries its own kubernetes.io/hostname value (true for the
    // template clones of an Estimate, node_info_utils.go:130).  Per-node consumers pass real nodes: when one of
    // them lacks the label its pods are in no hostname domain at all, which the node bits cannot express.
    if (e->opt.explicit_self_exclusion) {
        bool all_named = true;
        for (auto& g : e->groups) if (!g.labels.count(kHostname)) all_named = false;
        if (!all_named)
            for (size_t i = 0; i < G; ++i)
                if (!peg_blockers[i].empty() || peg_occ_bit[i] >= 0) {  // so far both hold hostname bits only
                    PodSpec& p = e->specs[(size_t)e->pegs[i].spec];
                    p.unsupported = true; p.why = "hostname anti-affinity with a node that has no kubernetes.io/hostname label";
                }
    }

    stage.mark("node_bits");
    // (4) group-wide exclusion bits: anti-affinity on non-hostname topology keys.  All nodes of a group
    // clone one template, so a domain == the whole group when the template carries the key.
    BitAlloc zbits;
    std::map<std::pair<int, std::string>, int> occ_z;     // (PEG, topology key) -> occupancy bit
    std::vector<std::vector<int>> z_block(G), z_mark(G);
    std::map<int, std::string> zbit_key;                  // bit -> topology key ("" = always valid)
    std::vector<int> static_zbit(G, -1);                  // per-PEG "blocked by the existing cluster" bit
    const bool per_node = e->opt.explicit_self_exclusion != 0;   // real nodes: domains are handled by the domain rules below
    if (!per_node) {
        auto occ = [&](int pg, const std::string& tk) {
            auto k = std::make_pair(pg, tk);
            auto it = occ_z.find(k);
            if (it != occ_z.end()) return it->second;
            const int b = zbits.next(); occ_z[k] = b; zbit_key[b] = tk; z_mark[(size_t)pg].push_back(b);
            return b;
        };
        for (size_t i = 0; i < G; ++i) {
            const PodSpec& a = e->specs[(size_t)e->pegs[i].spec];
            for (auto& t : a.anti) {
                if (t.topology_key == kHostname) continue;
                for (size_t j = 0; j < G; ++j) {
                    if (!term_matches(t, e->specs[(size_t)e->pegs[j].spec])) continue;
                    // i must not join a domain holding j; j must not join a domain holding i
                    z_block[i].push_back(occ((int)j, t.topology_key));
                    z_block[j].push_back(occ((int)i, t.topology_key));
                }
            }
        }
    }
    // existing cluster pods and pods preloaded on a template: static (PEG, group) blocks through
    // non-hostname topology keys.  Sparse: only PEGs / pods that carry such terms can interact.
    std::vector<std::vector<uint32_t>> existing_block(G);   // PEG -> groups where it is blocked
    if (!per_node) {
        auto has_zone_terms = [&](const PodSpec& p) {
            for (auto& t : p.anti) if (t.topology_key != kHostname) return true;
            return false;
        };
        bool others_have_terms = false;
        for (auto& x : e->existing) others_have_terms = others_have_terms || has_zone_terms(e->specs[(size_t)x.spec]);
        for (auto& g : e->groups) for (int32_t s : g.preloaded) others_have_terms = others_have_terms || has_zone_terms(e->specs[(size_t)s]);
        const bool any_others = !e->existing.empty();
        bool any_preloaded = false;
        for (auto& g : e->groups) any_preloaded = any_preloaded || !g.preloaded.empty();
        for (size_t i = 0; i < G && (any_others || any_preloaded); ++i) {
            const PodSpec& a = e->specs[(size_t)e->pegs[i].spec];
            if (!others_have_terms && !has_zone_terms(a)) continue;
            for (size_t gi = 0; gi < NG; ++gi) {
                const Group& g = e->groups[gi];
                bool blk = false;
                for (auto& x : e->existing) {
                    const PodSpec& b = e->specs[(size_t)x.spec];
                    auto same_domain = [&](const std::string& tk) {
                        if (tk == kHostname) return false;  // new nodes never share a hostname with an existing node
                        auto a1 = g.labels.find(tk); auto b1 = x.node_labels.find(tk);
                        return a1 != g.labels.end() && b1 != x.node_labels.end() && a1->second == b1->second;
                    };
                    for (auto& t : a.anti) if (same_domain(t.topology_key) && term_matches(t, b)) blk = true;
                    for (auto& t : b.anti) if (same_domain(t.topology_key) && term_matches(t, a)) blk = true;
                }
                for (int32_t s2 : g.preloaded) {  // preloaded pods share every non-hostname domain with the template's clones
                    const PodSpec& b = e->specs[(size_t)s2];
                    for (auto& t : a.anti) if (t.topology_key != kHostname && g.labels.count(t.topology_key) && term_matches(t, b)) blk = true;
                    for (auto& t : b.anti) if (t.topology_key != kHostname && g.labels.count(t.topology_key) && term_matches(t, a)) blk = true;
                }
                if (blk) existing_block[i].push_back((uint32_t)gi);
            }
            if (!existing_block[i].empty()) { static_zbit[i] = zbits.next(); zbit_key[static_zbit[i]] = ""; z_block[i].push_back(static_zbit[i]); }
        }
    }
    stage.mark("zone_bits");
    // (4a) template mode: required pod affinity.  satisfyPodAffinity (interpodaffinity/filtering.go:382-409) asks, per term, for a
    // pod matching ALL terms of the incoming pod in the node's domain of the term's key — or, when no such pod exists ANYWHERE and the
    // pod matches its own terms, lets the first pod of the series through (:396-407).  Counts only grow while an Estimate runs, and all
    // nodes of a group share the template's non-hostname domains, so for a (PEG, group) pair:
    //   * a term key the template lacks                                            -> the PEG never fits the group: the static
    //     (PEG, group) block the anti-affinity against existing pods already uses;
    //   * every term satisfied by the existing cluster / the template's preloaded pods (which sit on every clone: they share ALL
    //     its domains, the hostname included)                                      -> the affinity is a no-op for the whole Estimate;
    //   * the first-pod exception holds at snapshot time, every key a non-hostname key -> a no-op as well: whoever places the first
    //     matching pod (the PEG itself or a partner of the batch) satisfies every later one;
    //   * neither, non-hostname keys: the PEG waits for a PARTNER OF THE BATCH (a PEG matching all of its terms) to place a pod in
    //     the group — a group bit of NEED polarity (casim_pegs.zone_polarity): the PEG is forbidden while it is clear, every
    //     partner marks it, it starts set in the groups of the two cases above.  (SchedulablePodGroups never lists such a PEG: its
    //     sample pod fails on a fresh template node — K_feas says the same through the same bit; a caller's own list may.)
    //     Without a partner in the batch: never;
    //   * a hostname term that is not satisfied by the template's preloaded pods   -> the partner has to sit on the SAME node: a NODE
    //     bit of NEED polarity (casim_pegs.excl_polarity, ABI 8) — the PEG fits a node only while the bit is set there, every partner of
    //     the batch marks it, fresh nodes of a group whose template carries a partner start with it.  When the PEG matches its own terms
    //     and no matching pod exists in the cluster it marks the bit ITSELF: the packer then lets its first pod in by the exception while
    //     no node of the estimate carries the bit and makes the rest of the PEG join that pod's node (casim_pack.h, the series).
    //     Nobody in the batch who could be the partner and no exception: never.  (Round 3 sent this case to casim_estimate_on_cluster.)
    std::vector<uint8_t> aff_static(G, 0);
    std::vector<int> host_need_bit(G, -1);               // node bit of NEED polarity that blocks PEG i
    std::vector<std::vector<int>> host_marks(G);         // node bits of NEED polarity PEG j sets (it is a partner of their owners)
    std::vector<std::pair<uint32_t, int>> excl_preset;   // (group, bit): every fresh node of the group carries it
    std::vector<int> xneed_bits;
    std::vector<int> need_bits;   // group bits of NEED polarity
    std::vector<std::pair<uint32_t, int>> zone_preset;   // (group, bit): set from the start
    if (!per_node) {
        auto matches_all = [&](const PodSpec& owner, const PodSpec& q) {
            for (auto& t : owner.aff) if (!term_matches(t, q)) return false;
            return true;
        };
        for (size_t i = 0; i < G; ++i) {
            const PodSpec& a = e->specs[(size_t)e->pegs[i].spec];
            if (a.aff.empty()) continue;
            bool host = false;
            for (auto& t : a.aff) host = host || t.topology_key == kHostname;
            const bool self = matches_all(a, a);
            std::vector<size_t> partners;   // PEGs of the batch whose pods match all terms
            for (size_t j = 0; j < G; ++j) if (matches_all(a, e->specs[(size_t)e->pegs[j].spec])) partners.push_back(j);
            // len(affinityCounts) == 0 as far as the existing cluster goes: a matching pod on a node that carries one of the keys
            bool anywhere_cluster = false;
            for (auto& x : e->existing) {
                if (anywhere_cluster) break;
                if (!matches_all(a, e->specs[(size_t)x.spec])) continue;
                for (auto& t : a.aff) if (x.node_labels.count(t.topology_key)) { anywhere_cluster = true; break; }
            }
            std::vector<uint32_t> never, waits, sat_groups;
            bool dynamic = false, other_partner = false;
            for (size_t j : partners) other_partner = other_partner || j != i;
            for (size_t gi = 0; gi < NG; ++gi) {
                const Group& g = e->groups[gi];
                bool keys = true;
                for (auto& t : a.aff) keys = keys && (t.topology_key == kHostname || g.labels.count(t.topology_key) != 0);   // (clones always carry their own hostname)
                if (!keys) { never.push_back((uint32_t)gi); continue; }
                bool pre = false;
                for (int32_t s2 : g.preloaded) if (!pre && matches_all(a, e->specs[(size_t)s2])) pre = true;
                bool sat = true;
                for (auto& t : a.aff) {
                    bool found = pre;
                    if (!found && t.topology_key != kHostname) {   // (an existing node never shares a hostname with a new one)
                        auto a1 = g.labels.find(t.topology_key);
                        for (auto& x : e->existing) {
                            if (found) break;
                            auto b1 = x.node_labels.find(t.topology_key);
                            if (b1 != x.node_labels.end() && b1->second == a1->second && matches_all(a, e->specs[(size_t)x.spec])) found = true;
                        }
                    }
                    if (!found) { sat = false; break; }
                }
                if (sat) { sat_groups.push_back((uint32_t)gi); continue; }  // no-op for the whole Estimate
                const bool exception = self && !anywhere_cluster && !pre;
                // (a PEG that is its own only partner cannot start a series without the exception: nobody ever sets its bit)
                if (host) { if (exception || other_partner) dynamic = true; else never.push_back((uint32_t)gi); continue; }
                if (exception) continue;                                    // no-op: the first pod passes, the rest finds it
                if (partners.empty()) never.push_back((uint32_t)gi); else waits.push_back((uint32_t)gi);
            }
            if (dynamic) {
                const int b = xbits.next();
                host_need_bit[i] = b; xneed_bits.push_back(b);
                const bool starts_itself = self && !anywhere_cluster;   // the first-pod exception can hold: the PEG marks its own bit
                for (size_t j : partners) if (j != i || starts_itself) host_marks[j].push_back(b);
                for (uint32_t gi : sat_groups) excl_preset.emplace_back(gi, b);
            }
            aff_static[i] = 1;
            for (uint32_t gi : never) if (std::find(existing_block[i].begin(), existing_block[i].end(), gi) == existing_block[i].end()) existing_block[i].push_back(gi);
            if (!existing_block[i].empty() && static_zbit[i] < 0) { static_zbit[i] = zbits.next(); zbit_key[static_zbit[i]] = ""; z_block[i].push_back(static_zbit[i]); }
            if (!waits.empty()) {
                const int b = zbits.next();
                zbit_key[b] = ""; need_bits.push_back(b);
                z_block[i].push_back(b);
                for (size_t j : partners) z_mark[j].push_back(b);
                std::vector<uint8_t> w8(NG, 0);
                for (uint32_t gi : waits) w8[gi] = 1;
                for (size_t gi = 0; gi < NG; ++gi) if (!w8[gi]) zone_preset.emplace_back((uint32_t)gi, b);
            }
        }
    }
    e->Wz = zbits.words();
    e->Wx = xbits.words();   // ((4a) may have added node bits of NEED polarity)

    stage.mark("affinity_static");
    // (4b) per-node mode: domain rules (include/casim.h, casim_domain_rules) for PodTopologySpread and for required
    // anti-affinity on non-hostname keys.  Template mode (an Estimate): spread constraints are outside the subset.
    e->dr = decltype(e->dr)();
    e->fs.keys.clear(); e->fs.val_id.clear(); e->fs.rules.clear(); e->fs.class_rows.assign(G, std::array<int, 4>{{-1, -1, -1, -1}});
    if (!per_node) {
        for (size_t i = 0; i < G; ++i) {
            PodSpec& p = e->specs[(size_t)e->pegs[i].spec];
            if (!p.spread.empty()) { p.unsupported = true; p.why = "topologySpreadConstraints"; }
            // required pod affinity looks at the pods of the node's topology domain: a property of the snapshot, not of a
            // template — such groups are estimated on the whole snapshot (casim_estimate_on_cluster, rule kind 2 below)
            if (!p.aff.empty() && !aff_static[i]) { p.unsupported = true; p.why = "required pod affinity"; }   // ((4a) took the static ones)
        }
    } else {
        auto& dr = e->dr;
        const size_t words = (NG + 63) / 64;
        std::map<std::string, int> key_id;
        std::vector<std::string> keys;
        auto key_of = [&](const std::string& k) { auto it = key_id.find(k); if (it != key_id.end()) return it->second; key_id[k] = (int)keys.size(); keys.push_back(k); return (int)keys.size() - 1; };
        auto node_passes_affinity = [&](const PodSpec& p, const Group& g) {   // RequiredNodeAffinity.Match (nodeSelector + required term)
            for (auto& kv : p.node_selector) { auto it = g.labels.find(kv.first); if (it == g.labels.end() || it->second != kv.second) return false; }
            if (p.has_node_terms && !node_terms_match(p.node_terms, g.labels, g.name)) return false;
            return selector_matches(p.node_affinity, g.labels);
        };
        auto zone_conflict = [&](const PodSpec& a, const PodSpec& b, const std::string& k) {   // either direction, through key k
            for (auto& t : a.anti) if (t.topology_key == k && term_matches(t, b)) return true;
            for (auto& t : b.anti) if (t.topology_key == k && term_matches(t, a)) return true;
            return false;
        };
        // keys that matter: spread keys of the classes, non-hostname anti-affinity keys of classes and running pods
        std::set<std::string> aa_keys;
        for (size_t i = 0; i < G; ++i) for (auto& t : e->specs[(size_t)e->pegs[i].spec].anti) if (t.topology_key != kHostname) aa_keys.insert(t.topology_key);
        // (running pods through their content classes: only a class of its own can carry terms)
        for (int32_t c : rc.special) for (auto& t : e->specs[(size_t)rc.rep[(size_t)c]].anti) if (t.topology_key != kHostname) aa_keys.insert(t.topology_key);
        // required pod affinity (V/.../interpodaffinity/filtering.go:234-272,382-409): an existing / placed pod counts for the
        // class when it matches ALL of its affinity terms (podMatchesAllAffinityTerms), once per term, in the domain of its
        // node for that term's topology key; a node passes a term when its domain holds such a pod
        auto matches_all_aff = [&](const PodSpec& owner, const PodSpec& q) {
            if (owner.aff.empty()) return false;
            for (auto& t : owner.aff) if (!term_matches(t, q)) return false;
            return true;
        };
        struct Rule { int cls, key, kind, skew, mind, self, row; const Spread* sc; int ghost = 0; };
        std::vector<Rule> rules;
        std::vector<std::vector<uint64_t>> rows;
        struct SpreadClass { size_t cls; int row_of[4]; };
        std::vector<SpreadClass> spread_classes;   // their eligibility rows are filled after the loop, nodes in parallel
        for (size_t i = 0; i < G; ++i) {
            const PodSpec& p = e->specs[(size_t)e->pegs[i].spec];
            // eligibility of a node for a constraint (filtering.go:262-271, common.go:44-58): every constraint key of the
            // class present, the pod's required node affinity matches unless the constraint says nodeAffinityPolicy: Ignore,
            // and for nodeTaintsPolicy: Honor no untolerated NoSchedule / NoExecute taint.  One row per policy pair in use.
            int row_of[4] = {-1, -1, -1, -1};   // [affinity honoured ? 1 : 0][taints honoured ? 2 : 0]
            if (!p.spread.empty()) {
                for (auto& sc : p.spread) {
                    const int combo = (sc.affinity_honor ? 1 : 0) | (sc.taints_honor ? 2 : 0);
                    if (row_of[combo] < 0) { row_of[combo] = (int)rows.size(); rows.emplace_back(words, 0ull); }
                }
                spread_classes.push_back(SpreadClass{i, {row_of[0], row_of[1], row_of[2], row_of[3]}});
            }
            e->fs.class_rows[i] = std::array<int, 4>{{row_of[0], row_of[1], row_of[2], row_of[3]}};
            for (auto& sc : p.spread) {
                if (sc.taints_honor) dr.n_taint_rules++;
                Rule r{(int)i, key_of(sc.key), 0, sc.max_skew, sc.min_domains, 0, row_of[(sc.affinity_honor ? 1 : 0) | (sc.taints_honor ? 2 : 0)], &sc};
                r.self = (!sc.selector.empty() && selector_matches(sc.selector, p.labels)) ? 1 : 0;   // an empty selector counts nothing
                if (sc.taints_honor) {
                    // removal simulation: the candidate turns into a ghost carrying ToBeDeletedByClusterAutoscaler:NoSchedule
                    // (CA/simulator/cluster.go:240-252; the value is a timestamp: only an Exists toleration can match it, "0" here as in
                    // the oracle).  With nodeTaintsPolicy: Honor such a node is no member of the constraint's domains unless the pod
                    // tolerates the taint: the ghost LEAVES its domain for the simulation (rule_ghost_leaves).
                    const Taint ghost{"ToBeDeletedByClusterAutoscaler", "0", "NoSchedule"};
                    bool tol = false;
                    for (int32_t ti : p.tolerations) if (tolerates(e->tol_dict[(size_t)ti], ghost, e->opt.enable_taint_comparison_ops != 0)) { tol = true; break; }
                    r.ghost = tol ? 0 : 1;
                }
                rules.push_back(r);
            }
            for (auto& k : aa_keys) {
                bool any = false;
                for (size_t j = 0; j < G && !any; ++j) any = zone_conflict(p, e->specs[(size_t)e->pegs[j].spec], k);
                for (int32_t s2 : rc.rep) { if (any) break; any = zone_conflict(p, e->specs[(size_t)s2], k); }   // (one pod per content class)
                if (!any) continue;
                Rule r{(int)i, key_of(k), 1, 0, 0, zone_conflict(p, p, k) ? 1 : 0, -1, nullptr};
                rules.push_back(r);
            }
            for (auto& t : p.aff) {   // kind 2: one rule per affinity term (its topology key), all fed by the same pods
                Rule r{(int)i, key_of(t.topology_key), 2, 0, 0, matches_all_aff(p, p) ? 1 : 0, -1, nullptr};
                rules.push_back(r);
            }
        }
        // the rows: node ranges on 64-node word boundaries (no two threads share a word), every class with constraints per node
        enc_par_for(words, enc_threads(NG, 2048) > 1 ? enc_threads(NG, 2048) : 1, [&](size_t wlo, size_t whi, int) {
            for (size_t n = wlo * 64; n < whi * 64 && n < NG; ++n) {
                const Group& g = e->groups[n];
                for (auto& sc0 : spread_classes) {
                    const PodSpec& p = e->specs[(size_t)e->pegs[sc0.cls].spec];
                    const int* row_of = sc0.row_of;
                    bool keys_ok = true;
                    for (auto& sc : p.spread) keys_ok = keys_ok && g.labels.count(sc.key) != 0;
                    if (!keys_ok) continue;
                    const bool aff = node_passes_affinity(p, g);
                    bool tolerated = true;
                    for (auto& tn : g.taints) {
                        if (tn.effect != "NoSchedule" && tn.effect != "NoExecute") continue;
                        bool tol = false;
                        for (int32_t ti : p.tolerations) if (tolerates(e->tol_dict[(size_t)ti], tn, e->opt.enable_taint_comparison_ops != 0)) { tol = true; break; }
                        if (!tol) { tolerated = false; break; }
                    }
                    for (int combo = 0; combo < 4; ++combo) {
                        if (row_of[combo] < 0) continue;
                        if ((combo & 1) && !aff) continue;
                        if ((combo & 2) && !tolerated) continue;
                        rows[(size_t)row_of[combo]][n >> 6] |= 1ull << (n & 63);
                    }
                }
            }
        });
        stage.mark("dr_rules_rows");
        if (!rules.empty()) {
            dr.n_keys = (int32_t)keys.size(); dr.n_rules = (int32_t)rules.size(); dr.n_rows = (int32_t)rows.size();
            // domains: distinct values of each key over the nodes
            dr.node_domain.assign(keys.size() * NG, -1); dr.key_domains.assign(keys.size(), 0);
            dr.key_host.assign(keys.size(), 0);
            for (size_t k = 0; k < keys.size(); ++k) dr.key_host[k] = keys[k] == kHostname ? 1 : 0;
            e->fs.keys = keys;
            for (auto& R1 : rules) e->fs.rules.push_back(casim_encoder::FinalRule{R1.cls, R1.key, R1.kind, R1.row, R1.sc});
            for (size_t k = 0; k < keys.size(); ++k) {
                std::map<std::string, int> val_id;
                for (size_t n = 0; n < NG; ++n) {
                    auto it = e->groups[n].labels.find(keys[k]);
                    if (it == e->groups[n].labels.end()) continue;
                    auto v = val_id.find(it->second);
                    int id;
                    if (v == val_id.end()) { id = (int)val_id.size(); val_id[it->second] = id; } else id = v->second;
                    dr.node_domain[k * NG + n] = id;
                }
                dr.key_domains[k] = (int32_t)val_id.size();
                e->fs.val_id.push_back(val_id);
            }
            dr.r_off.assign(rules.size() + 1, 0);
            for (size_t r = 0; r < rules.size(); ++r) dr.r_off[r + 1] = dr.r_off[r] + dr.key_domains[(size_t)rules[r].key];
            dr.count_init.assign((size_t)dr.r_off.back(), 0); dr.exists.assign((size_t)dr.r_off.back(), 0);
            dr.dom_nodes.assign((size_t)dr.r_off.back(), 0); dr.node_contrib.assign(rules.size() * NG, 0);
            dr.class_off.assign(G + 1, 0); dr.inc_off.assign(G + 1, 0);
        stage.mark("dr_domains");
            // "does a running pod feed this rule" looks at the pod's namespace, labels and anti-affinity terms only: every (rule, content
            // class) pair is evaluated once, and a node's pods are walked ONCE, each adding to the rules its class feeds — instead of one
            // string-keyed selector evaluation per (rule, node, running pod) triple (4.8 M of them were 120 of the 185 ms of a full
            // finalize at 15 000 nodes / 150 000 running pods / 32 rules).
            const std::vector<int32_t>& spec_class = rc.cls;
            const std::vector<int32_t>& class_rep = rc.rep;
            std::vector<std::vector<int32_t>> class_feeds(class_rep.size());   // rules a running pod of the class feeds, ascending
            for (size_t r = 0; r < rules.size(); ++r) {
                const Rule& R0 = rules[r];
                const PodSpec& p = e->specs[(size_t)e->pegs[(size_t)R0.cls].spec];
                dr.r_class.push_back(R0.cls); dr.r_key.push_back(R0.key); dr.r_kind.push_back(R0.kind); dr.r_skew.push_back(R0.skew);
                dr.r_mind.push_back(R0.mind); dr.r_self.push_back(R0.self); dr.r_row.push_back(R0.row); dr.r_ghost.push_back((uint8_t)R0.ghost);
                dr.class_off[(size_t)R0.cls + 1]++;
                for (size_t c = 0; c < class_rep.size(); ++c) {
                    const PodSpec& q = e->specs[(size_t)class_rep[c]];
                    const bool feeds = R0.kind == 0 ? (!R0.sc->selector.empty() && q.ns == p.ns && selector_matches(R0.sc->selector, q.labels))
                                     : R0.kind == 1 ? zone_conflict(p, q, keys[(size_t)R0.key]) : matches_all_aff(p, q);
                    if (feeds) class_feeds[c].push_back((int32_t)r);
                }
                // the nodes of the rule's domains (a spread rule: the eligible ones)
                const int32_t* nd = &dr.node_domain[(size_t)R0.key * NG];
                const uint64_t* row = R0.kind == 0 ? rows[(size_t)R0.row].data() : nullptr;
                for (size_t n = 0; n < NG; ++n) {
                    if (nd[n] < 0) continue;
                    if (row && !((row[n >> 6] >> (n & 63)) & 1ull)) continue;
                    const size_t at = (size_t)dr.r_off[r] + (size_t)nd[n];
                    dr.exists[at] = 1; dr.dom_nodes[at]++;
                }
            }
            for (size_t n = 0; n < NG; ++n) {
                for (int32_t s2 : e->groups[n].preloaded) {
                    for (int32_t r : class_feeds[(size_t)spec_class[(size_t)s2]]) {
                        const Rule& R0 = rules[(size_t)r];
                        const int d = dr.node_domain[(size_t)R0.key * NG + n];
                        if (d < 0) continue;
                        if (R0.kind == 0 && !((rows[(size_t)R0.row][n >> 6] >> (n & 63)) & 1ull)) continue;
                        dr.count_init[(size_t)dr.r_off[(size_t)r] + (size_t)d]++; dr.node_contrib[(size_t)r * NG + n]++;
                    }
                }
            }
        stage.mark("dr_counters");
            for (size_t c = 0; c < G; ++c) dr.class_off[c + 1] += dr.class_off[c];
            // which rules a placed pod of class j feeds
            for (size_t j = 0; j < G; ++j) {
                const PodSpec& q = e->specs[(size_t)e->pegs[j].spec];
                for (size_t r = 0; r < rules.size(); ++r) {
                    const Rule& R0 = rules[r];
                    const PodSpec& p = e->specs[(size_t)e->pegs[(size_t)R0.cls].spec];
                    const bool feeds = R0.kind == 0 ? (!R0.sc->selector.empty() && q.ns == p.ns && selector_matches(R0.sc->selector, q.labels))
                                     : R0.kind == 1 ? zone_conflict(p, q, keys[(size_t)R0.key]) : matches_all_aff(p, q);
                    if (feeds) dr.inc_rule.push_back((int32_t)r);
                }
                dr.inc_off[j + 1] = (int32_t)dr.inc_rule.size();
            }
            dr.elig.assign(rows.size() * words, 0);
            for (size_t r = 0; r < rows.size(); ++r) for (size_t w = 0; w < words; ++w) dr.elig[r * words + w] = rows[r][w];
        }
    }

    stage.mark("domain_rules");
    // ---- flat PEG table ------------------------------------------------------------------
    const int Wt = e->Wt, Wl = e->Wl, Wx = e->Wx, Wz = e->Wz;
    e->req.assign(G * (size_t)R, 0); e->count.assign(G, 0); e->pflags.assign(G, 0);
    e->tol.assign(G * (size_t)Wt, 0); e->sel.assign(G * (size_t)Wl, 0);
    e->xblock.assign(G * (size_t)Wx, 0); e->xmark.assign(G * (size_t)Wx, 0); e->xports.assign(G * (size_t)Wx + 1, 0);
    e->zblock.assign(G * (size_t)Wz, 0); e->zmark.assign(G * (size_t)Wz, 0);
    e->fp_cpu.assign(G, 0.0); e->fp_mem.assign(G, 0.0);
    const Taint unsched{kUnschedulableTaint, "", "NoSchedule"};
    // A PEG tolerates a taint when ONE of its tolerations does (v1helper.TolerationsTolerateTaint): its row is the OR of what each
    // toleration tolerates.  PEG lists are mostly distinct, single tolerations are not (a few keys x operators x effects; interned when
    // they arrive), so the verdicts are worked out once per distinct toleration a PEG uses — [Wt words, then one word for the
    // unschedulable taint] — and ORed per PEG.
    const size_t TW = (size_t)Wt + 1;
    std::vector<uint64_t> tol_rows(e->tol_dict.size() * TW, 0);
    std::vector<uint8_t> tol_row_done(e->tol_dict.size(), 0);
    auto toleration_row = [&](int32_t ti) -> size_t {
        const size_t at = (size_t)ti * TW;
        if (tol_row_done[(size_t)ti]) return at;
        tol_row_done[(size_t)ti] = 1;
        const Toleration& t = e->tol_dict[(size_t)ti];
        for (auto& kv : taint_id) if (tolerates(t, kv.first, cmp_ops)) tol_rows[at + (size_t)(kv.second >> 6)] |= 1ull << (kv.second & 63);
        if (tolerates(t, unsched, cmp_ops)) tol_rows[at + (size_t)Wt] = 1;
        return at;
    };
    for (size_t i = 0; i < G; ++i) {
        const PodSpec& p = e->specs[(size_t)e->pegs[i].spec];
        for (int r = 0; r < R; ++r) e->req[i * (size_t)R + (size_t)r] = p.req[r];
        e->count[i] = e->pegs[i].count;
        uint32_t f = pflags[i];
        for (int32_t ti : p.tolerations) {
            const size_t at = toleration_row(ti);
            for (int w = 0; w < Wt; ++w) e->tol[i * (size_t)Wt + (size_t)w] |= tol_rows[at + (size_t)w];
            if (tol_rows[at + (size_t)Wt]) f |= CASIM_PEG_TOLERATES_UNSCHEDULABLE;
        }
        for (int id : spec_lreqs[(size_t)e->pegs[i].spec]) set_bit(e->sel, i, Wl, id);
        // ports: any host port conflicts with a second copy of the same pod
        bool has_port = false;
        for (auto& pt : p.ports) {
            if (pt.port <= 0) continue;
            has_port = true;
            const Port sp = sanitize(pt);
            auto own = port_bit.find(sp);
            if (own != port_bit.end()) set_bit(e->xmark, i, Wx, own->second);
            for (auto& pb : port_bit) if (ports_conflict(sp, pb.first)) { set_bit(e->xblock, i, Wx, pb.second); set_bit(e->xports, i, Wx, pb.second); }
        }
        if (has_port) f |= CASIM_PEG_SELF_EXCL_NODE;
        if (peg_occ_bit[i] >= 0) set_bit(e->xmark, i, Wx, peg_occ_bit[i]);
        for (int b : peg_blockers[i]) set_bit(e->xblock, i, Wx, b);
        if (host_need_bit[i] >= 0) set_bit(e->xblock, i, Wx, host_need_bit[i]);
        for (int b : host_marks[i]) set_bit(e->xmark, i, Wx, b);
        for (int b : z_block[i]) set_bit(e->zblock, i, Wz, b);
        for (int b : z_mark[i]) set_bit(e->zmark, i, Wz, b);
        // fastpath eligibility: no topology spread (unsupported anyway) and no non-hostname anti-affinity
        bool fp_ok = !p.unsupported;
        for (auto& t : p.anti) if (t.topology_key != kHostname) fp_ok = false;
        if (fp_ok) f |= CASIM_PEG_FASTPATH_OK;
        for (auto& t : p.anti)
            if (t.topology_key == kHostname && selector_matches(t.selector, p.labels)) { f |= CASIM_PEG_FASTPATH_AA_SELF; break; }
        if (p.unsupported) f |= CASIM_PEG_UNSUPPORTED;
        e->pflags[i] = f;
        e->fp_cpu[i] = p.fp_cpu; e->fp_mem[i] = p.fp_mem;
    }

    stage.mark("peg_table");
    // ---- flat group table ------------------------------------------------------------------
    e->alloc.assign(NG * (size_t)R, 0); e->init_req.assign(NG * (size_t)R, 0);
    e->allowed.assign(NG, 0); e->init_pods.assign(NG, 0); e->gflags.assign(NG, 0);
    e->taint.assign(NG * (size_t)Wt, 0); e->label.assign(NG * (size_t)Wl, 0);
    e->init_excl.assign(NG * (size_t)Wx, 0); e->init_zone.assign(NG * (size_t)Wz, 0); e->zone_valid.assign(NG * (size_t)Wz, 0);
    e->max_nodes.assign(NG, 0); e->existing_nodes.assign(NG, 0); e->last_index.assign(NG, 0);
    e->cap_cpu.assign(NG, 0.0); e->cap_mem.assign(NG, 0.0); e->waste_cpu.assign(NG, 0); e->waste_mem.assign(NG, 0);
    // (a row per node, nothing shared between rows: node ranges on up to four threads — at cluster scale the loop reads the requests of
    // every running pod's cold spec record)
    uint8_t explicit_in_slice[4] = {0, 0, 0, 0};
    enc_par_for(NG, enc_threads(NG, 2048), [&](size_t g_lo, size_t g_hi, int slice) {
    bool any_explicit = false;
    for (size_t gi = g_lo; gi < g_hi; ++gi) {
        const Group& g = e->groups[gi];
        for (int r = 0; r < R; ++r) e->alloc[gi * (size_t)R + (size_t)r] = g.alloc[r];
        e->allowed[gi] = g.allowed;
        e->gflags[gi] = g.unschedulable ? CASIM_NG_UNSCHEDULABLE : 0;
        e->max_nodes[gi] = g.max_nodes; e->existing_nodes[gi] = g.existing; e->last_index[gi] = g.last_index;
        e->cap_cpu[gi] = g.fp_cap_cpu; e->cap_mem[gi] = g.fp_cap_mem;
        e->waste_cpu[gi] = g.cap_cpu; e->waste_mem[gi] = g.cap_mem;
        for (auto& t : g.taints) { auto it = taint_id.find(t); if (it != taint_id.end()) set_bit(e->taint, gi, Wt, it->second); }
        for (size_t l = 0; l < lreqs.size(); ++l) if (lreqs[l].op == kNodeTerms ? node_terms_match(lreqs[l].terms, g.labels, g.name) : requirement_matches(lreqs[l], g.labels)) set_bit(e->label, gi, Wl, (int)l);
        for (int32_t s : g.preloaded) {
            const PodSpec& p = e->specs[(size_t)s];
            for (int r = 0; r < R; ++r) e->init_req[gi * (size_t)R + (size_t)r] += p.req[r];
            e->init_pods[gi] += 1;
            for (auto& pt : p.ports) {
                if (pt.port <= 0) continue;
                auto it = port_bit.find(sanitize(pt));
                if (it != port_bit.end()) set_bit(e->init_excl, gi, Wx, it->second);
            }
            auto ob = pre_occ_bit.find(s);
            if (ob != pre_occ_bit.end()) set_bit(e->init_excl, gi, Wx, ob->second);
        }
        for (auto& zk : zbit_key) {
            if (zk.second.empty() || g.labels.count(zk.second)) set_bit(e->zone_valid, gi, Wz, zk.first);
        }
        any_explicit = any_explicit || g.has_pegs;
    }
    explicit_in_slice[slice & 3] = any_explicit ? 1 : 0;
    });
    const bool any_explicit = explicit_in_slice[0] || explicit_in_slice[1] || explicit_in_slice[2] || explicit_in_slice[3];
    for (size_t i = 0; i < G; ++i) for (uint32_t gi : existing_block[i]) set_bit(e->init_zone, gi, Wz, static_zbit[i]);
    for (auto& pr : zone_preset) set_bit(e->init_zone, pr.first, Wz, pr.second);
    for (auto& pr : excl_preset) set_bit(e->init_excl, pr.first, Wx, pr.second);
    e->xpol.assign((size_t)Wx, 0ull);
    for (int b : xneed_bits) e->xpol[(size_t)(b >> 6)] |= 1ull << (b & 63);
    e->zpol.assign((size_t)Wz, 0ull);
    for (int b : need_bits) e->zpol[(size_t)(b >> 6)] |= 1ull << (b & 63);
    e->peg_off.clear(); e->peg_idx.clear();
    if (any_explicit) {
        e->peg_off.push_back(0);
        for (auto& g : e->groups) {
            for (int32_t pg : g.pegs) e->peg_idx.push_back(pg);
            e->peg_off.push_back((int32_t)e->peg_idx.size());
        }
        if (e->peg_idx.empty()) e->peg_idx.push_back(0);  // keep a valid pointer
    }
    e->dict[0] = (int)taint_id.size(); e->dict[1] = (int)lreqs.size(); e->dict[2] = xbits.n; e->dict[3] = zbits.n;
    e->finalized = true;
    stage.mark("group_table");
    // what casim_enc_refinalize works from
    e->fs.hostname_inert = hostname_inert && any_hostname_terms;
    e->fs.taint_id = taint_id; e->fs.lreqs = lreqs; e->fs.port_bit = port_bit; e->fs.pre_occ_bit = pre_occ_bit;
    e->fs.n_specs = NS; e->fs.NG = NG; e->fs.G = G;
    e->fs.dirty.assign(NG, 0); e->fs.dirty_list.clear();
    e->fs.valid = true; e->updating = false;
    return CASIM_OK;
}

}  // extern "C"

// ---- incremental re-encode (per-node mode) --------------------------------------------------------------------------------------
// The reference forks its snapshot in O(1) and adds pods in place (CA/simulator/clustersnapshot/store/delta.go:235-246,292-323); a
// full encode of 15 000 nodes / 150 000 pods costs tens of milliseconds per loop iteration (profiles/r02i_native_scale.jsonl).
// Between two iterations most nodes are unchanged: an update session re-describes ONLY the changed nodes (their labels, taints
// and running pods) and casim_enc_refinalize recomputes those rows of the node table and their share of the domain-rule
// counters against the dictionaries of the last full finalize.  Whatever would change a dictionary — a new rejecting taint, a
// label value that opens a new topology domain, a running spec that needs a node bit or a rule nobody has yet, a node without a
// hostname label, nodes added or removed, new classes — answers CASIM_ENC_NEEDS_FULL, and the caller runs casim_enc_finalize on the
// same encoder (every object it described is still there).
namespace {
bool node_passes_affinity_of(const PodSpec& p, const Group& g) {   // RequiredNodeAffinity.Match (nodeSelector + required term)
    for (auto& kv : p.node_selector) { auto it = g.labels.find(kv.first); if (it == g.labels.end() || it->second != kv.second) return false; }
    if (p.has_node_terms && !node_terms_match(p.node_terms, g.labels, g.name)) return false;
    return selector_matches(p.node_affinity, g.labels);
}
bool zone_conflict_of(const PodSpec& a, const PodSpec& b, const std::string& k) {
    for (auto& t : a.anti) if (t.topology_key == k && term_matches(t, b)) return true;
    for (auto& t : b.anti) if (t.topology_key == k && term_matches(t, a)) return true;
    return false;
}
bool matches_all_aff_of(const PodSpec& owner, const PodSpec& q) {
    if (owner.aff.empty()) return false;
    for (auto& t : owner.aff) if (!term_matches(t, q)) return false;
    return true;
}
}  // namespace

extern "C" {

int32_t casim_enc_begin_update(casim_encoder* e) {
    if (!e || !e->finalized || !e->fs.valid || e->updating) return CASIM_ERR_INVALID;
    if (!e->opt.explicit_self_exclusion) return CASIM_ERR_INVALID;   // per-node tables only (template tables are a few dozen rows)
    e->updating = true;
    return CASIM_OK;
}
int32_t casim_enc_group_reset(casim_encoder* e, int32_t group, const int64_t* alloc, int32_t allowed_pods, int64_t capacity_cpu_milli,
                              int64_t capacity_mem_bytes, int32_t unschedulable) {
    if (!e || !e->updating || group < 0 || (size_t)group >= e->groups.size() || !alloc) return CASIM_ERR_INVALID;
    Group& g = e->groups[(size_t)group];
    if (!e->fs.dirty[(size_t)group]) {
        // the node leaves the domain rules with everything it contributed (its row and counters are rebuilt by refinalize)
        auto& dr = e->dr;
        const size_t NG = e->fs.NG, words = (NG + 63) / 64, n = (size_t)group;
        for (size_t r = 0; r < e->fs.rules.size(); ++r) {
            const auto& R0 = e->fs.rules[r];
            const int d = dr.node_domain[(size_t)R0.key * NG + n];
            if (d < 0) continue;
            if (R0.kind == 0 && !((dr.elig[(size_t)R0.row * words + (n >> 6)] >> (n & 63)) & 1ull)) continue;
            const size_t at = (size_t)dr.r_off[r] + (size_t)d;
            dr.dom_nodes[at]--; dr.exists[at] = dr.dom_nodes[at] > 0 ? 1 : 0;
            dr.count_init[at] -= dr.node_contrib[r * NG + n];
            dr.node_contrib[r * NG + n] = 0;
        }
        for (size_t row = 0; row < (size_t)dr.n_rows; ++row) dr.elig[row * words + (n >> 6)] &= ~(1ull << (n & 63));
        e->fs.dirty[(size_t)group] = 1; e->fs.dirty_list.push_back(group);
    }
    for (int r = 0; r < CASIM_MAX_RES; ++r) g.alloc[r] = r < e->opt.n_res ? alloc[r] : 0;
    g.allowed = allowed_pods; g.cap_cpu = capacity_cpu_milli; g.cap_mem = capacity_mem_bytes; g.unschedulable = unschedulable != 0;
    g.fp_cap_cpu = (double)capacity_cpu_milli * 1e-3; g.fp_cap_mem = (double)capacity_mem_bytes;
    g.labels = Labels(); g.taints.clear(); g.preloaded.clear(); g.named_alloc.clear();
    return CASIM_OK;
}
int32_t casim_enc_set_peg_count(casim_encoder* e, int32_t peg, int32_t count) {
    if (!e || !e->updating || peg < 0 || (size_t)peg >= e->pegs.size() || count < 0) return CASIM_ERR_INVALID;
    e->pegs[(size_t)peg].count = count;
    e->count[(size_t)peg] = count;
    return CASIM_OK;
}

int32_t casim_enc_refinalize(casim_encoder* e, int32_t* changed_out, int32_t capacity, int32_t* n_changed_out) {
    if (!e || !e->updating) return CASIM_ERR_INVALID;
    auto& fs = e->fs; auto& dr = e->dr;
    const size_t NG = fs.NG, G = fs.G, R = (size_t)e->out_res, words = (NG + 63) / 64;
    const int Wt = e->Wt, Wl = e->Wl, Wx = e->Wx;
    const bool cmp_ops = e->opt.enable_taint_comparison_ops != 0;
    if (e->groups.size() != NG || e->pegs.size() != G) return CASIM_ENC_NEEDS_FULL;   // nodes or classes were added
    // the changed-node list must fit the caller's array BEFORE anything is recomputed: a truncated list would leave rows the caller
    // never ships to casim_cluster_update_nodes (a stale device image) with no way to get them back — the session stays open,
    // n_changed_out says what is needed
    if (changed_out && (capacity < 0 || (size_t)capacity < fs.dirty_list.size())) {
        if (n_changed_out) *n_changed_out = (int32_t)fs.dirty_list.size();
        return CASIM_ERR_INVALID;
    }
    // ---- 1. nothing may touch a dictionary ----
    bool hostname_bits = !fs.port_bit.empty() || !fs.pre_occ_bit.empty() || !fs.with_terms.empty();
    if (fs.running.size() < e->specs.size()) fs.running.resize(e->specs.size(), 0);
    for (int32_t gi : fs.dirty_list) {
        const Group& g = e->groups[(size_t)gi];
        for (auto& t : g.taints) if ((t.effect == "NoSchedule" || t.effect == "NoExecute") && !fs.taint_id.count(t)) return CASIM_ENC_NEEDS_FULL;
        if (hostname_bits && !g.labels.count(kHostname)) return CASIM_ENC_NEEDS_FULL;
        if (fs.hostname_inert && g.labels.count(kHostname)) return CASIM_ENC_NEEDS_FULL;   // (the first node with the label: the terms start to count)
        for (size_t k = 0; k < fs.keys.size(); ++k) {
            auto it = g.labels.find(fs.keys[k]);
            if (it != g.labels.end() && !fs.val_id[k].count(it->second)) return CASIM_ENC_NEEDS_FULL;   // a new topology domain
        }
        for (int32_t s2 : g.preloaded) {
            if (fs.running[(size_t)s2]) continue;
            const PodSpec& q = e->specs[(size_t)s2];
            // a spec no node ran at the last finalize: fine as long as it needs no node bit and no rule that does not exist yet
            for (auto& pt : q.ports) if (pt.port > 0 && !fs.port_bit.count(sanitize(pt))) return CASIM_ENC_NEEDS_FULL;
            if (!q.anti.empty() || !q.aff.empty() || !q.spread.empty()) return CASIM_ENC_NEEDS_FULL;
            for (size_t i : fs.with_terms)
                for (auto& t : e->specs[(size_t)e->pegs[i].spec].anti) if (t.topology_key == kHostname && term_matches(t, q)) return CASIM_ENC_NEEDS_FULL;
            for (size_t i = 0; i < G; ++i)
                for (auto& t : e->specs[(size_t)e->pegs[i].spec].anti) {
                    if (t.topology_key == kHostname || !term_matches(t, q)) continue;
                    bool have = false;
                    for (auto& R0 : fs.rules) if (R0.kind == 1 && (size_t)R0.cls == i && fs.keys[(size_t)R0.key] == t.topology_key) have = true;
                    if (!have) return CASIM_ENC_NEEDS_FULL;
                }
            fs.running[(size_t)s2] = 1;
        }
    }
    // ---- 2. rows of the node table ----
    for (int32_t gsel : fs.dirty_list) {
        const size_t gi = (size_t)gsel;
        const Group& g = e->groups[gi];
        for (size_t r = 0; r < R; ++r) { e->alloc[gi * R + r] = g.alloc[r]; e->init_req[gi * R + r] = 0; }
        e->allowed[gi] = g.allowed; e->init_pods[gi] = 0;
        e->gflags[gi] = g.unschedulable ? CASIM_NG_UNSCHEDULABLE : 0;
        e->max_nodes[gi] = g.max_nodes; e->existing_nodes[gi] = g.existing; e->last_index[gi] = g.last_index;
        e->cap_cpu[gi] = g.fp_cap_cpu; e->cap_mem[gi] = g.fp_cap_mem; e->waste_cpu[gi] = g.cap_cpu; e->waste_mem[gi] = g.cap_mem;
        for (int w = 0; w < Wt; ++w) e->taint[gi * (size_t)Wt + (size_t)w] = 0;
        for (int w = 0; w < Wl; ++w) e->label[gi * (size_t)Wl + (size_t)w] = 0;
        for (int w = 0; w < Wx; ++w) e->init_excl[gi * (size_t)Wx + (size_t)w] = 0;
        for (auto& t : g.taints) { auto it = fs.taint_id.find(t); if (it != fs.taint_id.end()) set_bit(e->taint, gi, Wt, it->second); }
        for (size_t l = 0; l < fs.lreqs.size(); ++l)
            if (fs.lreqs[l].op == kNodeTerms ? node_terms_match(fs.lreqs[l].terms, g.labels, g.name) : requirement_matches(fs.lreqs[l], g.labels)) set_bit(e->label, gi, Wl, (int)l);
        for (int32_t s2 : g.preloaded) {
            const PodSpec& p = e->specs[(size_t)s2];
            for (size_t r = 0; r < R; ++r) e->init_req[gi * R + r] += p.req[r];
            e->init_pods[gi] += 1;
            for (auto& pt : p.ports) {
                if (pt.port <= 0) continue;
                auto it = fs.port_bit.find(sanitize(pt));
                if (it != fs.port_bit.end()) set_bit(e->init_excl, gi, Wx, it->second);
            }
            auto ob = fs.pre_occ_bit.find(s2);
            if (ob != fs.pre_occ_bit.end()) set_bit(e->init_excl, gi, Wx, ob->second);
        }
        // ---- 3. the node's place in the domain rules ----
        if (!fs.rules.empty()) {
            for (size_t k = 0; k < fs.keys.size(); ++k) {
                auto it = g.labels.find(fs.keys[k]);
                dr.node_domain[k * NG + gi] = it == g.labels.end() ? -1 : fs.val_id[k].at(it->second);
            }
            for (size_t i = 0; i < G; ++i) {   // eligibility rows of the classes with spread constraints
                const PodSpec& p = e->specs[(size_t)e->pegs[i].spec];
                if (p.spread.empty()) continue;
                bool keys_ok = true;
                for (auto& sc : p.spread) keys_ok = keys_ok && g.labels.count(sc.key) != 0;
                if (!keys_ok) continue;
                const bool aff = node_passes_affinity_of(p, g);
                bool tolerated = true;
                for (auto& tn : g.taints) {
                    if (tn.effect != "NoSchedule" && tn.effect != "NoExecute") continue;
                    bool tol = false;
                    for (int32_t ti : p.tolerations) if (tolerates(e->tol_dict[(size_t)ti], tn, cmp_ops)) { tol = true; break; }
                    if (!tol) { tolerated = false; break; }
                }
                for (int combo = 0; combo < 4; ++combo) {
                    const int row = fs.class_rows[i][(size_t)combo];
                    if (row < 0) continue;
                    if ((combo & 1) && !aff) continue;
                    if ((combo & 2) && !tolerated) continue;
                    dr.elig[(size_t)row * words + (gi >> 6)] |= 1ull << (gi & 63);
                }
            }
            for (size_t r = 0; r < fs.rules.size(); ++r) {
                const auto& R0 = fs.rules[r];
                const PodSpec& p = e->specs[(size_t)e->pegs[(size_t)R0.cls].spec];
                const int d = dr.node_domain[(size_t)R0.key * NG + gi];
                if (d < 0) continue;
                if (R0.kind == 0 && !((dr.elig[(size_t)R0.row * words + (gi >> 6)] >> (gi & 63)) & 1ull)) continue;
                const size_t at = (size_t)dr.r_off[r] + (size_t)d;
                dr.exists[at] = 1; dr.dom_nodes[at]++;
                for (int32_t s2 : g.preloaded) {
                    const PodSpec& q = e->specs[(size_t)s2];
                    const bool feeds = R0.kind == 0 ? (!R0.sc->selector.empty() && q.ns == p.ns && selector_matches(R0.sc->selector, q.labels))
                                     : R0.kind == 1 ? zone_conflict_of(p, q, fs.keys[(size_t)R0.key]) : matches_all_aff_of(p, q);
                    if (feeds) { dr.count_init[at]++; dr.node_contrib[r * NG + gi]++; }
                }
            }
        }
    }
    const int32_t n = (int32_t)fs.dirty_list.size();
    if (n_changed_out) *n_changed_out = n;
    if (changed_out) for (int32_t k = 0; k < n && k < capacity; ++k) changed_out[k] = fs.dirty_list[(size_t)k];
    for (int32_t gi : fs.dirty_list) fs.dirty[(size_t)gi] = 0;
    fs.dirty_list.clear();
    fs.n_specs = e->specs.size();
    e->updating = false;
    return CASIM_OK;
}

// compact copies of n rows of the node table (what casim_cluster_update_nodes takes); valid until the next call / destroy
int32_t casim_enc_group_rows(casim_encoder* e, const int32_t* groups, int32_t n, casim_groups* out) {
    if (!e || !e->finalized || !out || n < 0 || (n > 0 && !groups)) return CASIM_ERR_INVALID;
    const size_t R = (size_t)e->out_res, Wt = (size_t)e->Wt, Wl = (size_t)e->Wl, Wx = (size_t)e->Wx, Wz = e->init_zone.size() / (e->groups.empty() ? 1 : e->groups.size()), N = (size_t)n;
    for (int32_t k = 0; k < n; ++k) if (groups[k] < 0 || (size_t)groups[k] >= e->groups.size()) return CASIM_ERR_INVALID;
    auto pick = [&](auto& dst, const auto& src, size_t width) {
        dst.resize(N * width + 1);
        for (size_t k = 0; k < N; ++k) for (size_t w = 0; w < width; ++w) dst[k * width + w] = src[(size_t)groups[k] * width + w];
    };
    pick(e->rows_alloc, e->alloc, R); pick(e->rows_init_req, e->init_req, R); pick(e->rows_allowed, e->allowed, 1); pick(e->rows_init_pods, e->init_pods, 1);
    pick(e->rows_gflags, e->gflags, 1); pick(e->rows_taint, e->taint, Wt); pick(e->rows_label, e->label, Wl); pick(e->rows_init_excl, e->init_excl, Wx);
    pick(e->rows_init_zone, e->init_zone, Wz); pick(e->rows_zone_valid, e->zone_valid, Wz);   // (template mode: group-wide exclusion words of the rows)
    pick(e->rows_max_nodes, e->max_nodes, 1); pick(e->rows_existing, e->existing_nodes, 1); pick(e->rows_last_index, e->last_index, 1);
    pick(e->rows_cap_cpu, e->cap_cpu, 1); pick(e->rows_cap_mem, e->cap_mem, 1); pick(e->rows_waste_cpu, e->waste_cpu, 1); pick(e->rows_waste_mem, e->waste_mem, 1);
    memset(out, 0, sizeof *out);
    out->n_groups = n;
    out->alloc = e->rows_alloc.data(); out->init_req = e->rows_init_req.data(); out->allowed_pods = e->rows_allowed.data(); out->init_pods = e->rows_init_pods.data();
    out->flags = e->rows_gflags.data(); out->taint_mask = e->rows_taint.data(); out->label_mask = e->rows_label.data(); out->init_excl = e->rows_init_excl.data();
    out->init_zone = e->rows_init_zone.data(); out->zone_valid = e->rows_zone_valid.data();
    out->max_nodes = e->rows_max_nodes.data(); out->existing_nodes = e->rows_existing.data(); out->last_index = e->rows_last_index.data();
    out->cap_cpu = e->rows_cap_cpu.data(); out->cap_mem = e->rows_cap_mem.data(); out->waste_cpu = e->rows_waste_cpu.data(); out->waste_mem = e->rows_waste_mem.data();
    return CASIM_OK;
}

}  // extern "C"

extern "C" {
int32_t casim_enc_tables(const casim_encoder* e, casim_pegs* p, casim_groups* g) {
    if (!e || !e->finalized || !p || !g) return CASIM_ERR_INVALID;
    memset(p, 0, sizeof *p); memset(g, 0, sizeof *g);
    p->n_pegs = (int32_t)e->pegs.size(); p->n_res = e->out_res;
    p->w_taint = e->Wt; p->w_label = e->Wl; p->w_excl = e->Wx; p->w_zone = e->Wz;
    p->req = e->req.data(); p->count = e->count.data(); p->flags = e->pflags.data();
    p->tol_mask = e->tol.data(); p->sel_mask = e->sel.data();
    p->excl_block = e->xblock.data(); p->excl_mark = e->xmark.data();
    p->zone_block = e->zblock.data(); p->zone_mark = e->zmark.data();
    p->zone_polarity = e->zpol.empty() ? nullptr : e->zpol.data();
    p->excl_polarity = e->xpol.empty() ? nullptr : e->xpol.data();
    p->fp_cpu = e->fp_cpu.data(); p->fp_mem = e->fp_mem.data();
    g->n_groups = (int32_t)e->groups.size();
    g->alloc = e->alloc.data(); g->init_req = e->init_req.data(); g->allowed_pods = e->allowed.data(); g->init_pods = e->init_pods.data();
    g->flags = e->gflags.data(); g->taint_mask = e->taint.data(); g->label_mask = e->label.data();
    g->init_excl = e->init_excl.data(); g->init_zone = e->init_zone.data(); g->zone_valid = e->zone_valid.data();
    g->max_nodes = e->max_nodes.data(); g->existing_nodes = e->existing_nodes.data(); g->last_index = e->last_index.data();
    g->cap_cpu = e->cap_cpu.data(); g->cap_mem = e->cap_mem.data(); g->waste_cpu = e->waste_cpu.data(); g->waste_mem = e->waste_mem.data();
    if (!e->peg_off.empty()) { g->peg_offsets = e->peg_off.data(); g->peg_index = e->peg_idx.data(); }
    return CASIM_OK;
}
int32_t casim_enc_domain_rules(const casim_encoder* e, casim_domain_rules* out) {
    if (!e || !e->finalized || !out) return CASIM_ERR_INVALID;
    memset(out, 0, sizeof *out);
    const auto& dr = e->dr;
    out->n_keys = dr.n_keys; out->n_rules = dr.n_rules; out->n_nodes = (int32_t)e->groups.size(); out->n_classes = (int32_t)e->pegs.size();
    out->n_elig_rows = dr.n_rows;
    out->n_taint_policy_rules = dr.n_taint_rules;
    if (dr.n_rules == 0) return CASIM_OK;
    out->node_domain = dr.node_domain.data(); out->key_domains = dr.key_domains.data(); out->key_is_hostname = dr.key_host.data();
    out->rule_class = dr.r_class.data(); out->rule_key = dr.r_key.data(); out->rule_kind = dr.r_kind.data();
    out->rule_max_skew = dr.r_skew.data(); out->rule_min_domains = dr.r_mind.data(); out->rule_self = dr.r_self.data();
    out->rule_elig_row = dr.r_row.data(); out->rule_offset = dr.r_off.data(); out->count_init = dr.count_init.data();
    out->rule_ghost_leaves = dr.r_ghost.data();
    out->domain_exists = dr.exists.data(); out->domain_nodes = dr.dom_nodes.data(); out->node_contrib = dr.node_contrib.data();
    out->elig_bits = dr.elig.data(); out->class_rule_off = dr.class_off.data();
    out->inc_off = dr.inc_off.data(); out->inc_rule = dr.inc_rule.data();
    return CASIM_OK;
}
const uint64_t* casim_enc_port_block(const casim_encoder* e) { return (e && e->finalized) ? e->xports.data() : nullptr; }
int32_t casim_enc_dict_sizes(const casim_encoder* e, int32_t sizes_out[4]) {
    if (!e || !e->finalized || !sizes_out) return CASIM_ERR_INVALID;
    for (int i = 0; i < 4; ++i) sizes_out[i] = e->dict[i];
    return CASIM_OK;
}

}  // extern "C"

// ---- f2: pod equivalence groups -------------------------------------------------------------------------------------------------
// BuildPodGroups / groupPodsBySchedulingProperties / match  (CA/core/scaleup/equivalence/groups.go:39-104).  match() compares a pod
// with the representant of each of its controller's groups: reflect.DeepEqual(labels) && PodSpecSemanticallyEqual(spec)
// (CA/utils/utils.go:63-119: projected volumes, hostname and env dropped, apiequality.Semantic: quantities by value, nil == empty).
// Here a spec is everything the shim told the encoder about it plus `extra`; two specs match when their canonical byte strings are
// equal.  The string is built once per distinct spec id and interned, so match() is one integer compare.
namespace {
void put(std::string& o, const std::string& s) { uint32_t n = (uint32_t)s.size(); o.append((const char*)&n, 4); o.append(s); }
void put_i(std::string& o, int64_t v) { o.append((const char*)&v, 8); }
void put_reqs(std::string& o, const std::vector<Requirement>& rs);
void put_node_terms(std::string& o, const std::vector<NodeTerm>& ts) {
    put_i(o, (int64_t)ts.size());
    for (auto& t : ts) { put_reqs(o, t.exprs); put_reqs(o, t.fields); }
}
void put_reqs(std::string& o, const std::vector<Requirement>& rs) {
    put_i(o, (int64_t)rs.size());
    for (auto& r : rs) {
        put(o, r.key); put_i(o, r.op); put_i(o, (int64_t)r.values.size());
        for (auto& v : r.values) put(o, v);
        put_node_terms(o, r.terms);
    }
}
void put_terms(std::string& o, const std::vector<Term>& ts) {
    put_i(o, (int64_t)ts.size());
    for (auto& t : ts) {
        put(o, t.topology_key); put_i(o, (int64_t)t.namespaces.size());
        for (auto& n : t.namespaces) put(o, n);
        put_reqs(o, t.selector); put_i(o, t.has_ns_sel); put_reqs(o, t.ns_sel);
    }
}
std::string canonical_spec(const PodSpec& p, int R) {
    std::string o;
    o.reserve(256);
    put(o, p.ns);
    for (int r = 0; r < R; ++r) put_i(o, p.req[r]);
    put_i(o, (int64_t)p.labels.v.size());
    for (auto& kv : p.labels.v) { put(o, kv.first); put(o, kv.second); }     // (sorted by key: a Go map has no order)
    put_i(o, (int64_t)p.tolerations.size());
    for (int32_t ti : p.tolerations) put_i(o, ti);   // (interned: equal ids == equal (key, operator, value, effect))
    std::vector<std::pair<std::string, std::string>> ns = p.node_selector;   // map[string]string
    std::sort(ns.begin(), ns.end());
    put_i(o, (int64_t)ns.size());
    for (auto& kv : ns) { put(o, kv.first); put(o, kv.second); }
    put_reqs(o, p.node_affinity); put_i(o, p.has_node_terms); put_node_terms(o, p.node_terms);
    put_i(o, (int64_t)p.ports.size());
    for (auto& h : p.ports) { put(o, h.ip); put(o, h.proto); put_i(o, h.port); }
    put_terms(o, p.anti); put_terms(o, p.aff);
    put_i(o, (int64_t)p.spread.size());
    for (auto& c : p.spread) { put_i(o, c.max_skew); put(o, c.key); put_i(o, c.min_domains); put_reqs(o, c.selector); put_i(o, c.taints_honor); put_i(o, c.affinity_honor); }
    put_i(o, p.unsupported); put(o, p.why); put(o, p.extra);
    int64_t f[2]; memcpy(f, &p.fp_cpu, 8); memcpy(f + 1, &p.fp_mem, 8); put_i(o, f[0]); put_i(o, f[1]);
    return o;
}
}  // namespace

extern "C" {

int32_t casim_enc_pod_set_spec_extra(casim_encoder* e, int32_t pod, const char* digest) {
    POD_CHECK(e, pod); e->specs[pod].extra = S(digest); return CASIM_OK;
}

int32_t casim_enc_group_pods(casim_encoder* e, int32_t n_pods, const int32_t* pod_spec, const char* const* controller_uid,
                             const uint8_t* daemonset, int32_t* group_out, int32_t* n_groups_out) {
    if (!e || n_pods < 0 || (n_pods > 0 && (!pod_spec || !group_out))) return CASIM_ERR_INVALID;
    const size_t NS = e->specs.size();
    for (int32_t i = 0; i < n_pods; ++i) if (pod_spec[i] < 0 || (size_t)pod_spec[i] >= NS) return CASIM_ERR_INVALID;
    // canonical id of a spec, built the first time a controller has to compare it
    std::vector<int32_t> canon(NS, -1);
    std::unordered_map<std::string, int32_t> interned;
    auto canon_of = [&](int32_t s) {
        if (canon[s] < 0) canon[s] = interned.emplace(canonical_spec(e->specs[s], CASIM_MAX_RES), (int32_t)interned.size()).first->second;
        return canon[s];
    };
    struct Eg { int32_t id, spec, canon; };                 // equivalenceGroup{id, representant}
    struct Ctl { int32_t n = 0; Eg eg[10]; };               // maxEquivalenceGroupsByController (groups.go:58)
    std::unordered_map<std::string_view, Ctl> by_controller;
    by_controller.reserve(1024);
    int32_t next = 0;
    for (int32_t i = 0; i < n_pods; ++i) {
        const char* uid = controller_uid ? controller_uid[i] : nullptr;
        if (!uid || !uid[0] || (daemonset && daemonset[i])) { group_out[i] = next++; continue; }   // groups.go:69-74
        Ctl& c = by_controller[std::string_view(uid)];
        const int32_t s = pod_spec[i];
        int32_t hit = -1;
        for (int32_t k = 0; k < c.n && hit < 0; ++k) if (c.eg[k].spec == s) hit = c.eg[k].id;       // same spec record: equal by construction
        if (hit < 0 && c.n > 0) {
            const int32_t cs = canon_of(s);
            for (int32_t k = 0; k < c.n && hit < 0; ++k) {
                if (c.eg[k].canon < 0) c.eg[k].canon = canon_of(c.eg[k].spec);
                if (c.eg[k].canon == cs) hit = c.eg[k].id;
            }
        }
        if (hit >= 0) { group_out[i] = hit; continue; }
        if (c.n < 10) c.eg[c.n++] = Eg{next, s, canon[s]};   // beyond 10 the pod still opens a group, nobody can join it (groups.go:81-90)
        group_out[i] = next++;
    }
    if (n_groups_out) *n_groups_out = next;
    return CASIM_OK;
}

int32_t casim_enc_add_grouped_pegs(casim_encoder* e, int32_t n_pods, const int32_t* pod_spec, const int32_t* group, int32_t n_groups,
                                   int32_t* peg_ids_out) {
    ENC_CHECK(e);
    if (n_pods < 0 || n_groups < 0 || (n_pods > 0 && (!pod_spec || !group))) return CASIM_ERR_INVALID;
    std::vector<int32_t> first(n_groups, -1), count(n_groups, 0);
    for (int32_t i = 0; i < n_pods; ++i) {
        if (group[i] < 0 || group[i] >= n_groups || pod_spec[i] < 0 || (size_t)pod_spec[i] >= e->specs.size()) return CASIM_ERR_INVALID;
        if (first[group[i]] < 0) first[group[i]] = pod_spec[i];   // the exemplar: Pods[0] (PodEquivalenceGroup.Exemplar)
        ++count[group[i]];
    }
    const int32_t base = (int32_t)e->pegs.size();
    for (int32_t g = 0; g < n_groups; ++g) if (first[g] < 0) return CASIM_ERR_INVALID;   // (an empty group: nothing is added at all)
    for (int32_t g = 0; g < n_groups; ++g) {
        e->pegs.push_back(Peg{first[g], count[g]});
        if (peg_ids_out) peg_ids_out[g] = base + g;
    }
    return base;
}

}  // extern "C"
