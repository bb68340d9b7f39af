/*
 * casim_oracle.c — CPU ORACLE (test infrastructure, NOT product code). See casim_oracle.h.
 *
 * Restates, object level and pod by pod, the reference path
 *   BinpackingNodeEstimator.Estimate -> PredicateSnapshot.SchedulePod* ->
 *   SchedulerPluginRunner.RunFilters* -> scheduler Filter plugins.
 * Each function cites the reference file:line it follows (`CA/` =
 * /root/reference/cluster-autoscaler/, `V/` = its vendor/k8s.io/).
 *
 * Deliberately naive: per-pod loops, per-node Filter runs, string-keyed maps rebuilt per pod —
 * the same asymptotic work the Go code does, so that timing it gives a "port"
 * CPU baseline.  Strings are interned to integer ids at insertion (id equality == string
 * equality); numeric Gt/Lt comparisons go back to the strings.
 */
#include "casim_oracle.h"

#include <errno.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------- */
/* small utilities                                                                        */
/* ------------------------------------------------------------------------------------- */
#define VEC(T) struct { T* v; int n, cap; }
#define VEC_PUSH(vec, item)                                                        \
    do {                                                                           \
        if ((vec).n == (vec).cap) {                                                \
            (vec).cap = (vec).cap ? (vec).cap * 2 : 4;                             \
            (vec).v = realloc((vec).v, sizeof(*(vec).v) * (size_t)(vec).cap);      \
        }                                                                          \
        (vec).v[(vec).n++] = (item);                                               \
    } while (0)
#define VEC_FREE(vec) do { free((vec).v); (vec).v = NULL; (vec).n = (vec).cap = 0; } while (0)

typedef VEC(int) ivec;

/* string interning */
typedef struct { char** s; int n, cap; int* table; int tcap; } strtab;

static unsigned long hash_str(const char* s) {
    unsigned long h = 1469598103934665603UL;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211UL; }
    return h;
}
static void strtab_rehash(strtab* t, int tcap) {
    free(t->table);
    t->tcap = tcap;
    t->table = malloc(sizeof(int) * (size_t)tcap);
    for (int i = 0; i < tcap; ++i) t->table[i] = -1;
    for (int i = 0; i < t->n; ++i) {
        unsigned long h = hash_str(t->s[i]) % (unsigned long)tcap;
        while (t->table[h] >= 0) h = (h + 1) % (unsigned long)tcap;
        t->table[h] = i;
    }
}
static int intern(strtab* t, const char* s) {
    if (!s) s = "";
    if (t->tcap == 0) strtab_rehash(t, 64);
    unsigned long h = hash_str(s) % (unsigned long)t->tcap;
    while (t->table[h] >= 0) {
        if (strcmp(t->s[t->table[h]], s) == 0) return t->table[h];
        h = (h + 1) % (unsigned long)t->tcap;
    }
    if (t->n == t->cap) { t->cap = t->cap ? t->cap * 2 : 32; t->s = realloc(t->s, sizeof(char*) * (size_t)t->cap); }
    t->s[t->n] = strdup(s);
    t->table[h] = t->n++;
    if (t->n * 2 > t->tcap) strtab_rehash(t, t->tcap * 2);
    return t->n - 1;
}

/* ------------------------------------------------------------------------------------- */
/* object model                                                                           */
/* ------------------------------------------------------------------------------------- */
enum { OP_IN, OP_NOTIN, OP_EXISTS, OP_DOESNOTEXIST, OP_GT, OP_LT, OP_BAD };
enum { TOL_EQUAL, TOL_EXISTS, TOL_LT, TOL_GT, TOL_BAD };

typedef struct { int key, value; } kv;
typedef struct { int key, value, effect; } taint;
typedef struct { int key, op, value, effect; } toleration;
typedef struct { int key, op; ivec values; } requirement;
typedef VEC(requirement) reqvec;
typedef struct { int ip, proto, port; } hostport;
/* v1.NodeSelectorTerm: matchExpressions (node labels) AND matchFields (metadata.name) */
typedef struct { reqvec exprs, fields; } node_term;
/* has_ns_sel: the term's namespaceSelector is set (empty = labels.Everything); auto_ns: namespaces holds the pod's own
 * namespace because neither field was given (V/kube-scheduler/framework/types.go:439-447) */
typedef struct { int topology_key; ivec namespaces; reqvec selector; int has_ns_sel, auto_ns; reqvec ns_sel; } aff_term;
typedef struct { int name; VEC(kv) labels; } ns_entry;
/* topologySpreadConstraint  V/.../podtopologyspread/common.go:34-41 (DoNotSchedule constraints only; nodeAffinityPolicy
 * Honor and nodeTaintsPolicy Ignore, the defaults :108-110) */
typedef struct { int max_skew, topology_key, min_domains; reqvec selector; int selector_set; int taints_honor; int affinity_ignore; } spread_constraint;

typedef struct {
    int ns;
    int64_t req[ORC_MAX_RES];
    VEC(kv) labels;
    VEC(toleration) tolerations;
    VEC(kv) node_selector;
    reqvec node_affinity; /* single required term, ANDed requirements */
    int has_node_affinity;
    VEC(node_term) node_terms; /* nodeSelectorTerms, ORed (set through orc_pod_node_affinity_term); excludes node_affinity */
    int has_node_terms;
    VEC(hostport) ports;
    VEC(aff_term) anti_terms;
    VEC(aff_term) aff_terms;   /* requiredDuringSchedulingIgnoredDuringExecution pod AFFINITY terms (explicit namespaces) */
    double fp_cpu, fp_mem;
    int fp_has_requests;
    int has_topology_spread;
    VEC(spread_constraint) spread;
} podspec;

typedef struct {
    int name;
    VEC(kv) labels;
    VEC(taint) taints;
    int unschedulable;
    int64_t alloc[ORC_MAX_RES];
    int allowed_pods;
    int64_t cap_cpu_milli, cap_mem;
    double fp_cap_cpu, fp_cap_mem; /* Capacity.{Cpu,Memory}().AsApproximateFloat64() */
    /* NodeInfo aggregates  V/kubernetes/pkg/scheduler/framework/types.go:166-214 */
    int64_t requested[ORC_MAX_RES];
    ivec pods;          /* pod spec ids, in AddPod order */
    VEC(hostport) used_ports;
    int n_pods_with_aa; /* len(PodsWithRequiredAntiAffinity) */
    int orig_id;        /* position at orc_snapshot_add time (stable id across removals) */
    int is_new;         /* created by this Estimate (estimationState.newNodeNames) */
    int new_pods;       /* pods placed by this Estimate (newNodesWithPods) */
} node;

struct orc {
    int n_res;
    strtab st;
    VEC(podspec) pods;
    VEC(node) nodes;     /* node objects (templates / existing) */
    VEC(node) snap;      /* the cluster snapshot list, insertion order */
    VEC(ns_entry) namespaces; /* the namespace lister */
    int taint_cmp_ops;
    int id_hostname, id_noschedule, id_noexecute, id_allip, id_tcp, id_empty, id_unsched_key;
    int64_t filter_runs;
    const char* last_fail_reason; /* reason of the last failing Filter run (SchedulingError.FailingPredicateReasons) */
    /* Fork()/Revert() support for callers that commit pods to snapshot nodes: pre-images of touched nodes */
    int snap_added;     /* nodes ever added to the snapshot (source of orig_id) */
    uint64_t shuffle;   /* != 0: ListNodeInfos() order is re-drawn for every scheduling attempt (Go map iteration) */
    int undo_on;
    struct { int n, cap; struct undo_rec { int orig_id; node copy; }* v; } undo;
};

/* ------------------------------------------------------------------------------------- */
/* construction                                                                           */
/* ------------------------------------------------------------------------------------- */
orc* orc_new(int n_res) {
    if (n_res < 2 || n_res > ORC_MAX_RES) return NULL;
    orc* o = calloc(1, sizeof(orc));
    o->n_res = n_res;
    o->id_empty = intern(&o->st, "");
    o->id_hostname = intern(&o->st, "kubernetes.io/hostname");
    o->id_noschedule = intern(&o->st, "NoSchedule");
    o->id_noexecute = intern(&o->st, "NoExecute");
    o->id_allip = intern(&o->st, "0.0.0.0");
    o->id_tcp = intern(&o->st, "TCP");
    o->id_unsched_key = intern(&o->st, "node.kubernetes.io/unschedulable");
    return o;
}

static void free_reqvec(reqvec* r) {
    for (int i = 0; i < r->n; ++i) VEC_FREE(r->v[i].values);
    VEC_FREE(*r);
}
static void node_free(node* n) {
    VEC_FREE(n->labels); VEC_FREE(n->taints); VEC_FREE(n->pods); VEC_FREE(n->used_ports);
}
void orc_free(orc* o) {
    if (!o) return;
    for (int i = 0; i < o->pods.n; ++i) {
        podspec* p = &o->pods.v[i];
        VEC_FREE(p->labels); VEC_FREE(p->tolerations); VEC_FREE(p->node_selector);
        free_reqvec(&p->node_affinity); VEC_FREE(p->ports);
        for (int t = 0; t < p->node_terms.n; ++t) { free_reqvec(&p->node_terms.v[t].exprs); free_reqvec(&p->node_terms.v[t].fields); }
        VEC_FREE(p->node_terms);
        for (int t = 0; t < p->anti_terms.n; ++t) { VEC_FREE(p->anti_terms.v[t].namespaces); free_reqvec(&p->anti_terms.v[t].selector); free_reqvec(&p->anti_terms.v[t].ns_sel); }
        VEC_FREE(p->anti_terms);
        for (int t = 0; t < p->aff_terms.n; ++t) { VEC_FREE(p->aff_terms.v[t].namespaces); free_reqvec(&p->aff_terms.v[t].selector); free_reqvec(&p->aff_terms.v[t].ns_sel); }
        VEC_FREE(p->aff_terms);
        for (int c = 0; c < p->spread.n; ++c) free_reqvec(&p->spread.v[c].selector);
        VEC_FREE(p->spread);
    }
    VEC_FREE(o->pods);
    for (int i = 0; i < o->nodes.n; ++i) node_free(&o->nodes.v[i]);
    VEC_FREE(o->nodes);
    for (int i = 0; i < o->snap.n; ++i) node_free(&o->snap.v[i]);
    VEC_FREE(o->snap);
    for (int i = 0; i < o->namespaces.n; ++i) VEC_FREE(o->namespaces.v[i].labels);
    VEC_FREE(o->namespaces);
    for (int i = 0; i < o->st.n; ++i) free(o->st.s[i]);
    free(o->st.s); free(o->st.table);
    free(o);
}
void orc_set_taint_comparison_ops(orc* o, int enabled) { o->taint_cmp_ops = enabled; }
void orc_set_list_shuffle(orc* o, uint64_t seed) { o->shuffle = seed; }

int orc_pod(orc* o, const char* ns, const int64_t* req) {
    podspec p; memset(&p, 0, sizeof p);
    p.ns = intern(&o->st, ns);
    for (int r = 0; r < o->n_res; ++r) p.req[r] = req[r];
    VEC_PUSH(o->pods, p);
    return o->pods.n - 1;
}
#define PODCHK(o, pod) if ((pod) < 0 || (pod) >= (o)->pods.n) return -1
#define NODECHK(o, nd) if ((nd) < 0 || (nd) >= (o)->nodes.n) return -1

int orc_pod_label(orc* o, int pod, const char* key, const char* value) {
    PODCHK(o, pod);
    int k = intern(&o->st, key), v = intern(&o->st, value);
    podspec* p = &o->pods.v[pod];
    for (int i = 0; i < p->labels.n; ++i) if (p->labels.v[i].key == k) { p->labels.v[i].value = v; return 0; }
    kv e = {k, v}; VEC_PUSH(p->labels, e); return 0;
}
static int parse_tol_op(const char* op) {
    if (!op || !*op || strcmp(op, "Equal") == 0) return TOL_EQUAL; /* empty operator means Equal, toleration.go:62 */
    if (strcmp(op, "Exists") == 0) return TOL_EXISTS;
    if (strcmp(op, "Lt") == 0) return TOL_LT;
    if (strcmp(op, "Gt") == 0) return TOL_GT;
    return TOL_BAD;
}
int orc_pod_toleration(orc* o, int pod, const char* key, const char* op, const char* value, const char* effect) {
    PODCHK(o, pod);
    toleration t = {intern(&o->st, key), parse_tol_op(op), intern(&o->st, value), intern(&o->st, effect)};
    VEC_PUSH(o->pods.v[pod].tolerations, t); return 0;
}
int orc_pod_node_selector(orc* o, int pod, const char* key, const char* value) {
    PODCHK(o, pod);
    kv e = {intern(&o->st, key), intern(&o->st, value)};
    VEC_PUSH(o->pods.v[pod].node_selector, e); return 0;
}
static int parse_req_op(const char* op) {
    if (!op) return OP_BAD;
    if (!strcmp(op, "In")) return OP_IN;
    if (!strcmp(op, "NotIn")) return OP_NOTIN;
    if (!strcmp(op, "Exists")) return OP_EXISTS;
    if (!strcmp(op, "DoesNotExist")) return OP_DOESNOTEXIST;
    if (!strcmp(op, "Gt")) return OP_GT;
    if (!strcmp(op, "Lt")) return OP_LT;
    return OP_BAD;
}
static requirement make_req(orc* o, const char* key, const char* op, const char* const* values, int n) {
    requirement r; memset(&r, 0, sizeof r);
    r.key = intern(&o->st, key); r.op = parse_req_op(op);
    for (int i = 0; i < n; ++i) VEC_PUSH(r.values, intern(&o->st, values[i]));
    return r;
}
int orc_pod_node_affinity_req(orc* o, int pod, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (o->pods.v[pod].has_node_terms) return -1;
    requirement r = make_req(o, key, op, values, n);
    VEC_PUSH(o->pods.v[pod].node_affinity, r);
    o->pods.v[pod].has_node_affinity = 1; return 0;
}
/* nodeAffinity.requiredDuringSchedulingIgnoredDuringExecution.nodeSelectorTerms[]: opens a new (empty) term */
int orc_pod_node_affinity_term(orc* o, int pod) {
    PODCHK(o, pod);
    if (o->pods.v[pod].has_node_affinity) return -1; /* one NodeSelector per pod: either API, not both */
    node_term t; memset(&t, 0, sizeof t);
    VEC_PUSH(o->pods.v[pod].node_terms, t);
    o->pods.v[pod].has_node_terms = 1;
    return o->pods.v[pod].node_terms.n - 1;
}
int orc_pod_node_term_req(orc* o, int pod, int term, int is_field, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].node_terms.n) return -1;
    requirement r = make_req(o, key, op, values, n);
    if (is_field) VEC_PUSH(o->pods.v[pod].node_terms.v[term].fields, r);
    else VEC_PUSH(o->pods.v[pod].node_terms.v[term].exprs, r);
    return 0;
}
int orc_pod_host_port(orc* o, int pod, const char* ip, const char* protocol, int port) {
    PODCHK(o, pod);
    hostport h = {intern(&o->st, ip), intern(&o->st, protocol), port};
    VEC_PUSH(o->pods.v[pod].ports, h); return 0;
}
int orc_pod_anti_affinity_term(orc* o, int pod, const char* topology_key, const char* const* namespaces, int n) {
    PODCHK(o, pod);
    aff_term t; memset(&t, 0, sizeof t);
    t.topology_key = intern(&o->st, topology_key);
    /* getNamespacesFromPodAffinityTerm  V/kubernetes/pkg/scheduler/framework/types.go (newAffinityTerm):
     * no namespaces and no namespaceSelector => the pod's own namespace */
    if (n == 0) { VEC_PUSH(t.namespaces, o->pods.v[pod].ns); t.auto_ns = 1; }
    for (int i = 0; i < n; ++i) VEC_PUSH(t.namespaces, intern(&o->st, namespaces[i]));
    VEC_PUSH(o->pods.v[pod].anti_terms, t);
    return o->pods.v[pod].anti_terms.n - 1;
}
/* a required pod AFFINITY term (PodAffinity.RequiredDuringSchedulingIgnoredDuringExecution); namespaces as above */
int orc_pod_affinity_term(orc* o, int pod, const char* topology_key, const char* const* namespaces, int n) {
    PODCHK(o, pod);
    aff_term t; memset(&t, 0, sizeof t);
    t.topology_key = intern(&o->st, topology_key);
    if (n == 0) { VEC_PUSH(t.namespaces, o->pods.v[pod].ns); t.auto_ns = 1; }
    for (int i = 0; i < n; ++i) VEC_PUSH(t.namespaces, intern(&o->st, namespaces[i]));
    VEC_PUSH(o->pods.v[pod].aff_terms, t);
    return o->pods.v[pod].aff_terms.n - 1;
}
int orc_aff_term_requirement(orc* o, int pod, int term, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].aff_terms.n) return -1;
    requirement r = make_req(o, key, op, values, n);
    VEC_PUSH(o->pods.v[pod].aff_terms.v[term].selector, r); return 0;
}
static ns_entry* ns_find(const orc* o, int name) {
    for (int i = 0; i < o->namespaces.n; ++i) if (o->namespaces.v[i].name == name) return &o->namespaces.v[i];
    return NULL;
}
/* the namespace lister: a namespace and, optionally, one of its labels (key NULL = just register it) */
int orc_namespace_label(orc* o, const char* name, const char* key, const char* value) {
    const int id = intern(&o->st, name);
    ns_entry* e = ns_find(o, id);
    if (!e) { ns_entry ne; memset(&ne, 0, sizeof ne); ne.name = id; VEC_PUSH(o->namespaces, ne); e = &o->namespaces.v[o->namespaces.n - 1]; }
    if (key) { kv l = {intern(&o->st, key), intern(&o->st, value)}; VEC_PUSH(e->labels, l); }
    return 0;
}
/* the term's namespaceSelector is set (possibly empty) */
int orc_term_namespace_selector(orc* o, int pod, int term) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].anti_terms.n) return -1;
    aff_term* t = &o->pods.v[pod].anti_terms.v[term];
    if (t->auto_ns) { t->namespaces.n = 0; t->auto_ns = 0; }
    t->has_ns_sel = 1; return 0;
}
int orc_term_namespace_requirement(orc* o, int pod, int term, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].anti_terms.n || !o->pods.v[pod].anti_terms.v[term].has_ns_sel) return -1;
    requirement r = make_req(o, key, op, values, n);
    VEC_PUSH(o->pods.v[pod].anti_terms.v[term].ns_sel, r); return 0;
}
/* the same for a REQUIRED AFFINITY term (V/kube-scheduler/framework/types.go:390-395, 439-447; the term is the incoming pod's: a
 * non-empty selector only ever selects namespaces the lister knows, interpodaffinity/plugin.go:144-157) */
int orc_aff_term_namespace_selector(orc* o, int pod, int term) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].aff_terms.n) return -1;
    aff_term* t = &o->pods.v[pod].aff_terms.v[term];
    if (t->auto_ns) { t->namespaces.n = 0; t->auto_ns = 0; }
    t->has_ns_sel = 1; return 0;
}
int orc_aff_term_namespace_requirement(orc* o, int pod, int term, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].aff_terms.n || !o->pods.v[pod].aff_terms.v[term].has_ns_sel) return -1;
    requirement r = make_req(o, key, op, values, n);
    VEC_PUSH(o->pods.v[pod].aff_terms.v[term].ns_sel, r); return 0;
}
int orc_term_requirement(orc* o, int pod, int term, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (term < 0 || term >= o->pods.v[pod].anti_terms.n) return -1;
    requirement r = make_req(o, key, op, values, n);
    VEC_PUSH(o->pods.v[pod].anti_terms.v[term].selector, r); return 0;
}
int orc_pod_fastpath_requests(orc* o, int pod, double cpu, double mem) {
    PODCHK(o, pod);
    o->pods.v[pod].fp_cpu = cpu; o->pods.v[pod].fp_mem = mem; o->pods.v[pod].fp_has_requests = 1; return 0;
}
int orc_pod_has_topology_spread(orc* o, int pod, int flag) { PODCHK(o, pod); o->pods.v[pod].has_topology_spread = flag; return 0; }
/* a DoNotSchedule topologySpreadConstraint; min_domains <= 0 means nil (treated as 1, common.go:107) */
int orc_pod_spread_constraint(orc* o, int pod, int max_skew, const char* topology_key, int min_domains) {
    PODCHK(o, pod);
    spread_constraint c; memset(&c, 0, sizeof c);
    c.max_skew = max_skew; c.topology_key = intern(&o->st, topology_key); c.min_domains = min_domains > 0 ? min_domains : 1;
    VEC_PUSH(o->pods.v[pod].spread, c);
    o->pods.v[pod].has_topology_spread = 1;
    return o->pods.v[pod].spread.n - 1;
}
/* nodeTaintsPolicy: Honor (common.go:52-56): nodes with a taint the pod does not tolerate are no domain members */
int orc_spread_taints_policy_honor(orc* o, int pod, int constraint, int honor) {
    PODCHK(o, pod);
    if (constraint < 0 || constraint >= o->pods.v[pod].spread.n) return -1;
    o->pods.v[pod].spread.v[constraint].taints_honor = honor;
    return 0;
}
/* nodeAffinityPolicy: Ignore (common.go:46-51): the pod's required node affinity / selector does not gate domain membership */
int orc_spread_affinity_policy_ignore(orc* o, int pod, int constraint, int ignore) {
    PODCHK(o, pod);
    if (constraint < 0 || constraint >= o->pods.v[pod].spread.n) return -1;
    o->pods.v[pod].spread.v[constraint].affinity_ignore = ignore;
    return 0;
}
/* one requirement of the constraint's labelSelector (matchLabels pair == In{value}); a constraint without any
 * requirement has an EMPTY selector, which counts nothing (countPodsMatchSelector, common.go:144-147) */
int orc_spread_requirement(orc* o, int pod, int constraint, const char* key, const char* op, const char* const* values, int n) {
    PODCHK(o, pod);
    if (constraint < 0 || constraint >= o->pods.v[pod].spread.n) return -1;
    requirement r = make_req(o, key, op, values, n);
    VEC_PUSH(o->pods.v[pod].spread.v[constraint].selector, r);
    o->pods.v[pod].spread.v[constraint].selector_set = 1;
    return 0;
}

int orc_node(orc* o, const char* name, const int64_t* alloc, int allowed_pods, int64_t cap_cpu_milli, int64_t cap_mem, int unschedulable) {
    node n; memset(&n, 0, sizeof n);
    n.name = intern(&o->st, name);
    for (int r = 0; r < o->n_res; ++r) n.alloc[r] = alloc[r];
    n.allowed_pods = allowed_pods; n.cap_cpu_milli = cap_cpu_milli; n.cap_mem = cap_mem;
    /* AsApproximateFloat64 of NewMilliQuantity(v): float64(v) * math.Pow10(-3)
     * (V/apimachinery/pkg/api/resource/quantity.go:468-483); override with orc_node_fastpath_capacity */
    n.fp_cap_cpu = (double)cap_cpu_milli * 1e-3; n.fp_cap_mem = (double)cap_mem;
    n.unschedulable = unschedulable;
    VEC_PUSH(o->nodes, n);
    return o->nodes.n - 1;
}
static void node_set_label(node* n, int k, int v) {
    for (int i = 0; i < n->labels.n; ++i) if (n->labels.v[i].key == k) { n->labels.v[i].value = v; return; }
    kv e = {k, v}; VEC_PUSH(n->labels, e);
}
int orc_node_fastpath_capacity(orc* o, int nd, double cpu, double mem) {
    NODECHK(o, nd); o->nodes.v[nd].fp_cap_cpu = cpu; o->nodes.v[nd].fp_cap_mem = mem; return 0;
}
int orc_node_label(orc* o, int nd, const char* key, const char* value) {
    NODECHK(o, nd); node_set_label(&o->nodes.v[nd], intern(&o->st, key), intern(&o->st, value)); return 0;
}
int orc_node_taint(orc* o, int nd, const char* key, const char* value, const char* effect) {
    NODECHK(o, nd);
    taint t = {intern(&o->st, key), intern(&o->st, value), intern(&o->st, effect)};
    VEC_PUSH(o->nodes.v[nd].taints, t); return 0;
}

/* HostPortInfo.sanitize  V/kube-scheduler/framework/types.go:633-640 */
static void port_sanitize(const orc* o, hostport* h) {
    if (h->ip == o->id_empty) h->ip = o->id_allip;
    if (h->proto == o->id_empty) h->proto = o->id_tcp;
}
/* HostPortInfo.Add  V/kube-scheduler/framework/types.go:552-571 */
static void ports_add(const orc* o, node* n, hostport h) {
    if (h.port <= 0) return;
    port_sanitize(o, &h);
    for (int i = 0; i < n->used_ports.n; ++i) {
        hostport* u = &n->used_ports.v[i];
        if (u->ip == h.ip && u->proto == h.proto && u->port == h.port) return;
    }
    VEC_PUSH(n->used_ports, h);
}
/* NodeInfo.AddPodInfo / update(+1)  V/kubernetes/pkg/scheduler/framework/types.go:361-371,439-463 */
static void node_add_pod(const orc* o, node* n, int pod) {
    const podspec* p = &o->pods.v[pod];
    VEC_PUSH(n->pods, pod);
    if (p->anti_terms.n > 0) n->n_pods_with_aa++;
    for (int r = 0; r < o->n_res; ++r) n->requested[r] += p->req[r]; /* Resource.Add :1028-1049 */
    for (int i = 0; i < p->ports.n; ++i) ports_add(o, n, p->ports.v[i]);
}
int orc_node_add_pod(orc* o, int nd, int pod) { NODECHK(o, nd); PODCHK(o, pod); node_add_pod(o, &o->nodes.v[nd], pod); return 0; }

static node node_clone(const node* s) {
    node d = *s;
    memset(&d.labels, 0, sizeof d.labels); memset(&d.taints, 0, sizeof d.taints);
    memset(&d.pods, 0, sizeof d.pods); memset(&d.used_ports, 0, sizeof d.used_ports);
    for (int i = 0; i < s->labels.n; ++i) VEC_PUSH(d.labels, s->labels.v[i]);
    for (int i = 0; i < s->taints.n; ++i) VEC_PUSH(d.taints, s->taints.v[i]);
    for (int i = 0; i < s->pods.n; ++i) VEC_PUSH(d.pods, s->pods.v[i]);
    for (int i = 0; i < s->used_ports.n; ++i) VEC_PUSH(d.used_ports, s->used_ports.v[i]);
    return d;
}
int orc_snapshot_add(orc* o, int nd) {
    NODECHK(o, nd);
    node c = node_clone(&o->nodes.v[nd]);
    c.orig_id = o->snap_added++;
    VEC_PUSH(o->snap, c);
    return o->snap.n - 1;
}

/* ------------------------------------------------------------------------------------- */
/* label requirement / selector matching                                                   */
/* ------------------------------------------------------------------------------------- */
static int labels_lookup(const kv* labels, int n, int key, int* value_out) {
    for (int i = 0; i < n; ++i) if (labels[i].key == key) { *value_out = labels[i].value; return 1; }
    return 0;
}
/* strconv.ParseInt(s, 10, 64) */
static int parse_int64(const char* s, int64_t* out) {
    if (!s || !*s) return 0;
    const char* p = s;
    if (*p == '+' || *p == '-') ++p;
    if (!*p) return 0;
    for (const char* q = p; *q; ++q) if (*q < '0' || *q > '9') return 0;
    errno = 0;
    long long v = strtoll(s, NULL, 10);
    if (errno == ERANGE) return 0;
    *out = v; return 1;
}
/* Requirement.Matches  V/apimachinery/pkg/labels/selector.go:247-292 */
static int requirement_matches(const orc* o, const requirement* r, const kv* labels, int n) {
    int val = 0, exists = labels_lookup(labels, n, r->key, &val);
    switch (r->op) {
    case OP_IN:
        if (!exists) return 0;
        for (int i = 0; i < r->values.n; ++i) if (r->values.v[i] == val) return 1;
        return 0;
    case OP_NOTIN:
        if (!exists) return 1;
        for (int i = 0; i < r->values.n; ++i) if (r->values.v[i] == val) return 0;
        return 1;
    case OP_EXISTS: return exists;
    case OP_DOESNOTEXIST: return !exists;
    case OP_GT: case OP_LT: {
        if (!exists) return 0;
        int64_t lv, rv;
        if (!parse_int64(o->st.s[val], &lv)) return 0;
        if (r->values.n != 1) return 0;
        if (!parse_int64(o->st.s[r->values.v[0]], &rv)) return 0;
        return (r->op == OP_GT && lv > rv) || (r->op == OP_LT && lv < rv);
    }
    default: return 0;
    }
}
/* internalSelector.Matches: every requirement must match (selector.go:386-393) */
static int selector_matches(const orc* o, const reqvec* sel, const kv* labels, int n) {
    for (int i = 0; i < sel->n; ++i) if (!requirement_matches(o, &sel->v[i], labels, n)) return 0;
    return 1;
}
/* AffinityTerm.Matches  V/kube-scheduler/framework/types.go:390-395: the pod's namespace is listed in the term OR matches
 * its namespaceSelector.  owner_is_incoming: the term belongs to the pod being scheduled; PreFilter replaced a NON-EMPTY
 * selector by the namespaces the lister returns for it (interpodaffinity/plugin.go:144-157, filtering.go:286-296) and
 * passes nil labels (:253), so an unlisted namespace never matches.  Otherwise the term belongs to a pod already on a node
 * and is evaluated against the incoming pod's namespace labels, an unlisted namespace counting as unlabelled
 * (plugin.go:161-169, filtering.go:298,213).  An empty selector is labels.Everything in both directions. */
static int term_matches_pod(const orc* o, const aff_term* t, const podspec* p, int owner_is_incoming) {
    int ns_ok = 0;
    for (int i = 0; i < t->namespaces.n; ++i) if (t->namespaces.v[i] == p->ns) { ns_ok = 1; break; }
    if (!ns_ok && t->has_ns_sel) {
        const ns_entry* e = ns_find(o, p->ns);
        if (t->ns_sel.n == 0) ns_ok = 1;
        else if (owner_is_incoming) ns_ok = e != NULL && selector_matches(o, &t->ns_sel, e->labels.v, e->labels.n);
        else ns_ok = selector_matches(o, &t->ns_sel, e ? e->labels.v : NULL, e ? e->labels.n : 0);
    }
    if (!ns_ok) return 0;
    return selector_matches(o, &t->selector, p->labels.v, p->labels.n);
}

/* ------------------------------------------------------------------------------------- */
/* Filter plugins                                                                          */
/* ------------------------------------------------------------------------------------- */
static const char* PL_UNSCHED = "NodeUnschedulable";
static const char* PL_TAINT = "TaintToleration";
static const char* PL_AFFINITY = "NodeAffinity";
static const char* PL_PORTS = "NodePorts";
static const char* PL_FIT = "NodeResourcesFit";
static const char* PL_IPA = "InterPodAffinity";
static const char* PL_PTS = "PodTopologySpread";
/* podtopologyspread/plugin.go: ErrReasonConstraintsNotMatch / ErrReasonNodeLabelNotMatch */
static const char* REASON_PTS_CONSTRAINTS = "node(s) didn't match pod topology spread constraints";
static const char* REASON_PTS_LABEL = "node(s) didn't match pod topology spread constraints (missing required label)";

/* Toleration.ToleratesTaint  V/api/core/v1/toleration.go:52-77 */
static int tolerates_taint(const orc* o, const toleration* t, const taint* tn) {
    if (t->effect != o->id_empty && t->effect != tn->effect) return 0;
    if (t->key != o->id_empty && t->key != tn->key) return 0;
    switch (t->op) {
    case TOL_EQUAL: return t->value == tn->value;
    case TOL_EXISTS: return 1;
    case TOL_LT: case TOL_GT: {
        if (!o->taint_cmp_ops) return 0;
        /* compareNumericValues  toleration.go:80-114 (IsDecimalInteger rejects signs other than '-') */
        int64_t tv, nv;
        if (!parse_int64(o->st.s[t->value], &tv) || o->st.s[t->value][0] == '+') return 0;
        if (!parse_int64(o->st.s[tn->value], &nv) || o->st.s[tn->value][0] == '+') return 0;
        return t->op == TOL_LT ? (nv < tv) : (nv > tv);
    }
    default: return 0;
    }
}
/* TaintToleration.Filter  V/kubernetes/pkg/scheduler/framework/plugins/tainttoleration/taint_toleration.go:119-132
 * FindMatchingUntoleratedTaint  V/component-helpers/scheduling/corev1/helpers.go:79-87
 * DoNotScheduleTaintsFilterFunc V/kubernetes/pkg/scheduler/framework/plugins/helper/taint.go:23-27 */
static int filter_taints(const orc* o, const podspec* p, const node* n) {
    for (int i = 0; i < n->taints.n; ++i) {
        const taint* tn = &n->taints.v[i];
        if (tn->effect != o->id_noschedule && tn->effect != o->id_noexecute) continue;
        int tolerated = 0;
        for (int j = 0; j < p->tolerations.n && !tolerated; ++j) tolerated = tolerates_taint(o, &p->tolerations.v[j], tn);
        if (!tolerated) return 0;
    }
    return 1;
}
/* NodeUnschedulable.Filter  V/.../nodeunschedulable/node_unschedulable.go:142-160: unschedulable nodes
 * pass only if the pod tolerates node.kubernetes.io/unschedulable:NoSchedule */
static int filter_unschedulable(const orc* o, const podspec* p, const node* n) {
    if (!n->unschedulable) return 1;
    taint tn = {o->id_unsched_key, o->id_empty, o->id_noschedule};
    for (int j = 0; j < p->tolerations.n; ++j) if (tolerates_taint(o, &p->tolerations.v[j], &tn)) return 1;
    return 0;
}
/* labels.NewRequirement  V/apimachinery/pkg/labels/selector.go:183-215, rules 1, 2, 4, 5 (key / value syntax is what the
 * API server admitted); nodeSelectorRequirementsAsSelector nodeaffinity.go:210-246: one bad requirement is a parse error of
 * the whole term */
static int requirement_parses(const orc* o, const requirement* r) {
    int64_t v;
    switch (r->op) {
    case OP_IN: case OP_NOTIN: return r->values.n > 0;
    case OP_EXISTS: case OP_DOESNOTEXIST: return r->values.n == 0;
    case OP_GT: case OP_LT: return r->values.n == 1 && parse_int64(o->st.s[r->values.v[0]], &v);
    default: return 0;
    }
}
/* nodeSelectorRequirementsAsFieldSelector nodeaffinity.go:260-291: In / NotIn with exactly one value, ANDed;
 * fields.Set.Get of a missing key is "" and extractNodeFields :155-161 only knows metadata.name */
static int field_requirement_parses(const requirement* r) { return (r->op == OP_IN || r->op == OP_NOTIN) && r->values.n == 1; }
static int field_requirement_matches(const orc* o, const requirement* r, const node* n) {
    const int have = !strcmp(o->st.s[r->key], "metadata.name") ? n->name : o->id_empty;
    return r->op == OP_IN ? have == r->values.v[0] : have != r->values.v[0];
}
/* LazyErrorNodeSelector.Match nodeaffinity.go:85-107 + nodeSelectorTerm.match :187-198: terms ORed, an empty term is dropped
 * (:60-64), a term with a parse error never matches, matchFields only looked at when the node has a name */
static int node_selector_terms_match(const orc* o, const podspec* p, const node* n) {
    for (int t = 0; t < p->node_terms.n; ++t) {
        const node_term* nt = &p->node_terms.v[t];
        if (nt->exprs.n == 0 && nt->fields.n == 0) continue;
        int ok = 1;
        for (int i = 0; i < nt->exprs.n && ok; ++i) ok = requirement_parses(o, &nt->exprs.v[i]);
        for (int i = 0; i < nt->fields.n && ok; ++i) ok = field_requirement_parses(&nt->fields.v[i]);
        if (!ok) continue;
        if (nt->exprs.n && !selector_matches(o, &nt->exprs, n->labels.v, n->labels.n)) continue;
        if (nt->fields.n && n->name != o->id_empty) {
            for (int i = 0; i < nt->fields.n && ok; ++i) ok = field_requirement_matches(o, &nt->fields.v[i], n);
            if (!ok) continue;
        }
        return 1;
    }
    return 0;
}
/* NodeAffinity.Filter  V/.../nodeaffinity/node_affinity.go:218-240;  RequiredNodeAffinity.Match
 * V/component-helpers/scheduling/corev1/nodeaffinity/nodeaffinity.go:323-333 */
static int filter_node_affinity(const orc* o, const podspec* p, const node* n) {
    for (int i = 0; i < p->node_selector.n; ++i) { /* labels.SelectorFromSet: all key==value */
        int val;
        if (!labels_lookup(n->labels.v, n->labels.n, p->node_selector.v[i].key, &val)) return 0;
        if (val != p->node_selector.v[i].value) return 0;
    }
    if (p->has_node_affinity && !selector_matches(o, &p->node_affinity, n->labels.v, n->labels.n)) return 0;
    if (p->has_node_terms) return node_selector_terms_match(o, p, n);
    return 1;
}
/* HostPortInfo.CheckConflict  V/kube-scheduler/framework/types.go:602-631 */
static int ports_conflict(const orc* o, const node* n, hostport h) {
    if (h.port <= 0) return 0;
    port_sanitize(o, &h);
    for (int i = 0; i < n->used_ports.n; ++i) {
        const hostport* u = &n->used_ports.v[i];
        if (u->proto != h.proto || u->port != h.port) continue;
        if (h.ip == o->id_allip) return 1;                  /* 0.0.0.0 checks every IP */
        if (u->ip == o->id_allip || u->ip == h.ip) return 1; /* else 0.0.0.0 and the same IP */
    }
    return 0;
}
/* NodePorts.Filter / fitsPorts  V/.../nodeports/node_ports.go:162-190 */
static int filter_ports(const orc* o, const podspec* p, const node* n) {
    for (int i = 0; i < p->ports.n; ++i) if (ports_conflict(o, n, p->ports.v[i])) return 0;
    return 1;
}
/* fitsRequest  V/.../noderesources/fit.go:678-765.  *reason gets the FIRST insufficient resource */
static unsigned g_last_fit_mask; /* every reason of the last failing fitsRequest: bit 0 "Too many pods", bit 1 + r "Insufficient <lane r>" */
static int filter_fit(const orc* o, const podspec* p, const node* n, const char** reason) {
    int ok = 1;
    g_last_fit_mask = 0;
    if (n->pods.n + 1 > n->allowed_pods) { ok = 0; g_last_fit_mask |= 1u; if (reason && !*reason) *reason = "Too many pods"; }
    int all_zero = 1;
    for (int r = 0; r < o->n_res; ++r) if (p->req[r] != 0) all_zero = 0;
    if (all_zero) return ok;
    static const char* names[ORC_MAX_RES] = {"Insufficient cpu", "Insufficient memory", "Insufficient ephemeral-storage",
        "Insufficient scalar-0", "Insufficient scalar-1", "Insufficient scalar-2", "Insufficient scalar-3", "Insufficient scalar-4"};
    for (int r = 0; r < o->n_res; ++r) {
        if (p->req[r] > 0 && p->req[r] > n->alloc[r] - n->requested[r]) { ok = 0; g_last_fit_mask |= 2u << r; if (reason && !*reason) *reason = names[r]; }
    }
    return ok;
}

/* InterPodAffinity pre-filter state: topologyToMatchedTermCount maps
 * V/.../interpodaffinity/filtering.go:59-76,204-309 */
typedef struct { int key, value; int64_t count; } tpcount;
typedef VEC(tpcount) tpmap;
static void tpmap_add(tpmap* m, int key, int value, int64_t c) {
    for (int i = 0; i < m->n; ++i) if (m->v[i].key == key && m->v[i].value == value) { m->v[i].count += c; return; }
    tpcount e = {key, value, c}; VEC_PUSH(*m, e);
}
static int64_t tpmap_get(const tpmap* m, int key, int value) {
    for (int i = 0; i < m->n; ++i) if (m->v[i].key == key && m->v[i].value == value) return m->v[i].count;
    return 0;
}
typedef struct { tpmap existing_anti; tpmap incoming_anti; tpmap incoming_aff; int skip; } ipa_state;

/* InterPodAffinity.PreFilter  filtering.go:274-309 (required pod AFFINITY terms are out of
 * the encoded subset; specs carrying them are never sent down this path) */
static void ipa_prefilter(const orc* o, const podspec* p, ipa_state* s) {
    memset(s, 0, sizeof *s);
    /* getExistingAntiAffinityCounts :204-231 — nodes from HavePodsWithRequiredAntiAffinityList */
    for (int i = 0; i < o->snap.n; ++i) {
        const node* n = &o->snap.v[i];
        if (n->n_pods_with_aa == 0) continue;
        for (int j = 0; j < n->pods.n; ++j) {
            const podspec* ep = &o->pods.v[n->pods.v[j]];
            for (int t = 0; t < ep->anti_terms.n; ++t) {
                const aff_term* term = &ep->anti_terms.v[t];
                if (!term_matches_pod(o, term, p, 0)) continue;
                int val;
                if (labels_lookup(n->labels.v, n->labels.n, term->topology_key, &val)) tpmap_add(&s->existing_anti, term->topology_key, val, 1);
            }
        }
    }
    /* getIncomingAffinityAntiAffinityCounts :234-272 — every node, every pod */
    if (p->anti_terms.n > 0) {
        for (int i = 0; i < o->snap.n; ++i) {
            const node* n = &o->snap.v[i];
            for (int j = 0; j < n->pods.n; ++j) {
                const podspec* ep = &o->pods.v[n->pods.v[j]];
                for (int t = 0; t < p->anti_terms.n; ++t) {
                    const aff_term* term = &p->anti_terms.v[t];
                    if (!term_matches_pod(o, term, ep, 1)) continue;
                    int val;
                    if (labels_lookup(n->labels.v, n->labels.n, term->topology_key, &val)) tpmap_add(&s->incoming_anti, term->topology_key, val, 1);
                }
            }
        }
    }
    /* the affinity half of getIncomingAffinityAntiAffinityCounts: an existing pod counts when it matches ALL of the incoming
     * pod's affinity terms (updateWithAffinityTerms / podMatchesAllAffinityTerms :125-133), once per term whose topology key
     * its node carries */
    if (p->aff_terms.n > 0) {
        for (int i = 0; i < o->snap.n; ++i) {
            const node* n = &o->snap.v[i];
            for (int j = 0; j < n->pods.n; ++j) {
                const podspec* ep = &o->pods.v[n->pods.v[j]];
                int all = 1;
                for (int t = 0; t < p->aff_terms.n && all; ++t) all = term_matches_pod(o, &p->aff_terms.v[t], ep, 1);
                if (!all) continue;
                for (int t = 0; t < p->aff_terms.n; ++t) {
                    int val;
                    if (labels_lookup(n->labels.v, n->labels.n, p->aff_terms.v[t].topology_key, &val)) tpmap_add(&s->incoming_aff, p->aff_terms.v[t].topology_key, val, 1);
                }
            }
        }
    }
    /* :300-305 Skip when nothing can interact */
    s->skip = (s->existing_anti.n == 0 && p->anti_terms.n == 0 && p->aff_terms.n == 0);
}
static void ipa_free(ipa_state* s) { VEC_FREE(s->existing_anti); VEC_FREE(s->incoming_anti); VEC_FREE(s->incoming_aff); }
/* satisfyPodAffinity  filtering.go:382-409: every term's topology label on the node and a matching pod in that domain — or
 * the pod is the first of a series with affinity to itself: no matching pod ANYWHERE (len(affinityCounts) == 0), it matches
 * all its own terms, and the node carries every topology key */
static int satisfy_pod_affinity(const orc* o, const podspec* p, const node* n, const ipa_state* s) {
    int pods_exist = 1;
    for (int t = 0; t < p->aff_terms.n; ++t) {
        int val;
        if (!labels_lookup(n->labels.v, n->labels.n, p->aff_terms.v[t].topology_key, &val)) return 0;
        if (tpmap_get(&s->incoming_aff, p->aff_terms.v[t].topology_key, val) <= 0) pods_exist = 0;
    }
    if (!pods_exist) {
        if (s->incoming_aff.n != 0) return 0;
        for (int t = 0; t < p->aff_terms.n; ++t) if (!term_matches_pod(o, &p->aff_terms.v[t], p, 1)) return 0;
        return 1;
    }
    return 1;
}
/* InterPodAffinity.Filter  filtering.go:412-432 with satisfyPodAntiAffinity :367-380 and
 * satisfyExistingPodsAntiAffinity :352-364 */
static int filter_ipa(const orc* o, const podspec* p, const node* n, const ipa_state* s) {
    if (s->skip) return 1;
    if (p->aff_terms.n > 0 && !satisfy_pod_affinity(o, p, n, s)) return -1;   /* ErrReasonAffinityRulesNotMatch */
    if (s->incoming_anti.n > 0) {
        for (int t = 0; t < p->anti_terms.n; ++t) {
            int val;
            if (labels_lookup(n->labels.v, n->labels.n, p->anti_terms.v[t].topology_key, &val))
                if (tpmap_get(&s->incoming_anti, p->anti_terms.v[t].topology_key, val) > 0) return 0;
        }
    }
    if (s->existing_anti.n > 0) {
        for (int i = 0; i < n->labels.n; ++i)
            if (tpmap_get(&s->existing_anti, n->labels.v[i].key, n->labels.v[i].value) > 0) return 0;
    }
    return 1;
}

/* PodTopologySpread.PreFilter = calPreFilterState  V/.../podtopologyspread/filtering.go:236-316 with the
 * NodeInclusionPolicy feature on (default): a node contributes to constraint i when it carries EVERY topology key
 * of the pod's constraints and matches the pod's required node affinity / selector (policy Honor); its matching
 * pods (same namespace, selector) are added to the domain of its topology value — a domain exists even at count 0. */
typedef struct { tpmap* counts; int n; } pts_state;
static int filter_node_affinity(const orc* o, const podspec* p, const node* n);
static int filter_taints(const orc* o, const podspec* p, const node* n);
static void pts_prefilter(const orc* o, const podspec* p, pts_state* s) {
    s->n = p->spread.n; s->counts = NULL;
    if (s->n == 0) return;
    s->counts = calloc((size_t)s->n, sizeof(tpmap));
    for (int i = 0; i < o->snap.n; ++i) {
        const node* n = &o->snap.v[i];
        int all = 1;
        for (int c = 0; c < s->n; ++c) { int val; if (!labels_lookup(n->labels.v, n->labels.n, p->spread.v[c].topology_key, &val)) all = 0; }
        if (!all) continue;                                           /* nodeLabelsMatchSpreadConstraints :268 */
        const int aff_ok = filter_node_affinity(o, p, n);
        for (int c = 0; c < s->n; ++c) {
            const spread_constraint* sc = &p->spread.v[c];
            /* matchNodeInclusionPolicies common.go:44-58: nodeAffinityPolicy Honor (default), nodeTaintsPolicy Ignore (default) */
            if (!sc->affinity_ignore && !aff_ok) continue;
            if (sc->taints_honor && !filter_taints(o, p, n)) continue;
            int val = 0; labels_lookup(n->labels.v, n->labels.n, sc->topology_key, &val);
            int64_t count = 0;
            if (sc->selector_set)                                     /* countPodsMatchSelector common.go:143-158 */
                for (int j = 0; j < n->pods.n; ++j) {
                    const podspec* ep = &o->pods.v[n->pods.v[j]];
                    if (ep->ns == p->ns && selector_matches(o, &sc->selector, ep->labels.v, ep->labels.n)) count++;
                }
            tpmap_add(&s->counts[c], sc->topology_key, val, count);
        }
    }
}
static void pts_free(pts_state* s) { for (int c = 0; c < s->n; ++c) VEC_FREE(s->counts[c]); free(s->counts); }
/* PodTopologySpread.Filter  filtering.go:319-366: matchNum + selfMatch - globalMin <= maxSkew per constraint;
 * minMatchNum :54-68 (fewer domains than minDomains => global minimum 0). Returns 0 ok, 1 constraints, 2 label. */
static int filter_pts(const orc* o, const podspec* p, const node* n, const pts_state* s) {
    for (int c = 0; c < s->n; ++c) {
        const spread_constraint* sc = &p->spread.v[c];
        int val;
        if (!labels_lookup(n->labels.v, n->labels.n, sc->topology_key, &val)) return 2;
        int64_t min = INT32_MAX;                                      /* newCriticalPaths: MaxInt32 */
        for (int i = 0; i < s->counts[c].n; ++i) if (s->counts[c].v[i].count < min) min = s->counts[c].v[i].count;
        if (s->counts[c].n < sc->min_domains) min = 0;
        int self = sc->selector_set && selector_matches(o, &sc->selector, p->labels.v, p->labels.n) ? 1 : 0;
        int64_t match = tpmap_get(&s->counts[c], sc->topology_key, val);
        if (match + self - min > sc->max_skew) return 1;
    }
    return 0;
}

/* NodeAffinity.PreFilter  V/.../nodeaffinity/node_affinity.go:172-210: when EVERY term carries a matchFields requirement
 * {metadata.name In names}, only the union (over terms) of the intersections (within a term) of those name sets stays
 * eligible.  0 = no restriction or the node is in the set, 1 = "PreFilter filtered the Node out"
 * (CA .../predicate/plugin_runner.go:163-166 and :99-103), 2 = the set is empty: "pod affinity terms conflict" (:203).
 * Same outcome as the Filter below (a node outside the set fails matchFields in every term); it only decides which
 * reason the caller sees. */
static int node_affinity_prefilter(const orc* o, const podspec* p, const node* n) {
    if (!p->has_node_terms || p->node_terms.n == 0) return 0;
    int any_name = 0, node_in = 0;
    for (int t = 0; t < p->node_terms.n; ++t) {
        const node_term* nt = &p->node_terms.v[t];
        int restricted = 0, in_all = 1, nonempty = 1;
        ivec inter; memset(&inter, 0, sizeof inter);
        for (int i = 0; i < nt->fields.n; ++i) {
            const requirement* r = &nt->fields.v[i];
            if (strcmp(o->st.s[r->key], "metadata.name") || r->op != OP_IN) continue;
            int has = 0;
            for (int v = 0; v < r->values.n; ++v) if (r->values.v[v] == n->name) has = 1;
            if (!restricted) { for (int v = 0; v < r->values.n; ++v) VEC_PUSH(inter, r->values.v[v]); }
            else { int k = 0; for (int a = 0; a < inter.n; ++a) { int keep = 0; for (int v = 0; v < r->values.n; ++v) if (r->values.v[v] == inter.v[a]) keep = 1; if (keep) inter.v[k++] = inter.v[a]; } inter.n = k; }
            restricted = 1; in_all = in_all && has;
        }
        nonempty = inter.n > 0;
        VEC_FREE(inter);
        if (!restricted) return 0;          /* a term without node-name affinity: every node stays eligible */
        if (nonempty) any_name = 1;
        if (in_all && nonempty) node_in = 1;
    }
    if (!any_name) return 2;
    return node_in ? 0 : 1;
}

/* frameworkImpl.RunFilterPlugins in default profile order
 * V/kubernetes/pkg/scheduler/framework/runtime/framework.go:1093-1126,
 * V/kubernetes/pkg/scheduler/apis/config/v1/default_plugins.go:34-51: first failing Filter wins */
static int run_filter_plugins(orc* o, const podspec* p, const node* n, const ipa_state* s, const pts_state* ts,
                              const char** plugin, const char** reason) {
    o->filter_runs++;
    const char* dummy = NULL;
    if (!reason) reason = &dummy;
    *reason = NULL;
    const char* failed = NULL;
    const int pre = node_affinity_prefilter(o, p, n);   /* PreFilter results are looked at before any Filter runs */
    if (pre) { failed = PL_AFFINITY; *reason = pre == 1 ? "PreFilter filtered the Node out" : "pod affinity terms conflict"; }
    else if (!filter_unschedulable(o, p, n)) { failed = PL_UNSCHED; *reason = "node(s) were unschedulable"; }
    /* NodeName: pending pods have empty spec.nodeName => pass (node_name.go:79-86) */
    else if (!filter_taints(o, p, n)) { failed = PL_TAINT; *reason = "node(s) had untolerated taint(s)"; }
    else if ((p->node_selector.n > 0 || p->has_node_affinity || p->has_node_terms) && !filter_node_affinity(o, p, n)) {
        failed = PL_AFFINITY; *reason = "node(s) didn't match Pod's node affinity/selector";
    } else if (p->ports.n > 0 && !filter_ports(o, p, n)) {
        failed = PL_PORTS; *reason = "node(s) didn't have free ports for the requested pod ports";
    } else if (!filter_fit(o, p, n, reason)) { failed = PL_FIT; }
    else {
        const int pts = ts ? filter_pts(o, p, n, ts) : 0;
        if (pts) { failed = PL_PTS; *reason = pts == 1 ? REASON_PTS_CONSTRAINTS : REASON_PTS_LABEL; }
        else {
            const int ipa = filter_ipa(o, p, n, s);
            if (ipa <= 0) { failed = PL_IPA; *reason = ipa < 0 ? "node(s) didn't match pod affinity rules" : "node(s) didn't satisfy anti-affinity rules"; }
        }
    }
    o->last_fail_reason = failed ? *reason : NULL;
    if (failed) { if (plugin) *plugin = failed; return 0; }
    return 1;
}

/* ------------------------------------------------------------------------------------- */
/* SchedulerPluginRunner                                                                   */
/* ------------------------------------------------------------------------------------- */
/* lastIndexOrderMapping.At  CA/simulator/clustersnapshot/scheduling_opts.go:54-59 */
int orc_last_index_at(int i, int offset, int last_index, int n) {
    if (n == 0) return -1;
    return (i + offset + last_index) % n;
}

/* RunFiltersUntilPassingNode  CA/simulator/clustersnapshot/predicate/plugin_runner.go:54-143,
 * parallelism 1.  accept_new_only mirrors Estimate's IsNodeAcceptable (binpacking_estimator.go:172-174). */
static uint64_t splitmix64_next(uint64_t* x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int run_filters_until_passing_ex(orc* o, int pod, int accept_new_only, int skip_idx, int* last_index) {
    const podspec* p = &o->pods.v[pod];
    ipa_state st; ipa_prefilter(o, p, &st);
    pts_state ts; pts_prefilter(o, p, &ts);
    int n = o->snap.n, found = -1, found_pos = -1;
    /* order-independence probe: the reference's stores enumerate nodes by Go map iteration, a fresh
     * order for every attempt (store/basic.go:41-47, store/delta.go:143-145); lastIndex is a position in THAT list */
    int* perm = NULL;
    if (o->shuffle && n > 0) {
        perm = malloc(sizeof(int) * (size_t)n);
        for (int i = 0; i < n; ++i) perm[i] = i;
        for (int i = n - 1; i > 0; --i) { int j = (int)(splitmix64_next(&o->shuffle) % (uint64_t)(i + 1)); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
    }
    for (int i = 0; i < n; ++i) {
        int pos = orc_last_index_at(i, 1, *last_index, n);
        if (pos < 0) break;
        int idx = perm ? perm[pos] : pos;
        found_pos = pos;
        const node* nd = &o->snap.v[idx];
        if (nd->unschedulable) continue;                    /* :108-110 */
        if (accept_new_only && !nd->is_new) continue;       /* :114 */
        if (idx == skip_idx) continue;
        if (run_filter_plugins(o, p, nd, &st, &ts, NULL, NULL)) { found = idx; break; }
    }
    ipa_free(&st); pts_free(&ts);
    free(perm);
    if (found >= 0) *last_index = found_pos;                /* MarkMatch :138 (position in the list of this attempt) */
    return found;
}
static int run_filters_until_passing(orc* o, int pod, int accept_new_only, int* last_index) {
    return run_filters_until_passing_ex(o, pod, accept_new_only, -1, last_index);
}
/* RunFiltersOnNode  plugin_runner.go:146-181 */
static int run_filters_on_node(orc* o, int pod, int idx, const char** plugin, const char** reason) {
    const podspec* p = &o->pods.v[pod];
    ipa_state st; ipa_prefilter(o, p, &st);
    pts_state ts; pts_prefilter(o, p, &ts);
    int ok = run_filter_plugins(o, p, &o->snap.v[idx], &st, &ts, plugin, reason);
    ipa_free(&st); pts_free(&ts);
    return ok;
}
int orc_run_filters_on_snapshot_node(orc* o, int index, int pod, const char** plugin_out, const char** reason_out) {
    if (index < 0 || index >= o->snap.n) return -1;
    PODCHK(o, pod);
    return run_filters_on_node(o, pod, index, plugin_out, reason_out);
}
int orc_run_filters_until_passing(orc* o, int pod, int* last_index) {
    PODCHK(o, pod);
    return run_filters_until_passing(o, pod, 0, last_index);
}
/* RunFiltersUntilPassingNode (plugin_runner.go:54-143) under an ARBITRARY NodeOrderMapping, given as data: At(i) = order[i] for
 * i < n_order, -1 beyond (a mapping may end the walk early: checkNode :93-97 cancels).  `acceptable` (may be NULL) is
 * SchedulingOptions.IsNodeAcceptable per snapshot index; visited_out receives the nodes IsNodeAcceptable was asked about, in order
 * (:108-116: after the Unschedulable short-circuit).  The earliest step that passes wins (:126-133 keeps the smallest i whatever the
 * goroutines' timing).  No production caller passes a NodeOrdering (the estimator and TrySchedulePods use the runner's lastIndex
 * mapping); the function exists so that the reference's three tests of the mapping contract (plugin_runner_test.go:296-446) pin the
 * oracle's loop. */
int orc_run_filters_until_passing_ordered(orc* o, int pod, const int* order, int n_order, const unsigned char* acceptable,
                                          int* visited_out, int* n_visited_out) {
    PODCHK(o, pod);
    const podspec* p = &o->pods.v[pod];
    ipa_state st; ipa_prefilter(o, p, &st);
    pts_state ts; pts_prefilter(o, p, &ts);
    int n = o->snap.n, found = -1, nv = 0;
    for (int i = 0; i < n; ++i) {                           /* ParallelizeUntil over len(nodeInfosList) steps */
        const int idx = i < n_order ? order[i] : -1;        /* nodeOrdering.At(i) */
        if (idx < 0 || idx >= n) break;                     /* :94-97 */
        const node* nd = &o->snap.v[idx];
        if (nd->unschedulable) continue;                    /* :108-110 */
        if (acceptable) { if (visited_out) visited_out[nv] = idx; nv++; if (!acceptable[idx]) continue; }   /* :114-116 */
        if (run_filter_plugins(o, p, nd, &st, &ts, NULL, NULL)) { found = idx; break; }
    }
    ipa_free(&st); pts_free(&ts);
    if (n_visited_out) *n_visited_out = nv;
    return found;
}

/* insufficientResources of the last NodeResourcesFit failure (fit.go:678-765 collects ALL of them; the Status carries one
 * reason per entry): bit 0 "Too many pods", bit 1 + r "Insufficient <lane r>" */
unsigned orc_last_fit_reasons(void) { return g_last_fit_mask; }

/* ------------------------------------------------------------------------------------- */
/* limiter + thresholds                                                                    */
/* ------------------------------------------------------------------------------------- */
/* getMinLimit  CA/estimator/threshold_based_limiter.go:45-53 */
int64_t orc_get_min_limit(int64_t base, int64_t target) {
    if (base < 0 || target < 0) return -1;
    if ((base == 0 || base > target) && target > 0) return target;
    return base;
}
/* StartEstimation :34-43 (duration part disabled: oracle runs with maxDuration 0) */
void orc_limiter_start(orc_limiter* l, int n, const int* node_limits) {
    l->nodes = 0; l->max_nodes = 0;
    for (int i = 0; i < n; ++i) l->max_nodes = (int)orc_get_min_limit(l->max_nodes, node_limits[i]);
}
/* PermissionToAddNode :57-69 */
int orc_limiter_permission(orc_limiter* l) {
    if (l->max_nodes < 0 || (l->max_nodes > 0 && l->nodes >= l->max_nodes)) return 0;
    l->nodes++;
    return 1;
}
/* sngCapacityThreshold  CA/estimator/sng_capacity_threshold.go:34-59 */
int orc_sng_capacity_limit(int has_context, int n, const int* max_size, const int* target_size) {
    if (!has_context) return 0;
    int total = 0;
    for (int i = 0; i < n; ++i) { int c = max_size[i] - target_size[i]; if (c > 0) total += c; }
    if (total <= 0) return -1;
    return total;
}
/* clusterCapacityThreshold  CA/estimator/cluster_capacity_threshold.go:33-41 */
int orc_cluster_capacity_limit(int has_context, int cluster_max, int current_nodes) {
    if (!has_context || cluster_max == 0) return 0;
    if (cluster_max < 0 || cluster_max <= current_nodes) return -1;
    return cluster_max - current_nodes;
}

/* ------------------------------------------------------------------------------------- */
/* orderer + fastpath chooser                                                              */
/* ------------------------------------------------------------------------------------- */
/* calculatePodScore  CA/estimator/decreasing_pod_orderer.go:64-88 */
double orc_pod_score(int64_t cpu_req, int64_t mem_req, int64_t cpu_alloc, int64_t mem_alloc) {
    double score = 0;
    if (cpu_alloc > 0) score += (double)cpu_req / (double)cpu_alloc;
    if (mem_alloc > 0) score += (double)mem_req / (double)mem_alloc;
    return score;
}
/* Order :46-62; ties keep input order (canonical rule, see header) */
void orc_order(int n, const int64_t* cpu_req, const int64_t* mem_req, const uint8_t* has_exemplar,
               int64_t cpu_alloc, int64_t mem_alloc, int32_t* order_out) {
    double* score = malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        score[i] = (has_exemplar && !has_exemplar[i]) ? 0.0 : orc_pod_score(cpu_req[i], mem_req[i], cpu_alloc, mem_alloc);
        order_out[i] = i;
    }
    for (int i = 1; i < n; ++i) { /* stable insertion sort, descending */
        int32_t x = order_out[i]; int j = i - 1;
        while (j >= 0 && score[order_out[j]] < score[x]) { order_out[j + 1] = order_out[j]; --j; }
        order_out[j + 1] = x;
    }
    free(score);
}
/* determineBestPEGToFastpath  CA/estimator/binpacking_estimator.go:433-473 */
int orc_best_fastpath_peg(int n, const int32_t* count, const double* cpu_req, const double* mem_req,
                          const uint8_t* aa_self_hostname, const uint8_t* fastpath_ok,
                          const uint8_t* has_requests, double cap_cpu, double cap_mem) {
    int max_saved = 0, best = -1;
    for (int i = 0; i < n; ++i) {
        if (count[i] == 0) continue; /* Exemplar() == nil */
        int by_aa = aa_self_hostname[i] ? count[i] : 0, by_cpu = 0, by_mem = 0;
        if (has_requests[i]) {
            by_cpu = (int)ceil((double)count[i] * cpu_req[i] / cap_cpu);
            by_mem = (int)ceil((double)count[i] * mem_req[i] / cap_mem);
        }
        int nodes = by_aa; if (by_cpu > nodes) nodes = by_cpu; if (by_mem > nodes) nodes = by_mem;
        int saved = 0;
        if (nodes > 0) saved = count[i] - count[i] / nodes;
        if (saved >= max_saved && fastpath_ok[i]) { best = i; max_saved = saved; }
    }
    return best;
}

/* ------------------------------------------------------------------------------------- */
/* Estimate                                                                                */
/* ------------------------------------------------------------------------------------- */
typedef struct { int idx; node copy; } saved_node;
typedef struct {
    orc* o;
    int template_node;
    int new_index;          /* estimationState.newNodeNameIndex */
    int last_node;          /* snapshot index of lastNodeName, -1 = "" */
    orc_limiter limiter;
    int last_index;         /* runner.defaultNodeOrdering.lastIndex */
    int fake_nodes;         /* fastpath fake nodes with pods */
    int fork_len;           /* snapshot length at Fork */
    VEC(saved_node) saved;  /* forked-snapshot nodes that received pods (hostname-spread retry only) */
    int scheduled;
    int64_t cpu_sum, mem_sum;
} est_state;

/* The request totals of an Estimate are two's-complement sums (tables whose requests add up past INT64_MAX wrap, here and in the
 * kernels alike; signed overflow itself is undefined in C: found by tests/tools/sanitize_cpu.sh). */
static int64_t wrap_madd(int64_t acc, int64_t k, int64_t q) { return (int64_t)((uint64_t)acc + (uint64_t)k * (uint64_t)q); }

/* labelSelectorMatches(term.LabelSelector, exemplar.Labels) && key == hostname
 * binpacking_estimator.go:444-450 */
static int peg_aa_self_hostname(const orc* o, const podspec* p) {
    for (int t = 0; t < p->anti_terms.n; ++t) {
        const aff_term* term = &p->anti_terms.v[t];
        if (term->topology_key == o->id_hostname && selector_matches(o, &term->selector, p->labels.v, p->labels.n)) return 1;
    }
    return 0;
}
/* shouldUseFastPath :411-425 / hasNonHostnamePodAntiAffinity :399-409 */
static int peg_fastpath_ok(const orc* o, const podspec* p) {
    if (p->has_topology_spread) return 0;
    for (int t = 0; t < p->anti_terms.n; ++t) if (p->anti_terms.v[t].topology_key != o->id_hostname) return 0;
    return 1;
}

/* addNewNodeToSnapshot :326-342 + SanitizedNodeInfo CA/simulator/node_info_utils.go:93-137 */
static void add_new_node(est_state* s) {
    orc* o = s->o;
    const node* tmpl = &o->nodes.v[s->template_node];
    node nn = node_clone(tmpl); /* DeepCopy; preloaded pods are copied with it (:111-118) */
    char buf[512];
    snprintf(buf, sizeof buf, "%s-e-%d", o->st.s[tmpl->name], s->new_index);
    nn.name = intern(&o->st, buf);
    node_set_label(&nn, o->id_hostname, nn.name); /* :130 */
    nn.is_new = 1; nn.new_pods = 0;
    VEC_PUSH(o->snap, nn);
    s->new_index++;
    s->last_node = o->snap.n - 1;
}
/* SchedulePod success path: createPodInfo + StorePodInfo  predicate_snapshot.go:281-286,
 * then estimationState.trackScheduledPod  binpacking_estimator.go:58-61 */
static void commit(est_state* s, int pod, int node_idx) {
    orc* o = s->o;
    if (node_idx < s->fork_len && o->snap.v[node_idx].new_pods == 0) {
        /* first pod of this Estimate on a node of the forked snapshot: keep the pre-fork NodeInfo for Revert */
        saved_node sv; sv.idx = node_idx; sv.copy = node_clone(&o->snap.v[node_idx]);
        VEC_PUSH(s->saved, sv);
    }
    node_add_pod(o, &o->snap.v[node_idx], pod);
    o->snap.v[node_idx].new_pods++;
    s->scheduled++;
    s->cpu_sum = wrap_madd(s->cpu_sum, 1, o->pods.v[pod].req[0]);
    s->mem_sum = wrap_madd(s->mem_sum, 1, o->pods.v[pod].req[1]);
}

/* tryToScheduleOnExistingNodes :163-186; returns index of first unscheduled pod */
static int try_existing(est_state* s, int pod, int count) {
    int index;
    for (index = 0; index < count; ++index) {
        int nd = run_filters_until_passing(s->o, pod, 1, &s->last_index);
        if (nd < 0) break;
        commit(s, pod, nd);
    }
    return index;
}
/* tryToScheduleOnNewNodes :190-269; returns newNodesAvailable; *placed += pods scheduled.
 * (the hostname-topology-spread retry :212-227 is outside the encoded subset) */
static int pod_uses_hostname_spread(const orc* o, const podspec* p) {  /* isPodUsingHostNameTopologyKey :358-369 */
    for (int c = 0; c < p->spread.n; ++c) if (p->spread.v[c].topology_key == o->id_hostname) return 1;
    return 0;
}
static int try_new_nodes(est_state* s, int pod, int count, int* placed) {
    orc* o = s->o;
    for (int i = 0; i < count; ++i) {
        int found = 0;
        if (s->last_node >= 0) {
            const char* reason = NULL;
            if (run_filters_on_node(o, pod, s->last_node, NULL, &reason)) { found = 1; commit(s, pod, s->last_node); (*placed)++; }
            else if (pod_uses_hostname_spread(o, &o->pods.v[pod]) && reason == REASON_PTS_CONSTRAINTS) {
                /* :212-227 the last node is full for the hostname spread: any OTHER node of the snapshot — new or
                 * already in the cluster — may take the pod (IsNodeAcceptable only excludes lastNodeName) */
                int nd = run_filters_until_passing_ex(o, pod, 0, s->last_node, &s->last_index);
                if (nd >= 0) { found = 1; commit(s, pod, nd); (*placed)++; }
            }
        }
        if (!found) {
            if (s->last_node >= 0 && o->snap.v[s->last_node].new_pods == 0) return 1; /* :234-236 */
            if (!orc_limiter_permission(&s->limiter)) return 0;                       /* :244-246 */
            add_new_node(s);                                                            /* :249 */
            if (!run_filters_on_node(o, pod, s->last_node, NULL, NULL)) break;         /* :257-263 */
            commit(s, pod, s->last_node); (*placed)++;
        }
    }
    return 1;
}
/* tryFastPath :274-324 */
static int try_fast_path(est_state* s, int pod, int count, int* placed) {
    orc* o = s->o;
    if (count == 0) return 1;
    if (!orc_limiter_permission(&s->limiter)) return 0;
    add_new_node(s);
    int i = 0;
    for (; i < count; ++i) {
        if (!run_filters_on_node(o, pod, s->last_node, NULL, NULL)) break;
        commit(s, pod, s->last_node); (*placed)++;
    }
    int per_node = i;
    if (per_node == 0) return 1;
    int size = count / per_node;
    if (per_node * size < count) size++;
    for (int j = 1; j < size; ++j) {
        if (!orc_limiter_permission(&s->limiter)) return 0;
        int k = per_node < count - i ? per_node : count - i;
        /* trackScheduledPod(pods[i+k], fakeNodeName): counted, not simulated */
        s->fake_nodes++;
        s->scheduled += k; *placed += k;
        s->cpu_sum = wrap_madd(s->cpu_sum, k, o->pods.v[pod].req[0]);
        s->mem_sum = wrap_madd(s->mem_sum, k, o->pods.v[pod].req[1]);
        i += k;
    }
    return 1;
}

int orc_estimate(orc* o, int template_node, int n_pegs, const int32_t* peg_pod, const int32_t* peg_count,
                 int max_nodes, int last_index, int fastpath, orc_estimate_result* out) {
    NODECHK(o, template_node);
    for (int i = 0; i < n_pegs; ++i) { PODCHK(o, peg_pod[i]); if (peg_count[i] < 0) return -1; }
    const node* tmpl = &o->nodes.v[template_node];
    o->filter_runs = 0;

    est_state s; memset(&s, 0, sizeof s);
    s.o = o; s.template_node = template_node; s.last_node = -1; s.last_index = last_index;
    /* limiter.StartEstimation :109 with one static threshold (maxNodes, duration 0) */
    orc_limiter_start(&s.limiter, 1, &max_nodes);

    /* podOrderer.Order :112 */
    int64_t* cpu = malloc(sizeof(int64_t) * (size_t)(n_pegs + 1));
    int64_t* mem = malloc(sizeof(int64_t) * (size_t)(n_pegs + 1));
    uint8_t* has = malloc((size_t)(n_pegs + 1));
    for (int i = 0; i < n_pegs; ++i) { cpu[i] = o->pods.v[peg_pod[i]].req[0]; mem[i] = o->pods.v[peg_pod[i]].req[1]; has[i] = peg_count[i] > 0; }
    orc_order(n_pegs, cpu, mem, has, tmpl->alloc[0], tmpl->alloc[1], out->order);

    /* fastpath: move the best PEG last :114-124 */
    int use_fast_last = 0;
    if (fastpath && n_pegs > 0) {
        int32_t* cnt = malloc(sizeof(int32_t) * (size_t)n_pegs);
        double* fc = malloc(sizeof(double) * (size_t)n_pegs); double* fm = malloc(sizeof(double) * (size_t)n_pegs);
        uint8_t* aa = malloc((size_t)n_pegs); uint8_t* ok = malloc((size_t)n_pegs); uint8_t* hr = malloc((size_t)n_pegs);
        for (int k = 0; k < n_pegs; ++k) {
            const podspec* p = &o->pods.v[peg_pod[out->order[k]]];
            cnt[k] = peg_count[out->order[k]]; fc[k] = p->fp_cpu; fm[k] = p->fp_mem; hr[k] = (uint8_t)p->fp_has_requests;
            aa[k] = (uint8_t)peg_aa_self_hostname(o, p); ok[k] = (uint8_t)peg_fastpath_ok(o, p);
        }
        /* Capacity.Cpu().AsApproximateFloat64(): milli value * 10^-3 (quantity.go:468-483) */
        int best = orc_best_fastpath_peg(n_pegs, cnt, fc, fm, aa, ok, hr, tmpl->fp_cap_cpu, tmpl->fp_cap_mem);
        if (best != -1) {
            int32_t b = out->order[best];
            for (int k = best; k + 1 < n_pegs; ++k) out->order[k] = out->order[k + 1];
            out->order[n_pegs - 1] = b;
            use_fast_last = 1;
        }
        free(cnt); free(fc); free(fm); free(aa); free(ok); free(hr);
    }
    free(cpu); free(mem); free(has);

    /* clusterSnapshot.Fork :126 */
    int fork_len = o->snap.n;
    s.fork_len = fork_len;

    int more = 1;
    for (int k = 0; k < n_pegs; ++k) {
        int pg = out->order[k], pod = peg_pod[pg], count = peg_count[pg];
        int done = try_existing(&s, pod, count);             /* :137 */
        int placed = done;
        if (more) {
            if (k == n_pegs - 1 && use_fast_last) more = try_fast_path(&s, pod, count - done, &placed);   /* :146 */
            else more = try_new_nodes(&s, pod, count - done, &placed);                                    /* :148 */
        }
        out->placed[k] = placed;
    }

    int with_pods = 0, added = 0;
    for (int i = fork_len; i < o->snap.n; ++i) {
        if (out->node_pods && added < out->node_pods_cap) out->node_pods[added] = o->snap.v[i].new_pods;
        added++;
        if (o->snap.v[i].new_pods > 0) with_pods++;
    }
    /* trackScheduledPod (:58-61) also records a node of the forked snapshot that took a pod in the retry above */
    with_pods += s.saved.n;
    out->node_count = with_pods + s.fake_nodes;  /* len(newNodesWithPods) :160 */
    out->pods_scheduled = s.scheduled;
    out->nodes_added = added;
    out->limiter_nodes = s.limiter.nodes;
    out->last_index_out = s.last_index;
    out->internal_error = 0;
    out->req_cpu_sum = s.cpu_sum; out->req_mem_sum = s.mem_sum;
    out->filter_runs = o->filter_runs;

    /* clusterSnapshot.Revert :127-129 */
    for (int i = fork_len; i < o->snap.n; ++i) node_free(&o->snap.v[i]);
    o->snap.n = fork_len;
    for (int i = 0; i < s.saved.n; ++i) {
        node_free(&o->snap.v[s.saved.v[i].idx]);
        o->snap.v[s.saved.v[i].idx] = s.saved.v[i].copy;
    }
    VEC_FREE(s.saved);
    return 0;
}

/* SchedulablePodGroups' CheckPredicates  CA/core/scaleup/orchestrator/orchestrator.go:542-552:
 * Fork; AddNodeInfo(template itself); CheckPredicates(exemplar, template name); Revert */
int orc_check_predicates(orc* o, int template_node, int pod, const char** plugin_out, const char** reason_out) {
    NODECHK(o, template_node); PODCHK(o, pod);
    node c = node_clone(&o->nodes.v[template_node]);
    VEC_PUSH(o->snap, c);
    int ok = run_filters_on_node(o, pod, o->snap.n - 1, plugin_out, reason_out);
    node_free(&o->snap.v[o->snap.n - 1]);
    o->snap.n--;
    return ok;
}

/* One whole scale-up simulation: the node-group loop of ScaleUpOrchestrator.ScaleUp
 * (CA/core/scaleup/orchestrator/orchestrator.go:161-186 -> SchedulablePodGroups :535-570 -> ComputeExpansionOption
 * :383-427 -> Estimate) restated in one call, so that the CPU baseline is not charged for per-call binding overhead:
 * per node group i, the PEGs whose exemplar passes CheckPredicates on the fresh template (:552), in input order, go
 * through orc_estimate.  Per-group scalars land in out[i]; order / placed of group i in order_out / placed_out at
 * [i * n_pegs, ...) (positions index the group's own schedulable list, whose PEG ids are in sched_out at the same
 * offset, n_sched_out[i] entries). */
/* chain != 0: the loop as the orchestrator really runs it on ONE snapshot — the plugin runner's lastIndex (plugin_runner.go:138 MarkMatch on
 * the runner's defaultNodeOrdering; the runner belongs to the snapshot, predicate_snapshot.go:64) is never reverted by the Fork / Revert around an
 * Estimate, so group i starts from the lastIndex group i - 1 left behind; last_index[0] is the loop's starting value, the other entries are
 * ignored.  chain == 0: every Estimate starts from its own last_index entry (independent calls). */
static int scale_up_simulation(orc* o, int chain, int n_groups, const int32_t* template_node, int n_pegs, const int32_t* peg_pod,
                            const int32_t* peg_count, const int32_t* max_nodes, const int32_t* last_index,
                            orc_estimate_result* out, int32_t* n_sched_out, int32_t* sched_out, int32_t* order_out,
                            int32_t* placed_out, int64_t* filter_runs_out) {
    int32_t carried = n_groups > 0 ? last_index[0] : 0;
    int32_t* pods = malloc(sizeof(int32_t) * (size_t)(n_pegs + 1));
    int32_t* cnts = malloc(sizeof(int32_t) * (size_t)(n_pegs + 1));
    int64_t runs = 0;
    int rc = 0;
    for (int i = 0; i < n_groups && rc == 0; ++i) {
        int32_t* ids = sched_out + (size_t)i * (size_t)n_pegs;
        int n = 0;
        for (int g = 0; g < n_pegs; ++g) {
            if (peg_count[g] <= 0) continue;                       /* Exemplar() == nil */
            o->filter_runs = 0;
            if (orc_check_predicates(o, template_node[i], peg_pod[g], NULL, NULL) == 1) { ids[n] = g; pods[n] = peg_pod[g]; cnts[n] = peg_count[g]; n++; }
            runs += 1;
        }
        n_sched_out[i] = n;
        memset(&out[i], 0, sizeof out[i]);
        out[i].order = order_out + (size_t)i * (size_t)n_pegs; out[i].placed = placed_out + (size_t)i * (size_t)n_pegs;
        rc = orc_estimate(o, template_node[i], n, pods, cnts, max_nodes[i], chain ? carried : last_index[i], 0, &out[i]);
        carried = out[i].last_index_out;
        runs += out[i].filter_runs;
    }
    free(pods); free(cnts);
    if (filter_runs_out) *filter_runs_out = runs;
    return rc;
}
int orc_scale_up_simulation(orc* o, int n_groups, const int32_t* template_node, int n_pegs, const int32_t* peg_pod,
                            const int32_t* peg_count, const int32_t* max_nodes, const int32_t* last_index,
                            orc_estimate_result* out, int32_t* n_sched_out, int32_t* sched_out, int32_t* order_out,
                            int32_t* placed_out, int64_t* filter_runs_out) {
    return scale_up_simulation(o, 0, n_groups, template_node, n_pegs, peg_pod, peg_count, max_nodes, last_index, out, n_sched_out, sched_out, order_out,
                               placed_out, filter_runs_out);
}
int orc_scale_up_simulation_chained(orc* o, int n_groups, const int32_t* template_node, int n_pegs, const int32_t* peg_pod,
                                    const int32_t* peg_count, const int32_t* max_nodes, const int32_t* last_index,
                                    orc_estimate_result* out, int32_t* n_sched_out, int32_t* sched_out, int32_t* order_out,
                                    int32_t* placed_out, int64_t* filter_runs_out) {
    return scale_up_simulation(o, 1, n_groups, template_node, n_pegs, peg_pod, peg_count, max_nodes, last_index, out, n_sched_out, sched_out, order_out,
                               placed_out, filter_runs_out);
}

/* ------------------------------------------------------------------------------------- */
/* HintingSimulator.TrySchedulePods  (filter-out-schedulable)                              */
/* ------------------------------------------------------------------------------------- */
int orc_snapshot_size(const orc* o) { return o->snap.n; }

/* pre-image of a snapshot node about to be modified inside a forked region (see orc_simulate_node_removals) */
static void undo_touch(orc* o, int idx) {
    if (!o->undo_on) return;
    const int id = o->snap.v[idx].orig_id;
    for (int i = 0; i < o->undo.n; ++i) if (o->undo.v[i].orig_id == id) return;
    if (o->undo.n == o->undo.cap) { o->undo.cap = o->undo.cap ? o->undo.cap * 2 : 16; o->undo.v = realloc(o->undo.v, sizeof(*o->undo.v) * (size_t)o->undo.cap); }
    o->undo.v[o->undo.n].orig_id = id;
    o->undo.v[o->undo.n].copy = node_clone(&o->snap.v[idx]);
    o->undo.n++;
}

/* SimilarPodsScheduling  CA/simulator/scheduling/similar_pods.go:50-97: per controller at most 10
 * remembered (spec, labels) entries; here an entry is a pod spec id */
typedef struct { int key; int specs[10]; int n; } similar_entry;

/* TrySchedulePods  CA/simulator/scheduling/hinting_simulator.go:53-82 with
 * tryScheduleUsingHints :86-110 and trySchedule :114-135 */
int orc_try_schedule_pods(orc* o, int n_pods, const int32_t* pod, const int32_t* hint, const int32_t* similar_key,
                          const uint8_t* acceptable, int break_on_failure, int* last_index, int32_t* node_out) {
    VEC(similar_entry) memo; memset(&memo, 0, sizeof memo);
    int scheduled = 0;
    for (int i = 0; i < n_pods; ++i) node_out[i] = -1;
    for (int i = 0; i < n_pods; ++i) {
        const int p = pod[i];
        if (p < 0 || p >= o->pods.n) { VEC_FREE(memo); return -1; }
        int node_idx = -1;
        /* tryScheduleUsingHints: hinted node still in the cluster, acceptable, and SchedulePod passes */
        if (hint && hint[i] >= 0 && hint[i] < o->snap.n && (!acceptable || acceptable[hint[i]])) {
            if (run_filters_on_node(o, p, hint[i], NULL, NULL)) node_idx = hint[i];
        }
        if (node_idx < 0) {
            /* trySchedule: similar pod already found unschedulable? */
            int similar_unschedulable = 0;
            const int key = similar_key ? similar_key[i] : -1;
            similar_entry* ent = NULL;
            if (key >= 0) {
                for (int m = 0; m < memo.n; ++m) if (memo.v[m].key == key) { ent = &memo.v[m]; break; }
                if (ent) for (int s = 0; s < ent->n; ++s) if (ent->specs[s] == p) similar_unschedulable = 1;
            }
            if (!similar_unschedulable) {
                /* SchedulePodOnAnyNodeMatching(pod, opts): every acceptable node, cyclic from lastIndex + 1 */
                const podspec* ps = &o->pods.v[p];
                ipa_state st; ipa_prefilter(o, ps, &st);
                pts_state ts; pts_prefilter(o, ps, &ts);
                const int n = o->snap.n;
                for (int k = 0; k < n; ++k) {
                    const int idx = orc_last_index_at(k, 1, *last_index, n);
                    if (idx < 0) break;
                    const node* nd = &o->snap.v[idx];
                    if (nd->unschedulable) continue;
                    if (acceptable && !acceptable[idx]) continue;
                    if (run_filter_plugins(o, ps, nd, &st, &ts, NULL, NULL)) { node_idx = idx; break; }
                }
                ipa_free(&st); pts_free(&ts);
                if (node_idx >= 0) *last_index = node_idx;
                else if (key >= 0) { /* SetUnschedulable :80-97 */
                    if (!ent) { similar_entry e; memset(&e, 0, sizeof e); e.key = key; VEC_PUSH(memo, e); ent = &memo.v[memo.n - 1]; }
                    if (ent->n < 10) ent->specs[ent->n++] = p;
                }
            }
        }
        if (node_idx >= 0) {
            undo_touch(o, node_idx);
            node_add_pod(o, &o->snap.v[node_idx], p);   /* SchedulePod commits the pod to the snapshot */
            node_out[i] = node_idx;
            scheduled++;
        } else if (break_on_failure) break;
    }
    VEC_FREE(memo);
    return scheduled;
}

/* ------------------------------------------------------------------------------------- */
/* scale-down: RemovalSimulator.SimulateNodeRemoval in planner order         */
/* ------------------------------------------------------------------------------------- */
static int snap_pos_of(const orc* o, int orig_id) {
    for (int i = 0; i < o->snap.n; ++i) if (o->snap.v[i].orig_id == orig_id) return i;
    return -1;
}
/* Planner.categorizeNodes loop  CA/core/scaledown/planner/planner.go:300-330 around
 * RemovalSimulator.SimulateNodeRemoval  CA/simulator/cluster.go:131-172 and findPlaceFor :190-231:
 *   per candidate, in order: Fork; the candidate becomes a pod-less tainted ghost (replaceWithTaintedGhostNode
 *   :243-265; canonical list order: the ghost keeps the candidate's list position); TrySchedulePods(its pods,
 *   breakOnFailure = true, acceptable = destination && not the candidate); all pods placed => removable:
 *   with `persist` the moves are committed, the ghost leaves the list (RemoveNodeInfo :230) and the node
 *   leaves the destination set (planner.go:318); otherwise Revert.  lastIndex lives in the plugin runner and is
 *   never reverted.
 * Node ids are positions at orc_snapshot_add time.  pods / hints: per candidate pod_offsets[k]..pod_offsets[k+1].
 * Pods that an earlier committed removal moved onto a later candidate are appended to that candidate's list (what
 *   GetPodsToMove sees in the committed snapshot) and reported as ext entries (candidate, pod, destination); the
 *   loop stops in front of such a candidate (*n_processed < K) when an arrived pod is pod_sticky or the ext arrays
 *   (ext_capacity) are full — the protocol of casim_simulate_node_removals, whose caller re-submits the rest.
 * cand_atomic[k] (may be NULL): the node belongs to an atomically scaled group; it does not count toward max_removable.
 * removable_out[k]: 1 removable, 0 no place, 2 not evaluated.  node_out[i]: node the i-th listed pod was placed
 * on in ITS candidate's simulation (-1: not placed).  final_node_out[i] (may be NULL): where the pod is at the end
 * (its candidate id if it never moved for good). */
int orc_simulate_node_removals(orc* o, int K, const int32_t* cand_node, const int32_t* pod_offsets, const int32_t* pod,
                               const int32_t* hint, const uint8_t* destination, const uint8_t* pod_sticky, const uint8_t* cand_atomic, int persist,
                               int max_removable, int ext_capacity, int* last_index, uint8_t* removable_out, int32_t* node_out,
                               int32_t* ext_cand_out, int32_t* ext_pod_out, int32_t* ext_node_out, int* n_ext_out,
                               int32_t* final_node_out, int* n_processed) {
    int n_ext = 0;
    const int N0 = o->snap_added;
    const int total = pod_offsets[K];
    uint8_t* dest = malloc((size_t)(N0 > 0 ? N0 : 1));
    for (int i = 0; i < N0; ++i) dest[i] = destination ? destination[i] : 1;
    for (int k = 0; k < K; ++k) removable_out[k] = 2;
    for (int i = 0; i < total; ++i) node_out[i] = -1;
    int32_t* where = malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));   /* current node (orig id) of every listed pod */
    for (int i = 0; i < total; ++i) where[i] = -1;   /* (offsets that do not start at 0 leave no entry unset) */
    for (int k = 0; k < K; ++k) for (int i = pod_offsets[k]; i < pod_offsets[k + 1]; ++i) where[i] = cand_node[k];
    ivec moves, move_dest; memset(&moves, 0, sizeof moves); memset(&move_dest, 0, sizeof move_dest);   /* committed moves, in order */
    int removed = 0, counted = 0, k = 0;
    for (; k < K; ++k) {
        /* planner.go:306-310: len(removableList) - atomicScaleDownNodesCount >= unneededNodesLimit() */
        if (max_removable > 0 && counted >= max_removable) break;
        const int Y = cand_node[k];
        const int ypos = snap_pos_of(o, Y);
        if (ypos < 0) { removable_out[k] = 0; continue; }                  /* NoNodeInfo :139-147 */
        /* the candidate's list: its own pods, then pods moved onto it earlier */
        VEC(int) list; memset(&list, 0, sizeof list);
        for (int i = pod_offsets[k]; i < pod_offsets[k + 1]; ++i) VEC_PUSH(list, i);
        int arrived = 0;
        for (int i = 0; i < total; ++i) if (where[i] == Y && !(i >= pod_offsets[k] && i < pod_offsets[k + 1])) arrived++;
        if (arrived) {
            /* GetPodsToMove on the committed snapshot lists the arrived pods after the node's own, in arrival order
             * (NodeInfo.AddPod appends) == order of the committed moves */
            int sticky = 0;
            for (int j = 0; j < moves.n; ++j) if (where[moves.v[j]] == Y && move_dest.v[j] == Y) { VEC_PUSH(list, moves.v[j]); if (pod_sticky && pod_sticky[moves.v[j]]) sticky = 1; }
            if (sticky || n_ext + arrived > ext_capacity) { VEC_FREE(list); break; }   /* protocol: hand back to the caller */
        }
        /* Fork */
        o->undo_on = 1; o->undo.n = 0;
        node saved_y = o->snap.v[ypos];                                    /* moved, restored or freed below */
        node ghost = node_clone(&saved_y);
        ghost.pods.n = 0; ghost.used_ports.n = 0; ghost.n_pods_with_aa = 0;
        memset(ghost.requested, 0, sizeof ghost.requested);
        { taint t = {intern(&o->st, "ToBeDeletedByClusterAutoscaler"), intern(&o->st, "0"), o->id_noschedule}; VEC_PUSH(ghost.taints, t); }
        o->snap.v[ypos] = ghost;
        const int n = o->snap.n, np = list.n;
        int32_t* pp = malloc(sizeof(int32_t) * (size_t)(np > 0 ? np : 1));
        int32_t* hh = malloc(sizeof(int32_t) * (size_t)(np > 0 ? np : 1));
        int32_t* oo = malloc(sizeof(int32_t) * (size_t)(np > 0 ? np : 1));
        uint8_t* acc = malloc((size_t)n);
        for (int i = 0; i < n; ++i) acc[i] = (uint8_t)(i != ypos && dest[o->snap.v[i].orig_id]);   /* isCandidateNode :191-193 */
        for (int i = 0; i < np; ++i) {
            pp[i] = pod[list.v[i]];
            /* hints of the caller refer to original pods; a pod that was moved carries the hint of its last move = the
             * node it sits on = the candidate itself, which is not acceptable */
            const int h = (hint && where[list.v[i]] == cand_node[k] && list.v[i] >= pod_offsets[k] && list.v[i] < pod_offsets[k + 1]) ? hint[list.v[i]] : -1;
            hh[i] = h >= 0 ? snap_pos_of(o, h) : -1;
        }
        const int placed = orc_try_schedule_pods(o, np, pp, hh, NULL, acc, 1, last_index, oo);
        const int ok = placed == np;
        for (int i = 0; i < np; ++i) {
            const int d = oo[i] >= 0 ? o->snap.v[oo[i]].orig_id : -1;
            if (list.v[i] >= pod_offsets[k] && list.v[i] < pod_offsets[k + 1]) node_out[list.v[i]] = d;
            else { ext_cand_out[n_ext] = k; ext_pod_out[n_ext] = list.v[i]; ext_node_out[n_ext] = d; n_ext++; }
        }
        o->undo_on = 0;
        if (ok && persist) {
            for (int i = 0; i < np; ++i) {
                where[list.v[i]] = o->snap.v[oo[i]].orig_id;
                VEC_PUSH(moves, list.v[i]); VEC_PUSH(move_dest, where[list.v[i]]);
            }
            for (int i = 0; i < o->undo.n; ++i) node_free(&o->undo.v[i].copy);
            node_free(&saved_y); node_free(&o->snap.v[ypos]);
            for (int i = ypos; i + 1 < o->snap.n; ++i) o->snap.v[i] = o->snap.v[i + 1];   /* RemoveNodeInfo */
            o->snap.n--;
            dest[Y] = 0;
            removed++;
            if (!(cand_atomic && cand_atomic[k])) counted++;               /* atomicScaleDownNode :321-324 */
        } else {
            for (int i = 0; i < o->undo.n; ++i) {                          /* Revert */
                const int pos = snap_pos_of(o, o->undo.v[i].orig_id);
                node_free(&o->snap.v[pos]);
                o->snap.v[pos] = o->undo.v[i].copy;
            }
            node_free(&o->snap.v[ypos]);
            o->snap.v[ypos] = saved_y;
            if (ok) { removed++; if (!(cand_atomic && cand_atomic[k])) counted++; }
        }
        o->undo.n = 0;
        removable_out[k] = (uint8_t)(ok ? 1 : 0);
        free(pp); free(hh); free(oo); free(acc); VEC_FREE(list);
    }
    if (final_node_out) for (int i = 0; i < total; ++i) final_node_out[i] = where[i];
    *n_processed = k;
    *n_ext_out = n_ext;
    VEC_FREE(moves); VEC_FREE(move_dest);
    free(dest); free(where);
    return removed;
}

/* ------------------------------------------------------------------------------------- */
/* expander filters                                                                        */
/* ------------------------------------------------------------------------------------- */
/* leastnodes.BestOptions  CA/expander/leastnodes/leastnodes.go:35-61 */
int orc_least_nodes(int n, const int32_t* node_count, uint8_t* sel) {
    int least = INT_MAX, cnt = 0;
    memset(sel, 0, (size_t)n);
    for (int i = 0; i < n; ++i) {
        if (node_count[i] == 0) continue;
        if (node_count[i] == least) { sel[i] = 1; cnt++; continue; }
        if (node_count[i] < least) { least = node_count[i]; memset(sel, 0, (size_t)n); sel[i] = 1; cnt = 1; }
    }
    return cnt;
}
/* mostpods.BestOptions  CA/expander/mostpods/mostpods.go:33-53 */
int orc_most_pods(int n, const int32_t* pod_count, uint8_t* sel) {
    int maxp = 0, cnt = 0;
    memset(sel, 0, (size_t)n);
    for (int i = 0; i < n; ++i) {
        if (pod_count[i] == maxp) { sel[i] = 1; cnt++; continue; }
        if (pod_count[i] > maxp) { maxp = pod_count[i]; memset(sel, 0, (size_t)n); sel[i] = 1; cnt = 1; }
    }
    return cnt;
}
/* leastwaste.BestOptions  CA/expander/waste/waste.go:37-73 (statement order kept: the
 * equality append runs BEFORE the "nil or smaller" replacement) */
int orc_least_waste(int n, const int32_t* node_count, const int64_t* req_cpu, const int64_t* req_mem,
                    const int64_t* node_cpu, const int64_t* node_mem, const uint8_t* has_node_info, uint8_t* sel) {
    double least = 0; int have = 0, cnt = 0;
    memset(sel, 0, (size_t)n);
    for (int i = 0; i < n; ++i) {
        if (has_node_info && !has_node_info[i]) continue;
        int64_t avail_cpu = node_cpu[i] * (int64_t)node_count[i];
        int64_t avail_mem = node_mem[i] * (int64_t)node_count[i];
        double wasted_cpu = (double)(avail_cpu - req_cpu[i]) / (double)avail_cpu;
        double wasted_mem = (double)(avail_mem - req_mem[i]) / (double)avail_mem;
        double score = wasted_cpu + wasted_mem;
        if (score == least) { sel[i] = 1; cnt++; have = 1; }
        if (!have || score < least) { least = score; memset(sel, 0, (size_t)n); sel[i] = 1; cnt = 1; have = 1; }
    }
    return cnt;
}
