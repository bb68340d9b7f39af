/*
 * casim_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's scale-up simulation path, object level
 * (strings, per-pod loops, per-node Filter runs), used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP
 * engine.  Nothing under kubernetes_autoscaler_amd/ may link, import or call it.
 *
 * Parity pin: the reference itself cannot run here (no Go toolchain), so this
 * oracle is pinned against the reference's own known-answer tests, transcribed as data in
 * tests/golden/reference_vectors.json (TestBinpackingEstimate rows 1-8 — the three
 * PodTopologySpread rows included —, BenchmarkBinpackingEstimate 2595/51000, TestTrySchedulePods,
 * TestPodSchedulesOnHintedNode, TestDetermineBestPodEquivalenceGroupToFastpath,
 * TestPodPriorityProcessor, TestThresholdBasedLimiter, TestMinLimit, TestSngCapacityThreshold,
 * TestNewClusterCapacityThreshold, TestLastIndexOrderMapping, TestRunFiltersOnNode,
 * TestRunFilterUntilPassingNode, TestDebugInfo taints, TestLeastNodes, TestLeastWaste,
 * TestFilterOutSchedulable, TestSimulateNodeRemoval incl. its two ghost-node PodTopologySpread rows,
 * the planner's TestUpdateClusterState and TestUpdateClusterStatUnneededNodesLimit, TestTopologySpreadTaintScheduling).
 * Taints / nodeSelector / anti-affinity INSIDE Estimate have no reference known-answer test
 * ("parity unpinned" for those rows); they are restated from the vendored plugin
 * sources cited below.
 *
 * Canonical determinism rules: node list = insertion order
 * [existing..., e-0, e-1, ...]; predicate parallelism 1; limiter duration 0; lastIndex an
 * explicit input; PEG score ties broken by input position (Go's sort.Slice is unstable; it
 * is insertion sort = stable for <= 12 elements).
 *
 * All `CA/` paths: /root/reference/cluster-autoscaler/ ; `V/`: its vendor/k8s.io/.
 */
#ifndef CASIM_ORACLE_H_
#define CASIM_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_RES 8

typedef struct orc orc;

orc* orc_new(int n_res);
void orc_free(orc* o);

/* ---- pod specs (the scheduling-relevant part of an exemplar pod) ---------------------- */
int orc_pod(orc* o, const char* ns, const int64_t* req);
int orc_pod_label(orc* o, int pod, const char* key, const char* value);
int orc_pod_toleration(orc* o, int pod, const char* key, const char* op, const char* value,
                       const char* effect);
int orc_pod_node_selector(orc* o, int pod, const char* key, const char* value);
int orc_pod_node_affinity_req(orc* o, int pod, const char* key, const char* op,
                              const char* const* values, int n_values);
int orc_pod_host_port(orc* o, int pod, const char* ip, const char* protocol, int port);
/* namespaceSelector of an anti-affinity term (orc_term_*) or of a required affinity term (orc_aff_term_*) + the namespace lister it is
 * resolved against (key NULL = register only) */
int orc_aff_term_namespace_selector(orc* o, int pod, int term);
int orc_aff_term_namespace_requirement(orc* o, int pod, int term, const char* key, const char* op, const char* const* values, int n);
int orc_namespace_label(orc* o, const char* name, const char* key, const char* value);
int orc_term_namespace_selector(orc* o, int pod, int term);
int orc_term_namespace_requirement(orc* o, int pod, int term, const char* key, const char* op,
                                   const char* const* values, int n);
/* nodeSelectorTerms (ORed): open a term, then add its matchExpressions (is_field 0) / matchFields (is_field 1).
 * Exclusive with orc_pod_node_affinity_req on the same pod. */
int orc_pod_node_affinity_term(orc* o, int pod);
int orc_pod_node_term_req(orc* o, int pod, int term, int is_field, const char* key, const char* op,
                          const char* const* values, int n);
int orc_pod_anti_affinity_term(orc* o, int pod, const char* topology_key,
                               const char* const* namespaces, int n_namespaces);
/* required pod AFFINITY term (explicit namespaces; n = 0 => the pod's own) and one requirement of its label selector:
 * InterPodAffinity.PreFilter / Filter, V/.../interpodaffinity/filtering.go:234-272,382-409 */
int orc_pod_affinity_term(orc* o, int pod, const char* topology_key, const char* const* namespaces, int n_namespaces);
int orc_aff_term_requirement(orc* o, int pod, int term, const char* key, const char* op, const char* const* values, int n_values);
int orc_term_requirement(orc* o, int pod, int term, const char* key, const char* op,
                         const char* const* values, int n_values);
int orc_pod_fastpath_requests(orc* o, int pod, double cpu, double mem);
int orc_pod_has_topology_spread(orc* o, int pod, int flag); /* only feeds shouldUseFastPath */
/* DoNotSchedule topologySpreadConstraint (PodTopologySpread Filter, V/.../podtopologyspread/filtering.go);
 * min_domains <= 0 = nil.  (Device side: domain rules of K_sched / K_est; the template-mode packer delegates.) */
int orc_pod_spread_constraint(orc* o, int pod, int max_skew, const char* topology_key, int min_domains);
int orc_spread_requirement(orc* o, int pod, int constraint, const char* key, const char* op,
                           const char* const* values, int n_values);
int orc_spread_taints_policy_honor(orc* o, int pod, int constraint, int honor);  /* nodeTaintsPolicy: Honor */
int orc_spread_affinity_policy_ignore(orc* o, int pod, int constraint, int ignore);  /* nodeAffinityPolicy: Ignore */

/* ---- node objects (a template or a node of the existing cluster) ---------------------- */
int orc_node(orc* o, const char* name, const int64_t* alloc, int allowed_pods,
             int64_t capacity_cpu_milli, int64_t capacity_mem, int unschedulable);
/* override Capacity.{Cpu,Memory}().AsApproximateFloat64() used by the fastpath chooser */
int orc_node_fastpath_capacity(orc* o, int node, double cpu, double mem);
int orc_node_label(orc* o, int node, const char* key, const char* value);
int orc_node_taint(orc* o, int node, const char* key, const char* value, const char* effect);
int orc_node_add_pod(orc* o, int node, int pod);
/* put a copy of `node` into the cluster snapshot (insertion order is the list order) */
int orc_snapshot_add(orc* o, int node);
void orc_set_taint_comparison_ops(orc* o, int enabled);
/* seed != 0: every scheduling attempt of Estimate sees the node list in a freshly drawn order, like the reference's
 * map-backed stores — used to check that a workload's (NodeCount, Pods) does not depend on the order */
void orc_set_list_shuffle(orc* o, uint64_t seed);

/* ---- BinpackingNodeEstimator.Estimate  (CA/estimator/binpacking_estimator.go:102-161) ---- */
typedef struct orc_estimate_result {
    int32_t node_count;      /* len(newNodesWithPods)                     */
    int32_t pods_scheduled;  /* len(scheduledPods)                        */
    int32_t nodes_added;     /* nodes added to the snapshot               */
    int32_t limiter_nodes;   /* limiter.nodes at exit                     */
    int32_t last_index_out;  /* lastIndexOrderMapping.lastIndex at exit   */
    int32_t internal_error;  /* 1 => reference would return (0, nil)      */
    int64_t req_cpu_sum, req_mem_sum; /* over scheduled pods               */
    int64_t filter_runs;     /* number of RunFilterPlugins node visits (work counter) */
    int32_t* order;          /* [n_pegs] input PEG index processed k-th   */
    int32_t* placed;         /* [n_pegs] pods scheduled of that PEG       */
    int32_t* node_pods;      /* [node_pods_cap] pods placed per new node, or NULL */
    int32_t node_pods_cap;
} orc_estimate_result;

int orc_estimate(orc* o, int template_node, int n_pegs, const int32_t* peg_pod,
                 const int32_t* peg_count, int max_nodes, int last_index, int fastpath,
                 orc_estimate_result* out);

/* One whole scale-up simulation (node-group loop of ScaleUpOrchestrator.ScaleUp, orchestrator.go:161-186, :535-570,
 * :383-427): per group the PEGs passing CheckPredicates on the fresh template, then Estimate.  Arrays of group i start at
 * i * n_pegs in sched_out / order_out / placed_out; order / placed index the group's own schedulable list. */
int orc_scale_up_simulation(orc* o, int n_groups, const int32_t* template_node, int n_pegs, const int32_t* peg_pod,
                            const int32_t* peg_count, const int32_t* max_nodes, const int32_t* last_index,
                            orc_estimate_result* out, int32_t* n_sched_out, int32_t* sched_out, int32_t* order_out,
                            int32_t* placed_out, int64_t* filter_runs_out);
/* the same loop with lastIndex carried from group to group (the plugin runner's state survives every Estimate: plugin_runner.go:138) */
int orc_scale_up_simulation_chained(orc* o, int n_groups, const int32_t* template_node, int n_pegs, const int32_t* peg_pod,
                            const int32_t* peg_count, const int32_t* max_nodes, const int32_t* last_index,
                            orc_estimate_result* out, int32_t* n_sched_out, int32_t* sched_out, int32_t* order_out,
                            int32_t* placed_out, int64_t* filter_runs_out);

/* CheckPredicates(exemplar, template) as in SchedulablePodGroups (orchestrator.go:535-570):
 * returns 1 pass / 0 fail; *plugin_out (may be NULL) receives a static plugin name. */
int orc_check_predicates(orc* o, int template_node, int pod, const char** plugin_out,
                         const char** reason_out);
/* every reason of the last NodeResourcesFit failure: bit 0 "Too many pods", bit 1 + r "Insufficient <lane r>" */
unsigned orc_last_fit_reasons(void);
/* RunFiltersOnNode against snapshot node #index (plugin_runner.go:146-181) */
int orc_run_filters_on_snapshot_node(orc* o, int index, int pod, const char** plugin_out,
                                     const char** reason_out);
/* RunFiltersUntilPassingNode over the whole snapshot, every node acceptable
 * (plugin_runner.go:54-143); returns the matched list index or -1; updates *last_index. */
int orc_run_filters_until_passing(orc* o, int pod, int* last_index);
/* the same loop under an arbitrary NodeOrderMapping given as data (plugin_runner_test.go:296-446); returns the snapshot index found or -1 */
int orc_run_filters_until_passing_ordered(orc* o, int pod, const int* order, int n_order, const unsigned char* acceptable,
                                          int* visited_out, int* n_visited_out);

/* ---- HintingSimulator.TrySchedulePods  (CA/simulator/scheduling/hinting_simulator.go:53-135) ----
 * Pending pods against the nodes ALREADY in the snapshot (filter-out-schedulable).
 * pod[i]          pod spec of the i-th pending pod (processing order)
 * hint[i]         snapshot index of the hinted node or -1                     (hints.go)
 * similar_key[i]  >= 0: controller-equivalence key for SimilarPodsScheduling, -1: no controller; a cached entry is a pod
 *                 spec id, so pods with equal spec + labels must be passed with ONE id (the reference matches by
 *                 PodSpecSemanticallyEqual + DeepEqual(labels), similar_pods.go:48-50)
 * acceptable      [snapshot size] IsNodeAcceptable per node, or NULL = every node
 * node_out[i]     snapshot index the pod was scheduled on, -1 if it stays pending
 * Pods are committed to the snapshot (the caller forks; see orc_snapshot_truncate). Returns the
 * number of scheduled pods. */
int orc_try_schedule_pods(orc* o, int n_pods, const int32_t* pod, const int32_t* hint, const int32_t* similar_key,
                          const uint8_t* acceptable, int break_on_failure, int* last_index, int32_t* node_out);
/* number of nodes in the snapshot; drop nodes added after `n` (Revert) — pods committed to older
 * nodes stay, so tests build a fresh scenario per case */
int orc_snapshot_size(const orc* o);

/* ---- scale-down: Planner.categorizeNodes loop around RemovalSimulator.SimulateNodeRemoval (SURVEY §8 f4;
 * CA/core/scaledown/planner/planner.go:300-330, CA/simulator/cluster.go:131-265).  See the .c file for the
 * argument meaning.  Node ids = positions at orc_snapshot_add time.  Returns the number of removable nodes. */
int orc_simulate_node_removals(orc* o, int n_candidates, const int32_t* cand_node, const int32_t* pod_offsets,
                               const int32_t* pod, const int32_t* hint, const uint8_t* destination, const uint8_t* pod_sticky, const uint8_t* cand_atomic,
                               int persist, int max_removable, int ext_capacity, int* last_index, uint8_t* removable_out,
                               int32_t* node_out, int32_t* ext_cand_out, int32_t* ext_pod_out, int32_t* ext_node_out,
                               int* n_ext_out, int32_t* final_node_out, int* n_processed);

/* ---- pieces with their own reference unit tests ----------------------------------------- */
/* getMinLimit (threshold_based_limiter.go:45-53) */
int64_t orc_get_min_limit(int64_t base, int64_t target);
/* sngCapacityThreshold.NodeLimit (sng_capacity_threshold.go:34-59); arrays describe the
 * node group itself (index 0) followed by its similar node groups */
int orc_sng_capacity_limit(int has_context, int n, const int* max_size, const int* target_size);
/* clusterCapacityThreshold.NodeLimit (cluster_capacity_threshold.go:33-41) */
int orc_cluster_capacity_limit(int has_context, int cluster_max, int current_nodes);
/* limiter state machine, duration disabled */
typedef struct { int max_nodes; int nodes; } orc_limiter;
void orc_limiter_start(orc_limiter* l, int n_thresholds, const int* node_limits);
int orc_limiter_permission(orc_limiter* l);
/* lastIndexOrderMapping.At (scheduling_opts.go:54-59) */
int orc_last_index_at(int i, int offset, int last_index, int n);
/* DecreasingPodOrderer score (decreasing_pod_orderer.go:64-88) */
double orc_pod_score(int64_t cpu_req, int64_t mem_req, int64_t cpu_alloc, int64_t mem_alloc);
/* DecreasingPodOrderer.Order on raw numbers: order_out[k] = input index; ties keep input order */
void orc_order(int n, const int64_t* cpu_req, const int64_t* mem_req, const uint8_t* has_exemplar,
               int64_t cpu_alloc, int64_t mem_alloc, int32_t* order_out);
/* determineBestPEGToFastpath (binpacking_estimator.go:433-473) on raw numbers */
int orc_best_fastpath_peg(int n, const int32_t* count, const double* cpu_req, const double* mem_req,
                          const uint8_t* aa_self_hostname, const uint8_t* fastpath_ok,
                          const uint8_t* has_requests, double cap_cpu, double cap_mem);
/* expander filters; sel_out[i] = 1 if option i survives; returns survivors */
int orc_least_nodes(int n, const int32_t* node_count, uint8_t* sel_out);
int orc_most_pods(int n, const int32_t* pod_count, uint8_t* sel_out);
int orc_least_waste(int n, const int32_t* node_count, const int64_t* req_cpu, const int64_t* req_mem,
                    const int64_t* node_cpu, const int64_t* node_mem, const uint8_t* has_node_info,
                    uint8_t* sel_out);

#ifdef __cplusplus
}
#endif
#endif
