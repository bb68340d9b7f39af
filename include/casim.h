/*
 * casim.h — C ABI of libcasim: the MI355X-native scale-up simulation engine.
 *
 * This header is the drop-in boundary for ONE path of the Cluster Autoscaler
 * (all `CA/` paths are relative to /root/reference/cluster-autoscaler/, `V/` to its
 * vendor/k8s.io/):
 *
 *   estimator.Estimator.Estimate              CA/estimator/estimator.go:53-56
 *     -> BinpackingNodeEstimator.Estimate     CA/estimator/binpacking_estimator.go:102-161
 *   ClusterSnapshot.CheckPredicates           CA/simulator/clustersnapshot/clustersnapshot.go:96
 *     (batched, as used by SchedulablePodGroups CA/core/scaleup/orchestrator/orchestrator.go:535-570)
 *   expander.Filter.BestOptions               CA/expander/expander.go:59-62
 *
 * A cgo shim binds exactly these entry points:
 *   - the ENCODER  (casim_enc_*): turns pod / node-template objects (strings) into the flat
 *     integer + bitmask tables the device consumes.  It replaces the per-call string work
 *     inside the scheduler Filter plugins (V/kubernetes/pkg/scheduler/framework/plugins/...).
 *   - the ENGINE   (casim_ctx_*, casim_problem_*): uploads the tables to HBM and runs the
 *     hand-written HIP kernels for gfx950.
 *
 * Conventions: every function returns int32 status (0 = ok, <0 = internal error, never
 * throws / aborts across the ABI) unless it returns a handle (NULL on failure; see
 * casim_last_error).  All buffers are caller-owned, read-only for the callee unless named
 * `out`.  No torch / C++ types appear in any signature.
 */
#ifndef CASIM_H_
#define CASIM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CASIM_ABI_VERSION 12  /* 2: casim_removal_candidates.cand_atomic, casim_domain_rules.n_taint_policy_rules
                               * 3: casim_groups.{peg_lo,peg_hi,global_id,n_sims,sim_offsets}, casim_best_option_sims,
                               *    casim_feasibility_reasons, casim_estimate_batch_timed, casim_mctx_*, casim_cluster_*
                               * 4: casim_options.n_streams (sub-batches on internal HIP streams), casim_enc_group_pods,
                               *    casim_enc_add_grouped_pegs, casim_enc_pod_set_spec_extra
                               * 5: casim_options.pack_build, casim_pack_build_info (two builds of the register packer + self-check),
                               *    casim_problem_info [5], [6], casim_option_query.join_stream, casim_prefetch_*, casim_enc_begin_update / _group_reset / _refinalize / _group_rows
                               * 6: casim_pegs.zone_polarity (group bits of NEED polarity: required pod affinity towards a partner of the batch);
                               *    later, without a new number (layouts unchanged, zero keeps its meaning): casim_options.no_front_kernel in one of
                               *    the two reserved words, casim_problem_info [7]
                               * 7: casim_cluster_forget_commits, casim_options.winners_only (the last reserved word)
                               * 8: casim_pegs.excl_polarity (node bits of NEED polarity: hostname-level required pod affinity inside casim_estimate_batch)
                               * 9: casim_enc_lane / _pod_set_request / _group_set_allocatable / _lane_count / _lane_name (resources by name),
                               *    casim_options.chain_last_index (+ 3 reserved words), casim_problem_time_feasibility
                               * 10: casim_pegs.req32 / req_unit (requests narrowed by the caller), casim_last_removals_info
                               * 11: casim_pod_columns / casim_enc_add_pods (the pods of a loop — namespace, requests, labels, tolerations, nodeSelector,
                               *     PEG sizes — in ONE crossing over an interned string table)
                               * 12: casim_enc_group_set_allocatable opens no lane and casim_enc_pod_set_request opens none for a request of zero (a
                               *     column no pod reads is not a column: real nodes' hugepages-*: 0 / attachable-volumes-* widen no table);
                               *     casim_last_chain_info (long chains stop at their fixed point); CASIM_PREFETCH_MISS_LAST_INDEX */

/* Resource lanes.  Lane 0 = cpu in millicores (Quantity.MilliValue), lane 1 = memory bytes,
 * lane 2 = ephemeral-storage bytes, lanes 3.. = scalar / extended resources (Quantity.Value),
 * mirroring framework.Resource  V/kubernetes/pkg/scheduler/framework/types.go:989-998. */
#define CASIM_MAX_RES 8
#define CASIM_RES_CPU 0
#define CASIM_RES_MEM 1
#define CASIM_RES_EPHEMERAL 2

/* ---- status codes ------------------------------------------------------------------- */
#define CASIM_OK 0
#define CASIM_ERR_INVALID (-1)      /* bad argument / inconsistent table sizes            */
#define CASIM_ERR_HIP (-2)          /* a HIP runtime call failed (see casim_last_error)   */
#define CASIM_ERR_NO_DEVICE (-3)    /* no gfx950 device visible: the engine NEVER falls   */
                                    /* back to a CPU path                                  */
#define CASIM_ERR_NOMEM (-4)
#define CASIM_ERR_NO_LANE (-5)      /* casim_enc_lane: every resource lane (CASIM_MAX_RES) is taken   */
#define CASIM_ENC_DELEGATED 1       /* casim_enc_pod_set_request / casim_enc_group_set_allocatable: no lane left for the name — the pod spec
                                       was marked CASIM_PEG_UNSUPPORTED (the allocatable entry dropped); not an error                  */
/* per node-group status written to casim_results.status                                  */
#define CASIM_NG_OK 0
#define CASIM_NG_UNSUPPORTED 1      /* a PEG of this group needs a predicate outside the   */
                                    /* encoded subset: the shim must delegate this group   */
                                    /* to the Go estimator (SURVEY §8b "error convention") */

/* ---- PEG flags (casim_pegs.flags) -------------------------------------------------------- */
#define CASIM_PEG_TOLERATES_UNSCHEDULABLE 0x1u /* tolerates node.kubernetes.io/unschedulable:NoSchedule
                                                  V/.../nodeunschedulable/node_unschedulable.go:142-160 */
#define CASIM_PEG_SELF_EXCL_NODE 0x2u  /* at most one pod of this PEG per node: hostname
                                          self-anti-affinity or a host port it conflicts with itself */
#define CASIM_PEG_SELF_EXCL_ZONE 0x4u  /* at most one pod of this PEG per node group: required
                                          anti-affinity to itself on a non-hostname topology key
                                          (all nodes of a group clone one template)   */
#define CASIM_PEG_FASTPATH_OK 0x8u     /* shouldUseFastPath  CA/estimator/binpacking_estimator.go:411 */
#define CASIM_PEG_FASTPATH_AA_SELF 0x10u /* numNodesByAntiAffinity = len(pods)  :444-450      */
#define CASIM_PEG_FLAG_MASK 0x3fu      /* the caller's bits; every other bit of casim_pegs.flags is reserved and must be zero (CASIM_ERR_INVALID) */
#define CASIM_PEG_UNSUPPORTED 0x20u    /* needs fallback (affinity, topology spread, DRA, ...)   */

/* ---- group flags (casim_groups.flags) ---------------------------------------------------- */
#define CASIM_NG_UNSCHEDULABLE 0x1u    /* template node.Spec.Unschedulable                       */

/* ---- expander kinds (casim_best_option) -------------------------------------------------- */
#define CASIM_EXPANDER_LEAST_NODES 0   /* CA/expander/leastnodes/leastnodes.go:35-61 */
#define CASIM_EXPANDER_LEAST_WASTE 1   /* CA/expander/waste/waste.go:37-73           */
#define CASIM_EXPANDER_MOST_PODS 2     /* CA/expander/mostpods/mostpods.go:33-53     */

/*
 * PEG table: one record per PodEquivalenceGroup (CA/estimator/estimator.go:37-48), structure of
 * arrays.  G records, R resource lanes, mask widths in 64-bit words.
 */
typedef struct casim_pegs {
    int32_t n_pegs;           /* G */
    int32_t n_res;            /* R, 1..CASIM_MAX_RES */
    int32_t w_taint;          /* words per toleration mask (taint dictionary)          */
    int32_t w_label;          /* words per nodeSelector mask (label-requirement dictionary) */
    int32_t w_excl;           /* words per node-local exclusion mask (ports + hostname anti-affinity) */
    int32_t w_zone;           /* words per group-wide exclusion mask (non-hostname anti-affinity)      */
    const int64_t* req;       /* [G][R] pod request per lane  (fit.go:321-331 computePodResourceRequest) */
    const int32_t* count;     /* [G]    len(peg.Pods)                                   */
    const uint32_t* flags;    /* [G]    CASIM_PEG_*                                     */
    const uint64_t* tol_mask; /* [G][w_taint] bit t set: some toleration tolerates taint t (toleration.go:52-77) */
    const uint64_t* sel_mask; /* [G][w_label] bit l set: the PEG requires label requirement l (nodeaffinity.go:306-333) */
    const uint64_t* excl_block; /* [G][w_excl] node bits that forbid this PEG on a node  */
    const uint64_t* excl_mark;  /* [G][w_excl] node bits this PEG sets once it is on a node */
    const uint64_t* zone_block; /* [G][w_zone] group bits that forbid this PEG in the group */
    const uint64_t* zone_mark;  /* [G][w_zone] group bits this PEG sets once placed in the group */
    const double* fp_cpu;     /* [G] first container cpu request, AsApproximateFloat64 (binpacking_estimator.go:451-455); may be NULL if fastpath unused */
    const double* fp_mem;     /* [G] first container memory request, AsApproximateFloat64 (:456-458) */
    const uint64_t* zone_polarity; /* [w_zone] or NULL (= all zero): group bits of NEED polarity.  A PEG whose zone_block holds such a
                                 bit is forbidden in the group while the bit is CLEAR (required pod affinity on a non-hostname key:
                                 "a pod matching all my terms sits in this domain" — interpodaffinity/filtering.go:382-409), a
                                 plain bit forbids it while SET (anti-affinity).  Marks set bits of either kind. (ABI 6) */
    const uint64_t* excl_polarity; /* [w_excl] or NULL (= all zero): NODE bits of NEED polarity (ABI 8) — required pod affinity on
                                 kubernetes.io/hostname: a PEG whose excl_block holds such a bit fits a node only while the bit is SET there
                                 ("a pod matching all my terms sits on this node"); every partner marks it.  A PEG that both needs and marks
                                 such a bit is a self-affine SERIES: while no simulated node of the group carries the bit (and the template's
                                 own pods do not), its first pod passes without it (the first-pod exception, filtering.go:396-407) and the
                                 rest of the PEG then has to join that pod's node. */
    const int32_t* req32;     /* ABI 10.  [G][R] or NULL: the requests as 32-bit multiples of a per-lane unit the CALLER knows — request of PEG g in
                                 lane r = req32[g * R + r] * req_unit[r] (milli-cpu, MiB-granular memory: what a Go shim holds anyway).  When set, `req`
                                 may be NULL (when both are set they must describe the same requests; the batch path reads req32): half the request bytes cross the link and the library's own gcd pass over the table (a device round trip in
                                 the middle of a big call's upload) is not needed.  Exact: the int64 table the kernels read is rebuilt on the device
                                 as req32 * req_unit.  Honoured by the Estimate batch path (casim_estimate_batch / _query / _multi,
                                 casim_problem_create); K_sched's and the resident cluster's entry points need `req` (CASIM_ERR_INVALID without). */
    const int64_t* req_unit;  /* [R], every entry > 0; read only with req32 */
} casim_pegs;

/*
 * Node-group table: one record per candidate node group = one Estimate() call
 * (CA/core/scaleup/orchestrator/orchestrator.go:409-413).  NG records.
 * The schedulable PEG subset of each group (SchedulablePodGroups, orchestrator.go:535) is a CSR
 * list; pass peg_offsets == NULL to let the engine compute it on the device (feasibility kernel).
 */
typedef struct casim_groups {
    int32_t n_groups;             /* NG */
    const int64_t* alloc;         /* [NG][R] node.Status.Allocatable per lane (types.go:527-537 SetNode)  */
    const int64_t* init_req;      /* [NG][R] requested by pods preloaded on the template (DaemonSets; node_info_utils.go:111-118) */
    const int32_t* allowed_pods;  /* [NG] Allocatable.AllowedPodNumber, in [0, 2^30]                     */
    const int32_t* init_pods;     /* [NG] pods preloaded on the template, in [0, 2^30]                   */
    const uint32_t* flags;        /* [NG] CASIM_NG_*                                                     */
    const uint64_t* taint_mask;   /* [NG][w_taint] NoSchedule/NoExecute taints of the template (helper/taint.go:23-27) */
    const uint64_t* label_mask;   /* [NG][w_label] label requirements the template satisfies            */
    const uint64_t* init_excl;    /* [NG][w_excl] node bits already set on a fresh node (ports of preloaded pods, ...) */
    const uint64_t* init_zone;    /* [NG][w_zone] group bits already set (matching pods in the existing cluster) */
    const uint64_t* zone_valid;   /* [NG][w_zone] group bits whose topology key exists on the template: only these
                                     are ever marked (a term on a key the node lacks never matches, filtering.go:155-163) */
    const int32_t* max_nodes;     /* [NG] limiter result after getMinLimit: <0 forbid, 0 unlimited, >0 cap (threshold_based_limiter.go:34-69) */
    const int32_t* existing_nodes;/* [NG] 0 <= E < 2^30: nodes already in the snapshot; they occupy list positions 0..E-1; else CASIM_ERR_INVALID */
    const int32_t* last_index;    /* [NG] in [-1, 2^30): lastIndexOrderMapping.lastIndex on entry (scheduling_opts.go:39-63; -1 = start at position 0); else CASIM_ERR_INVALID */
    const double* cap_cpu;        /* [NG] node.Status.Capacity cpu  AsApproximateFloat64 (fastpath chooser); may be NULL */
    const double* cap_mem;        /* [NG] node.Status.Capacity mem  AsApproximateFloat64; may be NULL    */
    const int64_t* waste_cpu;     /* [NG] node.Status.Capacity cpu MilliValue (least-waste, waste.go:85-90); may be NULL */
    const int64_t* waste_mem;     /* [NG] node.Status.Capacity memory Value; may be NULL                 */
    const int32_t* peg_offsets;   /* [NG+1] CSR offsets into peg_index, or NULL = compute on device      */
    const int32_t* peg_index;     /* [peg_offsets[NG]] PEG ids, in the order they reach Estimate()       */
    /* ---- batches of independent simulations (all optional; NULL / 0 = one simulation over all PEGs) ---------------
     * One launch can carry many scale-up simulations (RunOnce iterations of different clusters, or what-if variants):
     * group i only ever sees the PEGs [peg_lo[i], peg_hi[i]) — its simulation's pending pods — when the engine derives
     * the schedulable subsets on the device (peg_offsets == NULL), and the expander reduce (casim_best_option_sims)
     * runs per simulation over the groups [sim_offsets[s], sim_offsets[s+1]). */
    const int32_t* peg_lo;        /* [NG] first candidate PEG of the group, or NULL = 0                    */
    const int32_t* peg_hi;        /* [NG] one past its last candidate PEG, or NULL = G                     */
    const int32_t* global_id;     /* [NG] id of the group inside expander keys (node groups sharded over GPUs keep their
                                     cluster-wide index), or NULL = group_id_base + i                        */
    int32_t n_sims;               /* S, 0 = one simulation holding every group                             */
    const int32_t* sim_offsets;   /* [S+1] groups of simulation s, ascending, sim_offsets[0] = 0, [S] = NG  */
} casim_groups;

typedef struct casim_options {
    int32_t fastpath;             /* --fastpath-binpacking-enabled (flags.go:203), default 0 */
    int32_t force_generic_packer; /* 0 = the library picks: the register-resident packer on gcd-scaled int32 lanes, on two int64 lanes when a
                                     lane does not narrow (R <= 2, no negative request, amounts < 2^62), else the LDS store's generic one;
                                     1 = never a register-resident packer; 2 = skip the int32 narrowing (int64 lanes when eligible) — testing / A-B */
    int32_t node_pods;            /* 1 = keep the pods per simulated node (casim_results.node_pods): what
                                     estimationAnalyserFunc receives as newNodesWithPods (binpacking_estimator.go:157-159) */
    int32_t n_streams;            /* > 1: a batch of simulations (casim_groups.n_sims >= 2, subsets derived on the device from
                                     peg_lo / peg_hi) is cut by simulation into up to n_streams sub-batches that run on HIP streams
                                     of their own INSIDE the context (own memory pool and pinned staging each): the memory-latency-
                                     bound feasibility / order kernels of one part overlap the issue-bound packer of another, and in
                                     casim_estimate_batch the upload of one part overlaps the kernels and result copies of others.
                                     Results are identical to n_streams <= 1.  Ordering towards the context's stream:
                                     casim_problem_run forks from it (every internal stream waits for what it holds);
                                     casim_best_option_sims with dev_* outputs joins into it (work enqueued there afterwards sees
                                     the keys); host outputs / casim_problem_fetch synchronise.  Batches that cannot be cut (one
                                     simulation, explicit peg_offsets, node_pods) run as one part; casim_problem_info [4] tells.
                                     The internal streams want hardware queues of their own: export GPU_MAX_HW_QUEUES=8 in the host's
                                     environment before the process makes its first HIP call.  The library
                                     does NOT edit the process environment by itself; CASIM_SET_HW_QUEUES=1 opts in to its load-time
                                     constructor doing setenv("GPU_MAX_HW_QUEUES", "8", overwrite = 0).  Without either the context
                                     uses the queues there are (fewer concurrent parts; casim_problem_info [4], [6]). */
    int32_t pack_build;           /* which build of the register packer runs (csrc/casim_pack_tu.hip): CASIM_PACK_BUILD_AUTO (0, default) = the
                                     one the library's self-check left standing for the device, _PLAIN = compiled without the
                                     experimental LLVM option, _OPTION = compiled with it; see casim_pack_build_info */
    int32_t no_singleton_merge;   /* 1 = keep every PEG row as it is.  Default (0): adjacent rows that are identical in every column, hold ONE pod
                                     each and carry no exclusion state — controller-less pods, one PodEquivalenceGroup each
                                     (equivalence/groups.go:69-73; BenchmarkRunOnceScaleUp: 10 000 of them) — are estimated as one row of k
                                     pods with the one rule in which the two differ (lastIndex, csrc/casim_pipeline.h SingletonRuns) and
                                     written out member by member again: results are identical, 10 000 dependent PEG steps become one.
                                     Not applied with fastpath, to batches of simulations, or when opts is NULL.  (PEG flag bit 0x40 is
                                     reserved for this and must be zero in casim_pegs.flags.) */
    int32_t no_front_kernel;      /* 1 = a call of <= 1024 groups runs feasibility, list offsets, lists and PEG order as the four separate launches
                                     a batch uses instead of the one fused launch (csrc/casim_kernels.h front_kernel; testing / A-B).  Results
                                     are identical.  (Took one of the two reserved words of ABI 6: zero keeps its meaning.) */
    int32_t winners_only;         /* 1 = casim_estimate_batch_query with an expander query (q->best_out set) returns the PEG order / pods placed of the
                                     WINNING group of every simulation only ("per-PEG placed[] arrays only travel for the winning NG"):
                                     out->order / out->placed then hold, simulation after simulation, the list of group best_out[s] (nothing for a
                                     simulation without an option) — list s starts at sum over s' < s of (offsets_out[best_out[s'] + 1] -
                                     offsets_out[best_out[s']]) and PEG ids are the batch's own — compacted ON THE DEVICE, so the copy back is a few
                                     MB instead of 8 bytes per (group, PEG) pair; the per-group scalars and offsets_out stay complete.  A caller
                                     that serves every group's Estimate() from the batch (the prefetch cache of the Go shim) keeps 0.  Implies
                                     no_singleton_merge.  (Took the last reserved word of ABI 6: zero keeps its meaning.) */
    int32_t chain_last_index;     /* ABI 9.  1 = the groups of ONE simulation are estimated AS IF one after the other, in table order, each starting from
                                     the lastIndex its predecessor left: group i of a simulation runs with last_index_out of group i - 1 (the first
                                     group with casim_groups.last_index of its own; the column's other entries are ignored).  This is what the
                                     reference does when the orchestrator calls Estimate() group after group on one snapshot: lastIndex lives in
                                     the snapshot's plugin runner (CA/simulator/clustersnapshot/predicate/plugin_runner.go:138,
                                     predicate_snapshot.go:64) and is never reverted.  A group that comes back CASIM_NG_UNSUPPORTED hands its input
                                     on unchanged.  Default 0: every group starts from its own last_index entry (independent Estimate() calls).
                                     Simulations stay independent of each other, so batches keep their parallelism; inside a simulation the
                                     library runs the packer to a fixed point (at most groups-per-simulation passes, each re-estimating only
                                     the groups whose input changed) — results are exactly those of the sequential loop.  Chains of up to 24
                                     passes (batches of simulations) are enqueued whole, nothing is waited for; longer ones (ONE simulation with a
                                     group per node group: the Go shim's prefetch) go out in growing blocks and stop at the first block whose last
                                     pass marked nothing (casim_last_chain_info; CASIM_CHAIN_ASYNC_MAX moves the limit).  The whole simulation
                                     has to live in the problem (not with node groups sharded over devices: CASIM_ERR_INVALID). */
    int32_t reserved[3];          /* zero */
} casim_options;
#define CASIM_PACK_BUILD_AUTO 0
#define CASIM_PACK_BUILD_PLAIN 1
#define CASIM_PACK_BUILD_OPTION 2

/*
 * Results of one batch.  All arrays caller-allocated.  `order`/`placed` use the same CSR
 * offsets as the input (or the offsets returned by casim_problem_csr when computed on device):
 * for group i, sorted position k:  order[off[i]+k] = PEG id processed k-th
 * (DecreasingPodOrderer + fastpath move), placed[off[i]+k] = how many of its pods were
 * scheduled — always a prefix of the PEG.  The shim rebuilds
 * Estimate()'s []*Pod as concat_k PEG[order[k]].Pods[0:placed[k]].
 */
typedef struct casim_results {
    int32_t* node_count;      /* [NG] len(newNodesWithPods)  binpacking_estimator.go:160 */
    int32_t* pods_scheduled;  /* [NG] len(scheduledPods)                                  */
    int32_t* nodes_added;     /* [NG] nodes added to the snapshot (incl. empty ones)      */
    int32_t* limiter_nodes;   /* [NG] thresholdBasedEstimationLimiter.nodes at exit       */
    int32_t* last_index_out;  /* [NG] lastIndex at exit                                   */
    int32_t* status;          /* [NG] CASIM_NG_*                                          */
    int64_t* req_cpu_sum;     /* [NG] sum of lane-0 requests of scheduled pods (least-waste) */
    int64_t* req_mem_sum;     /* [NG] sum of lane-1 requests of scheduled pods            */
    int32_t* order;           /* [nnz]                                                    */
    int32_t* placed;          /* [nnz]                                                    */
    /* optional (casim_options.node_pods = 1, else ignored; may be NULL): pods on every node the estimate added, in creation
     * order — node j of group i is the reference's "<template>-e-<j>" (binpacking_estimator.go:326-342) and belongs to
     * newNodesWithPods iff its count is > 0 (estimationState.trackScheduledPod :58-61).  Compact: the nodes of group i are
     * node_pods[node_pods_offsets[i] .. node_pods_offsets[i+1]), nodes_added[i] entries; groups are cut off (and the call
     * still succeeds) once node_pods_capacity entries are used — size it as the sum of the groups' max_nodes.  The
     * fastpath's extrapolated nodes (tryFastPath :274-324) exist only as a count and are not listed. */
    int32_t* node_pods;           /* [node_pods_capacity] */
    int32_t* node_pods_offsets;   /* [NG+1] */
    int64_t node_pods_capacity;
} casim_results;

/* ======================================================================================
 * ENGINE
 * ==================================================================================== */
typedef struct casim_ctx casim_ctx;
typedef struct casim_problem casim_problem;
struct casim_option_query;

/* ABI version of the loaded library. */
int32_t casim_abi_version(void);
/* Human-readable message of the last failure on the calling thread ("" if none). */
const char* casim_last_error(void);

/*
 * Create an engine context bound to HIP device `device`.  `stream` is an optional
 * hipStream_t (passed as void*) on which every kernel and copy of this context is enqueued;
 * NULL = the context creates its own stream.  Safe to call from any OS thread (cgo): every
 * entry point re-binds the device on entry (SURVEY §8b "threading").
 * Returns NULL (casim_last_error explains) if no gfx950 device is visible — there is no
 * CPU fallback in this library.
 */
casim_ctx* casim_ctx_create(int32_t device, void* stream);
void casim_ctx_destroy(casim_ctx* ctx);
/* Number of HIP devices visible (0 when none / no driver). */
int32_t casim_device_count(void);
/* Page-locked host memory for table columns (hipHostMalloc): a column of >= 1 MiB that lives in page-locked memory — from here, from
 * hipHostRegister, from a pinned torch tensor — is copied to the device straight from the caller's array instead of through the library's
 * staging buffer (the host-side memcpy of a 4096-simulation call is ~1 ms per part of its 3.8 ms).  The library finds out by itself
 * (hipPointerGetAttributes); pageable columns work as before.  The arrays must stay untouched until the call that uploads them returns
 * (it does not return before the copies have finished).  NULL on failure. */
void* casim_host_alloc(size_t bytes);
void casim_host_free(void* p);

/*
 * Upload one batch (PEG table + node-group table) to HBM and size the device scratch.
 * The returned problem stays resident until destroyed; it can be run many times.
 */
casim_problem* casim_problem_create(casim_ctx* ctx, const casim_pegs* pegs,
                                    const casim_groups* groups, const casim_options* opts);
void casim_problem_destroy(casim_problem* p);

/*
 * Enqueue the whole hot path for the resident batch on the context's stream:
 *   [feasibility + CSR compaction, if peg_offsets was NULL] -> order -> pack.
 * Asynchronous; results stay in HBM until casim_problem_fetch.
 * Replaces the NG loop of prepareScaleUp (orchestrator.go:1049-1068).
 */
int32_t casim_problem_run(casim_problem* p);

/* Block until the stream is idle, then copy the results to the caller's buffers. */
int32_t casim_problem_fetch(casim_problem* p, casim_results* out);

/* Number of (group, PEG) pairs (nnz) and the CSR offsets actually used by the last run
 * (needed when the engine computed the schedulable subsets).  offsets_out has NG+1 slots or
 * is NULL.  Synchronises the stream. */
int32_t casim_problem_csr(casim_problem* p, int32_t* nnz_out, int32_t* offsets_out);

/* Which build of the register packer serves `device` in this process (csrc/casim_pack_tu.hip: the kernels exist twice, compiled with
 * and without an experimental LLVM option; right before the first AUTO launch of an instantiation — lanes x node slots x exclusion words — on a
 * device the library runs that instantiation's 16-18 corpus batches through both builds, ~15 ms, and retires the option build for the process on
 * any difference; CASIM_PACK_SELFCHECK=eager checks all 18 instantiations, 306 batches, when the first context is created).  out[0] = CASIM_PACK_BUILD_PLAIN / _OPTION (AUTO = no context created yet),
 * [1] = batches compared, [2] = batches that differed, [3] = 1 when the environment forced the build (CASIM_PACK_BUILD=plain|option). */
int32_t casim_pack_build_info(int32_t device, int32_t out[4]);

/* How the resident batch will be executed (for reports): info_out[0] = node slots per lane of the
 * register-resident packer (0 = generic int64 packer, node state in LDS / HBM), [1] = its lanes (2 / 4: gcd-scaled int32 lanes; 8: two int64
 * lanes, the batch did not narrow), [2] = 1 if the
 * generic packer keeps node state in LDS (0 = HBM slab), [3] = 1 if the schedulable subsets are
 * derived on the device, [4] = parts the batch runs as on internal streams (1 = not cut), [5] = how many runs so far had to fork
 * from the context's stream (it held pending work: see casim_options.n_streams), [6] = streams the context parked because
 * they shared a hardware queue with a lane it already had (the runtime maps streams onto GPU_MAX_HW_QUEUES queues), [7] bit 0 = feasibility,
 * list offsets, lists and PEG order run as ONE launch (front_kernel: calls of <= 1024 groups; casim_options.no_front_kernel), bit 1 = the
 * orderer ranks a simulation's PEGs once per (cpu, memory) allocatable pair and every group takes its list from that ranking (batches with long
 * candidate ranges whose pairs serve several groups: rank_shapes_kernel / order_ranked_kernel; CASIM_RANK_ONCE=0 / 1 forces it off / on). */
struct casim_cluster_estimate_result;
int32_t casim_problem_info(casim_problem* p, int32_t info_out[8]);
/* Replace the result of group `ng` of a batch that already ran (status becomes CASIM_NG_OK): how a group that the batch
 * delegated and casim_estimate_on_cluster then estimated joins the expander reduce of casim_best_option. */
int32_t casim_problem_set_group_result(casim_problem* p, int32_t ng, const struct casim_cluster_estimate_result* r);

/* upload + run + fetch in one call: the form the Go Estimate() wrapper uses once per loop. */
int32_t casim_estimate_batch(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups,
                             const casim_options* opts, casim_results* out);

/* casim_estimate_batch + the CSR offsets of order / placed + the expander reduce, enter -> return in ONE call: what a prefetching
 * shim calls once per scale-up loop and what SURVEY 8d defines the wall time on (H2D and D2H included).
 * offsets_out ([NG+1]) and q may be NULL.  With opts->n_streams > 1 the parts of the batch run end to end on their own streams:
 * the upload of one overlaps the kernels of another and the result copies of a third; a q then has to be per_sim. */
int32_t casim_estimate_batch_query(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                                   casim_results* out, int32_t* offsets_out, const struct casim_option_query* q);

/* casim_estimate_batch with the expander reduce and a phase breakdown, host clock, the stream drained after every
 * phase (so the phases add up to slightly more than an untimed call): phase_ms_out[0] = tables to HBM (allocation +
 * H2D), [1] = feasibility + CSR kernels, [2] = order kernel, [3] = pack kernel, [4] = expander reduce (skipped when
 * q == NULL), [5] = results to the host (D2H), [6] = the whole call enter -> return, [7] = 0.  What bench.py and
 * tools/casim_native report per config (wall time = enter -> return, H2D / D2H included). */
int32_t casim_estimate_batch_timed(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups,
                                   const casim_options* opts, casim_results* out, const struct casim_option_query* q,
                                   double phase_ms_out[8]);

/*
 * Batched CheckPredicates(exemplar, fresh template node): bit (i, g) of
 * out_bits[i * ceil(G/64) + g/64] is set iff PEG g passes every encoded Filter on an empty
 * node of group i (Appendix A `fits`).  Replaces the G x NG RunFiltersOnNode calls of
 * SchedulablePodGroups (orchestrator.go:552).  Synchronous.  For a PEG flagged CASIM_PEG_UNSUPPORTED a set bit means
 * "not rejected by the encoded subset" (the shim runs CheckPredicates for that pair); a clear bit is final.  The same
 * rule builds the per-group PEG lists when peg_offsets is NULL: such a PEG stays on the list of every group it may
 * fit, which makes those groups CASIM_NG_UNSUPPORTED — it never silently drops out of an estimate.
 */
/* With casim_groups.peg_lo / peg_hi (a batch of simulations) row i has ceil(max_i(peg_hi - peg_lo) / 64) words and its
 * bit k stands for PEG peg_lo[i] + k. */
int32_t casim_feasibility(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups,
                          uint64_t* out_bits);

/*
 * Why a PEG does not fit a fresh node of a group: the SchedulingError of CheckPredicates for every cell of the
 * SchedulablePodGroups matrix, i.e. what the orchestrator stores in eg.SchedulingErrors[nodeGroup.Id()]
 * (orchestrator.go:553-567; FailingPredicateError: plugin name + reasons, CA/simulator/clustersnapshot/scheduling_error.go:40-52,
 * filled by RunFiltersOnNode, plugin_runner.go:168-180).  One uint16 per (group, PEG): 0 = the pod fits; otherwise the low 4
 * bits name the FIRST failing Filter plugin in the scheduler's Filter order (default_plugins.go:34-51: NodeUnschedulable,
 * NodeName, TaintToleration, NodeAffinity, NodePorts, NodeResourcesFit, ..., PodTopologySpread, InterPodAffinity) and the upper
 * bits carry that plugin's reasons: for NodeResourcesFit ALL of them, as fitsRequest reports them (fit.go:678-765) —
 * CASIM_REASON_TOO_MANY_PODS and CASIM_REASON_INSUFFICIENT(lane) per lane.  The other plugins have one fixed reason string each
 * (INTEGRATION.md lists them).  port_block = casim_enc_port_block (tells NodePorts conflicts with pods preloaded on the
 * template from hostname anti-affinity against them; NULL = every node-local exclusion is reported as InterPodAffinity).
 * Layout [NG][L], L = n_pegs, or max(peg_hi - peg_lo) with entry k standing for PEG peg_lo[i] + k.  Synchronous.
 * A PEG flagged CASIM_PEG_UNSUPPORTED that passes the encoded subset gets CASIM_PLUGIN_UNKNOWN (the shim runs Go CheckPredicates).
 */
#define CASIM_PLUGIN_NONE 0
#define CASIM_PLUGIN_NODE_AFFINITY_PREFILTER 1   /* "PreFilter filtered the Node out" (per-node mode only) */
#define CASIM_PLUGIN_NODE_UNSCHEDULABLE 2        /* "node(s) were unschedulable" */
#define CASIM_PLUGIN_TAINT_TOLERATION 3          /* "node(s) had untolerated taint(s)" */
#define CASIM_PLUGIN_NODE_AFFINITY 4             /* "node(s) didn't match Pod's node affinity/selector" */
#define CASIM_PLUGIN_NODE_PORTS 5                /* "node(s) didn't have free ports for the requested pod ports" */
#define CASIM_PLUGIN_NODE_RESOURCES_FIT 6        /* "Too many pods", "Insufficient <resource>" ... */
#define CASIM_PLUGIN_POD_TOPOLOGY_SPREAD 7
#define CASIM_PLUGIN_INTER_POD_AFFINITY 8        /* "node(s) didn't satisfy anti-affinity rules" / "... didn't match pod affinity rules" */
#define CASIM_PLUGIN_UNKNOWN 15                  /* outside the encoded subset: ask the Go path */
#define CASIM_PLUGIN_MASK 0xfu
#define CASIM_REASON_TOO_MANY_PODS 0x10u
#define CASIM_REASON_INSUFFICIENT(lane) (0x20u << (lane))
int32_t casim_feasibility_reasons(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups,
                                  const uint64_t* port_block, uint16_t* out_codes);

/*
 * Expander reduce on the device results of the last run: applies the filter chain `kinds`
 * (each CASIM_EXPANDER_*) in order to the options with node_count > 0 and pods_scheduled > 0
 * (orchestrator.go:1057-1063), as chainStrategy.BestOption does (factory/chain.go:36-45).
 * best_ng_out   = lowest-index group of the surviving set (deterministic stand-in for the
 *                 random fallback, random.go:49-56), -1 if no option;
 * n_best_out    = size of the surviving set (so a shim can apply its own random pick);
 * best_set_out  = [NG] 1 for surviving groups, may be NULL;
 * key_out       = packed sortable int64 key of the winner (smaller = better, group index in
 *                 the low 20 bits) for a cross-GPU min all-reduce; may be NULL.
 * key_out       = [10] int64 key block of the winner, smaller = better: [0] packed
 *                 (first filter's metric << 20 | global group id; one all-reduce(min) is exact for the
 *                 integer metrics), [1..8] the winner's metric under each filter (order-preserving
 *                 int64; the chain is the lexicographic min over (m_1..m_k, id)), [9] global group id;
 * dev_key_out   = optional DEVICE pointer (void*) to 10 int64 receiving the same block, so that an
 *                 RCCL collective can consume it without a host round trip; may be NULL.
 * group_id_base = added to the local group index inside the key (rank offset when the
 *                 groups are sharded across GPUs).
 * Synchronous unless every host out pointer is NULL.
 */
int32_t casim_best_option(casim_problem* p, const int32_t* kinds, int32_t n_kinds,
                          int32_t group_id_base, int32_t* best_ng_out, int32_t* n_best_out,
                          uint8_t* best_set_out, int64_t* key_out, void* dev_key_out);

/*
 * General form of casim_best_option: an optional validity mask and one reduce PER SIMULATION of a batch
 * (casim_groups.sim_offsets).  valid[i] == 0 takes option i out of the competition before the first filter: how the
 * shim applies prepareScaleUp's all-or-nothing rule (orchestrator.go:1057-1063 drops an option whose Pods are fewer than the
 * pods it was asked to place BEFORE ExpanderStrategy.BestOption sees the list).
 * Outputs are indexed by simulation (S = n_sims when per_sim, else 1): best_out[s] = lowest surviving group index
 * (index inside the launch) or -1, n_best_out[s] = survivors, key_out[s][10] = the winner's key block,
 * packed_out[s] = key block entry [0] (first filter's metric << 20 | global group id): ONE all-reduce(min) over the
 * [S] packed keys picks every simulation's winner across GPUs when the first filter is an integer metric.
 * dev_* are optional DEVICE pointers receiving the same data without a host round trip (RCCL operands).
 * Synchronous unless every host out pointer is NULL.
 */
typedef struct casim_option_query {
    const int32_t* kinds;       /* [n_kinds] CASIM_EXPANDER_* */
    int32_t n_kinds;
    int32_t group_id_base;      /* added to the group index inside keys unless casim_groups.global_id is set */
    int32_t per_sim;            /* 1 = one reduce per simulation, 0 = one over every group */
    const uint8_t* valid;       /* [NG] or NULL */
    int32_t* best_out;          /* [S] or NULL */
    int32_t* n_best_out;        /* [S] or NULL */
    uint8_t* best_set_out;      /* [NG] or NULL */
    int64_t* key_out;           /* [S][10] or NULL */
    int64_t* packed_out;        /* [S] or NULL */
    void* dev_key_out;          /* device [S][10] int64 or NULL */
    void* dev_packed_out;       /* device [S] int64 or NULL */
    void* join_stream;          /* hipStream_t or NULL.  With dev_* outputs the keys are ordered against a stream: by default the CONTEXT's
                                   stream waits for them (a collective enqueued there sees them) — and, being busy, makes the next
                                   casim_problem_run fork from it: every internal stream then waits for all the others' previous step.  A
                                   caller whose steps are independent passes the stream its collective runs on instead: only THAT stream
                                   waits, the context's stream stays idle and the sub-batches keep running ahead of each other (the
                                   caller then owns the reuse of its key buffers: bench.py keeps two and waits on the host for the
                                   all-reduce of two steps ago). */
} casim_option_query;
int32_t casim_best_option_sims(casim_problem* p, const casim_option_query* q);

/*
 * Several devices behind ONE caller.  The scale-up loop is a single goroutine (orchestrator.go:1053) and an
 * estimator.Estimator lives in that process, so multi-GPU has to work without one process per device: a casim_mctx owns one
 * context (device + stream) per listed device; casim_estimate_batch_multi block-partitions the node groups of every simulation
 * over them (rotated by simulation index; PEG table replicated, no pods x nodes data crosses devices), runs them concurrently
 * and settles the expander's choice with ONE collective: all-reduce(min) over the per-simulation packed keys
 * (first filter's metric << 20 | cluster-wide group id) — RCCL over xGMI when use_rccl != 0 (librccl.so is loaded at run time:
 * ncclCommInitAll over the listed devices, ncclAllReduce(ncclMin, ncclInt64) inside one group call), a host loop otherwise.
 * Chains whose first filter is not an integer metric (least-waste) take the per-device key blocks to the host and pick the
 * lexicographic minimum there.  Results come back in the caller's group order; offsets_out ([NG+1], may be NULL) receives
 * the CSR offsets of order / placed.  q->best_out[s] is the caller's index of simulation s's winner; n_best_out is 1 / 0
 * (survivors are not counted across devices); q->dev_* are ignored.  A device named twice (tests on a 1-GPU box) is
 * allowed without RCCL.
 */
typedef struct casim_mctx casim_mctx;
casim_mctx* casim_mctx_create(const int32_t* devices, int32_t n_devices, int32_t use_rccl);
void casim_mctx_destroy(casim_mctx* m);
/* n devices, 1 if RCCL communicators exist, 1 if the last batch's keys were reduced by RCCL (0 = on the host), and the
 * number of node groups each device held in the last batch ([n devices]); any pointer may be NULL. */
int32_t casim_mctx_info(const casim_mctx* m, int32_t* n_devices_out, int32_t* uses_rccl_out, int32_t* last_reduced_by_rccl_out,
                        int32_t* groups_per_device_out);
int32_t casim_estimate_batch_multi(casim_mctx* m, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                                   casim_results* out, int32_t* offsets_out, const struct casim_option_query* q);

/*
 * HintingSimulator.TrySchedulePods (CA/simulator/scheduling/hinting_simulator.go:53-135) — the
 * filter-out-schedulable pass (CA/core/podlistprocessor/filter_out_schedulable.go:48-103): pending pods,
 * one by one in the caller's (priority) order, against the nodes ALREADY in the cluster snapshot.
 *   classes  = one casim_pegs record per distinct pending-pod spec (count is ignored);
 *   nodes    = one casim_groups record per existing node: alloc, init_req / init_pods / init_excl = what its
 *              running pods hold (encode them as preloaded pods), taint / label masks of THAT node,
 *              CASIM_NG_UNSCHEDULABLE; the limiter / CSR fields are ignored.  The encoder must run with
 *              explicit_self_exclusion = 1.
 * Per pod: the hinted node first (tryScheduleUsingHints :86-110: RunFiltersOnNode, no lastIndex update), else
 * the first passing node in cyclic order from lastIndex + 1 among acceptable, schedulable nodes
 * (RunFiltersUntilPassingNode, plugin_runner.go:54-143), with the SimilarPodsScheduling memo (:115-118; see
 * similar_key below).
 * Returns CASIM_OK, <0 on error, or CASIM_NG_UNSUPPORTED (> 0) when a class needs a predicate outside the
 * encoded subset or a group-wide (non-hostname) exclusion: the shim then runs the Go simulator.
 */
/*
 * Domain rules: the Filters whose verdict on a node depends on what the OTHER nodes of its topology domain hold —
 * PodTopologySpread (V/.../podtopologyspread/filtering.go:236-366; DoNotSchedule constraints, nodeAffinityPolicy Honor,
 * nodeTaintsPolicy Ignore) and required anti-affinity on non-hostname topology keys
 * (V/.../interpodaffinity/filtering.go:204-432).  Built by the encoder in per-node mode
 * (explicit_self_exclusion = 1, casim_enc_domain_rules) from the pod specs and the nodes' labels / running pods.
 * A rule belongs to one class and one topology key and owns one counter per domain of that key:
 *   kind 0 (spread): counter = pods matching the constraint's selector (same namespace) on ELIGIBLE nodes of the domain
 *          (eligible = passes the class's nodeSelector / required affinity and carries every topology key of the
 *          class's constraints); a node passes iff it carries the key and
 *          counter[its domain] + self - min over existing domains (0 if fewer than min_domains) <= max_skew;
 *   kind 1 (conflict): counter = pods in the domain that are anti-affine with the class through this key, either
 *          direction; a node passes iff it lacks the key or counter[its domain] == 0.
 *   kind 2 (affinity): one rule per required pod-affinity term; counter = pods in the domain that match ALL affinity terms
 *          of the class; a node passes iff it carries the key and counter[its domain] > 0 for every such rule — or, when
 *          some counter is 0, iff every counter of the class's affinity rules is 0 in every domain AND rule_self (the class
 *          matches its own terms): the "first pod of a self-affine series" exception (filtering.go:396-407).
 * inc_*: which rules a placed pod of class c increments (on the node's domain, if the node is eligible for the rule).
 */
typedef struct casim_domain_rules {
    int32_t n_keys, n_rules, n_nodes, n_classes, n_elig_rows;
    const int32_t* node_domain;      /* [n_keys][n_nodes] domain id of the node for the key, -1 = label missing */
    const int32_t* key_domains;      /* [n_keys] number of domains                                            */
    const uint8_t* key_is_hostname;  /* [n_keys] the key is kubernetes.io/hostname                             */
    const int32_t* rule_class;       /* [n_rules] rules are sorted by class                                    */
    const int32_t* rule_key;         /* [n_rules]                                                              */
    const int32_t* rule_kind;        /* [n_rules] 0 spread, 1 conflict, 2 affinity                             */
    const int32_t* rule_max_skew;    /* [n_rules]                                                              */
    const int32_t* rule_min_domains; /* [n_rules]                                                              */
    const int32_t* rule_self;        /* [n_rules] 1 = a pod of the class increments its own rule               */
    const int32_t* rule_elig_row;    /* [n_rules] row of elig_bits (kind 0) or -1 = every node carrying the key */
    const int64_t* rule_offset;      /* [n_rules + 1] slice of count_init / domain_exists                      */
    const int32_t* count_init;       /* counters from the pods already running                                 */
    const uint8_t* domain_exists;    /* kind 0: an eligible node carries this domain (it counts for the minimum) */
    const int32_t* domain_nodes;     /* same slices: how many eligible nodes carry the domain (a removed node leaves it) */
    const int32_t* node_contrib;     /* [n_rules][n_nodes] what the pods running on the node add to the rule's counter
                                        of the node's domain (the removal simulation takes them out with the node) */
    const uint64_t* elig_bits;       /* [n_elig_rows][ceil(n_nodes / 64)]                                      */
    const int32_t* class_rule_off;   /* [n_classes + 1] rules of class c: [class_rule_off[c], class_rule_off[c+1]) */
    const int32_t* inc_off;          /* [n_classes + 1]                                                        */
    const int32_t* inc_rule;         /* [inc_off[n_classes]]                                                   */
    int32_t n_taint_policy_rules;    /* spread rules whose eligibility row honours node taints (see casim_enc_spread_set_taints_policy) */
    const uint8_t* rule_ghost_leaves; /* [n_rules] or NULL: 1 = a removal candidate leaves this rule's domains while it is simulated —
                                        a spread rule with nodeTaintsPolicy: Honor whose class does not tolerate the ghost's
                                        ToBeDeletedByClusterAutoscaler:NoSchedule taint (CA/simulator/cluster.go:240-252) */
} casim_domain_rules;

typedef struct casim_pod_sequence {
    int32_t n_pods;                  /* P */
    const int32_t* pod_class;        /* [P] class (PEG id) of each pending pod, processing order          */
    const int32_t* hint_node;        /* [P] node index hinted for the pod (hints.go) or -1; may be NULL    */
    const uint8_t* node_acceptable;  /* [N] SchedulingOptions.IsNodeAcceptable; may be NULL (= all)        */
    int32_t break_on_failure;        /* stop at the first pod that fits nowhere                            */
    int32_t last_index;              /* lastIndexOrderMapping.lastIndex on entry                           */
    const struct casim_domain_rules* rules; /* PodTopologySpread / non-hostname anti-affinity; NULL = none  */
    const int32_t* similar_key;      /* [P] or NULL: controller of the pod (drain.ControllerRef UID as a dense id >= 0),
                                        -1 = none or a DaemonSet pod.  SimilarPodsScheduling (similar_pods.go:38-98)
                                        caches "this spec of this controller found no node" for at most 10 specs per
                                        controller.  Needed for exactness only when spread rules exist: a spread
                                        constraint can start passing again once other domains fill up, so whether a
                                        failed spec is re-tried matters; for every other class a node that failed once
                                        fails again, cached or not. */
} casim_pod_sequence;

int32_t casim_try_schedule_pods(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                const casim_pod_sequence* seq, int32_t* node_out /*[P], -1 = stays pending*/,
                                int32_t* last_index_out, int32_t* n_scheduled_out);
/* Same on resident data, `iters` times, HIP-event timed (bench / profiles). */
int32_t casim_time_try_schedule_pods(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                     const casim_pod_sequence* seq, int32_t iters, float* ms_out);

/*
 * Scale-down removal simulation: the Planner.categorizeNodes loop
 * (CA/core/scaledown/planner/planner.go:300-330) around RemovalSimulator.SimulateNodeRemoval
 * (CA/simulator/cluster.go:131-265).  classes / nodes as for casim_try_schedule_pods (the nodes carry ALL their
 * running pods, the candidates' own pods included).  Per candidate, in order: the node becomes an unacceptable
 * ghost, its pods (GetPodsToMove's list, drainability and PDB rules stay on the host) go through TrySchedulePods
 * with breakOnFailure on the acceptable destinations; all placed => removable and, with `persist` (the planner's
 * NewRemovalSimulator(..., true)), the moves are committed, the node leaves the list (lastIndex positions shift
 * like the reference's list) and the destination set; otherwise the simulation is reverted.  lastIndex is never
 * reverted (it lives in the plugin runner).
 * A candidate that received pods from an earlier committed removal lists them again after its own pods, in arrival
 * order — what GetPodsToMove sees in the committed snapshot.  Those "ext" pods are reported in the ext_* arrays
 * (which candidate's simulation, which pod, where it went).  The call stops in front of such a candidate
 * (n_processed < K; the caller applies the results so far and re-submits the rest) when one of the arrived pods
 * is marked pod_sticky (the host has to re-run its drainability / PDB rules for it) or the ext arrays are full.
 * removable[k]: 1 removable, 0 no place to move the pods, 2 not evaluated.  node_out[i]: destination of pod i in
 * its own candidate's simulation (also for a failed one: the reference keeps those hints), -1 not placed.
 */
typedef struct casim_removal_candidates {
    int32_t n_candidates;            /* K */
    const int32_t* cand_node;        /* [K] node index, planner order, distinct */
    const int32_t* pod_offsets;      /* [K+1] pods to move of candidate k: [pod_offsets[k], pod_offsets[k+1]) */
    const int32_t* pod_class;        /* [total] */
    const int32_t* hint_node;        /* [total] or NULL */
    const uint8_t* destination;      /* [N] podDestinations membership; NULL = every node */
    const uint8_t* pod_sticky;       /* [total] or NULL: 1 = may not move a second time without the host */
    const uint8_t* cand_atomic;      /* [K] or NULL: 1 = node of an atomically scaled group (ZeroOrMaxNodeScaling): its
                                      * removal does not count toward max_removable (planner.go:306, :321-324) */
    int32_t persist;                 /* canPersist */
    int32_t max_removable;           /* stop after this many removable nodes (what is LEFT of unneededNodesLimit);
                                      * <= 0 = no limit.  A limit that is already used up (planner.go:303 breaks before
                                      * the first candidate, also when unneededNodesLimit() is 0) is the caller's early
                                      * return: there is nothing to simulate, so no call is made */
    int32_t last_index;
    int32_t ext_capacity;            /* entries of the ext_* result arrays; 0 = stop at the first candidate with arrivals.  (The one-wave removal kernel —
                                      * casim_last_removals_info — keeps its log of committed moves in LDS.  The worst case is `pods + ext_capacity`
                                      * entries; when that does not fit, the log gets what LDS has left — if that holds at least half of the call's
                                      * pods; moves onto nodes that were removed since are squeezed out when it fills up — and a call that still
                                      * outgrows it is run again with the log in HBM (or, where that is not possible, through the general loop): same
                                      * results, the first attempt's time lost.  Calls that list more than 65 536 pods keep the log in HBM from the start.) */
    const struct casim_domain_rules* rules; /* the encoder's domain rules (PodTopologySpread, zone anti-affinity); NULL = none */
} casim_removal_candidates;

typedef struct casim_removal_results {
    uint8_t* removable;              /* [K] */
    int32_t* node_out;               /* [total] */
    int32_t* ext_candidate;          /* [ext_capacity] candidate whose simulation listed the pod again */
    int32_t* ext_pod;                /* [ext_capacity] flat index of that pod */
    int32_t* ext_node;               /* [ext_capacity] its destination in that simulation, -1 not placed */
    int32_t n_ext;                   /* out: ext entries written */
    int32_t last_index;              /* out */
    int32_t n_processed;             /* out: candidates [0, n_processed) have their final answer */
} casim_removal_results;

int32_t casim_simulate_node_removals(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                     const casim_removal_candidates* cand, casim_removal_results* out);
int32_t casim_time_node_removals(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                 const casim_removal_candidates* cand, int32_t iters, float* ms_out);
/* Which kernel the calling thread's last removal simulation (either entry point, the resident cluster's too) ran as: info_out[0] 1 = the one-wave
 * kernel over per-class fit masks (removals_lean_kernel: clusters without domain rules and node-local exclusion words, <= 64 pod classes,
 * <= 4 resource lanes, node state within the LDS budget), 0 = K_sched's general transaction loop; [1] threads of the workgroup; [2] 1 = node
 * state in LDS; [3] runs of the call.  Results are identical either way; CASIM_NO_LEAN_REMOVALS=1 in the environment keeps K_sched (A/B). */
int32_t casim_last_removals_info(int32_t info_out[4]);
/* The calling thread's last run with casim_options.chain_last_index: info_out[0] = passes the fixed point is bounded by (groups per
 * simulation - 1), [1] = passes enqueued, [2] = times the host read the marks of a pass back (one wait each; 0 for a chain enqueued
 * whole), [3] = 1 when the chain was short enough to be enqueued whole. */
int32_t casim_last_chain_info(int32_t info_out[4]);

/*
 * Resident cluster (second half): the snapshot's node table stays in HBM for a whole RunOnce iteration.
 * The reference threads ONE ClusterSnapshot through the iteration — filter-out-schedulable adds the pods it places
 * (SchedulePod), the scale-down planner simulates removals inside Fork / Revert (CA/simulator/clustersnapshot/store/
 * delta.go:292-323,442-463, planner.go:286-336).  casim_cluster_create uploads the class table (every pod spec the
 * iteration will meet: pending pods AND the pods of removal candidates, one encoder session) and the node table once;
 *   casim_cluster_try_schedule_pods(..., commit)   forks from the committed image; commit != 0 folds the placements into it
 *                                                  (requested += request, pod count, host-port / anti-affinity bits), 0 = Revert;
 *   casim_cluster_simulate_node_removals           runs on the committed image and never persists (the planner's own Fork / Revert);
 *   casim_cluster_update_nodes                     replaces the records of single nodes between calls (a delta: only those rows travel).
 * Same results as the non-resident entry points on the equivalent tables; status conventions as there.
 * Domain rules (casim_pod_sequence.rules, casim_removal_candidates.rules) after a commit: the caller keeps passing the rules its
 * encoder built from the snapshot BEFORE the commits; the cluster adds the pods it committed since (class, node remembered per
 * pod) to count_init / node_contrib itself, with the kernels' increment rule, so that spread / zone anti-affinity / pod-affinity
 * counters see them the way the reference's one ClusterSnapshot does.  casim_cluster_update_nodes makes it forget the pods it
 * committed to the replaced nodes: their new records, and counters derived with them, describe those nodes from then on.
 * casim_cluster_forget_commits: for a caller whose NEXT rules were built from a snapshot that already holds the committed pods
 * (a shim that rebuilds its domain rules from a fresh snapshot every loop and keeps the resident cluster): the cluster stops
 * adding the pods committed so far to count_init / node_contrib — without it they would be counted twice, silently.  The image
 * itself (requests, pod counts, exclusion bits) is untouched; pods committed afterwards are remembered again.
 * casim_cluster_stats: out[0] full uploads (1), [1] node rows replaced by deltas, [2] commits, [3] nodes.
 */
typedef struct casim_cluster casim_cluster;
casim_cluster* casim_cluster_create(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes);
void casim_cluster_destroy(casim_cluster* c);
int32_t casim_cluster_update_nodes(casim_cluster* c, int32_t n, const int32_t* node_index, const casim_groups* rows);
int32_t casim_cluster_try_schedule_pods(casim_cluster* c, const casim_pod_sequence* seq, int32_t commit, int32_t* node_out,
                                        int32_t* last_index_out, int32_t* n_scheduled_out);
int32_t casim_cluster_simulate_node_removals(casim_cluster* c, const casim_removal_candidates* cand, casim_removal_results* out);
int32_t casim_cluster_fetch_nodes(casim_cluster* c, int64_t* init_req_out, int32_t* init_pods_out, uint64_t* init_excl_out);
int32_t casim_cluster_stats(const casim_cluster* c, int64_t out[4]);
int32_t casim_cluster_forget_commits(casim_cluster* c);

/*
 * BinpackingNodeEstimator.Estimate on the whole snapshot (CA/estimator/binpacking_estimator.go:102-342):
 * the path for node groups whose PEGs carry domain rules (PodTopologySpread, anti-affinity on non-hostname keys), which
 * the template-mode batch (casim_estimate_batch) flags CASIM_NG_UNSUPPORTED.  `nodes` = the n_existing nodes of the
 * cluster snapshot, in list order, with their running pods, followed by clones of the template (one record per node the
 * limiter may grant; hostname label = the name the estimator would give it); `classes` = the PEGs (count = pods).
 * Encoder: per-node mode (explicit_self_exclusion = 1).  Fastpath is not modelled (a PEG with spread constraints is
 * never fast-pathed, :411-425): callers running with fastpath on keep delegating such groups.
 * Result as casim_results for one group; node_count also counts nodes of the cluster that took a pod in the
 * hostname-spread retry (:212-227, trackScheduledPod :58-61).
 */
typedef struct casim_cluster_estimate {
    int32_t n_existing;              /* E */
    int32_t max_nodes;               /* limiter: 0 no limit, < 0 no node may be added */
    int32_t last_index;
    const struct casim_domain_rules* rules;   /* casim_enc_domain_rules; NULL = none */
    const uint64_t* port_block;      /* [n_pegs][w_excl] casim_enc_port_block: the NodePorts part of excl_block */
} casim_cluster_estimate;

typedef struct casim_cluster_estimate_result {
    int32_t node_count, pods_scheduled, nodes_added, limiter_nodes, last_index_out, status;
    int64_t req_cpu_sum, req_mem_sum;
    int32_t* order;                  /* [n_pegs] PEG processed k-th */
    int32_t* placed;                 /* [n_pegs] pods of that PEG that were scheduled */
} casim_cluster_estimate_result;

int32_t casim_estimate_on_cluster(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                  const casim_cluster_estimate* params, casim_cluster_estimate_result* out);

/* Measurement helpers (used by bench.py): run `iters` times, bracketed by HIP events on the
 * context's stream; returns the mean per-run milliseconds of the whole pipeline and of the
 * named kernel classes.  kernel_ms_out: [0]=feasibility+csr [1]=order [2]=pack (may be NULL). */
int32_t casim_problem_time(casim_problem* p, int32_t iters, float* total_ms_out,
                           float* kernel_ms_out);
/* The feasibility launch (SchedulablePodGroups matrix) of the problem alone: `iters` launches back to back between two HIP events on the
 * launch stream, average per launch.  info_out (may be NULL): [0] 1 = the streaming kernel of round 5 (csrc/casim_kernels.h
 * feas_stream_kernel; batches of simulations on narrowed int32 lanes), [1] its lean instantiation, [2] bit 0: no upper-half terms, bit 1: NodeUnschedulable rides on a spare mask bit,
 * [3] workgroups per launch.  What bench.py's roofline_feasibility row is measured with. */
int32_t casim_problem_time_feasibility(casim_problem* p, int32_t iters, float* ms_per_launch_out, int32_t info_out[4]);
/* The same measurement WITHOUT stopping the stream: casim_problem_run_marked is casim_problem_run with HIP events recorded
 * around the three kernel classes (nothing waits; up to 64 marked runs are kept, older ones are overwritten);
 * casim_problem_marked_ms waits for the stream and returns the mean milliseconds over the marked runs (and forgets
 * them).  For kernels that overlap with work of other streams — sub-batches of one batch on several contexts — this is
 * what a kernel trace of the same loop sees. */
int32_t casim_problem_run_marked(casim_problem* p);
int32_t casim_problem_marked_ms(casim_problem* p, float* total_ms_out, float* kernel_ms_out /*[3]*/, int32_t* n_runs_out);
/* Device-to-device copy bandwidth probe (GB/s) used as the "achievable HBM" reference. */
int32_t casim_copy_bandwidth(casim_ctx* ctx, int64_t bytes, int32_t iters, double* gbps_out);

/* Read-only stream of `bytes` with `lane_bytes` (4 or 16) per lane and step: read bandwidth, and — under rocprofv3 --pmc
 * FETCH_SIZE — the calibration of the counter on that access width (kernel name stream_probe_kernel<4|16>).
 * lane_bytes == 0: the scalar form — every wave walks its own region with one 32-byte scalar load per step (kernel
 * stream_probe_scalar_kernel), the register packer's record fetch. */
int32_t casim_stream_probe(casim_ctx* ctx, int64_t bytes, int32_t lane_bytes, int32_t iters, double* gbps_out);

/* ======================================================================================
 * ENCODER  (host side, C++ inside; replaces the string work of the Filter plugins)
 * ==================================================================================== */
typedef struct casim_encoder casim_encoder;

typedef struct casim_encoder_options {
    int32_t n_res;                 /* R: lanes every object carries (>= 2)                     */
    int32_t enable_taint_comparison_ops; /* TaintTolerationComparisonOperators gate (toleration.go:66-72); default 0 */
    int32_t explicit_self_exclusion;     /* 1 = self-conflicts (own host port, hostname self-anti-affinity) also get
                                            node bits; required by casim_try_schedule_pods (pod-by-pod placement) */
    int32_t reserved[5];
} casim_encoder_options;

casim_encoder* casim_enc_create(const casim_encoder_options* opts);
void casim_enc_destroy(casim_encoder* e);

/* ---- node-group templates ------------------------------------------------------------ */
/* Adds a node group; returns its index (>= 0) or <0.  alloc = Allocatable lanes, capacity_* =
 * node.Status.Capacity (least-waste / fastpath chooser). */
int32_t casim_enc_add_group(casim_encoder* e, const char* template_name, const int64_t* alloc,
                            int32_t allowed_pods, int64_t capacity_cpu_milli,
                            int64_t capacity_mem_bytes, int32_t unschedulable);
int32_t casim_enc_group_add_label(casim_encoder* e, int32_t group, const char* key, const char* value);
int32_t casim_enc_group_add_taint(casim_encoder* e, int32_t group, const char* key, const char* value,
                                  const char* effect);
/* override Capacity.{Cpu,Memory}().AsApproximateFloat64() (default: milli * 1e-3, bytes) */
int32_t casim_enc_group_set_fastpath_capacity(casim_encoder* e, int32_t group, double cpu, double mem);
/* limiter inputs: result of StartEstimation's getMinLimit fold (the shim runs the
 * reference's own thresholds), E and lastIndex. */
int32_t casim_enc_group_set_limits(casim_encoder* e, int32_t group, int32_t max_nodes,
                                   int32_t existing_nodes, int32_t last_index);
/* A pod preloaded on the template (DaemonSet pod): `pod` is a pod spec id from
 * casim_enc_add_pod_spec. */
int32_t casim_enc_group_add_preloaded_pod(casim_encoder* e, int32_t group, int32_t pod_spec);
/* Many plain running pods in ONE call (per-node mode at cluster scale: 150 000 running pods are 600 000 calls and, from cgo, a C
 * string per label otherwise).  Pod i: namespace strings[ns[i]], requests req[i * n_res ..], labels
 * (strings[label_key[k]], strings[label_val[k]]) for k in [label_off[i], label_off[i + 1]), running on node group[i] (>= 0: added as
 * casim_enc_group_add_preloaded_pod does; -1: only the spec is created).  Pods with tolerations, selectors, host ports or
 * (anti-)affinity terms get those through the calls above on the returned ids.  Returns the spec id of pod 0 (the ids are
 * consecutive), or a negative CASIM_ERR_*; nothing is added on an error.  Not inside an update session. */
int32_t casim_enc_add_running_pods(casim_encoder* e, int32_t n_pods, const int32_t* group, const int32_t* ns, const int64_t* req,
                                   const int32_t* label_off, const int32_t* label_key, const int32_t* label_val,
                                   const char* const* strings, int32_t n_strings);
/* Restrict the group to an explicit PEG list (order = order the PEGs reach Estimate).  If never
 * called for any group, the engine derives the schedulable subsets on the device. */
int32_t casim_enc_group_set_pegs(casim_encoder* e, int32_t group, const int32_t* pegs, int32_t n);

/* ---- pod specs and PEGs -------------------------------------------------------------- */
/* A pod spec = the scheduling-relevant part of one exemplar pod.  Returns its id. */
int32_t casim_enc_add_pod_spec(casim_encoder* e, const char* namespace_, const int64_t* req);
/* ---- resources BY NAME (ABI 9) --------------------------------------------------------------------------------------------
 * The reference keeps cpu / memory / ephemeral-storage in fields of their own and EVERYTHING else by name in
 * Resource.ScalarResources (V/kubernetes/pkg/scheduler/framework/types.go:989-998); fitsRequest walks the pod's map
 * (V/.../noderesources/fit.go:731-763: zero quantities skipped, the rest compared with Allocatable[name] - Requested[name]) and
 * AddPodInfo accumulates name by name (types.go:444-448).  Lane numbers are an encoding detail that stays BEHIND the ABI: a binding
 * hands over every entry of the pod's request list and of node.Status.Allocatable under its Kubernetes name and cannot drop or
 * mis-order a lane.  The positional arrays of casim_enc_add_pod_spec / casim_enc_add_group cover lanes [0, casim_encoder_options.n_res)
 * and stay what they were; named values are written on top of them.
 *
 * casim_enc_lane: "cpu" -> 0 (value in millicores), "memory" -> 1, "ephemeral-storage" -> 2 (bytes), any other name -> its lane,
 *   assigned in first-use order from max(3, n_res) on (Quantity.Value units); CASIM_ERR_NO_LANE when all CASIM_MAX_RES lanes are
 *   taken or the name is new after casim_enc_finalize (update sessions re-encode rows of fixed width); "pods" / NULL / "" ->
 *   CASIM_ERR_INVALID.  The caller decides WHICH names count — the scheduler's own rule is schedutil.IsScalarResourceName
 *   (extended resources, hugepages-*, attachable-volumes-*, prefixed native names: types.go Resource.Add) — the encoder takes every
 *   name it is given.
 * casim_enc_pod_set_request: the pod's request for the name (>= 0).  A NON-ZERO request opens the name's lane; a request of zero opens
 *   none (fitsRequest skips zero quantities, fit.go:733).  CASIM_OK, or CASIM_ENC_DELEGATED (1) when no lane is left and
 *   value != 0: the pod spec is marked CASIM_PEG_UNSUPPORTED, so every group that lists it comes back CASIM_NG_UNSUPPORTED and the
 *   shim runs the reference path — a request is NEVER silently ignored.  A NEGATIVE return (CASIM_ERR_INVALID: negative value, bad
 *   id) means the request was NOT recorded: the caller must fail closed (casim_enc_pod_mark_unsupported, or give the call up).
 * casim_enc_group_set_allocatable: the template's Allocatable[name] ("pods" sets allowed_pods).  Allocatable NEVER opens a lane: a
 *   column no pod reads is not a column (real nodes list hugepages-1Gi: 0, hugepages-2Mi: 0, attachable-volumes-*; one lane per
 *   listed name used to push every table past the four lanes the register packer takes).  A name without a lane is kept aside and
 *   written into its lane by casim_enc_finalize if some pod's request has opened one by then — pods and groups may arrive in any
 *   order.  After finalize (update sessions: fixed table width) a name without a lane answers CASIM_ENC_DELEGATED (harmless: every
 *   pod that asks for the name is delegated).
 * casim_enc_lane_count: lanes the tables carry (casim_pegs.n_res after finalize); casim_enc_lane_name: the name of a lane or NULL. */
int32_t casim_enc_lane(casim_encoder* e, const char* resource_name);
int32_t casim_enc_pod_set_request(casim_encoder* e, int32_t pod, const char* resource_name, int64_t value);
int32_t casim_enc_group_set_allocatable(casim_encoder* e, int32_t group, const char* resource_name, int64_t value);
int32_t casim_enc_lane_count(const casim_encoder* e);
const char* casim_enc_lane_name(const casim_encoder* e, int32_t lane);
int32_t casim_enc_pod_add_label(casim_encoder* e, int32_t pod, const char* key, const char* value);
/* op: "", "Equal", "Exists", "Lt", "Gt";  effect: "", "NoSchedule", "NoExecute", "PreferNoSchedule" */
int32_t casim_enc_pod_add_toleration(casim_encoder* e, int32_t pod, const char* key, const char* op,
                                     const char* value, const char* effect);
int32_t casim_enc_pod_add_node_selector(casim_encoder* e, int32_t pod, const char* key, const char* value);
/* One requirement of the (single) required node-affinity term; op in In/NotIn/Exists/
 * DoesNotExist/Gt/Lt; values = n_values C strings. */
int32_t casim_enc_pod_add_node_affinity_req(casim_encoder* e, int32_t pod, const char* key,
                                            const char* op, const char* const* values, int32_t n_values);
/* nodeSelectorTerms of requiredDuringSchedulingIgnoredDuringExecution, ORed
 * (LazyErrorNodeSelector.Match, V/component-helpers/scheduling/corev1/nodeaffinity/nodeaffinity.go:85-107): the first call
 * opens a term and returns its index (>= 0), the second adds one of its matchExpressions (is_field 0) or matchFields
 * (is_field 1: key metadata.name, op In / NotIn, one value, :260-291).  The encoder folds the whole list into one
 * per-node bit, so the kernels see it as one more label requirement.  Exclusive with casim_enc_pod_add_node_affinity_req
 * on the same pod (CASIM_ERR_INVALID).  Template mode (an Estimate): a term that reads metadata.name or
 * kubernetes.io/hostname marks the pod CASIM_PEG_UNSUPPORTED, because simulated nodes get fresh names
 * (CA/simulator/node_info_utils.go:93-137). */
int32_t casim_enc_pod_add_node_affinity_term(casim_encoder* e, int32_t pod);
int32_t casim_enc_node_term_add_requirement(casim_encoder* e, int32_t pod, int32_t term, int32_t is_field,
                                            const char* key, const char* op, const char* const* values,
                                            int32_t n_values);
/* protocol "" = TCP, ip "" = 0.0.0.0  (V/kube-scheduler/framework/types.go:633-640 sanitize) */
int32_t casim_enc_pod_add_host_port(casim_encoder* e, int32_t pod, const char* ip, const char* protocol,
                                    int32_t port);
/* Required pod anti-affinity term: topology key, explicit namespaces (n = 0 => the pod's own
 * namespace, types.go getNamespacesFromPodAffinityTerm), and a label selector given as
 * requirements (matchLabels k=v is op "In" with one value).  Returns the term id. */
int32_t casim_enc_pod_add_anti_affinity_term(casim_encoder* e, int32_t pod, const char* topology_key,
                                             const char* const* namespaces, int32_t n_namespaces);
int32_t casim_enc_term_add_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key,
                                       const char* op, const char* const* values, int32_t n_values);
/* REQUIRED pod affinity term (PodAffinity.RequiredDuringSchedulingIgnoredDuringExecution; V/.../interpodaffinity/
 * filtering.go:234-272 getIncomingAffinityAntiAffinityCounts, :382-409 satisfyPodAffinity): topology key, explicit namespaces
 * (n = 0 => the pod's own), label selector through casim_enc_aff_term_add_requirement.  A pod "matches" when it matches ALL
 * terms of the spec; a node passes when, for every term, it carries the topology key and its domain holds a matching pod —
 * or the pod is the first of a series with affinity to itself (no matching pod anywhere, it matches its own terms, the node
 * carries every key).  Evaluated on the device in per-node mode (TrySchedulePods, the removal loop, Estimate on the snapshot)
 * as domain rules of kind 2.  Template mode (an Estimate): when every term's key is a non-hostname key and no PEG of the batch can
 * become a partner (nobody matches all terms, the PEG itself included), the verdict of every (PEG, group) pair is fixed by the
 * existing cluster (casim_enc_add_existing_pod) and the template's preloaded pods — satisfied: the term is a no-op, else the PEG
 * never fits the group — and the group stays in the template-mode packer; otherwise (hostname keys, partners inside the batch,
 * a self-affine series) the spec is flagged CASIM_PEG_UNSUPPORTED, which sends its node groups to casim_estimate_on_cluster.  A term's namespaceSelector: casim_enc_aff_term_set_namespace_selector (below). */
int32_t casim_enc_pod_add_affinity_term(casim_encoder* e, int32_t pod, const char* topology_key,
                                        const char* const* namespaces, int32_t n_namespaces);
int32_t casim_enc_aff_term_add_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key,
                                           const char* op, const char* const* values, int32_t n_values);
/* first-container requests as float64 for the fastpath chooser */
/* DoNotSchedule topologySpreadConstraint of the pod spec (min_domains <= 0 = nil); returns its index.  Evaluated on the
 * device only in per-node mode (explicit_self_exclusion); in template mode such a spec is flagged UNSUPPORTED. */
int32_t casim_enc_pod_add_spread_constraint(casim_encoder* enc, int32_t pod, int32_t max_skew, const char* topology_key,
                                            int32_t min_domains);
int32_t casim_enc_spread_add_requirement(casim_encoder* enc, int32_t pod, int32_t constraint, const char* key, const char* op,
                                         const char* const* values, int32_t n_values);
/* nodeTaintsPolicy: Honor (podtopologyspread/common.go:52-56): nodes with a NoSchedule / NoExecute taint the pod does
 * not tolerate are no members of the constraint's domains.  Default (0) = Ignore.  Evaluated by TrySchedulePods and
 * by the estimator on the cluster; the removal loop (whose ghost node gains a taint inside a simulation) answers
 * CASIM_NG_UNSUPPORTED for rules carrying it. */
int32_t casim_enc_spread_set_taints_policy(casim_encoder* enc, int32_t pod, int32_t constraint, int32_t honor);
/* nodeAffinityPolicy (common.go:46-51): Honor (1, the default) = only nodes matching the pod's nodeSelector / required node
 * affinity are members of the constraint's domains; Ignore (0) = every node carrying the constraint keys. */
int32_t casim_enc_spread_set_affinity_policy(casim_encoder* enc, int32_t pod, int32_t constraint, int32_t honor);
/* namespaceSelector of a required anti-affinity term (AffinityTerm.Matches, V/kube-scheduler/framework/types.go:390-395: a
 * pod is selected when its namespace is listed in the term OR its namespace's labels match the selector).
 * casim_enc_add_namespace / casim_enc_namespace_add_label describe the namespace lister (name -> labels);
 * casim_enc_term_set_namespace_selector marks the field as set on term `term` of `pod` (an empty selector selects every
 * namespace; the pod's own namespace no longer stands in, types.go:439-447) and
 * casim_enc_term_add_namespace_requirement adds one of its requirements (matchLabels pair == In{value}).  finalize resolves
 * the selectors into namespace sets (interpodaffinity/plugin.go:144-157).  When a non-empty selector exists and some pod's
 * namespace was not listed, every PEG is flagged CASIM_PEG_UNSUPPORTED (the reference treats that corner differently for
 * arriving and resident pods, plugin.go:161-169). */
int32_t casim_enc_add_namespace(casim_encoder* e, const char* name);
int32_t casim_enc_namespace_add_label(casim_encoder* e, const char* name, const char* key, const char* value);
int32_t casim_enc_term_set_namespace_selector(casim_encoder* e, int32_t pod, int32_t term);
int32_t casim_enc_term_add_namespace_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key,
                                                 const char* op, const char* const* values, int32_t n_values);
/* the same for a required AFFINITY term (index returned by casim_enc_pod_add_affinity_term): the term is the incoming pod's, so a non-empty
 * selector selects among the namespaces fed with casim_enc_add_namespace — what the namespace lister returns to InterPodAffinity.PreFilter
 * (plugin.go:144-157); an empty selector selects every namespace. */
int32_t casim_enc_aff_term_set_namespace_selector(casim_encoder* e, int32_t pod, int32_t term);
int32_t casim_enc_aff_term_add_namespace_requirement(casim_encoder* e, int32_t pod, int32_t term, const char* key, const char* op,
                                                     const char* const* values, int32_t n_values);
int32_t casim_enc_pod_set_fastpath_requests(casim_encoder* e, int32_t pod, double cpu, double mem);
/* Mark the spec as carrying a predicate outside the encoded subset (required pod affinity,
 * topology spread, volumes, DRA claims...). */
int32_t casim_enc_pod_mark_unsupported(casim_encoder* e, int32_t pod, const char* why);
/* A PodEquivalenceGroup: `count` pods sharing pod spec `pod`.  Returns the PEG id. */
int32_t casim_enc_add_peg(casim_encoder* e, int32_t pod_spec, int32_t count);

/* Bulk form for resource-only PEGs (no labels, tolerations, selectors, ports or affinity): one cgo
 * crossing for n PEGs.  req = [n][n_res] lanes, count = [n]; ids_out (may be NULL) receives the
 * PEG ids.  Returns the id of the first PEG added or <0. */
int32_t casim_enc_add_resource_pegs(casim_encoder* e, const char* namespace_, int32_t n, const int64_t* req,
                                    const int32_t* count, int32_t* ids_out);

/* Bulk form for pods in general (ABI 11): what nearly every pending pod carries — namespace, requests, labels, tolerations, nodeSelector,
 * the first container's requests — for n_pods pods in ONE call, strings by index into a table the caller interned (clusters repeat a few
 * dozen keys and values over thousands of pods; from cgo that is one C string per DISTINCT string instead of one per use).  C2 of the bench
 * (400 PEGs, 20 node groups) is 3 875 casim_enc_* calls pod by pod and 280 with this one; the encoder also sees equal tolerations as
 * equal indices without hashing them.  Pod i becomes spec record first + i (the return value is `first`; ids are consecutive) exactly as
 * casim_enc_add_pod_spec + casim_enc_pod_add_label / _add_toleration / _add_node_selector / _set_fastpath_requests in that order
 * would have built it; peg_count[i] >= 0 also makes it the exemplar of a PEG of that size (casim_enc_add_peg; PEG ids are consecutive
 * in pod order, peg_ids_out[i] = the id or -1).  Everything rarer — host ports, (anti-)affinity terms, node-affinity terms, spread
 * constraints, requests by name, unsupported marks, the spec digest — goes through the per-pod calls on the returned ids afterwards.
 * A string index of -1 is NULL / "".  Offsets: [n_pods + 1], starting at 0, non-decreasing; a NULL offset column = no pod has any.
 * Nothing is added on an error (CASIM_ERR_INVALID: an index out of range, offsets that do not rise, a PEG size below -1).  Not after
 * casim_enc_finalize. */
typedef struct casim_pod_columns {
    int32_t n_pods, n_strings;
    const char* const* strings;     /* [n_strings] NUL-terminated, distinct or not                                          */
    const int32_t* ns;              /* [n_pods] namespace                                                                   */
    const int64_t* req;             /* [n_pods][n_res] positional lanes (casim_encoder_options.n_res)                       */
    const double* fastpath_req;     /* [n_pods][2] Containers[0] cpu, memory as float64 (binpacking_estimator.go:451-458);
                                       NULL = derived from lanes 0 / 1 as casim_enc_add_resource_pegs does                   */
    const int32_t* peg_count;       /* [n_pods] or NULL: >= 0 = exemplar of a PEG of that many pods, -1 = spec record only   */
    const int32_t* label_off;       /* [n_pods + 1] or NULL                                                                 */
    const int32_t* label_key;       /* labels of pod i: k in [label_off[i], label_off[i + 1])                               */
    const int32_t* label_val;
    const int32_t* tol_off;         /* [n_pods + 1] or NULL; tolerations in the pod's order                                 */
    const int32_t* tol_key;
    const int32_t* tol_op;          /* "Exists" / "Equal" / "" / "Lt" / "Gt" as casim_enc_pod_add_toleration takes them      */
    const int32_t* tol_value;
    const int32_t* tol_effect;
    const int32_t* sel_off;         /* [n_pods + 1] or NULL; nodeSelector pairs                                             */
    const int32_t* sel_key;
    const int32_t* sel_val;
} casim_pod_columns;
int32_t casim_enc_add_pods(casim_encoder* e, const casim_pod_columns* pods, int32_t* peg_ids_out);

/* f2 — pod equivalence groups: equivalence.BuildPodGroups / groupPodsBySchedulingProperties / match
 * (CA/core/scaleup/equivalence/groups.go:39-104) over n_pods pending pods IN LIST ORDER.  pod_spec[i] = the spec record of pod i
 * (pods the shim knows to be identical may share one record; different records are compared by content: namespace, requests,
 * labels (reflect.DeepEqual on the label map), every scheduling field registered through casim_enc_pod_*, and the `extra` digest
 * the shim sets with casim_enc_pod_set_spec_extra for the sanitized PodSpec fields the encoder does not model — what
 * utils.PodSpecSemanticallyEqual compares, CA/utils/utils.go:63-119).  controller_uid[i] = drain.ControllerRef(pod).UID, NULL or ""
 * when the pod has no controller; daemonset[i] != 0 for pod_utils.IsDaemonSetPod (NULL = none).  Both kinds are singletons; a
 * controller keeps at most 10 joinable groups, a further distinct spec opens a group nobody can join (groups.go:58,81-90).
 * group_out[i] = group id in creation order (the reference's ids are map keys: "Group ID is meaningless"); *n_groups_out = count.
 * Callable before or after finalize. */
int32_t casim_enc_pod_set_spec_extra(casim_encoder* e, int32_t pod, const char* digest);
int32_t casim_enc_group_pods(casim_encoder* e, int32_t n_pods, const int32_t* pod_spec, const char* const* controller_uid,
                             const uint8_t* daemonset, int32_t* group_out, int32_t* n_groups_out);
/* One PEG per group of casim_enc_group_pods, in group order: exemplar = the group's first pod (PodEquivalenceGroup.Exemplar,
 * CA/estimator/estimator.go:42-51), count = its size.  peg_ids_out (may be NULL) [n_groups].  Returns the first PEG id or <0. */
int32_t casim_enc_add_grouped_pegs(casim_encoder* e, int32_t n_pods, const int32_t* pod_spec, const int32_t* group, int32_t n_groups,
                                   int32_t* peg_ids_out);

/* Pods already running in the cluster that may interact with anti-affinity on non-hostname
 * topology keys: (pod spec, label set of the node it runs on given as a group-like label
 * list).  v0 accepts them only to detect interactions; see DESIGN.md. */
int32_t casim_enc_add_existing_pod(casim_encoder* e, int32_t pod_spec, const char* const* node_label_keys,
                                   const char* const* node_label_values, int32_t n_labels);

/* ---- incremental re-encode (per-node mode: explicit_self_exclusion = 1) ------------------------------------------------
 * The reference forks its snapshot in O(1) and adds pods in place (CA/simulator/clustersnapshot/store/delta.go:235-246,292-323);
 * re-walking 15 000 nodes / 150 000 pods through the encoder costs tens of milliseconds per loop iteration.  Between two iterations
 * most nodes are unchanged, so a FINALIZED encoder can be kept and updated:
 *   casim_enc_begin_update(e);
 *   for every node whose pods / labels / taints changed:
 *       casim_enc_group_reset(e, node, alloc, ...);                     // forgets its labels, taints and running pods
 *       casim_enc_group_add_label / _add_taint / _add_preloaded_pod ... // describe it again (new pod specs: casim_enc_add_pod_spec + casim_enc_pod_*)
 *   casim_enc_set_peg_count(e, class, n) for classes whose number of pending pods changed;
 *   rc = casim_enc_refinalize(e, changed, cap, &n_changed);
 * CASIM_OK: only those rows of the node table and their share of the domain-rule counters were recomputed, against the
 * dictionaries of the last full finalize; the table pointers of casim_enc_tables / casim_enc_domain_rules stay valid and show the
 * new state; casim_enc_group_rows packs the changed rows for casim_cluster_update_nodes.
 * CASIM_ENC_NEEDS_FULL (> 0): the update would change a dictionary — a new NoSchedule / NoExecute taint, a label value that opens a
 * new topology domain of a rule key, a running spec that needs a node bit or a rule nobody has yet (ports, (anti-)affinity,
 * spread constraints of its own, or a class's anti-affinity term matching it), a node without a hostname label next to hostname
 * bits, nodes or classes added — the session stays open and casim_enc_finalize rebuilds everything from the objects the encoder
 * holds (nothing has to be described again).  Results after an update are identical to a full finalize of the same objects
 * (tests/test_incremental_encode.py compares every column).
 * casim_enc_refinalize with changed_out != NULL and capacity < the number of changed nodes: CASIM_ERR_INVALID, nothing recomputed, the
 * session stays open and *n_changed_out holds the capacity needed (a truncated list would leave the device image stale). */
#define CASIM_ENC_NEEDS_FULL 65
int32_t casim_enc_begin_update(casim_encoder* e);
int32_t casim_enc_group_reset(casim_encoder* e, int32_t group, const int64_t* alloc, int32_t allowed_pods, int64_t capacity_cpu_milli,
                              int64_t capacity_mem_bytes, int32_t unschedulable);
int32_t casim_enc_set_peg_count(casim_encoder* e, int32_t peg, int32_t count);
int32_t casim_enc_refinalize(casim_encoder* e, int32_t* changed_out, int32_t capacity, int32_t* n_changed_out);
/* n rows of the node table as a compact casim_groups (pointers owned by the encoder, valid until the next call).  Besides node deltas
 * (casim_cluster_update_nodes) this is the PER-CALL mode of an estimator shim: the tables of a scale-up loop are encoded ONCE (every PEG,
 * every candidate group), and one Estimate() = this call with n = 1 + the caller's own peg_offsets / peg_index / max_nodes /
 * existing_nodes / last_index written into the returned struct + casim_estimate_batch: no encoder work per Estimate() at all. */
int32_t casim_enc_group_rows(casim_encoder* e, const int32_t* groups, int32_t n, casim_groups* rows_out);

/* Build the dictionaries and the flat tables.  After finalize the views below stay valid
 * until the encoder is destroyed. */
int32_t casim_enc_finalize(casim_encoder* e);
int32_t casim_enc_tables(const casim_encoder* e, casim_pegs* pegs_out, casim_groups* groups_out);
/* Dictionary sizes (bits in use) for reporting: taints, label requirements, node bits, zone bits */
/* per-node mode: the domain rules of the finalized tables (pointers owned by the encoder); n_rules == 0 when none */
int32_t casim_enc_domain_rules(const casim_encoder* enc, casim_domain_rules* out);
/* [n_pegs][w_excl]: the bits of excl_block that come from host ports (NodePorts runs before PodTopologySpread, the
 * hostname anti-affinity bits after it: casim_estimate_on_cluster needs to tell them apart) */
const uint64_t* casim_enc_port_block(const casim_encoder* enc);
int32_t casim_enc_dict_sizes(const casim_encoder* e, int32_t sizes_out[4]);

/* ======================================================================================
 * PREFETCH CACHE of the estimator shim (host code, csrc/casim_prefetch.cpp)
 * ======================================================================================
 * The estimator.Estimator interface hands the estimator ONE node group per call (CA/estimator/estimator.go:53-56;
 * ComputeExpansionOption, CA/core/scaleup/orchestrator/orchestrator.go:383-427).  The shim's NodeGroupListProcessor wrapper sees
 * every candidate group, every template and the pending pods BEFORE that loop (orchestrator.go:121-123): it runs ONE batch over
 * all of them (casim_prefetch_fill) and each Estimate() is a casim_prefetch_lookup.  A lookup hits only when the call asks exactly
 * the question the batch answered: same group key, same PEG list AS A SET (keys are the caller's 64-bit identities: exemplar
 * pod pointer, group id hash + template generation; the orchestrator's list order comes out of a Go map and only breaks score
 * ties), same max_nodes (the limiter's answer after StartEstimation), same E and lastIndex.  Everything else returns CASIM_PREFETCH_MISS with the reason and the shim takes the per-call path (one
 * casim_estimate_batch with one group record): the cache is an accelerator, never the source of truth.
 * A hit may carry status CASIM_NG_UNSUPPORTED: the batch delegated that group (casim_estimate_on_cluster or the Go estimator).
 */
typedef struct casim_prefetch casim_prefetch;
typedef struct casim_prefetch_result {
    int32_t node_count, pods_scheduled, nodes_added, limiter_nodes, last_index_out, status;   /* as casim_results, one group */
    int64_t req_cpu_sum, req_mem_sum;
    int32_t n_pegs;        /* entries written to order_out / placed_out (= the PEG list's length) */
    int32_t miss_reason;   /* CASIM_PREFETCH_MISS_* when the call returned CASIM_PREFETCH_MISS */
} casim_prefetch_result;
#define CASIM_PREFETCH_MISS 64          /* return value of casim_prefetch_lookup (> 0: not an error) */
#define CASIM_PREFETCH_MISS_GROUP 1     /* the batch did not hold this node group (processor not installed, group added since) */
#define CASIM_PREFETCH_MISS_PEGS 2      /* the PEG list differs from the group's schedulable subset of the batch */
#define CASIM_PREFETCH_MISS_LIMITS 3    /* max_nodes / existing_nodes differ from what the batch ran with */
#define CASIM_PREFETCH_MISS_LAST_INDEX 4 /* ABI 12: ONLY last_index differs — a chained batch (casim_options.chain_last_index) whose order the calls left (a
                                          * group ran elsewhere or was skipped): the caller may fill the cache again with the REST of the loop's groups,
                                          * chained from the runner's lastIndex of now (the Go shim's rechain), instead of one per-call trip per group */
casim_prefetch* casim_prefetch_create(casim_ctx* ctx);
void casim_prefetch_destroy(casim_prefetch* p);
void casim_prefetch_clear(casim_prefetch* p);   /* a new loop iteration: forget the previous batch */
const char* casim_prefetch_error(const casim_prefetch* p);
/* One batch over every PEG (peg_key[G]) and every group (group_key[NG]); groups->peg_offsets == NULL lets the device derive the
 * schedulable subsets (SchedulablePodGroups), an explicit CSR keeps the caller's lists.  Replaces the cache's content. */
int32_t casim_prefetch_fill(casim_prefetch* p, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                            const uint64_t* group_key, const uint64_t* peg_key);
/* order_out[k] = position IN THE CALLER'S LIST of the PEG processed k-th, placed_out[k] = how many of its pods were scheduled
 * (Estimate()'s []*Pod = concat_k list[order_out[k]].Pods[0:placed_out[k]]); both [n_pegs], may be NULL. */
int32_t casim_prefetch_lookup(casim_prefetch* p, uint64_t group_key, const uint64_t* peg_keys, int32_t n_pegs, int32_t max_nodes,
                              int32_t existing_nodes, int32_t last_index, casim_prefetch_result* out, int32_t* order_out, int32_t* placed_out);
/* out[0] fills, [1] groups cached in total, [2] hits, [3] misses: unknown group, [4] misses: PEG list, [5] misses: limits or lastIndex,
 * [6] of those: lastIndex alone (CASIM_PREFETCH_MISS_LAST_INDEX) */
int32_t casim_prefetch_stats(const casim_prefetch* p, int64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* CASIM_H_ */
