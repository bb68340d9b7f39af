"""ctypes mirror of include/casim.h (struct layouts and function prototypes only — loading the
shared library happens in _ffi.py).  Kept separate so that test harnesses that consume the same
structs can reuse the layouts without loading the product library."""
import ctypes as C

ABI_VERSION = 12
MAX_RES = 8

OK = 0
ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_NOMEM, ERR_NO_LANE = -1, -2, -3, -4, -5
ENC_DELEGATED = 1
NG_OK, NG_UNSUPPORTED = 0, 1

PEG_TOLERATES_UNSCHEDULABLE = 0x1
PEG_SELF_EXCL_NODE = 0x2
PEG_SELF_EXCL_ZONE = 0x4
PEG_FASTPATH_OK = 0x8
PEG_FASTPATH_AA_SELF = 0x10
PEG_UNSUPPORTED = 0x20
NGF_UNSCHEDULABLE = 0x1

PACK_BUILD_AUTO, PACK_BUILD_PLAIN, PACK_BUILD_OPTION = 0, 1, 2
EXPANDER_LEAST_NODES, EXPANDER_LEAST_WASTE, EXPANDER_MOST_PODS = 0, 1, 2

# casim_feasibility_reasons codes: plugin in the low 4 bits, NodeResourcesFit reasons above
PLUGIN_MASK = 0xF
PLUGIN_NAMES = {0: "", 1: "NodeAffinity", 2: "NodeUnschedulable", 3: "TaintToleration", 4: "NodeAffinity", 5: "NodePorts",
                6: "NodeResourcesFit", 7: "PodTopologySpread", 8: "InterPodAffinity", 15: "<unknown>"}
PLUGIN_REASONS = {1: "PreFilter filtered the Node out", 2: "node(s) were unschedulable", 3: "node(s) had untolerated taint(s)",
                  4: "node(s) didn't match Pod's node affinity/selector", 5: "node(s) didn't have free ports for the requested pod ports",
                  7: "node(s) didn't match pod topology spread constraints", 8: "node(s) didn't satisfy anti-affinity rules"}
REASON_TOO_MANY_PODS = 0x10


def reason_insufficient(lane: int) -> int:
    return 0x20 << lane

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
f64p = C.POINTER(C.c_double)


class Pegs(C.Structure):
    _fields_ = [
        ("n_pegs", C.c_int32), ("n_res", C.c_int32), ("w_taint", C.c_int32), ("w_label", C.c_int32),
        ("w_excl", C.c_int32), ("w_zone", C.c_int32),
        ("req", i64p), ("count", i32p), ("flags", u32p),
        ("tol_mask", u64p), ("sel_mask", u64p), ("excl_block", u64p), ("excl_mark", u64p),
        ("zone_block", u64p), ("zone_mark", u64p), ("fp_cpu", f64p), ("fp_mem", f64p),
        ("zone_polarity", u64p), ("excl_polarity", u64p),
        ("req32", i32p), ("req_unit", i64p),   # ABI 10: requests narrowed by the caller (req may then be NULL)
    ]


class Groups(C.Structure):
    _fields_ = [
        ("n_groups", C.c_int32),
        ("alloc", i64p), ("init_req", i64p), ("allowed_pods", i32p), ("init_pods", i32p), ("flags", u32p),
        ("taint_mask", u64p), ("label_mask", u64p), ("init_excl", u64p), ("init_zone", u64p), ("zone_valid", u64p),
        ("max_nodes", i32p), ("existing_nodes", i32p), ("last_index", i32p),
        ("cap_cpu", f64p), ("cap_mem", f64p), ("waste_cpu", i64p), ("waste_mem", i64p),
        ("peg_offsets", i32p), ("peg_index", i32p),
        ("peg_lo", i32p), ("peg_hi", i32p), ("global_id", i32p), ("n_sims", C.c_int32), ("sim_offsets", i32p),
    ]


class PrefetchResult(C.Structure):
    _fields_ = [("node_count", C.c_int32), ("pods_scheduled", C.c_int32), ("nodes_added", C.c_int32), ("limiter_nodes", C.c_int32),
                ("last_index_out", C.c_int32), ("status", C.c_int32), ("req_cpu_sum", C.c_int64), ("req_mem_sum", C.c_int64),
                ("n_pegs", C.c_int32), ("miss_reason", C.c_int32)]


ENC_NEEDS_FULL = 65
PREFETCH_MISS = 64
PREFETCH_MISS_GROUP, PREFETCH_MISS_PEGS, PREFETCH_MISS_LIMITS, PREFETCH_MISS_LAST_INDEX = 1, 2, 3, 4


class Options(C.Structure):
    _fields_ = [("fastpath", C.c_int32), ("force_generic_packer", C.c_int32), ("node_pods", C.c_int32), ("n_streams", C.c_int32), ("pack_build", C.c_int32), ("no_singleton_merge", C.c_int32), ("no_front_kernel", C.c_int32), ("winners_only", C.c_int32), ("chain_last_index", C.c_int32), ("reserved", C.c_int32 * 3)]


class Results(C.Structure):
    _fields_ = [
        ("node_count", i32p), ("pods_scheduled", i32p), ("nodes_added", i32p), ("limiter_nodes", i32p),
        ("last_index_out", i32p), ("status", i32p), ("req_cpu_sum", i64p), ("req_mem_sum", i64p),
        ("order", i32p), ("placed", i32p),
        ("node_pods", i32p), ("node_pods_offsets", i32p), ("node_pods_capacity", C.c_int64),
    ]


class OptionQuery(C.Structure):
    _fields_ = [("kinds", i32p), ("n_kinds", C.c_int32), ("group_id_base", C.c_int32), ("per_sim", C.c_int32), ("valid", u8p),
                ("best_out", i32p), ("n_best_out", i32p), ("best_set_out", u8p), ("key_out", i64p), ("packed_out", i64p),
                ("dev_key_out", C.c_void_p), ("dev_packed_out", C.c_void_p), ("join_stream", C.c_void_p)]


class EncoderOptions(C.Structure):
    _fields_ = [("n_res", C.c_int32), ("enable_taint_comparison_ops", C.c_int32), ("explicit_self_exclusion", C.c_int32),
                ("reserved", C.c_int32 * 5)]


class DomainRules(C.Structure):
    _fields_ = [("n_keys", C.c_int32), ("n_rules", C.c_int32), ("n_nodes", C.c_int32), ("n_classes", C.c_int32), ("n_elig_rows", C.c_int32),
                ("node_domain", i32p), ("key_domains", i32p), ("key_is_hostname", u8p), ("rule_class", i32p), ("rule_key", i32p), ("rule_kind", i32p),
                ("rule_max_skew", i32p), ("rule_min_domains", i32p), ("rule_self", i32p), ("rule_elig_row", i32p), ("rule_offset", i64p),
                ("count_init", i32p), ("domain_exists", u8p), ("domain_nodes", i32p), ("node_contrib", i32p), ("elig_bits", u64p), ("class_rule_off", i32p), ("inc_off", i32p),
                ("inc_rule", i32p), ("n_taint_policy_rules", C.c_int32), ("rule_ghost_leaves", u8p)]


class PodSequence(C.Structure):
    _fields_ = [("n_pods", C.c_int32), ("pod_class", i32p), ("hint_node", i32p), ("node_acceptable", u8p),
                ("break_on_failure", C.c_int32), ("last_index", C.c_int32), ("rules", C.POINTER(DomainRules)),
                ("similar_key", i32p)]


class ClusterEstimate(C.Structure):
    _fields_ = [("n_existing", C.c_int32), ("max_nodes", C.c_int32), ("last_index", C.c_int32), ("rules", C.POINTER(DomainRules)),
                ("port_block", u64p)]


class ClusterEstimateResult(C.Structure):
    _fields_ = [("node_count", C.c_int32), ("pods_scheduled", C.c_int32), ("nodes_added", C.c_int32), ("limiter_nodes", C.c_int32),
                ("last_index_out", C.c_int32), ("status", C.c_int32), ("req_cpu_sum", C.c_int64), ("req_mem_sum", C.c_int64),
                ("order", i32p), ("placed", i32p)]


class RemovalCandidates(C.Structure):
    _fields_ = [("n_candidates", C.c_int32), ("cand_node", i32p), ("pod_offsets", i32p), ("pod_class", i32p), ("hint_node", i32p),
                ("destination", u8p), ("pod_sticky", u8p), ("cand_atomic", u8p), ("persist", C.c_int32), ("max_removable", C.c_int32),
                ("last_index", C.c_int32), ("ext_capacity", C.c_int32), ("rules", C.POINTER(DomainRules))]


class RemovalResults(C.Structure):
    _fields_ = [("removable", u8p), ("node_out", i32p), ("ext_candidate", i32p), ("ext_pod", i32p), ("ext_node", i32p),
                ("n_ext", C.c_int32), ("last_index", C.c_int32), ("n_processed", C.c_int32)]


cstr = C.c_char_p
cstrp = C.POINTER(C.c_char_p)


class PodColumns(C.Structure):
    """casim_pod_columns (ABI 11): the pods of a loop for casim_enc_add_pods, strings by index into `strings`."""
    _fields_ = [("n_pods", C.c_int32), ("n_strings", C.c_int32), ("strings", cstrp), ("ns", i32p), ("req", i64p), ("fastpath_req", f64p),
                ("peg_count", i32p), ("label_off", i32p), ("label_key", i32p), ("label_val", i32p),
                ("tol_off", i32p), ("tol_key", i32p), ("tol_op", i32p), ("tol_value", i32p), ("tol_effect", i32p),
                ("sel_off", i32p), ("sel_key", i32p), ("sel_val", i32p)]


# name -> (restype, argtypes): every symbol include/casim.h declares
PROTOTYPES = {
    "casim_abi_version": (C.c_int32, []),
    "casim_last_error": (C.c_char_p, []),
    "casim_ctx_create": (C.c_void_p, [C.c_int32, C.c_void_p]),
    "casim_ctx_destroy": (None, [C.c_void_p]),
    "casim_device_count": (C.c_int32, []),
    "casim_host_alloc": (C.c_void_p, [C.c_size_t]),
    "casim_host_free": (None, [C.c_void_p]),
    "casim_problem_create": (C.c_void_p, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(Options)]),
    "casim_problem_destroy": (None, [C.c_void_p]),
    "casim_problem_run": (C.c_int32, [C.c_void_p]),
    "casim_problem_fetch": (C.c_int32, [C.c_void_p, C.POINTER(Results)]),
    "casim_problem_csr": (C.c_int32, [C.c_void_p, i32p, i32p]),
    "casim_problem_info": (C.c_int32, [C.c_void_p, i32p]),
    "casim_problem_set_group_result": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(ClusterEstimateResult)]),
    "casim_estimate_batch": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(Options), C.POINTER(Results)]),
    "casim_feasibility": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), u64p]),
    "casim_feasibility_reasons": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), u64p, C.POINTER(C.c_uint16)]),
    "casim_best_option": (C.c_int32, [C.c_void_p, i32p, C.c_int32, C.c_int32, i32p, i32p, u8p, i64p, C.c_void_p]),
    "casim_best_option_sims": (C.c_int32, [C.c_void_p, C.POINTER(OptionQuery)]),
    "casim_estimate_batch_query": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(Options), C.POINTER(Results), i32p,
                                               C.POINTER(OptionQuery)]),
    "casim_estimate_batch_timed": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(Options), C.POINTER(Results),
                                               C.POINTER(OptionQuery), f64p]),
    "casim_mctx_create": (C.c_void_p, [i32p, C.c_int32, C.c_int32]),
    "casim_mctx_destroy": (None, [C.c_void_p]),
    "casim_mctx_info": (C.c_int32, [C.c_void_p, i32p, i32p, i32p, i32p]),
    "casim_estimate_batch_multi": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(Options), C.POINTER(Results), i32p,
                                               C.POINTER(OptionQuery)]),
    "casim_cluster_create": (C.c_void_p, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups)]),
    "casim_cluster_destroy": (None, [C.c_void_p]),
    "casim_cluster_update_nodes": (C.c_int32, [C.c_void_p, C.c_int32, i32p, C.POINTER(Groups)]),
    "casim_cluster_try_schedule_pods": (C.c_int32, [C.c_void_p, C.POINTER(PodSequence), C.c_int32, i32p, i32p, i32p]),
    "casim_cluster_simulate_node_removals": (C.c_int32, [C.c_void_p, C.POINTER(RemovalCandidates), C.POINTER(RemovalResults)]),
    "casim_cluster_fetch_nodes": (C.c_int32, [C.c_void_p, i64p, i32p, u64p]),
    "casim_cluster_stats": (C.c_int32, [C.c_void_p, i64p]),
    "casim_cluster_forget_commits": (C.c_int32, [C.c_void_p]),
    "casim_problem_time": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "casim_problem_time_feasibility": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), i32p]),
    "casim_problem_run_marked": (C.c_int32, [C.c_void_p]),
    "casim_problem_marked_ms": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "casim_try_schedule_pods": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(PodSequence), i32p, i32p, i32p]),
    "casim_time_try_schedule_pods": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(PodSequence), C.c_int32,
                                                 C.POINTER(C.c_float)]),
    "casim_simulate_node_removals": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(RemovalCandidates),
                                                 C.POINTER(RemovalResults)]),
    "casim_time_node_removals": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(RemovalCandidates), C.c_int32,
                                             C.POINTER(C.c_float)]),
    "casim_last_removals_info": (C.c_int32, [i32p]),
    "casim_last_chain_info": (C.c_int32, [i32p]),
    "casim_pack_build_info": (C.c_int32, [C.c_int32, i32p]),
    "casim_prefetch_create": (C.c_void_p, [C.c_void_p]),
    "casim_prefetch_destroy": (None, [C.c_void_p]),
    "casim_prefetch_clear": (None, [C.c_void_p]),
    "casim_prefetch_error": (C.c_char_p, [C.c_void_p]),
    "casim_prefetch_fill": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(Options), u64p, u64p]),
    "casim_prefetch_lookup": (C.c_int32, [C.c_void_p, C.c_uint64, u64p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(PrefetchResult), i32p, i32p]),
    "casim_prefetch_stats": (C.c_int32, [C.c_void_p, i64p]),
    "casim_copy_bandwidth": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, f64p]),
    "casim_stream_probe": (C.c_int32, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, f64p]),
    "casim_enc_create": (C.c_void_p, [C.POINTER(EncoderOptions)]),
    "casim_enc_destroy": (None, [C.c_void_p]),
    "casim_enc_add_group": (C.c_int32, [C.c_void_p, cstr, i64p, C.c_int32, C.c_int64, C.c_int64, C.c_int32]),
    "casim_enc_group_add_label": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr]),
    "casim_enc_group_add_taint": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr, cstr]),
    "casim_enc_group_set_fastpath_capacity": (C.c_int32, [C.c_void_p, C.c_int32, C.c_double, C.c_double]),
    "casim_enc_group_set_limits": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "casim_enc_group_add_preloaded_pod": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "casim_enc_add_running_pods": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_char_p), C.c_int32]),
    "casim_enc_group_set_pegs": (C.c_int32, [C.c_void_p, C.c_int32, i32p, C.c_int32]),
    "casim_enc_add_pod_spec": (C.c_int32, [C.c_void_p, cstr, i64p]),
    "casim_enc_lane": (C.c_int32, [C.c_void_p, cstr]),
    "casim_enc_pod_set_request": (C.c_int32, [C.c_void_p, C.c_int32, cstr, C.c_int64]),
    "casim_enc_group_set_allocatable": (C.c_int32, [C.c_void_p, C.c_int32, cstr, C.c_int64]),
    "casim_enc_lane_count": (C.c_int32, [C.c_void_p]),
    "casim_enc_lane_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "casim_enc_pod_add_label": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr]),
    "casim_enc_pod_add_toleration": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr, cstr, cstr]),
    "casim_enc_pod_add_node_selector": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr]),
    "casim_enc_pod_add_node_affinity_req": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_pod_add_node_affinity_term": (C.c_int32, [C.c_void_p, C.c_int32]),
    "casim_enc_add_namespace": (C.c_int32, [C.c_void_p, cstr]),
    "casim_enc_namespace_add_label": (C.c_int32, [C.c_void_p, cstr, cstr, cstr]),
    "casim_enc_term_set_namespace_selector": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "casim_enc_term_add_namespace_requirement": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_aff_term_set_namespace_selector": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "casim_enc_aff_term_add_namespace_requirement": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_node_term_add_requirement": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_pod_add_host_port": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstr, C.c_int32]),
    "casim_enc_pod_add_anti_affinity_term": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstrp, C.c_int32]),
    "casim_enc_term_add_requirement": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_pod_add_affinity_term": (C.c_int32, [C.c_void_p, C.c_int32, cstr, cstrp, C.c_int32]),
    "casim_enc_aff_term_add_requirement": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_pod_add_spread_constraint": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, cstr, C.c_int32]),
    "casim_enc_spread_add_requirement": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, cstr, cstr, cstrp, C.c_int32]),
    "casim_enc_spread_set_taints_policy": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "casim_enc_spread_set_affinity_policy": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "casim_enc_domain_rules": (C.c_int32, [C.c_void_p, C.POINTER(DomainRules)]),
    "casim_enc_port_block": (u64p, [C.c_void_p]),
    "casim_estimate_on_cluster": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups), C.POINTER(ClusterEstimate),
                                              C.POINTER(ClusterEstimateResult)]),
    "casim_enc_pod_set_fastpath_requests": (C.c_int32, [C.c_void_p, C.c_int32, C.c_double, C.c_double]),
    "casim_enc_pod_mark_unsupported": (C.c_int32, [C.c_void_p, C.c_int32, cstr]),
    "casim_enc_add_peg": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "casim_enc_add_resource_pegs": (C.c_int32, [C.c_void_p, cstr, C.c_int32, i64p, i32p, i32p]),
    "casim_enc_add_pods": (C.c_int32, [C.c_void_p, C.POINTER(PodColumns), i32p]),
    "casim_enc_pod_set_spec_extra": (C.c_int32, [C.c_void_p, C.c_int32, cstr]),
    "casim_enc_group_pods": (C.c_int32, [C.c_void_p, C.c_int32, i32p, cstrp, u8p, i32p, i32p]),
    "casim_enc_add_grouped_pegs": (C.c_int32, [C.c_void_p, C.c_int32, i32p, i32p, C.c_int32, i32p]),
    "casim_enc_add_existing_pod": (C.c_int32, [C.c_void_p, C.c_int32, cstrp, cstrp, C.c_int32]),
    "casim_enc_finalize": (C.c_int32, [C.c_void_p]),
    "casim_enc_begin_update": (C.c_int32, [C.c_void_p]),
    "casim_enc_group_reset": (C.c_int32, [C.c_void_p, C.c_int32, i64p, C.c_int32, C.c_int64, C.c_int64, C.c_int32]),
    "casim_enc_set_peg_count": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32]),
    "casim_enc_refinalize": (C.c_int32, [C.c_void_p, i32p, C.c_int32, i32p]),
    "casim_enc_group_rows": (C.c_int32, [C.c_void_p, i32p, C.c_int32, C.POINTER(Groups)]),
    "casim_enc_tables": (C.c_int32, [C.c_void_p, C.POINTER(Pegs), C.POINTER(Groups)]),
    "casim_enc_dict_sizes": (C.c_int32, [C.c_void_p, i32p]),
}


def bind(lib):
    """Attach prototypes; raises AttributeError if the library misses a declared symbol."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
