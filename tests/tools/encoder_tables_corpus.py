"""Hashes of the encoder's tables over a corpus of workloads: one line per workload, FNV-1a over every column of casim_pegs /
casim_groups (tools/native_trace.tables_fnv) + the domain rules.  Run it with two builds of the library (CASIM_LIB_PATH) and diff the
outputs: an encoder change that claims "same tables" has to produce the same file.

    python tests/tools/encoder_tables_corpus.py > /tmp/after.txt
    CASIM_LIB_PATH=/tmp/libcasim_base.so python tests/tools/encoder_tables_corpus.py > /tmp/before.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from kubernetes_autoscaler_amd import workloads  # noqa: E402
from kubernetes_autoscaler_amd.encoder import Encoder  # noqa: E402
from kubernetes_autoscaler_amd.scheduling import encode_pending_pods  # noqa: E402
import native_trace as nt  # noqa: E402


def estimate_encoder(w):
    enc = Encoder(lanes=w.lanes)
    for pg in w.pegs:
        enc.add_peg(pg)
    for info in w.existing:
        for p in info.pods:
            enc.add_existing_pod(p, info.node.labels)
    for g in w.groups:
        enc.add_group(g.template, max_nodes=g.max_nodes, existing_nodes=len(w.existing), last_index=g.last_index,
                      pegs=list(g.pegs) if g.pegs is not None else None)
    enc.finalize()
    return enc


def rules_hash(enc):
    """FNV-1a over the domain rules (casim_enc_domain_rules) as the struct of include/casim.h sizes its columns."""
    r = getattr(enc, "rules", None)
    if r is None or (r.n_rules == 0 and r.n_keys == 0):
        return "-"
    import ctypes as C
    h = 0xcbf29ce484222325

    def eat(ptr, n, width):
        nonlocal h
        if not ptr or n <= 0:
            return
        for b in C.string_at(ptr, n * width):
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF

    nk, nr, nn, nc, ne = r.n_keys, r.n_rules, r.n_nodes, r.n_classes, r.n_elig_rows
    for v in (nk, nr, nn, nc, ne, r.n_taint_policy_rules):
        h = ((h ^ (v & 0xFFFFFFFF)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    eat(r.node_domain, nk * nn, 4); eat(r.key_domains, nk, 4); eat(r.key_is_hostname, nk, 1)
    for name in ("rule_class", "rule_key", "rule_kind", "rule_max_skew", "rule_min_domains", "rule_self", "rule_elig_row"):
        eat(getattr(r, name), nr, 4)
    eat(r.rule_offset, nr + 1, 8)
    total = r.rule_offset[nr] if nr > 0 and r.rule_offset else 0
    eat(r.count_init, total, 4); eat(r.domain_exists, total, 1); eat(r.domain_nodes, total, 4)
    eat(r.node_contrib, nr * nn, 4); eat(r.elig_bits, ne * ((nn + 63) // 64), 8)
    eat(r.class_rule_off, nc + 1, 4); eat(r.inc_off, nc + 1, 4)
    eat(r.inc_rule, r.inc_off[nc] if nc > 0 and r.inc_off else 0, 4)
    eat(r.rule_ghost_leaves, nr, 1)
    return f"{h:016x}"


def main():
    rows = []

    def est(name, w):
        enc = estimate_encoder(w)
        rows.append(f"{name} {nt.tables_fnv(enc.pegs, enc.groups)} {rules_hash(enc)}")

    def pend(name, w):
        enc, _ = encode_pending_pods(w.nodes, w.pods)
        rows.append(f"{name} {nt.tables_fnv(enc.pegs, enc.groups)} {rules_hash(enc)}")

    for name in ("config_c0", "config_c1", "config_c2", "config_c3", "config_c4", "config_r2", "config_retry_mix"):
        est(name, getattr(workloads, name)())
    for s in range(1, 4):
        est(f"c2+{s}", workloads.config_c2(seed_offset=s))
        est(f"c4+{s}", workloads.config_c4(seed_offset=s))
    for seed in range(300):
        est(f"fuzz{seed}", workloads.fuzz(seed))
    for seed in range(60):
        est(f"fuzz_singleton{seed}", workloads.fuzz_singleton_runs(seed))
        est(f"fuzz_estimate_domains{seed}", workloads.fuzz_estimate_domains(seed))
    for seed in range(150):
        pend(f"fuzz_pending{seed}", workloads.fuzz_pending(seed))
        pend(f"fuzz_pending_domains{seed}", workloads.fuzz_pending_domains(seed))
    for seed in range(40):
        w = workloads.fuzz_pending(seed)
        workloads.add_random_node_affinity_terms(seed, w.pods, w.nodes)
        pend(f"fuzz_pending_terms{seed}", w)
        w = workloads.fuzz_pending_domains(seed)
        workloads.add_random_pod_affinity(seed, w.pods)
        pend(f"fuzz_pending_aff{seed}", w)
    pend("pending_scale", workloads.pending_scale(400, 4000))
    print("\n".join(rows))


if __name__ == "__main__":
    main()
