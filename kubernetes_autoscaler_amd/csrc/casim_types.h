// casim_types.h — plain-old-data argument blocks shared by the host runtime and the kernels.
#pragma once
#include <stdint.h>

#define CASIM_KMAX_RES 8

// Device-resident copy of casim_pegs + casim_groups (include/casim.h), structure of arrays.
struct DevTables {
    int32_t G, R, Wt, Wl, Wx, Wz, NG;
    int32_t fastpath;
    // PEG table
    const int64_t* req;      // [G][R]
    const int32_t* count;    // [G]
    const uint32_t* pflags;  // [G]
    const uint64_t* tol;     // [G][Wt]
    const uint64_t* sel;     // [G][Wl]
    const uint64_t* xblock;  // [G][Wx]
    const uint64_t* xmark;   // [G][Wx]
    const uint64_t* zblock;  // [G][Wz]
    const uint64_t* zmark;   // [G][Wz]
    const uint64_t* zpol;    // [Wz] group bits of NEED polarity (casim_pegs.zone_polarity; all zero when the caller passed none)
    const uint64_t* xpol;    // [Wx] node bits of NEED polarity (casim_pegs.excl_polarity; all zero when the caller passed none; null outside the template-mode pipeline)
    const double* fp_cpu;    // [G] or null
    const double* fp_mem;    // [G] or null
    // node-group table
    const int64_t* alloc;     // [NG][R]
    const int64_t* init_req;  // [NG][R]
    const int32_t* allowed;   // [NG]
    const int32_t* init_pods; // [NG]
    const uint32_t* gflags;   // [NG]
    const uint64_t* taint;    // [NG][Wt]
    const uint64_t* label;    // [NG][Wl]
    const uint64_t* init_excl;// [NG][Wx]
    const uint64_t* init_zone;// [NG][Wz]
    const uint64_t* zone_valid;// [NG][Wz]
    const int32_t* max_nodes; // [NG]
    const int32_t* existing;  // [NG]
    const int32_t* last_index;// [NG]
    const double* cap_cpu;    // [NG] or null
    const double* cap_mem;    // [NG] or null
    const int64_t* waste_cpu; // [NG] or null
    const int64_t* waste_mem; // [NG] or null
    // schedulable subsets (CSR), device memory
    const int32_t* peg_off;   // [NG+1]
    const int32_t* peg_idx;   // [nnz]
    // candidate PEG range of each group (device-side CSR only): its simulation's PEGs, [0, G) without batching
    const int32_t* peg_lo;    // [NG]
    const int32_t* peg_hi;    // [NG]
    const int32_t* global_id; // [NG] id of the group inside expander keys, or null = group_id_base + index
    const int32_t* sim_off;   // [n_sims + 1] groups of each simulation, or null = one simulation
    int32_t n_sims;
    int32_t lists_from_feas;  // 1 = the per-group PEG lists come from the feasibility kernel: every listed PEG passed the template-level Filters
    // fixed-stride lists (batches, front_sim_kernel): peg_off is STATIC — group ng owns the region [peg_off[ng], peg_off[ng] + (peg_hi - peg_lo)) of
    // order / placed / the record array — and its list length is peg_cnt[ng]; null = compact CSR, length = peg_off[ng + 1] - peg_off[ng]
    const int32_t* peg_cnt;   // [NG] or null
    // casim_options.chain_last_index: null in the first packer pass; in the fix-up passes chain_redo[ng] != 0 marks the groups whose input
    // lastIndex changed (chain_fix_kernel) — every other group's wave leaves at once
    const int32_t* chain_redo;// [NG] or null
};

struct DevResults {
    int32_t* node_count;
    int32_t* pods;
    int32_t* nodes_added;
    int32_t* limiter_nodes;
    int32_t* last_index_out;
    int32_t* status;
    int64_t* cpu_sum;
    int64_t* mem_sum;
    int32_t* order;      // [nnz]
    int32_t* placed;     // [nnz]
    uint8_t* fast_last;  // [NG] 1 => fastpath applies to the last PEG of the sorted list
    // PEG records in processing order, written by order_kernel, streamed by pack_kernel
    int32_t* s_count;    // [nnz]
    uint32_t* s_flags;   // [nnz] CASIM_PEG_* | CASIM_KFLAG_STATIC_OK
    int64_t* s_req;      // [nnz][R]   (generic packer; null when the register packer runs)
    // register packer: ONE record per PEG in processing order instead of the three arrays above (see casim_peg_record
    // below): order_kernel writes it, the packer reads it with one scalar load per PEG
    uint32_t* rec;           // [nnz][rec_dw] or null
    int32_t rec_dw;          // 8 (R <= 2) or 16 (R <= 4, or the int64 record)
    int32_t rec_i64;         // 1: 16-dword records of the int64 register store (two int64 requests as they came, no gcd scaling; req32 / fresh32 null)
    int32_t rec_xw;          // 1: 16-dword records = the 8-dword record of two int32 lanes + the PEG's node-local exclusion words (see below)
    const int32_t* req32;    // [G][R]  gcd-scaled requests (FastScratch::req32)
    const int32_t* fresh32;  // [NG][R] gcd-scaled free resources of an empty node (FastScratch::fresh32)
    // optional (casim_options.node_pods): pods per simulated node, group i at node_pods[node_pods_off[i] ..), node bound entries
    int32_t* node_pods;
    const int64_t* node_pods_off;
};

// kernel-internal flag bit (not part of the ABI): the template-level Filters pass for (PEG, group)
#define CASIM_KFLAG_STATIC_OK 0x80000000u
// kernel-internal PEG flag (never set by a caller): this table row stands for a RUN of adjacent, identical, controller-less pods — k
// singleton PodEquivalenceGroups that the host merged into one row of k pods (casim_pipeline.h, SingletonRuns).  The packer
// places them with the closed forms of one PEG and applies the one rule in which k singleton PEGs differ from a PEG of k pods:
// every pod after the first on a NEW node reaches it through tryToScheduleOnExistingNodes, which moves lastIndex there.
#define CASIM_KFLAG_SINGLETON_RUN 0x40u

// PEG record of the register packer (dwords; RL = 2 or 4 request lanes, record = 8 or 16 dwords):
//   [0] pods of the PEG
//   [1] CASIM_PEG_* flags (bits 0-6) | CASIM_REC_SIMPLE | pods of this PEG that fit an EMPTY node << 8 | CASIM_REC_A2_SIMPLE | CASIM_REC_A2_OK | CASIM_KFLAG_STATIC_OK
//   [2 .. 2+RL) gcd-scaled requests   [2+RL .. 2+3RL) their reciprocals as IEEE doubles (lo, hi), 0.0 for a zero request
// int64 register store (DevResults::rec_i64, 16 dwords): [0], [1] the same; [2..5] two int64 requests (lo, hi) as the boundary carries them;
//   [6..9] their reciprocals; [10..15] zero.  CASIM_REC_SIMPLE there means only "both lanes are requested" (no magnitude condition).
// Two int32 lanes WITH exclusion words (DevResults::rec_xw, round 6; 16 dwords): [0..7] as the 8-dword record; [8..11] xblock[0], xblock[1]
//   (lo, hi each), [12..15] xmark[0], xmark[1] — the PEG's row of casim_pegs.excl_block / excl_mark, zero past Wx.  Until then the packer
//   fetched these words from the mask tables at the head of every PEG step, behind the record: dependent loads nobody had issued ahead — 87 % of
//   the anti-affinity packer's time on BASELINE config C4 (profiles/r15c_pack_phase_profile_c4.txt).  In the record they arrive with it, a step ahead.
// The fresh-node capacity is < 2^21 by eligibility (casim_pipeline.h: pod slots of an empty node).  Everything the packer
// would otherwise derive per PEG with scalar compares is a bit here (the kernel is bound by SCALAR issue, r02n PMC):
#define CASIM_REC_SIMPLE 0x80u            /* every request lane of the record is in (0, 2^30): the branch-free quotient sweep applies */
#define CASIM_REC_A2_OK 0x40000000u       /* template-level Filters pass AND the PEG has pods: existing simulated nodes are worth a visit */
#define CASIM_REC_A2_SIMPLE 0x20000000u   /* A2_OK and SIMPLE in one bit: the test at the head of a PEG step behind a dry limiter */
#define CASIM_REC_FRESH_SHIFT 8
#define CASIM_REC_FRESH_MAX 0x1fffff
#define CASIM_REC_FLAG_MASK 0xc00000ffu

// Per-group scratch geometry of the packer: simulated-node state lives in LDS when every
// group of the launch fits, else in an HBM scratch slab.
struct PackScratch {
    const int32_t* node_cap;   // [NG] node slots reserved for the group (multiple of 64)
    const int64_t* state_off;  // [NG] byte offset into gstate (global variant)
    char* gstate;              // HBM slab (global variant) or null
    int32_t retry_only;        // 1 = only the groups the register packer gave up on (status CASIM_NG_RETRY_INTERNAL) are packed
};
// kernel-internal status (never leaves the library): the register packer ran out of node slots (> 1024 simulated nodes) — the
// generic packer's retry launch, enqueued right behind it, packs the group
#define CASIM_NG_RETRY_INTERNAL 0x7f

// Inputs of the register-resident fast packer: every resource lane divided by the gcd of all its
// values (requests, allocatable, preloaded requests) — exact, and small enough for int32.
struct FastScratch {
    const int32_t* req32;    // [G][R]  req / scale
    const int32_t* fresh32;  // [NG][R] (alloc - init_req) / scale
    const int64_t* scale;    // [R]     gcd per lane (>= 1)
    int64_t* prof;           // [NG][8] phase ticks (CASIM_PACK_PROF builds) or null
};

struct OrderScratch {
    const int64_t* off;  // [NG] byte offset into gbuf (global variant)
    char* gbuf;
    int64_t* prof;       // [NG][4] phase ticks (CASIM_PACK_PROF builds) or null
    int32_t lds_list_cap; // > 0: the launch's LDS holds lists up to this length, longer ones use gbuf (LDS variant only)
};

// bytes of packer state per simulated node (8-byte fields first)
static inline int64_t casim_pack_state_bytes(int R, int Wx, int Wz, int64_t cap) {
    return cap * (8ll * R + 8ll * Wx + 12ll) + 64ll * 8ll * (Wz > 0 ? Wz : 1);
}
