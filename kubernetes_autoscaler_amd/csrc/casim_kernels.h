// casim_kernels.h — the hot-path kernels of the scale-up simulation engine (gfx950, wave64).
//
// Reference path restated here (closed forms, not a translation):
//   K_feas   `fits(peg, fresh node)` for every PEG x node group  — CheckPredicates as used by
//            SchedulablePodGroups, CA/core/scaleup/orchestrator/orchestrator.go:535-570
//   K_csr*   compaction of the feasibility bit-matrix into per-group PEG lists
//   K_order  DecreasingPodOrderer.Order  CA/estimator/decreasing_pod_orderer.go:46-88 and
//            determineBestPEGToFastpath  CA/estimator/binpacking_estimator.go:433-473
//   K_pack   BinpackingNodeEstimator.Estimate  CA/estimator/binpacking_estimator.go:102-342
//            (tryToScheduleOnExistingNodes :163, tryToScheduleOnNewNodes :190, tryFastPath :274,
//            limiter CA/estimator/threshold_based_limiter.go:57-69, node order
//            CA/simulator/clustersnapshot/scheduling_opts.go:54-59) with the Filter arithmetic
//            of NodeResourcesFit (V/.../noderesources/fit.go:678-765), TaintToleration,
//            NodeAffinity/nodeSelector, NodeUnschedulable, NodePorts and hostname / group-wide
//            InterPodAffinity folded into integer compares and bitmask tests.
//   K_option expander filter chain  CA/expander/{leastnodes,waste,mostpods}/*.go
//
// All integer work: no MFMA.  One wavefront simulates one node group; every simulated node is
// owned by exactly one lane (node m <-> lane m & 63), so all node state is lane-private and the
// only cross-lane traffic is ballots / wave reductions (no barriers in K_pack).
#pragma once
#include "casim_device.h"
#include "casim_types.h"

#include "../../include/casim.h"

namespace casim {

// ------------------------------------------------------------------------------------------
// shared predicate pieces
// ------------------------------------------------------------------------------------------
// Static (template-level) Filters of PEG g against group ng:
//   TaintToleration  : every NoSchedule/NoExecute taint tolerated  (taint & ~tol) == 0
//   NodeAffinity     : every required label requirement satisfied  (sel & ~label) == 0
//   NodeUnschedulable: unschedulable template needs the toleration flag
CS_DEVICE bool static_filters_pass(const DevTables& t, int g, int ng) {
    const uint64_t* tol = t.tol + (int64_t)g * t.Wt;
    const uint64_t* tnt = t.taint + (int64_t)ng * t.Wt;
    for (int w = 0; w < t.Wt; ++w)
        if (tnt[w] & ~tol[w]) return false;
    const uint64_t* sel = t.sel + (int64_t)g * t.Wl;
    const uint64_t* lab = t.label + (int64_t)ng * t.Wl;
    for (int w = 0; w < t.Wl; ++w)
        if (sel[w] & ~lab[w]) return false;
    if ((t.gflags[ng] & CASIM_NG_UNSCHEDULABLE) && !(t.pflags[g] & CASIM_PEG_TOLERATES_UNSCHEDULABLE)) return false;
    return true;
}

// How many pods with request `req` fit into (free, slots), clamped to `clampk`.
// fitsRequest: pod count first, then every lane with req > 0 needs req <= alloc - requested
// (fit.go:681-765); k pods fit iff k <= slots and k*req <= free for each such lane.
// `rq` (optional) = 1.0 / (double)req[r], precomputed once per PEG: the quotient is then ONE f64
// multiply + an exact +-1 fix-up instead of a ~100-instruction emulated 64-bit division.
// Exactness: f < 2^53 converts exactly; the quotient d < c <= 2^31, so the estimate's absolute error
// is < 2^-20 and trunc() is within +-1 of floor(f/q); the remainder test restores the exact floor.
CS_DEVICE uint32_t capacity_of(const int64_t* fr, int stride, int32_t slots, int R, const int64_t* req, uint32_t clampk,
                               const double* rq = nullptr) {
    if (slots <= 0) return 0;
    uint32_t c = (uint32_t)slots < clampk ? (uint32_t)slots : clampk;
    for (int r = 0; r < CASIM_KMAX_RES; ++r) {
        if (r >= R) break;
        const int64_t q = req[r];
        if (q > 0) {
            const int64_t f = fr[(int64_t)r * stride];
            if (f < q) return 0;
            // division only when the running bound does not already fit: c*q <= f  => floor(f/q) >= c
            const unsigned __int128 cq = (unsigned __int128)c * (uint64_t)q;
            if (cq > (unsigned __int128)(uint64_t)f) {
                uint64_t d;
                if (rq && f < (1ll << 53)) {
                    uint32_t e = (uint32_t)((double)f * rq[r]);
                    const int64_t rem = f - (int64_t)((uint64_t)e * (uint64_t)q);
                    if (rem < 0) e -= 1;
                    else if (rem >= q) e += 1;
                    d = e;
                } else d = (uint64_t)f / (uint64_t)q;
                c = (uint32_t)d;  // d < c here
            }
        }
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// K_feas: feasibility bit-matrix [NG][ceil(G/64)]
// ------------------------------------------------------------------------------------------
// grid = (ceil(G/256), NG), block = 256.  Thread -> one PEG; a wave's ballot is one output word.
CS_DEVICE bool fits_fresh_node(const DevTables& t, int g, int ng) {
    // A PEG that needs a predicate outside the encoded subset is still judged on the encoded part: failing it is final
    // (the full Filter set only rejects more); passing it puts the PEG on the group's list, where it turns the group
    // CASIM_NG_UNSUPPORTED (pack_unsupported) instead of silently vanishing from the estimate.
    if (!static_filters_pass(t, g, ng)) return false;
    int64_t fr[CASIM_KMAX_RES];
    for (int r = 0; r < CASIM_KMAX_RES; ++r)
        fr[r] = r < t.R ? t.alloc[(int64_t)ng * t.R + r] - t.init_req[(int64_t)ng * t.R + r] : 0;
    const int32_t slots = t.allowed[ng] - t.init_pods[ng];
    if (capacity_of(fr, 1, slots, t.R, t.req + (int64_t)g * t.R, 1u) == 0) return false;
    const uint64_t* xb = t.xblock + (int64_t)g * t.Wx;
    const uint64_t* ix = t.init_excl + (int64_t)ng * t.Wx;
    for (int w = 0; w < t.Wx; ++w) {
        // (NEED bits, casim_pegs.excl_polarity: forbidden while clear — except the ones the PEG marks itself: at snapshot time its first pod
        // passes by the first-pod exception, the encoder only builds such a series when no partner exists anywhere)
        const uint64_t pol = t.xpol ? t.xpol[w] : 0ull;
        const uint64_t b = xb[w] & ~(t.xmark[(int64_t)g * t.Wx + w] & pol);
        if ((b & ix[w]) != (b & pol)) return false;
    }
    const uint64_t* zb = t.zblock + (int64_t)g * t.Wz;
    const uint64_t* iz = t.init_zone + (int64_t)ng * t.Wz;
    for (int w = 0; w < t.Wz; ++w)
        if (zb[w] & (iz[w] ^ t.zpol[w])) return false;   // (a NEED bit forbids while clear)
    return true;
}

// Bit k of row ng stands for PEG peg_lo[ng] + k: a group only ever sees the PEGs of its own simulation.
CS_GLOBAL void feas_kernel(DevTables t, uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg) {
    const int ng = cs::bid_y();
    const int k = cs::bid() * cs::nthreads() + cs::tid();
    const int lo = t.peg_lo[ng], hi = t.peg_hi[ng];
    bool ok = false;   // (a block beyond the group's range still clears its words)
    if (lo + k < hi) ok = fits_fresh_node(t, lo + k, ng);
    const uint64_t b = cs::ballot(ok);
    if (cs::lane() == 0 && (k >> 6) < Wg) bits[(int64_t)ng * Wg + (k >> 6)] = b;
}

// K_feas, simulation-major form for batches: grid = (ceil(L/256), n_sims).  A thread keeps ONE PEG record in registers and walks the
// node groups of its simulation (group records are wave-uniform: scalar loads), one ballot word per group.  The group-major
// kernel above re-read every PEG record once per group and launched 20x the blocks: 0.29 ms of a 2.0 ms step at 4096 C2
// simulations (profiles/r02a_rocpd_summary.txt).  Requires every group of a simulation to share its PEG range and mask widths <= 1
// word (checked by the host, which falls back to feas_kernel otherwise).
// req32 / fresh32 (optional): the gcd-scaled int32 lanes of the register packer (exact: the gcd divides every value of a lane),
// two 32-bit compares per cell instead of 64-bit subtract + compare chains.
// The group records of the simulation are staged in LDS first (dynamic LDS: 128 bytes per group of the largest simulation): one
// coalesced load phase per block, then every group costs a few LDS broadcast reads — as global loads behind the `bits` store
// of the previous group each field was its own dependent round trip (0.146 ms at 4096 x 20 groups, three times the issue time).
// kLean: the batch carries no exclusion words of either kind and at most two (narrowed) resource lanes — the headline's shape: the cell loses
// its two mask tests and the compares of lanes 2 and 3 (12 of ~41 vector instructions per group and wave)
template <bool kLean>
CS_GLOBAL void feas_sim_kernel(DevTables t, uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg, const int32_t* CS_RESTRICT req32,
                               const int32_t* CS_RESTRICT fresh32) {
    const int sim = cs::bid_y();
    const int g0 = t.sim_off[sim], g1 = t.sim_off[sim + 1];
    if (g1 <= g0) return;
    const int lo = t.peg_lo[g0], hi = t.peg_hi[g0];
    const int k = cs::bid() * cs::nthreads() + cs::tid();
    const bool live = lo + k < hi;
    const int g = live ? lo + k : (hi > lo ? lo : 0);
    const bool narrow = req32 != nullptr && t.R <= 4;
    // ---- stage: record i = 16 x uint64: [0] taint [1] label [2] node-local exclusion [3] group-wide exclusion
    //      [4] flags | free pod slots << 32   [5..6] scaled free lanes (int32 pairs, narrow)   [8..15] free lanes (int64, wide)
    uint64_t* grec = (uint64_t*)cs::dyn_smem();
    for (int i = cs::tid(); i < (g1 - g0) * 16; i += cs::nthreads()) {
        const int ng = g0 + (i >> 4), f = i & 15;
        uint64_t v = 0;
        if (f == 0) v = t.Wt ? t.taint[(int64_t)ng * t.Wt] : 0ull;
        else if (f == 1) v = t.Wl ? t.label[(int64_t)ng * t.Wl] : ~0ull;
        else if (f == 2) v = t.Wx ? (t.init_excl[(int64_t)ng * t.Wx] ^ (t.xpol ? t.xpol[0] : 0ull)) : 0ull;   // (NEED bits inverted once, like the group-wide word)
        else if (f == 3) v = t.Wz ? (t.init_zone[(int64_t)ng * t.Wz] ^ t.zpol[0]) : 0ull;   // (NEED bits inverted once: the cell test stays one AND)
        else if (f == 4) v = (uint64_t)t.gflags[ng] | ((uint64_t)(uint32_t)(t.allowed[ng] - t.init_pods[ng]) << 32);
        else if (f == 5 || f == 6) {
            if (narrow) {
                const int r = (f - 5) * 2;
                const uint32_t a = r < t.R ? (uint32_t)fresh32[(int64_t)ng * t.R + r] : 0u, b2 = r + 1 < t.R ? (uint32_t)fresh32[(int64_t)ng * t.R + r + 1] : 0u;
                v = (uint64_t)a | ((uint64_t)b2 << 32);
            }
        } else if (f >= 8) {
            const int r = f - 8;
            if (!narrow && r < t.R) v = (uint64_t)(t.alloc[(int64_t)ng * t.R + r] - t.init_req[(int64_t)ng * t.R + r]);
        }
        grec[i] = v;
    }
    // the PEG record, once
    int64_t req[CASIM_KMAX_RES];
    int32_t rq32[4] = {0, 0, 0, 0};
    for (int r = 0; r < CASIM_KMAX_RES; ++r) {
        if (narrow) { if (r < 4) rq32[r] = (live && r < t.R) ? req32[(int64_t)g * t.R + r] : 0; req[r] = 0; }
        else req[r] = (live && r < t.R) ? t.req[(int64_t)g * t.R + r] : 0;
    }
    const uint32_t pf = live ? t.pflags[g] : 0u;
    const uint64_t tol = (live && t.Wt) ? t.tol[(int64_t)g * t.Wt] : 0ull, sel = (live && t.Wl) ? t.sel[(int64_t)g * t.Wl] : 0ull;
    // (a NEED bit the PEG marks itself does not count on a fresh node: fits_fresh_node)
    const uint64_t xb = (live && t.Wx) ? (t.xblock[(int64_t)g * t.Wx] & ~(t.xmark[(int64_t)g * t.Wx] & (t.xpol ? t.xpol[0] : 0ull))) : 0ull, zb = (live && t.Wz) ? t.zblock[(int64_t)g * t.Wz] : 0ull;
    cs::sync();
    // (branch-free: every test is evaluated and the verdicts are ANDed as lane masks — written with && the compiler built an
    // exec-mask region per test, ~85 scalar instructions per group, 12 % of all scalar instructions of a batch step)
    const bool tolerates_unsched = (pf & CASIM_PEG_TOLERATES_UNSCHEDULABLE) != 0;
    for (int ng = g0; ng < g1; ++ng) {
        const uint64_t* gr = grec + (int64_t)(ng - g0) * 16;   // wave-uniform address: LDS broadcast reads
        const uint64_t fl = gr[4];
        bool ok = live & ((gr[0] & ~tol) == 0) & ((sel & ~gr[1]) == 0);
        if constexpr (!kLean) ok = ok & ((xb & gr[2]) == 0) & ((zb & gr[3]) == 0);
        ok = ok & (tolerates_unsched | (((uint32_t)fl & CASIM_NG_UNSCHEDULABLE) == 0));
        ok = ok & ((int32_t)(fl >> 32) > 0);                                      // fitsRequest: pod count first (fit.go:681-690)
        if (narrow) {   // wave-uniform; lanes past R carry a zero request, which passes
            const uint64_t f01 = gr[5], f23 = gr[6];
            bool fit = (rq32[0] <= 0) | (rq32[0] <= (int32_t)(uint32_t)f01);
            fit = fit & ((rq32[1] <= 0) | (rq32[1] <= (int32_t)(f01 >> 32)));
            if constexpr (!kLean) {
                fit = fit & ((rq32[2] <= 0) | (rq32[2] <= (int32_t)(uint32_t)f23));
                fit = fit & ((rq32[3] <= 0) | (rq32[3] <= (int32_t)(f23 >> 32)));
            }
            ok = ok & fit;   // (a pod without requests, whose lane tests the reference skips, :699, passes every one of them)
        } else {
            bool fit = true;
#pragma unroll
            for (int r = 0; r < CASIM_KMAX_RES; ++r) fit = fit & ((req[r] <= 0) | (req[r] <= (int64_t)gr[8 + r]));   // every requested lane fits once (:699-752)
            ok = ok & fit;
        }
        const uint64_t b = cs::ballot(ok);
        if (cs::lane() == 0 && (k >> 6) < Wg) bits[(int64_t)ng * Wg + (k >> 6)] = b;
    }
}

// ------------------------------------------------------------------------------------------
// K_feas, round 5: the streaming form (feas_stream_kernel) — the kernel BASELINE.md section 4 prices against the HBM roofline
// ------------------------------------------------------------------------------------------
// Same cells, same output rows as feas_sim_kernel (narrowed int32 lanes, one word per mask kind, every group of a simulation on one PEG
// range), rebuilt around what that kernel spent its time on: ~29 vector instructions + 4 LDS reads per (wave, group) — every field of the
// group record its own LDS read, every test its own 64-bit compare, the ballot word through two v_mov and a lane-0 store.  Here
//   * the group record is built ONCE per problem (feas_group_records_kernel, at init) with the wave-uniform gates folded INTO the data: a
//     group without a free pod slot gets f0 = INT32_MIN (every lane fails the lane-0 compare: "pod count first", fit.go:681-690), the label
//     word is stored inverted, and the NodeUnschedulable test rides on a SPARE BIT of the taint or label word when the batch's dictionaries
//     leave one (group side: "template is unschedulable", PEG side: "does not tolerate it") — else it is one more AND-OR term;
//   * a block copies its simulation's records into LDS with the SAME burst of loads that fetches its PEG columns: after one wait the group
//     loop touches no memory but LDS (a first version walked the records with scalar loads two ahead: every record a dependent miss of its
//     own, waves parked 80 % of their 4.5 us lives — profiles/r10c_feas_rocpd_summary.txt — and half the issue slots empty);
//   * the PEG side is pre-inverted once per lane: ~tol, requests with rq <= 0 replaced by INT32_MIN + 1 (a lane nobody asks for passes every
//     free amount, fit.go:699, and still fails the INT32_MIN gate);
//   * all static Filters of a cell are ONE accumulated word, x = (taint & ~tol) | (sel & ~label) [| upper halves | exclusion words], built
//     from v_and_or_b32 and tested by ONE compare; the requests by one 32-bit compare per lane; the lane masks meet on the SCALAR unit;
//   * the ballot words of up to 64 groups are parked in the lanes of two VGPRs (v_writelane, lane j = group j) and leave with ONE store
//     instruction per 64 groups.
// Lean cell, dictionaries in the lower halves, a spare bit for NodeUnschedulable (BASELINE config C2): ONE ds_read_b128 and 7 vector
// instructions per (wave, group) — v_and, v_and_or, v_cmp_eq, 2 x v_cmp_ge, 2 x v_writelane; 10 with every term.  grid = ONE dimension,
// XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs, so the blocks of a simulation take ids that agree mod 8 — its group
// records are fetched into ONE XCD's L2.
// Algorithmic bytes (history: docs/HISTORY.md 17a): per PEG the columns the cell needs as this kernel reads them — 4 R (narrowed requests) + 4
// (flags) + 8 (tolerations) + 8 (selector) [+ 8 + 8 exclusion words] — per group its 64-byte record, per cell one bit.
#define CASIM_FEAS_REC_DW 16
// record (dwords): [0] taint lo  [1] ~label lo  [2] f0 | INT32_MIN gate  [3] f1      (all a lean cell with narrow dictionaries reads)
//                  [4] taint hi  [5] ~label hi  [6] unschedulable (0 / 1)  [7] 0
//                  [8,9] node-local exclusion ^ NEED polarity  [10,11] group-wide exclusion ^ NEED polarity  [12] f2  [13] f3  [14,15] 0
// us_word: 0 = the NodeUnschedulable bit rides in the taint word, 1 = in the label word, at bit us_bit; -1 = nowhere ([6] is its own term)
CS_GLOBAL void feas_group_records_kernel(DevTables t, const int32_t* CS_RESTRICT fresh32, uint32_t* CS_RESTRICT rec /*[NG][16]*/, int us_word, int us_bit) {
    const int ng = cs::bid() * cs::nthreads() + cs::tid();
    if (ng >= t.NG) return;
    uint32_t* r = rec + (int64_t)ng * CASIM_FEAS_REC_DW;
    uint64_t taint = t.Wt ? t.taint[(int64_t)ng * t.Wt] : 0ull, nlabel = t.Wl ? ~t.label[(int64_t)ng * t.Wl] : 0ull;
    const uint64_t ex = t.Wx ? (t.init_excl[(int64_t)ng * t.Wx] ^ (t.xpol ? t.xpol[0] : 0ull)) : 0ull;
    const uint64_t zn = t.Wz ? (t.init_zone[(int64_t)ng * t.Wz] ^ t.zpol[0]) : 0ull;
    int32_t f[4];
    for (int k = 0; k < 4; ++k) f[k] = k < t.R ? fresh32[(int64_t)ng * t.R + k] : 0x7fffffff;
    if (t.allowed[ng] - t.init_pods[ng] <= 0) f[0] = (int32_t)0x80000000;
    const uint64_t unsched = (t.gflags[ng] & CASIM_NG_UNSCHEDULABLE) ? 1ull : 0ull;
    if (us_word == 0) taint = (taint & ~(1ull << us_bit)) | (unsched << us_bit);
    if (us_word == 1) nlabel = (nlabel & ~(1ull << us_bit)) | (unsched << us_bit);
    r[0] = (uint32_t)taint; r[1] = (uint32_t)nlabel; r[2] = (uint32_t)f[0]; r[3] = (uint32_t)f[1];
    r[4] = (uint32_t)(taint >> 32); r[5] = (uint32_t)(nlabel >> 32); r[6] = (uint32_t)unsched; r[7] = 0;
    r[8] = (uint32_t)ex; r[9] = (uint32_t)(ex >> 32); r[10] = (uint32_t)zn; r[11] = (uint32_t)(zn >> 32);
    r[12] = (uint32_t)f[2]; r[13] = (uint32_t)f[3]; r[14] = 0; r[15] = 0;
}

// OR of two word arrays (the groups' taint words, the PEGs' selector words) -> out[0], out[1]: which bits of the cells' `x` can ever be set.
// (The other two operands do not matter: a zero taint bit kills its term whatever ~tol holds, a zero selector bit whatever ~label holds.)
// The host reads them behind the wait a resident problem's init ends with: upper halves unused -> the kHi terms go; a bit nobody uses ->
// NodeUnschedulable rides there.
CS_GLOBAL void mask_or_kernel(const uint64_t* CS_RESTRICT a, int64_t na, const uint64_t* CS_RESTRICT b, int64_t nb, uint64_t* CS_RESTRICT out /*[2]*/) {
    const int64_t stride = (int64_t)cs::nblocks() * cs::nthreads();
    uint64_t acc_a = 0, acc_b = 0;
    for (int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid(); i < na; i += stride) acc_a |= a[i];
    for (int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid(); i < nb; i += stride) acc_b |= b[i];
    if (acc_a) cs::atomic_or_u64(out, acc_a);
    if (acc_b) cs::atomic_or_u64(out + 1, acc_b);
}

// kLean: no exclusion words, at most two request lanes.  kHi: some taint / label-requirement bit lies in an upper half.  kUnschedTerm:
// NodeUnschedulable as a term of its own (no spare bit, or nobody looked: one-shot calls) — else the records carry it at (us_word, us_bit).
// Dynamic LDS: (groups of the largest simulation, rounded up to 4) x 64 bytes.
template <bool kLean, bool kHi, bool kUnschedTerm>
CS_GLOBAL CS_LAUNCH_BOUNDS(256, 1) void feas_stream_kernel(DevTables t, uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg, const int32_t* CS_RESTRICT req32,
                                                          const uint32_t* CS_RESTRICT grec /*[NG][16]*/, int gx /* blocks per simulation */, int n_sims,
                                                          int us_word, int us_bit) {
    // workgroup id -> (simulation, block of the simulation): ids that agree mod 8 run on one XCD
    const int id = cs::bid();
    const int chunk = id / (8 * gx), within = id - chunk * (8 * gx);
    const int sim = chunk * 8 + (within & 7), bx = within >> 3;
    if (sim >= n_sims) return;
    const int g0 = cs::uniform_i32(t.sim_off[sim]), g1 = cs::uniform_i32(t.sim_off[sim + 1]);
    if (g1 <= g0) return;
    const int lo = cs::uniform_i32(t.peg_lo[g0]), hi = cs::uniform_i32(t.peg_hi[g0]);
    const int k = bx * cs::nthreads() + cs::tid();
    const int word = cs::uniform_i32(k >> 6);
    const bool live = lo + k < hi;
    const int g = live ? lo + k : (hi > lo ? lo : 0);
    const int R = t.R;
    // ---- the PEG, once: pre-inverted so that a cell is AND-OR terms and signed compares (these loads and the record copy below are in flight together)
    int32_t rq[4];
    if (R == 2) {   // (the usual shape: one 8-byte load)
        rq[0] = req32[(int64_t)g * 2]; rq[1] = req32[(int64_t)g * 2 + 1]; rq[2] = 0; rq[3] = 0;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) rq[r] = (r < R && r < (kLean ? 2 : 4)) ? req32[(int64_t)g * R + r] : 0;
    }
    const uint32_t pf = t.pflags[g];
    const uint64_t tol = t.Wt ? t.tol[(int64_t)g * t.Wt] : 0ull, sel = t.Wl ? t.sel[(int64_t)g * t.Wl] : 0ull;
    uint64_t xb = 0, zb = 0;
    if constexpr (!kLean) {
        // (a NEED bit the PEG marks itself does not count on a fresh node: fits_fresh_node)
        xb = t.Wx ? (t.xblock[(int64_t)g * t.Wx] & ~(t.xmark[(int64_t)g * t.Wx] & (t.xpol ? t.xpol[0] : 0ull))) : 0ull;
        zb = t.Wz ? t.zblock[(int64_t)g * t.Wz] : 0ull;
    }
    // ---- the simulation's group records -> LDS (16-byte pieces; the lean cell reads the first 32 bytes of a record, all 64 are copied: one layout)
    uint32_t* lrec = (uint32_t*)cs::dyn_smem();
    {
        const int n4 = (g1 - g0) * 4;
        const uint32_t* src = grec + (int64_t)g0 * CASIM_FEAS_REC_DW;
        for (int i = cs::tid(); i < n4; i += cs::nthreads()) {
            const cs::Words<4> q = cs::load4(src + (int64_t)i * 4);
            cs::store4(lrec + (int64_t)i * 4, q);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rq[r] = rq[r] > 0 ? rq[r] : (int32_t)0x80000001;
    const uint32_t no_unsched = (pf & CASIM_PEG_TOLERATES_UNSCHEDULABLE) ? 0u : 1u;
    uint64_t ntol = ~tol, sel2 = sel;
    if constexpr (!kUnschedTerm) {
        // the PEG's side of the NodeUnschedulable bit: set = "does not tolerate" (ANDed with the group's "is unschedulable")
        const uint64_t b = (uint64_t)no_unsched << us_bit;
        if (us_word == 0) ntol = (ntol & ~(1ull << us_bit)) | b; else sel2 = (sel2 & ~(1ull << us_bit)) | b;
    }
    const uint32_t ntol_lo = (uint32_t)ntol, ntol_hi = (uint32_t)(ntol >> 32), sel_lo = (uint32_t)sel2, sel_hi = (uint32_t)(sel2 >> 32);
    const uint32_t xb_lo = (uint32_t)xb, xb_hi = (uint32_t)(xb >> 32), zb_lo = (uint32_t)zb, zb_hi = (uint32_t)(zb >> 32);
    cs::sync();
    if (word >= Wg) return;                       // (wave-uniform, behind the barrier: a wave past the row's last word has nothing to write)
    const uint64_t live_mask = cs::ballot(live);
    uint32_t out_lo = 0, out_hi = 0;              // lane j: the ballot word of group base + j
    uint64_t* const row0 = bits + (int64_t)g0 * Wg + word;
    const int lane = cs::lane();
    // one cell row: the group's record out of LDS (a wave-uniform address: broadcast reads), the static word, the compares
    struct Rec { cs::Words<4> a, b, c, d; };
    auto fetch = [&](const uint32_t* q) -> Rec {
        Rec r;
        r.a = cs::load4(q);
        if constexpr (kHi || kUnschedTerm) r.b = cs::load4(q + 4);
        if constexpr (!kLean) { r.c = cs::load4(q + 8); r.d = cs::load4(q + 12); }
        return r;
    };
    auto cell = [&](const Rec& r) -> uint64_t {
        uint32_t x = r.a.w[0] & ntol_lo;
        x = cs::and_or_vvv(r.a.w[1], sel_lo, x);
        if constexpr (kHi) { x = cs::and_or_vvv(r.b.w[0], ntol_hi, x); x = cs::and_or_vvv(r.b.w[1], sel_hi, x); }
        if constexpr (kUnschedTerm) x = cs::and_or_vvv(r.b.w[2], no_unsched, x);
        if constexpr (kLean) {
            return cs::ballot(x == 0) & cs::ballot(rq[0] <= (int32_t)r.a.w[2]) & cs::ballot(rq[1] <= (int32_t)r.a.w[3]);
        } else {
            x = cs::and_or_vvv(r.c.w[0], xb_lo, x); x = cs::and_or_vvv(r.c.w[1], xb_hi, x); x = cs::and_or_vvv(r.c.w[2], zb_lo, x); x = cs::and_or_vvv(r.c.w[3], zb_hi, x);
            return cs::ballot(x == 0) & cs::ballot(rq[0] <= (int32_t)r.a.w[2]) & cs::ballot(rq[1] <= (int32_t)r.a.w[3]) &
                   cs::ballot(rq[2] <= (int32_t)r.d.w[0]) & cs::ballot(rq[3] <= (int32_t)r.d.w[1]);
        }
    };
    for (int base = g0; base < g1; base += 64) {
        const uint32_t n = cs::scalar_min_u32((uint32_t)(g1 - base), 64u);
        const uint32_t* q = lrec + (int64_t)(base - g0) * CASIM_FEAS_REC_DW;
        // four groups per step (immediate LDS offsets), each record read a group ahead of its use.  A count that is no multiple of four
        // runs up to three records over and the look-ahead one more — inside the LDS allocation, which is rounded up to four records plus
        // four; the words of the extra cells land in lanes >= n, which are not stored
        Rec cur = fetch(q);
        for (uint32_t j = 0; j < n; j += 4) {
#pragma unroll
            for (uint32_t u = 0; u < 4; ++u) {
                const Rec nxt = fetch(q + (int64_t)(j + u + 1) * CASIM_FEAS_REC_DW);
                cs::write_lane2_u32(out_lo, out_hi, cell(cur), j + u);
                cur = nxt;
            }
        }
        // one store instruction for up to 64 rows; lanes past the row's end of the simulation drop out (dead PEGs: masked here, once)
        if (lane < (int)n) row0[(int64_t)(base - g0 + lane) * Wg] = (((uint64_t)out_hi << 32) | out_lo) & live_mask;
    }
}

// K_reason: the SchedulingError of every cell of the SchedulablePodGroups matrix (see casim_feasibility_reasons): first
// failing Filter plugin in the scheduler's Filter order + the reasons of NodeResourcesFit.  Same geometry as K_feas.
CS_DEVICE uint32_t fresh_node_verdict(const DevTables& t, const uint64_t* CS_RESTRICT port_block, int g, int ng) {
    const uint32_t pf = t.pflags[g];
    if ((t.gflags[ng] & CASIM_NG_UNSCHEDULABLE) && !(pf & CASIM_PEG_TOLERATES_UNSCHEDULABLE)) return CASIM_PLUGIN_NODE_UNSCHEDULABLE;
    const uint64_t* tol = t.tol + (int64_t)g * t.Wt;
    const uint64_t* tnt = t.taint + (int64_t)ng * t.Wt;
    for (int w = 0; w < t.Wt; ++w) if (tnt[w] & ~tol[w]) return CASIM_PLUGIN_TAINT_TOLERATION;
    const uint64_t* sel = t.sel + (int64_t)g * t.Wl;
    const uint64_t* lab = t.label + (int64_t)ng * t.Wl;
    for (int w = 0; w < t.Wl; ++w) if (sel[w] & ~lab[w]) return CASIM_PLUGIN_NODE_AFFINITY;
    const uint64_t* xb = t.xblock + (int64_t)g * t.Wx;
    const uint64_t* ix = t.init_excl + (int64_t)ng * t.Wx;
    bool other_excl = false;
    for (int w = 0; w < t.Wx; ++w) {
        const uint64_t pol = t.xpol ? t.xpol[w] : 0ull;
        const uint64_t b = xb[w] & ~(t.xmark[(int64_t)g * t.Wx + w] & pol);
        const uint64_t hit = (b & ix[w]) ^ (b & pol);   // a plain bit that is set, a NEED bit (required pod affinity on the hostname) that is clear
        if (port_block && (hit & port_block[(int64_t)g * t.Wx + w])) return CASIM_PLUGIN_NODE_PORTS;
        other_excl = other_excl || hit != 0;
    }
    // NodeResourcesFit: every reason, as fitsRequest collects them (fit.go:681-765)
    uint32_t fit = 0;
    if (t.init_pods[ng] + 1 > t.allowed[ng]) fit |= CASIM_REASON_TOO_MANY_PODS;
    bool all_zero = true;
    for (int r = 0; r < t.R; ++r) all_zero = all_zero && t.req[(int64_t)g * t.R + r] == 0;
    if (!all_zero)
        for (int r = 0; r < t.R; ++r) {
            const int64_t q = t.req[(int64_t)g * t.R + r];
            if (q > 0 && q > t.alloc[(int64_t)ng * t.R + r] - t.init_req[(int64_t)ng * t.R + r]) fit |= CASIM_REASON_INSUFFICIENT(r);
        }
    if (fit) return CASIM_PLUGIN_NODE_RESOURCES_FIT | fit;
    if (other_excl) return CASIM_PLUGIN_INTER_POD_AFFINITY;
    const uint64_t* zb = t.zblock + (int64_t)g * t.Wz;
    const uint64_t* iz = t.init_zone + (int64_t)ng * t.Wz;
    for (int w = 0; w < t.Wz; ++w) if (zb[w] & (iz[w] ^ t.zpol[w])) return CASIM_PLUGIN_INTER_POD_AFFINITY;
    if (pf & CASIM_PEG_UNSUPPORTED) return CASIM_PLUGIN_UNKNOWN;
    return CASIM_PLUGIN_NONE;
}
CS_GLOBAL void reason_kernel(DevTables t, const uint64_t* CS_RESTRICT port_block, uint16_t* CS_RESTRICT out /*[NG][L]*/, int L) {
    const int ng = cs::bid_y();
    const int k = cs::bid() * cs::nthreads() + cs::tid();
    if (k >= L) return;
    const int lo = t.peg_lo[ng], hi = t.peg_hi[ng];
    out[(int64_t)ng * L + k] = lo + k < hi ? (uint16_t)fresh_node_verdict(t, port_block, lo + k, ng) : (uint16_t)0;
}

// K_csr_count: nnz per group; one block (256 threads) per group.
CS_GLOBAL void csr_count_kernel(const uint64_t* CS_RESTRICT bits, int Wg, int32_t* CS_RESTRICT counts) {
    const int ng = cs::bid();
    uint32_t c = 0;
    for (int w = cs::tid(); w < Wg; w += cs::nthreads()) c += (uint32_t)cs::popc64(bits[(int64_t)ng * Wg + w]);
    c = cs::wave_sum_u32(c);
    uint32_t* sm = (uint32_t*)cs::dyn_smem();
    const int wave = cs::tid() >> 6, nw = (cs::nthreads() + 63) >> 6;
    if (cs::lane() == 0) sm[wave] = c;
    cs::sync();
    if (cs::tid() == 0) {
        uint32_t s = 0;
        for (int i = 0; i < nw; ++i) s += sm[i];
        counts[ng] = (int32_t)s;
    }
}
// K_csr_scan: exclusive scan of the per-group counts -> offsets[NG+1], in two steps so that a batch of thousands of
// simulations (tens of thousands of groups) is scanned by many blocks with coalesced loads instead of one block walking
// serial chunks (0.128 ms of a 2.3 ms step at NG = 81920, profiles/r02a_rocpd_summary.txt):
//   csr_scan_local_kernel  block b scans its 1024 groups (one element per thread: wave scan by lane exchange + LDS across
//                          waves), writes the block-local exclusive prefix and its block total; with rows of <= 16 words the
//                          popcount of the row (K_csr_count) is folded in;
//   csr_fill_kernel        (below) adds the totals of the blocks in front of a group's block while it fills the group's list and
//                          writes the final offsets (a fix-up launch of its own until the end of round 2).
CS_DEVICE uint32_t block_exclusive_scan(uint32_t mine, uint32_t* sm /*[nw + 1]*/, uint32_t* total) {
    const int tid = cs::tid(), lane = cs::lane(), wave = tid >> 6, nw = (cs::nthreads() + 63) >> 6;
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)cs::readlane_u64(incl, lane >= d ? lane - d : lane);
        if (lane >= d) incl += o;
    }
    if (lane == 63) sm[wave] = incl;
    cs::sync();
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const uint32_t v = sm[w]; if (w < wave) base += v; tot += v; }
    *total = tot;
    return base + incl - mine;
}
CS_GLOBAL void csr_scan_local_kernel(const int32_t* CS_RESTRICT counts, const uint64_t* CS_RESTRICT bits, int Wg, int NG,
                                     int32_t* CS_RESTRICT offsets, int32_t* CS_RESTRICT block_sums, int32_t* CS_RESTRICT counts_out) {
    const int i = cs::bid() * cs::nthreads() + cs::tid();
    uint32_t mine = 0;
    if (i < NG) {
        if (bits) { for (int w = 0; w < Wg; ++w) mine += (uint32_t)cs::popc64(bits[(int64_t)i * Wg + w]); counts_out[i] = (int32_t)mine; }
        else mine = (uint32_t)counts[i];
    }
    uint32_t total = 0;
    const uint32_t excl = block_exclusive_scan(mine, (uint32_t*)cs::dyn_smem(), &total);
    if (i < NG) offsets[i] = (int32_t)excl;
    if (cs::tid() == 0) block_sums[cs::bid()] = (int32_t)total;
}
// K_csr_fill: PEG ids of each group in ascending order; one wave per group, 64 words per step.
// local_offsets (optional): the block-local exclusive offsets of csr_scan_local_kernel together with its block totals — the wave
// adds the totals of the blocks in front of its own (a handful of values) and writes the FINAL offset of its group (the last
// group also the grand total): the separate fix-up launch of the scan is folded into this kernel (one launch less in every
// stream's chain of a batch step; the chain's length is the step's).
CS_GLOBAL void csr_fill_kernel(const uint64_t* CS_RESTRICT bits, int Wg, int32_t* CS_RESTRICT offsets,
                               int32_t* CS_RESTRICT idx, const int32_t* CS_RESTRICT peg_lo,
                               const int32_t* CS_RESTRICT local_offsets, const int32_t* CS_RESTRICT block_sums, int scan_threads, int NG,
                               const int32_t* CS_RESTRICT counts) {
    const int ng = cs::bid();
    const int lane = cs::lane();
    const int lo = peg_lo[ng];
    int32_t base;
    if (local_offsets) {
        const int b = ng / scan_threads;
        uint32_t before = 0;
        for (int k = lane; k < b; k += 64) before += (uint32_t)block_sums[k];
        base = local_offsets[ng] + (int32_t)cs::wave_sum_u32(before);
        if (lane == 0) { offsets[ng] = base; if (ng == NG - 1) offsets[NG] = base + counts[ng]; }
    } else base = offsets[ng];
    for (int w0 = 0; w0 < Wg; w0 += 64) {
        const int w = w0 + lane;
        uint64_t word = w < Wg ? bits[(int64_t)ng * Wg + w] : 0ull;
        const uint32_t pc = (uint32_t)cs::popc64(word);
        // exclusive prefix of pc over lanes (log-step scan through wave exchange)
        uint32_t incl = pc;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)cs::readlane_u64(incl, lane >= d ? lane - d : lane);
            if (lane >= d) incl += o;
        }
        int32_t pos = base + (int32_t)(incl - pc);
        while (word) {
            const int b = cs::ffs64(word);
            word &= word - 1;
            idx[pos++] = lo + w * 64 + b;
        }
        base += (int32_t)cs::readlane_u64(incl, 63);
    }
}

// ------------------------------------------------------------------------------------------
// K_order: PEG order per group (+ fastpath chooser)
// ------------------------------------------------------------------------------------------
// score = cpuReq/cpuAlloc + memReq/memAlloc in IEEE double, each term only if alloc > 0
// (decreasing_pod_orderer.go:76-81).  Sorted descending; ties keep input order (canonical rule:
// Go's sort.Slice is unstable, insertion-sort-stable for <= 12 elements).
CS_DEVICE double peg_score(const DevTables& t, int g, int ng) {
    if (t.count[g] <= 0) return 0.0;  // Exemplar() == nil
    const int64_t ca = t.alloc[(int64_t)ng * t.R + 0], ma = t.alloc[(int64_t)ng * t.R + 1];
    double s = 0.0;
    if (ca > 0) s += (double)t.req[(int64_t)g * t.R + 0] / (double)ca;
    if (ma > 0) s += (double)t.req[(int64_t)g * t.R + 1] / (double)ma;
    return s;
}
CS_DEVICE uint64_t desc_key(double s) {
    uint64_t u = cs::double_bits(s);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ull);  // order-preserving map of doubles
    return ~u;                                          // ascending key == descending score
}
// simulationsSaved of determineBestPEGToFastpath (:436-466); -1 = not eligible
CS_DEVICE int32_t fastpath_saved(const DevTables& t, int g, int ng) {
    const int32_t n = t.count[g];
    if (n <= 0 || !(t.pflags[g] & CASIM_PEG_FASTPATH_OK)) return -1;
    int32_t by_aa = (t.pflags[g] & CASIM_PEG_FASTPATH_AA_SELF) ? n : 0;
    int32_t by_cpu = 0, by_mem = 0;
    if (t.fp_cpu && t.cap_cpu) by_cpu = (int32_t)ceil((double)n * t.fp_cpu[g] / t.cap_cpu[ng]);
    if (t.fp_mem && t.cap_mem) by_mem = (int32_t)ceil((double)n * t.fp_mem[g] / t.cap_mem[ng]);
    int32_t nodes = by_aa > by_cpu ? by_aa : by_cpu;
    nodes = nodes > by_mem ? nodes : by_mem;
    return nodes > 0 ? n - n / nodes : 0;
}

struct alignas(16) RecQuad { uint32_t a, b, c, d; };
template <int N> struct IntTag { static constexpr int value = N; };
// One block per group.  Bitonic sort of (key, position) pairs in LDS, or in an HBM scratch slab when the group's PEG list
// does not fit (kLds == false).
// NPAD > 0: the block is ONE wave and the padded list length is the constant NPAD (64 / 128 / 256): every loop over the
// list unrolls (the gathers of a thread's 1-4 PEGs are all in flight together instead of one dependent chain per PEG) and
// the sorting network is straight-line code with constant masks and strides — no inner pair loop, no loop control, half
// the address arithmetic (sort phase 18.9 k -> 13.0 k cycles per group, profiles/r02v).
// NPAD == 0: any block size / list length (the loops are runtime loops).
// kWave: ONE WAVE of a larger block orders the group on its own (front_sim_kernel: the waves of a simulation's block take its groups in
// turn) — lane instead of thread index, wave_sync instead of the block barrier, `smem` = the wave's own LDS scratch laid out as
// pos[NPAD] | gid[NPAD] | red[64], with gid ALREADY holding the list (PEG ids in ascending order); NPAD > 0 only.
// smem != null without kWave: the block's scratch starts there instead of at the dynamic LDS base.
// kSorted (with kWave, NPAD == 0): `smem` holds gid[Gn] = the group's PEGs ALREADY in processing order (order_ranked_kernel: ranked once per
// allocatable pair, compacted against the group's feasibility row) — no scores, no sort, the records only; never with the fastpath.
template <bool kLds, int NPAD, bool kWave = false, bool kSorted = false>
CS_DEVICE void order_group(const DevTables& t, const DevResults& res, const OrderScratch& os, const int ng, const int off, const int Gn, char* smem = nullptr) {
    static_assert(!kWave || NPAD > 0 || kSorted, "a single wave sorts in its registers");
    static_assert(!kSorted || (kWave && NPAD == 0), "a sorted list is handed over by ONE wave");
    const int tid = kWave ? cs::lane() : cs::tid();
    const int nt = (NPAD > 0 || kWave) ? 64 : cs::nthreads();
    auto barrier = [&]() { if constexpr (kWave) cs::wave_sync(); else cs::sync(); };
    int npad = NPAD > 0 ? NPAD : 1;
    if (NPAD == 0) while (npad < Gn) npad <<= 1;
#if defined(CASIM_PACK_PROF) && !defined(CASIM_HOST_EMU)
    uint64_t oprof[4] = {0, 0, 0, 0}; uint64_t oprof_last = __builtin_amdgcn_s_memtime();
#define CASIM_OPROF(i) do { const uint64_t _n = __builtin_amdgcn_s_memtime(); oprof[i] += _n - oprof_last; oprof_last = _n; } while (0)
#else
#define CASIM_OPROF(i)
#endif
    char* base = smem ? smem : (kLds ? cs::dyn_smem() : os.gbuf + os.off[ng]);
    uint64_t* keys = (uint64_t*)base;           // [npad]
    int32_t* pos = kWave ? (int32_t*)base : (int32_t*)(keys + npad);     // [npad]  (kWave: no key array, the keys live in registers)
    int32_t* gid = kSorted ? (int32_t*)base : pos + npad;   // [npad] PEG id of list position i (read back after the sort: one dependent
                                                            //        global gather less on the way to the records)
    if constexpr (kSorted) {
        // (nothing to do: the list came in processing order)
    } else if constexpr (NPAD > 0) {
        // ONE wave, the list in REGISTERS: element i = tid + 64 * q lives in slot q of lane tid.  A compare-exchange with distance
        // j < 64 fetches the partner from lane tid ^ j through the LDS crossbar (ds_bpermute: three dwords per element, no LDS
        // memory, no barrier), with j >= 64 it swaps two slots of the same lane.  As (key, position) arrays in LDS every stage was
        // four reads, a compare, four predicated writes and a barrier — ~70 instructions for two elements per lane, 28 stages for
        // 128 entries: 60 % of this kernel, which is bound by instruction issue (profiles/r05d_sched_phase_profile.txt, [order prof]).
        constexpr int QN = NPAD / 64;
        uint64_t ek[QN];
        int32_t ep[QN];
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int i = tid + 64 * q;
            if (i < Gn) {
                int g;
                if constexpr (kWave) g = gid[i];   // (the caller built the list in the wave's scratch)
                else { g = t.peg_idx[off + i]; gid[i] = g; }
                ek[q] = desc_key(peg_score(t, g, ng));
                ep[q] = i;
            } else { ek[q] = ~0ull; ep[q] = 0x7fffffff; }
        }
        CASIM_OPROF(0);   // scores
#pragma unroll
        for (int k = 2; k <= NPAD; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                if (j >= 64) {
                    const int dq = j >> 6;
#pragma unroll
                    for (int q = 0; q < QN; ++q) {
                        if (q & dq) continue;
                        const bool up = ((64 * q) & k) == 0;   // (k > j >= 64: a constant of the slot)
                        const bool gt = ek[q] > ek[q | dq] || (ek[q] == ek[q | dq] && ep[q] > ep[q | dq]);
                        if (gt == up) { const uint64_t tk = ek[q]; ek[q] = ek[q | dq]; ek[q | dq] = tk; const int32_t tp = ep[q]; ep[q] = ep[q | dq]; ep[q | dq] = tp; }
                    }
                } else {
                    const bool low = (tid & j) == 0;   // I am the lower index of my pair
#pragma unroll
                    for (int q = 0; q < QN; ++q) {
                        const bool up = ((tid + 64 * q) & k) == 0;
                        const uint64_t pk = cs::readlane_u64(ek[q], tid ^ j);
                        const int32_t pp = (int32_t)cs::shfl_u32((uint32_t)ep[q], tid ^ j);
                        const bool gt = ek[q] > pk || (ek[q] == pk && ep[q] > pp);   // mine sorts after the partner
                        // the lower index keeps the smaller of the two when the run ascends, the larger when it descends
                        if (gt == (low == up)) { ek[q] = pk; ep[q] = pp; }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) pos[tid + 64 * q] = ep[q];
        barrier();
    } else {
#pragma unroll
    for (int i = tid; i < npad; i += nt) {
        if (i < Gn) {
            const int g = t.peg_idx[off + i];
            gid[i] = g;
            keys[i] = desc_key(peg_score(t, g, ng));
            pos[i] = i;
        } else {
            keys[i] = ~0ull;
            pos[i] = 0x7fffffff;
        }
    }
    cs::sync();
    CASIM_OPROF(0);   // scores
    // one thread per PAIR (i, i | j): every lane of every wave works in every pass (with one thread per element
    // half of them only tested l > i, and the kernel is bound by instruction issue)
    auto exchange = [&](const int p, const int k, const int j) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int l = i | j;
        const uint64_t ki = keys[i], kl = keys[l];
        const int32_t pi = pos[i], pl = pos[l];
        const bool gt = ki > kl || (ki == kl && pi > pl);  // (i) sorts after (l)
        const bool up = (i & k) == 0;
        if (gt == up) { keys[i] = kl; keys[l] = ki; pos[i] = pl; pos[l] = pi; }
    };
        for (int k = 2; k <= npad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int p = tid; p < (npad >> 1); p += nt) exchange(p, k, j);
                cs::sync();
            }
        }
    }
    CASIM_OPROF(1);   // sort
    // fastpath: the eligible PEG with the largest simulationsSaved, last one on ties, goes last
    int best = -1;
    if (!kSorted && t.fastpath && Gn > 0) {
        int64_t* red = (int64_t*)(gid + npad);  // [nt] 8-byte aligned (16 bytes per entry behind an 8-byte aligned base)
        int64_t mine = -1;
#pragma unroll
        for (int i = tid; i < (NPAD > 0 ? NPAD : Gn); i += nt) {
            if (NPAD > 0 && i >= Gn) continue;
            const int32_t sv = fastpath_saved(t, gid[pos[i]], ng);
            if (sv >= 0) {
                const int64_t v = ((int64_t)sv << 32) | (uint32_t)i;
                mine = v > mine ? v : mine;
            }
        }
        red[tid] = mine;
        barrier();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) { const int64_t o = red[tid + s]; if (o > red[tid]) red[tid] = o; }
            barrier();
        }
        const int64_t top = red[0];
        if (top >= 0) best = (int)(uint32_t)(top & 0xffffffffll);
        barrier();
    }
    // (the tables and result arrays through a fresh view of the kernel arguments: see cs::kernarg_view — every kernel that calls order_group
    // declares (DevTables, DevResults, ...) first)
    const DevTables& te_ = cs::kernarg_view(t, 0);
    const DevResults& re_ = cs::kernarg_view(res, (sizeof(DevTables) + 7) & ~(size_t)7);
    // RULE for every kernel that calls order_group: its first two explicit arguments are (DevTables, DevResults), passed on UNMODIFIED —
    // the views above re-read them at byte offsets 0 and sizeof(DevTables) rounded up.  Nothing in the language enforces it, so
    // the A/B library of tests/ab is built with -DCASIM_CHECK_KERNARGS: every group then compares the views with the parameters it was
    // handed and traps on a difference (tests/test_gpu_ab_structurizer.py runs the whole corpus through that build on the MI355X).
#if defined(CASIM_CHECK_KERNARGS) && !defined(CASIM_HOST_EMU)
    {
        static_assert(sizeof(DevTables) % 4 == 0 && sizeof(DevResults) % 4 == 0, "compared word by word");
        bool same = true;
        for (size_t i = 0; i < sizeof(DevTables) / 4; ++i) same = same && ((const uint32_t*)&te_)[i] == ((const uint32_t*)&t)[i];
        for (size_t i = 0; i < sizeof(DevResults) / 4; ++i) same = same && ((const uint32_t*)&re_)[i] == ((const uint32_t*)&res)[i];
        if (!same) __builtin_trap();
    }
#endif
#pragma unroll
    for (int i = tid; i < (NPAD > 0 ? NPAD : Gn); i += nt) {
        if (NPAD > 0 && i >= Gn) continue;
        int src = i;
        if (best >= 0) {
            if (i == Gn - 1) src = best;
            else if (i >= best) src = i + 1;
        }
        // Emit the PEG record in PROCESSING order (structure of arrays): the packer then streams its
        // group's records with coalesced loads, 64 records per wave-load, and never chases indices.
        // The template-level Filters (taints, nodeSelector / affinity, unschedulable) are constant per
        // (PEG, group): evaluate them once here and hand them over as one flag bit.
        const int g = kSorted ? gid[src] : gid[pos[src]];
        re_.order[off + i] = g;
        // (lists derived by the feasibility kernel only hold PEGs that passed these Filters already)
        const uint32_t flags = (te_.pflags[g] & ~CASIM_KFLAG_STATIC_OK) | ((te_.lists_from_feas || static_filters_pass(te_, g, ng)) ? CASIM_KFLAG_STATIC_OK : 0u);
        if (re_.rec) {
            // register packer: one record (casim_types.h) = everything the packer needs to know about the PEG, computed HERE
            // by the record's own thread — the scaled requests, their reciprocals (the packer's quotient estimate) and how
            // many pods of the PEG fit an EMPTY node of this group (fitsRequest on the template, fit.go:681-765)
            auto emit = [&](auto rl_tag, auto xw_tag) {   // (RL as a constant: the record is built in registers, not in a scratch array)
                constexpr int RL = decltype(rl_tag)::value, DW = RL == 2 ? 8 : 16;
                uint32_t w[DW];
#pragma unroll
                for (int k = 0; k < DW; ++k) w[k] = 0;
                const int32_t slots = te_.allowed[ng] - te_.init_pods[ng];
                uint32_t cf = slots > 0 ? (uint32_t)slots : 0u;
                bool simple = true;
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const int32_t q = r < te_.R ? re_.req32[(int64_t)g * te_.R + r] : 0;
                    simple = simple && q > 0 && q < (1 << 30);
                    w[2 + r] = (uint32_t)q;
                    // (the reciprocal is only ever a quotient estimate with an exact +-1 fix-up behind it: no IEEE division — 4 of them per lane
                    // pass were ~60 of this kernel's ~800 vector instructions — and the empty node's capacity by the same estimate instead of
                    // the compiler's 32-bit division sequence)
                    const double rqd = q > 0 ? cs::estimate_rcp_f64((double)q) : 0.0;
                    const uint64_t rq = cs::double_bits(rqd);
                    w[2 + RL + 2 * r] = (uint32_t)rq; w[2 + RL + 2 * r + 1] = (uint32_t)(rq >> 32);
                    if (q > 0) {
                        const int32_t f = re_.fresh32[(int64_t)ng * te_.R + r];
                        uint32_t e = 0u;
                        if (f >= q) {
                            e = (uint32_t)((double)(uint32_t)f * rqd);
                            const int64_t rem = (int64_t)f - (int64_t)((uint64_t)e * (uint64_t)(uint32_t)q);   // (q may be >= 2^30: 64-bit remainder)
                            e = rem < 0 ? e - 1u : (rem >= (int64_t)q ? e + 1u : e);
                        }
                        cf = e < cf ? e : cf;
                    }
                }
                w[0] = (uint32_t)te_.count[g];
                if (!(flags & CASIM_KFLAG_STATIC_OK)) cf = 0u;   // the template-level Filters fail: no pod of the PEG fits an empty node of this group (the packer's a3 reads only this)
                // (RunFiltersUntilPassingNode skips Spec.Unschedulable nodes before any Filter runs, plugin_runner.go:108-110: in such a group the
                // simulated nodes are never worth a visit — decided here, the packer tests ONE constant bit)
                const bool a2_ok = (flags & CASIM_KFLAG_STATIC_OK) && te_.count[g] > 0 && !(te_.gflags[ng] & CASIM_NG_UNSCHEDULABLE);
                // (a record that carries exclusion words: CASIM_REC_A2_SIMPLE also says "and every one of them is zero" — the packer's dry loop then
                // decides such a PEG by the lean store's three compares, before any word logic: casim_pack.h, `idle`)
                uint64_t xb[2] = {0, 0}, xm[2] = {0, 0};
                if constexpr (decltype(xw_tag)::value) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) if (q < te_.Wx) { xb[q] = te_.xblock[(int64_t)g * te_.Wx + q]; xm[q] = te_.xmark[(int64_t)g * te_.Wx + q]; }
                }
                const bool wordless = (xb[0] | xb[1] | xm[0] | xm[1]) == 0ull;
                w[1] = (flags & (CASIM_REC_FLAG_MASK & ~(CASIM_REC_SIMPLE | CASIM_REC_A2_OK))) | (cf << CASIM_REC_FRESH_SHIFT) | (simple ? CASIM_REC_SIMPLE : 0u) |
                       (a2_ok ? CASIM_REC_A2_OK : 0u) | ((a2_ok && simple && wordless) ? CASIM_REC_A2_SIMPLE : 0u);
                constexpr int DWOUT = decltype(xw_tag)::value ? 16 : DW;
                RecQuad* out = (RecQuad*)(re_.rec + (int64_t)(off + i) * DWOUT);   // 16-byte stores (records are 32 / 64 bytes)
#pragma unroll
                for (int k = 0; k < DW / 4; ++k) out[k] = RecQuad{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
                if constexpr (decltype(xw_tag)::value) {   // DevResults::rec_xw: the PEG's node-local exclusion words ride with the record (casim_types.h)
                    out[2] = RecQuad{(uint32_t)xb[0], (uint32_t)(xb[0] >> 32), (uint32_t)xb[1], (uint32_t)(xb[1] >> 32)};
                    out[3] = RecQuad{(uint32_t)xm[0], (uint32_t)(xm[0] >> 32), (uint32_t)xm[1], (uint32_t)(xm[1] >> 32)};
                }
            };
            // int64 register store: the two requests as they came (no gcd scaling), 64-bit quotient for the empty node's capacity
            auto emit64 = [&]() {
                uint32_t w[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) w[k] = 0;
                const int32_t slots = te_.allowed[ng] - te_.init_pods[ng];
                uint32_t cf = slots > 0 ? (uint32_t)slots : 0u;
                bool simple = true;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int64_t q = r < te_.R ? te_.req[(int64_t)g * te_.R + r] : 0;
                    simple = simple && q > 0;
                    w[2 + 2 * r] = (uint32_t)(uint64_t)q; w[2 + 2 * r + 1] = (uint32_t)((uint64_t)q >> 32);
                    const uint64_t rq = cs::double_bits(q > 0 ? cs::estimate_rcp_f64((double)q) : 0.0);
                    w[6 + 2 * r] = (uint32_t)rq; w[6 + 2 * r + 1] = (uint32_t)(rq >> 32);
                    if (q > 0) {
                        const int64_t f = te_.alloc[(int64_t)ng * te_.R + r] - te_.init_req[(int64_t)ng * te_.R + r];
                        const uint64_t e = f >= q ? (uint64_t)f / (uint64_t)q : 0ull;
                        cf = e < (uint64_t)cf ? (uint32_t)e : cf;
                    }
                }
                w[0] = (uint32_t)te_.count[g];
                if (!(flags & CASIM_KFLAG_STATIC_OK)) cf = 0u;   // the template-level Filters fail: no pod of the PEG fits an empty node of this group (the packer's a3 reads only this)
                // (RunFiltersUntilPassingNode skips Spec.Unschedulable nodes before any Filter runs, plugin_runner.go:108-110: in such a group the
                // simulated nodes are never worth a visit — decided here, the packer tests ONE constant bit)
                const bool a2_ok = (flags & CASIM_KFLAG_STATIC_OK) && te_.count[g] > 0 && !(te_.gflags[ng] & CASIM_NG_UNSCHEDULABLE);
                w[1] = (flags & (CASIM_REC_FLAG_MASK & ~(CASIM_REC_SIMPLE | CASIM_REC_A2_OK))) | (cf << CASIM_REC_FRESH_SHIFT) | (simple ? CASIM_REC_SIMPLE : 0u) |
                       (a2_ok ? CASIM_REC_A2_OK : 0u) | ((a2_ok && simple) ? CASIM_REC_A2_SIMPLE : 0u);
                RecQuad* out = (RecQuad*)(re_.rec + (int64_t)(off + i) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = RecQuad{w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
            };
            if (re_.rec_i64) emit64(); else if (re_.rec_dw == 8) emit(IntTag<2>{}, IntTag<0>{}); else if (re_.rec_xw) emit(IntTag<2>{}, IntTag<1>{}); else emit(IntTag<4>{}, IntTag<0>{});
        }
        if (re_.s_count) {   // the generic packer's three arrays (also next to the records when it stands by for retries)
            re_.s_count[off + i] = te_.count[g];
            re_.s_flags[off + i] = flags;
            for (int r = 0; r < te_.R; ++r) re_.s_req[(int64_t)(off + i) * te_.R + r] = te_.req[(int64_t)g * te_.R + r];
        }
    }
    if (tid == 0) res.fast_last[ng] = best >= 0 ? 1 : 0;
    CASIM_OPROF(2);   // records
#if defined(CASIM_PACK_PROF) && !defined(CASIM_HOST_EMU)
    if (os.prof && tid == 0) for (int i = 0; i < 4; ++i) os.prof[(int64_t)ng * 4 + i] = (int64_t)oprof[i];
#endif
}
template <bool kLds>
CS_DEVICE void order_dispatch(const DevTables& t, const DevResults& res, const OrderScratch& os, const int ng, const int off, const int Gn) {
    if (kLds && cs::nthreads() == 64 && Gn <= 256) {   // one wave per group (batches of simulations): straight-line networks
        if (Gn <= 64) order_group<kLds, 64>(t, res, os, ng, off, Gn);
        else if (Gn <= 128) order_group<kLds, 128>(t, res, os, ng, off, Gn);
        else order_group<kLds, 256>(t, res, os, ng, off, Gn);
    } else if (kLds && os.lds_list_cap > 0 && Gn > os.lds_list_cap) {
        // a batch sizes its LDS for the lists the one-wave networks take (what the occupancy of this latency-bound kernel
        // hangs on); the few longer lists of the launch sort in the HBM scratch slab
        order_group<false, 0>(t, res, os, ng, off, Gn);
    } else order_group<kLds, 0>(t, res, os, ng, off, Gn);
}
template <bool kLds>
CS_GLOBAL void order_kernel(DevTables t, DevResults res, OrderScratch os) {
    const int ng = cs::bid();
    const int off = t.peg_off[ng];
    order_dispatch<kLds>(t, res, os, ng, off, t.peg_off[ng + 1] - off);
}

// ------------------------------------------------------------------------------------------
// K_front: feasibility row + list offsets + list + PEG order of ONE group per block, in ONE launch
// ------------------------------------------------------------------------------------------
// A single Estimate call (one simulation, tens of groups, hundreds of PEGs) is bound by the launches in its chain, not by work: feas,
// scan, fill and order were four dependent launches of a few microseconds each.  Here block ng does all four for its group.  The one
// thing a group needs from the others — the number of list entries in front of its own — travels through `ticket`: every block
// publishes (epoch << 32 | count) with a device-scope release store as soon as its row is done, then waits for the words of the blocks
// in front of it to carry this launch's epoch (the run counter of the problem: a second run on the same tables does not read the first's)
// — for a bounded number of polls; what has not arrived by then it counts itself (see step 3 below), so no block waits without bound.  Workgroups of a launch are dispatched in ascending order, so everything a resident block waits for is resident
// or finished (the assumption every single-pass chained scan makes); nobody waits for a block behind it.  `ticket` starts as zeros
// (the host keeps it inside the upload slab), epochs start at 1.  The batch geometry keeps its separate kernels: tens of thousands of groups
// would serialise on the tickets (a last-block scan inside feas_sim_kernel cost 1.05 -> 2.77 ms per step, profiles/r02s_packer_notes.txt).
// LDS (dynamic, shared with order_group which takes it over afterwards): [Wg] ballot words, then [2 waves + 3] counters.
template <bool kLds>
CS_GLOBAL void front_kernel(DevTables t, DevResults res, OrderScratch os, uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg,
                            int32_t* offsets /*[NG + 1] == t.peg_off*/, int32_t* idx /*[nnz bound] == t.peg_idx*/,
                            uint64_t* ticket /*[NG]*/, uint32_t epoch, int NG, uint32_t spin_limit) {
    const int ng = cs::bid(), tid = cs::tid(), lane = cs::lane(), wave = tid >> 6, nw = (cs::nthreads() + 63) >> 6;
    const int lo = t.peg_lo[ng], hi = t.peg_hi[ng];
    uint64_t* words = (uint64_t*)cs::dyn_smem();
    uint32_t* cnt = (uint32_t*)(words + Wg);   // [0, nw) per-wave counts, [nw] the entries in front of this group, [nw + 1, nw + 3) tickets missing, [nw + 3, 2 nw + 3) recount
    // 1. the row (feas_kernel): wave w takes words w, w + nw, ..
    uint32_t mine = 0;
    for (int w = wave; w < Wg; w += nw) {
        const int k = w * 64 + lane;
        bool ok = false;
        if (lo + k < hi) ok = fits_fresh_node(t, lo + k, ng);
        const uint64_t b = cs::ballot(ok);
        if (lane == 0) { bits[(int64_t)ng * Wg + w] = b; words[w] = b; }
        mine += (uint32_t)cs::popc64(b);
    }
    if (lane == 0) cnt[wave] = mine;
    cs::sync();
    uint32_t total = 0;
    for (int w = 0; w < nw; ++w) total += cnt[w];
    // 2. publish the count, 3. collect the counts in front: wave 0 polls a lane per predecessor, at most `spin_limit` times each.  A
    // ticket that has not arrived by then is NOT waited for any longer: the whole block counts that group's row itself (the same
    // predicate, the same count its owner will publish).  So no block ever holds its slot waiting for another one without bound —
    // several front kernels of different streams sharing a saturated chip cannot starve each other's lowest blocks into a circular
    // wait — and in the usual case (every block resident within microseconds) nothing is counted twice.
    if (wave == 0 && lane == 0) { cs::publish_u64(ticket + ng, ((uint64_t)epoch << 32) | total); cnt[nw] = 0; }
    uint32_t before = 0;   // (wave 0: per-lane partial sums of the tickets that arrived)
    for (int j0 = 0; j0 < ng; j0 += 64) {
        if (wave == 0) {
            const int j = j0 + lane;
            bool have = j >= ng;
            if (!have) {
                uint64_t v = 0;
                for (uint32_t spin = 0; spin < spin_limit; ++spin) {
                    v = cs::poll_u64(ticket + j);
                    if ((uint32_t)(v >> 32) == epoch) { have = true; break; }
                }
                if (have) before += (uint32_t)v;
            }
            const uint64_t missing = cs::ballot(!have);
            if (lane == 0) { cnt[nw + 1] = (uint32_t)missing; cnt[nw + 2] = (uint32_t)(missing >> 32); }
        }
        cs::sync();
        uint64_t missing = (uint64_t)cnt[nw + 1] | ((uint64_t)cnt[nw + 2] << 32);   // (the same for every thread of the block)
        while (missing) {
            const int jm = j0 + cs::ffs64(missing);
            missing &= missing - 1;
            const int lo_j = t.peg_lo[jm], hi_j = t.peg_hi[jm];
            uint32_t c = 0;
            for (int w = wave; w * 64 < hi_j - lo_j; w += nw) {
                const int k = w * 64 + lane;
                bool ok = false;
                if (lo_j + k < hi_j) ok = fits_fresh_node(t, lo_j + k, jm);
                c += (uint32_t)cs::popc64(cs::ballot(ok));
            }
            if (lane == 0) cnt[nw + 3 + wave] = c;
            cs::sync();
            if (tid == 0) { uint32_t sum = 0; for (int w = 0; w < nw; ++w) sum += cnt[nw + 3 + w]; cnt[nw] += sum; }
            cs::sync();
        }
        cs::sync();   // (cnt[nw + 1 ..] is rewritten by the next chunk of predecessors)
    }
    if (wave == 0) {
        before = cs::wave_sum_u32(before);
        if (lane == 0) {
            before += cnt[nw];
            cnt[nw] = before;
            offsets[ng] = (int32_t)before;
            if (ng == NG - 1) offsets[NG] = (int32_t)(before + total);
        }
    }
    cs::sync();
    const int32_t base = (int32_t)cnt[nw];
    // 4. the list (csr_fill_kernel): PEG ids ascending — a lane per bit, its place = set bits in the words in front + in the lanes below
    for (int w = wave; w < Wg; w += nw) {
        uint32_t front = 0;
        for (int j = lane; j < w; j += 64) front += (uint32_t)cs::popc64(words[j]);
        front = cs::wave_sum_u32(front);
        const uint64_t b = words[w];
        if ((b >> lane) & 1ull) idx[base + (int32_t)front + cs::mbcnt(b)] = lo + w * 64 + lane;
    }
    cs::sync();   // the list is read back by other waves of the block (and the LDS changes hands)
    // 5. the order (order_kernel)
    order_dispatch<kLds>(t, res, os, ng, base, (int)total);
}

// ------------------------------------------------------------------------------------------
// K_front_sim: feasibility rows + lists + PEG order of every group of ONE SIMULATION per block, fixed-stride lists
// ------------------------------------------------------------------------------------------
// The batch geometry of round 3 ran four launches per sub-batch in front of the packer — feas_sim_kernel, csr_scan_local_kernel,
// csr_fill_kernel, order_kernel: 682 scalar + 1337 vector instructions per group, none of them sequential — because
// a group's list position depended on the counts of every group in front of it.  With FIXED-STRIDE lists it does not: group ng owns the
// region [peg_off[ng], peg_off[ng] + (peg_hi - peg_lo)) of order / placed / records (peg_off is static: the host's prefix sum of the
// candidate range lengths, the bound every one of those arrays was sized for anyway) and its length travels apart (peg_cnt).  So ONE
// block per simulation does everything: (A) every thread keeps one PEG in registers and walks the simulation's group records staged in
// LDS (feas_sim_kernel's loop), the ballot words stay in LDS; (B) the waves take the groups in turn — list from the words (a lane per
// bit), the register sorting network of order_group, the records — with no block barrier: each wave works in its own LDS scratch;
// (C) the rare list beyond the one-wave networks (> 256 PEGs) is ordered by the whole block afterwards.  No bit matrix round trip, no
// scan, no fill, no dependent launches.  Compaction for callers that fetch every list happens at fetch time (compact_lists_kernel).
// LDS: [hdr: cnt per group][grec 16 x u64 per group][words Wg per group][per wave: pos | gid | red], phase C reuses everything behind hdr.
CS_HOST_DEVICE size_t front_sim_wave_scratch() { return (size_t)(256 * 8 + 64 * 8); }
template <bool kLds>
CS_GLOBAL void front_sim_kernel(DevTables t, DevResults res, OrderScratch os, uint64_t* CS_RESTRICT bits /*[NG][Wg] or null*/, int Wg,
                                const int32_t* CS_RESTRICT req32, const int32_t* CS_RESTRICT fresh32, int32_t* CS_RESTRICT cnt_out /*[NG] == t.peg_cnt*/,
                                int32_t* CS_RESTRICT idx /*[nnz bound], lists beyond 256 entries only*/, int max_groups) {
    const int sim = cs::bid();
    const int g0 = t.sim_off[sim], g1 = t.sim_off[sim + 1];
    const int ngroups = g1 - g0;
    if (ngroups <= 0) return;
    const int lo = t.peg_lo[g0], hi = t.peg_hi[g0];
    const int tid = cs::tid(), lane = cs::lane(), wave = tid >> 6, nw = (cs::nthreads() + 63) >> 6;
    const int k = tid;
    const bool live = lo + k < hi;
    const int g = live ? lo + k : (hi > lo ? lo : 0);
    const bool narrow = req32 != nullptr && t.R <= 4;
    char* smem = cs::dyn_smem();
    int32_t* cnt_lds = (int32_t*)smem;                                         // [max_groups]
    uint64_t* grec = (uint64_t*)(smem + (((size_t)max_groups * 4 + 15) & ~(size_t)15));   // [ngroups][16]
    uint64_t* words = grec + (size_t)max_groups * 16;                          // [ngroups][Wg]
    char* scratch0 = (char*)(words + (size_t)max_groups * Wg);
    // ---- A. stage the group records (feas_sim_kernel's layout), then the rows
    for (int i = tid; i < ngroups * 16; i += cs::nthreads()) {
        const int ng = g0 + (i >> 4), f = i & 15;
        uint64_t v = 0;
        if (f == 0) v = t.Wt ? t.taint[(int64_t)ng * t.Wt] : 0ull;
        else if (f == 1) v = t.Wl ? t.label[(int64_t)ng * t.Wl] : ~0ull;
        else if (f == 2) v = t.Wx ? (t.init_excl[(int64_t)ng * t.Wx] ^ (t.xpol ? t.xpol[0] : 0ull)) : 0ull;   // (NEED bits inverted once, like the group-wide word)
        else if (f == 3) v = t.Wz ? (t.init_zone[(int64_t)ng * t.Wz] ^ t.zpol[0]) : 0ull;
        else if (f == 4) v = (uint64_t)t.gflags[ng] | ((uint64_t)(uint32_t)(t.allowed[ng] - t.init_pods[ng]) << 32);
        else if (f == 5 || f == 6) {
            if (narrow) {
                const int r = (f - 5) * 2;
                const uint32_t a = r < t.R ? (uint32_t)fresh32[(int64_t)ng * t.R + r] : 0u, b2 = r + 1 < t.R ? (uint32_t)fresh32[(int64_t)ng * t.R + r + 1] : 0u;
                v = (uint64_t)a | ((uint64_t)b2 << 32);
            }
        } else if (f >= 8) {
            const int r = f - 8;
            if (!narrow && r < t.R) v = (uint64_t)(t.alloc[(int64_t)ng * t.R + r] - t.init_req[(int64_t)ng * t.R + r]);
        }
        grec[i] = v;
    }
    int64_t req[CASIM_KMAX_RES];
    int32_t rq32[4] = {0, 0, 0, 0};
    for (int r = 0; r < CASIM_KMAX_RES; ++r) {
        if (narrow) { if (r < 4) rq32[r] = (live && r < t.R) ? req32[(int64_t)g * t.R + r] : 0; req[r] = 0; }
        else req[r] = (live && r < t.R) ? t.req[(int64_t)g * t.R + r] : 0;
    }
    const uint32_t pf = live ? t.pflags[g] : 0u;
    const uint64_t tol = (live && t.Wt) ? t.tol[(int64_t)g * t.Wt] : 0ull, sel = (live && t.Wl) ? t.sel[(int64_t)g * t.Wl] : 0ull;
    // (a NEED bit the PEG marks itself does not count on a fresh node: fits_fresh_node)
    const uint64_t xb = (live && t.Wx) ? (t.xblock[(int64_t)g * t.Wx] & ~(t.xmark[(int64_t)g * t.Wx] & (t.xpol ? t.xpol[0] : 0ull))) : 0ull, zb = (live && t.Wz) ? t.zblock[(int64_t)g * t.Wz] : 0ull;
    cs::sync();
    const bool tolerates_unsched = (pf & CASIM_PEG_TOLERATES_UNSCHEDULABLE) != 0;
    for (int gl = 0; gl < ngroups; ++gl) {
        const uint64_t* gr = grec + (int64_t)gl * 16;
        const uint64_t fl = gr[4];
        bool ok = live & ((gr[0] & ~tol) == 0) & ((sel & ~gr[1]) == 0) & ((xb & gr[2]) == 0) & ((zb & gr[3]) == 0);
        ok = ok & (tolerates_unsched | (((uint32_t)fl & CASIM_NG_UNSCHEDULABLE) == 0));
        ok = ok & ((int32_t)(fl >> 32) > 0);
        if (narrow) {
            const uint64_t f01 = gr[5], f23 = gr[6];
            bool fit = (rq32[0] <= 0) | (rq32[0] <= (int32_t)(uint32_t)f01);
            fit = fit & ((rq32[1] <= 0) | (rq32[1] <= (int32_t)(f01 >> 32)));
            fit = fit & ((rq32[2] <= 0) | (rq32[2] <= (int32_t)(uint32_t)f23));
            fit = fit & ((rq32[3] <= 0) | (rq32[3] <= (int32_t)(f23 >> 32)));
            ok = ok & fit;
        } else {
            bool fit = true;
#pragma unroll
            for (int r = 0; r < CASIM_KMAX_RES; ++r) fit = fit & ((req[r] <= 0) | (req[r] <= (int64_t)gr[8 + r]));
            ok = ok & fit;
        }
        const uint64_t b = cs::ballot(ok);
        if (lane == 0 && wave < Wg) { words[(int64_t)gl * Wg + wave] = b; if (bits) bits[(int64_t)(g0 + gl) * Wg + wave] = b; }
    }
    cs::sync();
    // ---- B. the waves take the groups in turn: list, order, records — no block barrier in here
    char* my = scratch0 + (size_t)wave * front_sim_wave_scratch();
    int32_t* my_gid = (int32_t*)my + 256;   // (order_group<.., NPAD, true>: pos[NPAD] | gid[NPAD]; NPAD <= 256 -> gid of the 256-layout is the furthest)
    for (int gl = wave; gl < ngroups; gl += nw) {
        const int ng = g0 + gl;
        const int base = t.peg_off[ng];
        int total = 0;
        for (int w = 0; w < Wg; ++w) total += cs::popc64(words[(int64_t)gl * Wg + w]);
        if (lane == 0) { cnt_lds[gl] = total; cnt_out[ng] = total; }
        if (total > 256) {   // (ordered by the whole block in phase C: its list goes to global memory)
            int run = 0;
            for (int w = 0; w < Wg; ++w) {
                const uint64_t b = words[(int64_t)gl * Wg + w];
                if ((b >> lane) & 1ull) idx[base + run + cs::mbcnt(b)] = lo + w * 64 + lane;
                run += cs::popc64(b);
            }
            continue;
        }
        // the layout order_group expects depends on its NPAD: gid sits right behind pos[NPAD]
        const int npad = total <= 64 ? 64 : (total <= 128 ? 128 : 256);
        int32_t* gid = (int32_t*)my + npad;
        int run = 0;
        for (int w = 0; w < Wg; ++w) {
            const uint64_t b = words[(int64_t)gl * Wg + w];
            if ((b >> lane) & 1ull) gid[run + cs::mbcnt(b)] = lo + w * 64 + lane;
            run += cs::popc64(b);
        }
        cs::wave_sync();
        if (npad == 64) order_group<true, 64, true>(t, res, os, ng, base, total, my);
        else if (npad == 128) order_group<true, 128, true>(t, res, os, ng, base, total, my);
        else order_group<true, 256, true>(t, res, os, ng, base, total, my);
        cs::wave_sync();   // (the scratch changes hands: the wave's next group)
    }
    (void)my_gid;
    cs::sync();
    // ---- C. lists beyond the one-wave networks: the whole block, one after the other (block-uniform loop: cnt_lds is the same for everybody)
    for (int gl = 0; gl < ngroups; ++gl) {
        const int total = cnt_lds[gl];
        if (total <= 256) continue;
        order_group<kLds, 0>(t, res, os, g0 + gl, t.peg_off[g0 + gl], total, kLds ? (char*)grec : nullptr);
        cs::sync();
    }
}

// The same lists with the parallelism of order_kernel (one wave per GROUP, NG blocks): feas_sim_kernel writes the bit matrix as before, this
// kernel reads its group's row (a few words), builds the list in LDS and orders it — the scan and fill launches are gone, the wave count of
// the ordering step is not (front_sim_kernel's 7-wave blocks, three groups per wave one after the other, overlapped WORSE with the packers
// of the other streams: 1.13 ms per step against 1.03, profiles/r08f_*).
template <bool kLds>
CS_GLOBAL void order_strided_kernel(DevTables t, DevResults res, OrderScratch os, const uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg,
                                    int32_t* CS_RESTRICT cnt_out /*[NG] == t.peg_cnt*/, int32_t* CS_RESTRICT idx) {
    const int ng = cs::bid(), lane = cs::lane();
    const int base = t.peg_off[ng], lo = t.peg_lo[ng];
    const uint64_t* row = bits + (int64_t)ng * Wg;
    int total = 0;
    for (int w = 0; w < Wg; ++w) total += cs::popc64(row[w]);
    if (cs::tid() == 0) cnt_out[ng] = total;
    char* smem = cs::dyn_smem();
    if (cs::nthreads() == 64 && total <= 256) {
        const int npad = total <= 64 ? 64 : (total <= 128 ? 128 : 256);
        int32_t* gid = (int32_t*)smem + npad;
        int run = 0;
        for (int w = 0; w < Wg; ++w) {
            const uint64_t b = row[w];
            if ((b >> lane) & 1ull) gid[run + cs::mbcnt(b)] = lo + w * 64 + lane;
            run += cs::popc64(b);
        }
        cs::wave_sync();
        if (npad == 64) order_group<true, 64, true>(t, res, os, ng, base, total, smem);
        else if (npad == 128) order_group<true, 128, true>(t, res, os, ng, base, total, smem);
        else order_group<true, 256, true>(t, res, os, ng, base, total, smem);
        return;
    }
    // a long list (or a block of several waves): through global memory and the general network
    for (int w = cs::tid() >> 6; w < Wg; w += (cs::nthreads() + 63) >> 6) {
        int run = 0;
        for (int j = 0; j < w; ++j) run += cs::popc64(row[j]);
        const uint64_t b = row[w];
        if ((b >> lane) & 1ull) idx[base + run + cs::mbcnt(b)] = lo + w * 64 + lane;
    }
    cs::sync();
    if (kLds && os.lds_list_cap > 0 && total > os.lds_list_cap) order_group<false, 0>(t, res, os, ng, base, total);
    else order_group<kLds, 0>(t, res, os, ng, base, total);
}

// ---- rank once per (simulation, allocatable pair) (VERDICT r4 next #5b) ----------------------------------------------------------------------
// The orderer's score depends on the PEG and on the template's (cpu, memory) allocatable only (decreasing_pod_orderer.go:76-81): the 64 node
// groups of a C3 simulation are 5 such pairs.  order_strided_kernel sorts every group's list (~250 of the simulation's 1000 PEGs: a 256-key
// register network per group, the general LDS network beyond that) — 59 % of the batched C3 step's kernel time (profiles/r11l).  Instead:
//   rank_shapes_kernel   one block per (simulation, pair): ALL PEGs of the simulation by (score descending, id ascending) — the canonical tie
//                        rule: lists are built in ascending id order — -> ranks[pair][k]
//   order_ranked_kernel  one wave per group: walks its pair's ranks 64 at a time, keeps the PEGs whose bit is set in the group's feasibility
//                        row (ballot + mbcnt: the order survives) and emits the records (order_group<.., kSorted>)
// A subsequence of a sorted sequence is sorted: the lists are exactly those of the per-group sort.  The host turns it on when the candidate
// ranges are long and a pair serves several groups (csrc/casim_pipeline.h: rank_once_; CASIM_RANK_ONCE=0 / 1 forces it off / on); for the
// headline's 20 groups x ~110 PEGs it loses (five 512-key sorts against twenty 128-key networks).
// `list` names the pairs this launch ranks.  pair_base == null: every listed pair sorts.  Else (second launch, the pairs whose allocatable is
// PROPORTIONAL to their simulation's base pair: k x (cpu, memory) — instance families): the scores are the base pair's times a constant, so the
// base ranking almost always is this pair's too; the block CHECKS that (adjacent entries of the base ranking in (score descending, id
// ascending) order under ITS scores: one pass) and points pair_src at the base ranking, or sorts when the check fails (rounding made a tie).
CS_GLOBAL void rank_shapes_kernel(DevTables t, const int32_t* CS_RESTRICT list, const int32_t* CS_RESTRICT pair_rep /*[n_pairs] one group of the pair*/,
                                  const int32_t* CS_RESTRICT pair_base /*[n_pairs] or null*/, int32_t* CS_RESTRICT pair_src /*[n_pairs] whose ranking the pair's groups read*/,
                                  int32_t* CS_RESTRICT ranks /*[n_pairs][stride]*/, int stride) {
    const int pair = list[cs::bid()], tid = cs::tid(), nt = cs::nthreads();
    const int ng = pair_rep[pair];
    const int lo = t.peg_lo[ng], n = t.peg_hi[ng] - lo;
    int npad = 64;
    while (npad < n) npad <<= 1;
    uint64_t* keys = (uint64_t*)cs::dyn_smem();   // [npad]
    int32_t* pos = (int32_t*)(keys + npad);       // [npad]
    if (pair_base) {
        const int32_t* rb = ranks + (int64_t)pair_base[pair] * stride;
        bool bad = false;
        for (int i = tid; i + 1 < n; i += nt) {
            const int a = rb[i], b = rb[i + 1];
            const uint64_t ka = desc_key(peg_score(t, a, ng)), kb = desc_key(peg_score(t, b, ng));
            bad = bad || ka > kb || (ka == kb && a > b);
        }
        uint32_t* flag = (uint32_t*)keys;   // [waves of the block]
        const uint64_t bb = cs::ballot(bad);
        if (cs::lane() == 0) flag[tid >> 6] = bb != 0ull ? 1u : 0u;
        cs::sync();
        bool any = false;
        for (int w = 0; w < (nt + 63) / 64; ++w) any = any || flag[w] != 0u;
        cs::sync();   // (the words are the sort's key array in a moment)
        if (!any) { if (tid == 0) pair_src[pair] = pair_base[pair]; return; }
    }
    if (tid == 0) pair_src[pair] = pair;
    for (int i = tid; i < npad; i += nt) {
        if (i < n) { keys[i] = desc_key(peg_score(t, lo + i, ng)); pos[i] = i; }
        else { keys[i] = ~0ull; pos[i] = 0x7fffffff; }
    }
    cs::sync();
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (npad >> 1); p += nt) {   // one thread per pair (i, i | j), as in order_group
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int l = i | j;
                const uint64_t ki = keys[i], kl = keys[l];
                const int32_t pi = pos[i], pl = pos[l];
                const bool gt = ki > kl || (ki == kl && pi > pl);
                const bool up = (i & k) == 0;
                if (gt == up) { keys[i] = kl; keys[l] = ki; pos[i] = pl; pos[l] = pi; }
            }
            cs::sync();
        }
    }
    for (int i = tid; i < n; i += nt) ranks[(int64_t)pair * stride + i] = lo + pos[i];
}
// LDS: [Wg rounded up to even] row words, then gid[stride]
CS_GLOBAL void order_ranked_kernel(DevTables t, DevResults res, OrderScratch os, const uint64_t* CS_RESTRICT bits /*[NG][Wg]*/, int Wg, int32_t* CS_RESTRICT cnt_out /*[NG] == t.peg_cnt*/,
                                   const int32_t* CS_RESTRICT pair_of_group /*[NG]*/, const int32_t* CS_RESTRICT pair_src /*[n_pairs]*/, const int32_t* CS_RESTRICT ranks, int stride) {
    const int ng = cs::bid(), lane = cs::lane();
    const int base = t.peg_off[ng], lo = t.peg_lo[ng], n = t.peg_hi[ng] - lo;
    char* smem = cs::dyn_smem();
    uint64_t* rowl = (uint64_t*)smem;
    int32_t* gid = (int32_t*)(rowl + ((Wg + 1) & ~1));
    for (int w = lane; w < Wg; w += 64) rowl[w] = bits[(int64_t)ng * Wg + w];
    cs::wave_sync();
    const int32_t* rk = ranks + (int64_t)pair_src[pair_of_group[ng]] * stride;
    int run = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int k = c0 + lane;
        const int p = k < n ? rk[k] : lo;
        const int rel = p - lo;
        const bool keep = k < n && ((rowl[rel >> 6] >> (rel & 63)) & 1ull);
        const uint64_t b = cs::ballot(keep);
        if (keep) gid[run + cs::mbcnt(b)] = p;
        run += cs::popc64(b);
    }
    if (lane == 0) cnt_out[ng] = run;
    cs::wave_sync();
    order_group<true, 0, true, true>(t, res, os, ng, base, run, (char*)gid);
}

// K_compact: fixed-stride lists -> the compact CSR a caller fetches (casim_problem_fetch of every list; never in the resident loop)
CS_GLOBAL void count_offsets_kernel(const int32_t* CS_RESTRICT cnt /*[NG]*/, int NG, int32_t* CS_RESTRICT coff /*[NG + 1]*/) {
    uint32_t* sm = (uint32_t*)cs::dyn_smem();
    uint32_t carry = 0;
    for (int s0 = 0; s0 < NG; s0 += cs::nthreads()) {
        const int s = s0 + cs::tid();
        const uint32_t mine = s < NG ? (uint32_t)cnt[s] : 0u;
        uint32_t total = 0;
        const uint32_t excl = block_exclusive_scan(mine, sm, &total);
        if (s < NG) coff[s] = (int32_t)(carry + excl);
        carry += total;
        cs::sync();
    }
    if (cs::tid() == 0) coff[NG] = (int32_t)carry;
}
// id_add: a part of a cut batch numbers its PEGs from 0; the caller sees the whole batch's numbering (casim_streams.h) — added here, where the
// ids pass through registers anyway, instead of by a host loop over millions of fetched entries
CS_GLOBAL void compact_lists_kernel(const int32_t* CS_RESTRICT off, const int32_t* CS_RESTRICT cnt, const int32_t* CS_RESTRICT coff,
                                    const int32_t* CS_RESTRICT order, const int32_t* CS_RESTRICT placed, int32_t* CS_RESTRICT corder,
                                    int32_t* CS_RESTRICT cplaced, int32_t id_add) {
    const int ng = cs::bid();
    const int a = off[ng], n = cnt[ng], c = coff[ng];
    for (int i = cs::tid(); i < n; i += cs::nthreads()) { corder[c + i] = order[a + i] + id_add; cplaced[c + i] = placed[a + i]; }
}

// ------------------------------------------------------------------------------------------
// K_pack: one wavefront = one Estimate()  -> casim_pack.h (included at the end of this file)
// ------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------
// K_option: expander filter chain over the groups of one launch
// ------------------------------------------------------------------------------------------
// Each filter keeps the options whose metric equals the best one among the current survivors
// (leastnodes.go:35-61, mostpods.go:33-53, waste.go:37-73), then chainStrategy stops as soon as
// one survivor is left (chain.go:36-45).  Single block; NG is small.
struct OptionArgs {
    const int32_t* node_count; const int32_t* pods; const int32_t* status;
    const int64_t* cpu_sum; const int64_t* mem_sum;
    const int64_t* waste_cpu; const int64_t* waste_mem;
    int32_t NG;
    int32_t kinds[8]; int32_t n_kinds;
    int32_t group_id_base;
    const int32_t* global_id;  // [NG] or null = group_id_base + index
    const int32_t* sim_off;    // [n blocks + 1] groups of each simulation (one block per simulation), or null = one block over [0, NG)
    uint8_t* valid;            // [NG] or null: 1 = the option may compete (all-or-nothing filter, orchestrator.go:1057-1063)
    uint8_t* best_set;      // [NG] out
    int32_t* out;           // [n blocks][2]: best index (within the launch), survivors
    int64_t* key_out;       // [n blocks][10]: key block of the winner (see option_kernel)
    int64_t* packed_out;    // [n blocks] or null: key_out[.][0] gathered, the operand of ONE all-reduce(min) over every simulation
};

CS_DEVICE uint64_t option_metric(const OptionArgs& a, int kind, int i) {
    // smaller is better, as an order-preserving uint64
    if (kind == CASIM_EXPANDER_LEAST_NODES) return (uint64_t)(uint32_t)a.node_count[i];
    if (kind == CASIM_EXPANDER_MOST_PODS) return (uint64_t)(0x7fffffff - a.pods[i]);
    // least-waste: (availCPU-reqCPU)/availCPU + (availMem-reqMem)/availMem, avail = capacity*count (waste.go:48-54)
    const int64_t avc = a.waste_cpu ? a.waste_cpu[i] * (int64_t)a.node_count[i] : 0;
    const int64_t avm = a.waste_mem ? a.waste_mem[i] * (int64_t)a.node_count[i] : 0;
    const double wc = (double)(avc - a.cpu_sum[i]) / (double)avc;
    const double wm = (double)(avm - a.mem_sum[i]) / (double)avm;
    const double sc = wc + wm;
    uint64_t u = cs::double_bits(sc);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

CS_GLOBAL void option_kernel(OptionArgs a) {
    const int tid = cs::tid(), nt = cs::nthreads();
    uint64_t* red = (uint64_t*)cs::dyn_smem();  // [nt]
    // one block = one simulation: its groups are [g0, g1)
    const int sim = cs::bid();
    const int g0 = a.sim_off ? a.sim_off[sim] : 0, g1 = a.sim_off ? a.sim_off[sim + 1] : a.NG;
    a.out += 2 * sim; a.key_out += 10 * sim;
    // valid options: something was scheduled on at least one node (orchestrator.go:1057-1063)
    for (int i = g0 + tid; i < g1; i += nt)
        a.best_set[i] = (a.status[i] == CASIM_NG_OK && a.node_count[i] > 0 && a.pods[i] > 0 && (!a.valid || a.valid[i])) ? 1 : 0;
    cs::sync();
#ifndef CASIM_HOST_EMU
#pragma unroll
#endif
    for (int f = 0; f < 8; ++f) {   // constant bound + unroll: kinds[f] is a kernel-argument scalar, not a scratch copy of the array
        if (f >= a.n_kinds) break;
        uint64_t mine = ~0ull;
        for (int i = g0 + tid; i < g1; i += nt)
            if (a.best_set[i]) { const uint64_t m = option_metric(a, a.kinds[f], i); mine = m < mine ? m : mine; }
        red[tid] = mine;
        cs::sync();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s && red[tid + s] < red[tid]) red[tid] = red[tid + s];
            cs::sync();
        }
        const uint64_t best = red[0];
        cs::sync();
        for (int i = g0 + tid; i < g1; i += nt)
            if (a.best_set[i] && option_metric(a, a.kinds[f], i) != best) a.best_set[i] = 0;
        cs::sync();
        // survivors
        uint32_t sv = 0;
        for (int i = g0 + tid; i < g1; i += nt) sv += a.best_set[i];
        red[tid] = sv;
        cs::sync();
        for (int s = nt >> 1; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; cs::sync(); }
        const uint64_t survivors = red[0];
        cs::sync();
        if (survivors <= 1) break;
    }
    // lowest surviving index + count
    uint64_t first = ~0ull; uint32_t sv = 0;
    for (int i = g0 + tid; i < g1; i += nt) if (a.best_set[i]) { if ((uint64_t)i < first) first = (uint64_t)i; sv++; }
    red[tid] = first;
    cs::sync();
    for (int s = nt >> 1; s > 0; s >>= 1) { if (tid < s && red[tid + s] < red[tid]) red[tid] = red[tid + s]; cs::sync(); }
    const uint64_t bi = red[0];
    cs::sync();
    red[tid] = sv;
    cs::sync();
    for (int s = nt >> 1; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; cs::sync(); }
    if (tid == 0) {
        a.out[0] = bi == ~0ull ? -1 : (int32_t)bi;
        a.out[1] = (int32_t)red[0];
        // key block (10 x int64, smaller = better) for the cross-GPU reduce:
        //   [0]      packed (first filter's metric << 20 | global group id): one all-reduce(min) is exact
        //            for the integer metrics (least-nodes / most-pods, < 2^43)
        //   [1..8]   the winner's metric under each filter of the chain, order-preserving int64
        //            (the chain == lexicographic min over (m_1, .., m_k, id))
        //   [9]      global group id of the winner
        const int64_t none = 0x7fffffffffffffffll;
        for (int i = 0; i < 10; ++i) a.key_out[i] = none;
        if (bi != ~0ull) {
#ifndef CASIM_HOST_EMU
#pragma unroll
#endif
            for (int f = 0; f < 8; ++f)
                if (f < a.n_kinds) a.key_out[1 + f] = (int64_t)(option_metric(a, a.kinds[f], (int)bi) ^ 0x8000000000000000ull);
            const int64_t gid = a.global_id ? (int64_t)a.global_id[bi] : (int64_t)a.group_id_base + (int64_t)bi;
            a.key_out[9] = gid;
            if (a.n_kinds > 0) {
                uint64_t m = option_metric(a, a.kinds[0], (int)bi);
                if (m > 0x7ffffffffffull) m = 0x7ffffffffffull;
                a.key_out[0] = (int64_t)((m << 20) | (uint64_t)(gid & 0xfffff));
            } else a.key_out[0] = gid;
        }
        if (a.packed_out) a.packed_out[sim] = a.key_out[0];
    }
}

// ------------------------------------------------------------------------------------------
// K_gcd / K_scale: the exact int32 image of the request table, on the device
// ------------------------------------------------------------------------------------------
// The register packer computes on int32 lanes: every resource lane divided by the gcd of all its values (exact).  For a batch of a
// million PEGs the host pass over the request table — one modulo per value while the gcd settles, then the quotients — was the longest
// stage of an enter -> return call (2.1 of ~4.5 ms per part, profiles/r05p_init_stages.txt) and its result, 8 more bytes per PEG, travelled
// over the link as well.  The int64 table is on the device anyway: gcd_reduce_kernel folds it to one partial per block (gcd, largest
// magnitude, "some value is negative" per lane; the host folds the few hundred partials together with the group table and decides),
// scale_requests_kernel writes the quotients.  Same arithmetic as the host pass (casim_pipeline.h), same results by test.
struct GcdPartial { uint64_t sc[4]; uint64_t amax[4]; uint64_t neg; };
CS_DEVICE uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { const uint64_t x = a % b; a = b; b = x; } return a; }
CS_DEVICE uint64_t gcd_fold(uint64_t sc, uint64_t a /* |value| */) {
    if (sc == 1 || a == 0) return sc;
    if (sc == 0) return a;
    uint64_t m;
    if ((sc & (sc - 1)) == 0) m = a & (sc - 1);                                                // byte-granular lanes settle on a power of two
    else if (a <= 0xffffffffull && sc <= 0xffffffffull) m = (uint64_t)((uint32_t)a % (uint32_t)sc);   // milli-cpu lanes: the short division
    else m = a % sc;
    return m ? gcd_u64(sc, m) : sc;
}
CS_GLOBAL void gcd_reduce_kernel(const int64_t* CS_RESTRICT req /*[G][R]*/, int64_t G, int R, GcdPartial* CS_RESTRICT out /*[nblocks]*/) {
    uint64_t sc[4] = {0, 0, 0, 0}, amax[4] = {0, 0, 0, 0};
    bool neg = false;
    const int64_t stride = (int64_t)cs::nblocks() * cs::nthreads();
    for (int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid(); i < G; i += stride) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= R) break;
            const int64_t v = req[i * R + r];
            neg = neg || v < 0;
            const uint64_t a = v < 0 ? (v == INT64_MIN ? (uint64_t)INT64_MAX : (uint64_t)(-v)) : (uint64_t)v;
            sc[r] = gcd_fold(sc[r], a);
            amax[r] = a > amax[r] ? a : amax[r];
        }
    }
    // wave: butterfly over the lanes (a partner's gcd is folded like a value), block: one partial per wave through LDS
    const int lane = cs::lane(), wave = cs::tid() >> 6, nw = (cs::nthreads() + 63) >> 6;
    for (int d = 32; d > 0; d >>= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint64_t os = cs::readlane_u64(sc[r], lane ^ d), oa = cs::readlane_u64(amax[r], lane ^ d);
            sc[r] = gcd_fold(sc[r], os);
            amax[r] = oa > amax[r] ? oa : amax[r];
        }
    }
    const bool wneg = cs::ballot(neg) != 0;
    uint64_t* sm = (uint64_t*)cs::dyn_smem();   // [nw][9]
    if (lane == 0) { for (int r = 0; r < 4; ++r) { sm[wave * 9 + r] = sc[r]; sm[wave * 9 + 4 + r] = amax[r]; } sm[wave * 9 + 8] = wneg ? 1 : 0; }
    cs::sync();
    if (cs::tid() == 0) {
        GcdPartial p;
        for (int r = 0; r < 4; ++r) { p.sc[r] = 0; p.amax[r] = 0; }
        p.neg = 0;
        for (int w = 0; w < nw; ++w) {
            for (int r = 0; r < 4; ++r) { p.sc[r] = gcd_fold(p.sc[r], sm[w * 9 + r]); p.amax[r] = sm[w * 9 + 4 + r] > p.amax[r] ? sm[w * 9 + 4 + r] : p.amax[r]; }
            p.neg |= sm[w * 9 + 8];
        }
        out[cs::bid()] = p;
    }
}
// exact division by scale = 2^tz * odd: arithmetic shift, then multiply by the inverse of `odd` modulo 2^64 (every value is a multiple)
struct ScaleParams { uint64_t inv[4]; int32_t tz[4]; };
CS_GLOBAL void scale_requests_kernel(const int64_t* CS_RESTRICT req, int64_t n /* G * R */, int R, ScaleParams sp, int32_t* CS_RESTRICT out) {
    const int64_t stride = (int64_t)cs::nblocks() * cs::nthreads();
    for (int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid(); i < n; i += stride) {
        const int r = (int)(i % R);
        out[i] = (int32_t)(int64_t)((uint64_t)(req[i] >> sp.tz[r]) * sp.inv[r]);
    }
}

// casim_pegs.req32 / req_unit (ABI 10): the caller shipped 32-bit multiples of a per-lane unit.  The int64 table every kernel reads is rebuilt here
// (exact: a product), and — only when the node-group columns force a finer scale than the caller's unit — the packer's int32 table as
// req32 * (unit / scale), which the host checked to fit
struct UnitParams { int64_t unit[CASIM_KMAX_RES]; int32_t factor[CASIM_KMAX_RES]; };
CS_GLOBAL void expand_requests_kernel(const int32_t* CS_RESTRICT req32, int64_t n /* G * R */, int R, UnitParams up, int64_t* CS_RESTRICT out64, int32_t* CS_RESTRICT out32 /* or null */) {
    const int64_t stride = (int64_t)cs::nblocks() * cs::nthreads();
    for (int64_t i = (int64_t)cs::bid() * cs::nthreads() + cs::tid(); i < n; i += stride) {
        const int r = (int)(i % R);
        const int32_t v = req32[i];
        out64[i] = (int64_t)v * up.unit[r];
        if (out32) out32[i] = v * up.factor[r];
    }
}

// ------------------------------------------------------------------------------------------
// K_winners: the lists of the winning groups only (casim_options.winners_only)
// ------------------------------------------------------------------------------------------
// After the expander's reduce: best[s][0] = winning group of simulation s (index inside the launch, -1 = no option).  One block scans the
// winners' list lengths into woff[S + 1]; one wave per simulation then copies order / placed of the winner to woff[s].  What leaves the
// device afterwards is sum(len(winner)) entries instead of every (group, PEG) pair — 3.6 MB instead of 69 MB for 4096 C2 simulations.
CS_GLOBAL void winner_offsets_kernel(const int32_t* CS_RESTRICT best /*[S][2]*/, const int32_t* CS_RESTRICT off /*[NG + 1]*/, const int32_t* CS_RESTRICT cnt /*[NG] or null*/,
                                     int S, int32_t* CS_RESTRICT woff /*[S + 1]*/) {
    uint32_t* sm = (uint32_t*)cs::dyn_smem();   // [nw + 1] of the scan + [1] carry
    const int nw = (cs::nthreads() + 63) >> 6;
    uint32_t carry = 0;
    for (int s0 = 0; s0 < S; s0 += cs::nthreads()) {
        const int s = s0 + cs::tid();
        uint32_t mine = 0;
        if (s < S) { const int b = best[2 * s]; if (b >= 0) mine = (uint32_t)(cnt ? cnt[b] : off[b + 1] - off[b]); }
        uint32_t total = 0;
        const uint32_t excl = block_exclusive_scan(mine, sm, &total);
        if (s < S) woff[s] = (int32_t)(carry + excl);
        carry += total;
        cs::sync();   // (sm is rewritten by the next chunk)
    }
    if (cs::tid() == 0) woff[S] = (int32_t)carry;
    (void)nw;
}
CS_GLOBAL void gather_winners_kernel(const int32_t* CS_RESTRICT best, const int32_t* CS_RESTRICT off, const int32_t* CS_RESTRICT cnt, const int32_t* CS_RESTRICT woff,
                                     const int32_t* CS_RESTRICT order, const int32_t* CS_RESTRICT placed, int32_t* CS_RESTRICT worder,
                                     int32_t* CS_RESTRICT wplaced) {
    const int s = cs::bid();
    const int b = best[2 * s];
    if (b < 0) return;
    const int a = off[b], n = cnt ? cnt[b] : off[b + 1] - a, w = woff[s];
    for (int i = cs::tid(); i < n; i += cs::nthreads()) { worder[w + i] = order[a + i]; wplaced[w + i] = placed[a + i]; }
}

// casim_options.chain_last_index: the groups of a simulation as ONE sequence of Estimate() calls on one snapshot — lastIndex lives in the
// snapshot's plugin runner and survives every Estimate (CA/simulator/clustersnapshot/predicate/plugin_runner.go:138: MarkMatch;
// predicate_snapshot.go:64; the Fork / Revert around an estimate does not touch it), so group i starts where group i - 1 ended.
// The packer keeps its parallelism (one wave per group) and iterates to the sequential loop's fixed point instead: after every pass a
// thread per simulation compares each group's INPUT with its predecessor's current OUTPUT, rewrites the input where they differ and marks
// the group for the next pass.  After pass p the first p + 1 groups of every simulation are final (induction over the chain), and a pass
// without marks proves the whole chain consistent — i.e. equal to the sequential loop; the host enqueues (groups per simulation - 1)
// fix-up passes, the bound.  A wave of an unmarked group leaves at its first instruction (pack_unsupported).
CS_GLOBAL void chain_fix_kernel(DevTables t, DevResults res, int32_t* CS_RESTRICT last_index_rw /* = t.last_index */, int32_t* CS_RESTRICT redo /*[NG]*/,
                                int32_t* CS_RESTRICT marks /*[1], += groups marked*/) {
    const int sim = cs::bid() * cs::nthreads() + cs::tid();
    const int n_sims = t.sim_off ? t.n_sims : 1;
    if (sim >= n_sims) return;
    const int g0 = t.sim_off ? t.sim_off[sim] : 0, g1 = t.sim_off ? t.sim_off[sim + 1] : t.NG;
    int n = 0;
    for (int i = g0; i < g1; ++i) {
        int r = 0;
        if (i > g0) {
            const int32_t want = res.last_index_out[i - 1];
            if (last_index_rw[i] != want) { last_index_rw[i] = want; r = 1; ++n; }
        }
        redo[i] = r;
    }
    if (n && marks) cs::atomic_add_i32(marks, n);
}

}  // namespace casim

#include "casim_pack.h"
