// casim_pack.h — K_pack: one wavefront simulates one BinpackingNodeEstimator.Estimate()
// (CA/estimator/binpacking_estimator.go:102-342; `CA/` = /root/reference/cluster-autoscaler/).
//
// Reference semantics, restated as closed forms over a PodEquivalenceGroup of k identical pods:
//   a2  tryToScheduleOnExistingNodes (:163-186): pods visit the simulated nodes round-robin in the
//       cyclic order that starts right after lastIndex (scheduling_opts.go:54-59; MarkMatch moves the
//       start to the matched node, plugin_runner.go:138).  With per-node capacities c_j, after t full
//       rounds node j holds min(c_j, t) pods: find the largest T with
//       S(T) = sum_j min(c_j, T) <= k by bisection on wave sums, hand the remaining k - S(T) pods to
//       the first nodes (rotated order) with c_j > T, and read the new lastIndex off the last one.
//   a3  tryToScheduleOnNewNodes (:190-269): next-fit on the newest node, then ceil(r / c_new) fresh
//       nodes, gated by the limiter (threshold_based_limiter.go:57-69), with the three exits of
//       SURVEY N2 (empty newest node / limiter refusal / pod does not fit a fresh node).
//   a4  tryFastPath (:274-324): one simulated node + arithmetic for the rest (last PEG only).
// Filters: NodeResourcesFit (fit.go:678-765) as integer compares in capacity_of(); TaintToleration /
// NodeAffinity / NodeUnschedulable folded into one precomputed flag by order_kernel; NodePorts and
// InterPodAffinity (anti-affinity) as node-local / group-wide exclusion bits.
//
// Mapping to the wave: simulated node m is owned by lane (m & 63), slot (m >> 6).  Node state is
// lane-private, so the kernel needs no barrier; all cross-lane traffic is ballots, v_mbcnt, v_readlane
// and DPP reductions.  Two node stores share ONE algorithm body:
//   RegStore<R, NPT>  node state in VGPRs as int32 (lanes pre-divided by their gcd on the host — exact,
//                     see casim_pipeline.h), up to 64*NPT nodes, no exclusion bits: the fast path
//   MemStore<kLds, RMAX>  int64 state in LDS (or an HBM slab), any R / node count / exclusion masks
// PEG records arrive in processing order (written by order_kernel).  Memory store: 64 records per coalesced
// wave-load, one per lane, broadcast field by field with v_readlane.  Register store: ONE scalar load per PEG
// (s_load_dwordx8 / x16 of the 32- or 64-byte record of casim_types.h, issued when the previous PEG is done) puts every
// field straight into scalar registers: no VALU, no LDS (the record fetch was 31 % of the kernel's cycles as LDS reads +
// v_readfirstlane, profiles/r02m_pack_phase_profile.txt).
//
// The register packer is bound by SCALAR instruction issue (profiles/r02n .. r03*: a SIMD issues one scalar and one vector
// instruction per ~4 cycles, and a PEG step carried more scalar than vector instructions).  What follows from that, and
// shapes the code below: wave-uniform flags are integers / record bits tested where they are needed (a C++ bool that
// lives across branches becomes a 64-bit lane mask and costs 5-6 scalar instructions per test), only lane MASKS cross
// wave-uniform branches, uniform divisions run on the vector unit, and the PEGs behind a dry limiter — most of a
// scale-up — run in a loop of their own that contains no a3 / a4 code.
#pragma once
#include "casim_device.h"
#include "casim_types.h"

// Optional phase timing (s_memtime ticks per phase, per group) for tools/prof_pack.sh builds only.
#if defined(CASIM_PACK_PROF) && !defined(CASIM_HOST_EMU)
#define CASIM_PROF_DECL uint64_t prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t prof_last = __builtin_amdgcn_s_memtime()
#define CASIM_PROF(i) do { const uint64_t _n = __builtin_amdgcn_s_memtime(); prof_acc[i] += _n - prof_last; prof_last = _n; } while (0)
#define CASIM_PROF_STORE(p) do { if ((p) && cs::lane() == 0) for (int _i = 0; _i < 8; ++_i) (p)[(int64_t)cs::bid() * 8 + _i] = (int64_t)prof_acc[_i]; } while (0)
#else
#define CASIM_PROF_DECL
#define CASIM_PROF(i)
#define CASIM_PROF_STORE(p) (void)(p)
#endif

// Waves per SIMD the register allocator of the 256-node fast packer must leave room for (a lower bound for the
// occupancy: the kernel needs 36 VGPRs by now and runs 7-8 waves per SIMD).  History on MI355X (C1 x 16384):
//   r01f  nested chunk / record loops, records in lanes: 148 VGPRs natural (3 waves) 8.2 M sims/s; bounded to 128 by
//         spilling 68 B/lane 9.2 M but 2.6x the algorithmic HBM traffic; scheduling fences -> 123 VGPRs, no scratch, 9.0 M
//   r01s  one flat PEG loop + records parked in LDS + wave-uniform regions kept out of the CFG structurizer:
//         the node state is no longer copied through every merge point (58 -> 49..55 VGPRs, 8 waves); with the scalar
//         work trimmed as well (the kernel was bound by SALU issue as much as by the VALU) 13.5 M sims/s.
//   r02m+ records by scalar load instead of the LDS parking, scalar-issue economy throughout (see the head of this file).
#ifndef CASIM_FAST_WAVES
#define CASIM_FAST_WAVES 4
#endif

namespace casim {

struct CsTrue { static constexpr bool value = true; };
struct CsFalse { static constexpr bool value = false; };

// kVal: the PEG's exclusion words (and their polarity words) are VALUES of the view — the register stores, whose words live in scalar
// registers (a view that points at local copies keeps them in scratch memory: the optimiser does not see through a pointer stored in a
// struct) — instead of pointers into the mask tables (the memory store: any number of words)
template <class L, int RMAX, bool kVal = false>
struct PegView {
    L req[RMAX];
    double rq[RMAX];  // 1 / req: quotient estimate of capacity_of
    const uint64_t* xblock;
    const uint64_t* xmark;
    uint64_t xb[2], xm[2];  // first words of each, cached for the register store (up to two exclusion words)
    // node bits of NEED polarity (casim_pegs.excl_polarity: hostname-level required pod affinity): a node is blocked when
    // (state & block) != need, need = block & polarity — a plain bit forbids while set, a NEED bit while clear.  `waive`: the first pod of a
    // self-affine series enters by the first-pod exception, the NEED bits it marks itself do not count (block & mark & polarity dropped).
    uint64_t xp[2] = {0, 0};          // need words of xb (register store)
    const uint64_t* xpol = nullptr;   // [Wx] polarity words or null (memory store: need computed per word)
    bool waive = false;
    uint64_t vb[2] = {0, 0}, vm[2] = {0, 0}, vp[2] = {0, 0};   // kVal: block / mark / polarity words
    CS_DEVICE uint64_t bw(int w) const { if constexpr (kVal) return vb[w]; else return xblock[w]; }
    CS_DEVICE uint64_t mw(int w) const { if constexpr (kVal) return vm[w]; else return xmark[w]; }
    CS_DEVICE uint64_t pw(int w) const { if constexpr (kVal) return vp[w]; else return xpol[w]; }
    CS_DEVICE bool has_pol() const { if constexpr (kVal) return true; else return xpol != nullptr; }
    CS_DEVICE uint64_t block_word(int w) const {
        uint64_t b = bw(w);
        if (has_pol() && waive) b &= ~(mw(w) & pw(w));
        return b;
    }
    CS_DEVICE uint64_t need_word(int w) const { return has_pol() ? (block_word(w) & pw(w)) : 0ull; }
};
template <class L, int RMAX, bool kVal = false>
struct FreshNode {
    L free[RMAX];  // alloc - requested by pods preloaded on the template
    int32_t slots;           // allowed pods - preloaded pods
    const uint64_t* excl;    // node bits already set on a fresh node
    uint64_t ve[2] = {0, 0}; // kVal: the same as values (filled by pack_body)
    CS_DEVICE uint64_t ex(int w) const { if constexpr (kVal) return ve[w]; else return excl[w]; }
};

// How many pods with request `req` fit into (free, slots), clamped to `clampk`
// (fitsRequest, fit.go:681-765: pod count first, then every lane with req > 0 needs req <= free).
// Quotients use one f64 multiply by the PEG's precomputed reciprocal + an exact +-1 fix-up (see
// capacity_of in casim_kernels.h for the error bound); int64 lanes above 2^53 fall back to a division.
template <class L, int RMAX, class PV /* a PegView<L, RMAX, .> */>
CS_DEVICE uint32_t capacity_lanes(const L* fr, int32_t slots, int R, const PV& pv, uint32_t clampk) {
    if constexpr (sizeof(L) == 4) {
        // int32 lanes: straight-line code (selects, no lane-divergent branch) so that the slots of one
        // sweep interleave; the only branches are on the wave-uniform request.
        uint32_t c = slots > 0 ? ((uint32_t)slots < clampk ? (uint32_t)slots : clampk) : 0u;
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const L q = pv.req[r];
            if (r < R && q > 0) {  // wave-uniform
                const L f = fr[r];
                const bool fits = f >= q;
                const uint32_t fpos = fits ? (uint32_t)f : 0u;
                uint32_t e = (uint32_t)((double)fpos * pv.rq[r]);  // f < 2^31, e <= f: exact up to +-1
                if (q < (1 << 30)) {   // the remainder lies in (-q, 2q): wrapping 32-bit arithmetic, sign bit + carry fix-up
                    const int32_t rem = (int32_t)(fpos - e * (uint32_t)q);
                    e = e - ((uint32_t)rem >> 31) + (rem >= q ? 1u : 0u);
                } else {
                    const int64_t rem = (int64_t)fpos - (int64_t)((uint64_t)e * (uint64_t)(uint32_t)q);
                    e = rem < 0 ? e - 1 : (rem >= (int64_t)q ? e + 1 : e);
                }
                c = e < c ? e : c;   // fits == false gives e == 0
            }
        }
        return c;
    }
    if (slots <= 0) return 0;
    uint32_t c = (uint32_t)slots < clampk ? (uint32_t)slots : clampk;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        if (r < R) {
            const L q = pv.req[r];
            if (q > 0) {
                const L f = fr[r];
                if (f < q) return 0;
                bool need;
                if constexpr (sizeof(L) == 4) need = (uint64_t)c * (uint64_t)(uint32_t)q > (uint64_t)(uint32_t)f;
                else need = (unsigned __int128)c * (uint64_t)q > (unsigned __int128)(uint64_t)f;
                if (need) {  // floor(f / q) < c
                    uint32_t e;
                    if (sizeof(L) == 4 || (int64_t)f < (1ll << 53)) {
                        e = (uint32_t)((double)f * pv.rq[r]);
                        const int64_t rem = (int64_t)f - (int64_t)((uint64_t)e * (uint64_t)q);
                        if (rem < 0) e -= 1;
                        else if (rem >= (int64_t)q) e += 1;
                    } else e = (uint32_t)((uint64_t)f / (uint64_t)q);
                    c = e;
                }
            }
        }
    }
    return c;
}

// ---- node store: int64 state in LDS / HBM -------------------------------------------------------
// RMAX_: resource lanes carried in registers (2 for the common cpu + memory batch, else CASIM_KMAX_RES): every lane costs
// four scalar registers of PEG record (request + reciprocal), and these kernels live on the edge of the SGPR file.
template <bool kLds, int RMAX_ = CASIM_KMAX_RES>
struct MemStore {
    using Lane = int64_t;
    static constexpr int kNPT = 0;          // 0 = runtime slot count
    static constexpr int kRMax = RMAX_;
    static constexpr bool kHasExcl = true;
    static constexpr bool kHasZone = true;
    static constexpr int kZoneWords = 0;    // group-wide exclusion words live in LDS (per-lane copies), any number
    static constexpr int kRecDw = 0;        // PEG records: three arrays, 64 records per wave-load, fields broadcast by v_readlane
    static constexpr bool kRecWords = false;
    using Peg = PegView<int64_t, RMAX_>;
    using Fresh = FreshNode<int64_t, RMAX_>;
    int R, Wx, cap;
    int64_t* sfree;   // [R][cap]
    uint64_t* sexcl;  // [Wx][cap]
    int32_t* sslots;  // [cap]
    int32_t* snpods;  // [cap]
    int32_t* sctmp;   // [cap]

    CS_DEVICE uint32_t capacity(int, int m, const Peg& pv, uint32_t clampk, bool selfx) const {
        for (int w = 0; w < Wx; ++w)
            if ((sexcl[(int64_t)w * cap + m] & pv.block_word(w)) != pv.need_word(w)) return 0;  // NodePorts / hostname anti-affinity / a missing partner (NEED bits)
        Lane fr[RMAX_];
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) fr[r] = r < R ? sfree[(int64_t)r * cap + m] : 0;
        uint32_t k = capacity_lanes<Lane, RMAX_>(fr, sslots[m], R, pv, clampk);
        if (selfx && k > 1) k = 1;
        return k;
    }
    // place x pods on node m (NodeInfo.AddPodInfo / update, types.go:361-371,439-463)
    CS_DEVICE void commit(int, int m, uint32_t x, const Peg& pv) {
        // (unrolled with a guard: a runtime index would put the PEG record into scratch memory)
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) if (r < R) sfree[(int64_t)r * cap + m] -= (int64_t)x * pv.req[r];
        sslots[m] -= (int32_t)x;
        snpods[m] += (int32_t)x;
        for (int w = 0; w < Wx; ++w) sexcl[(int64_t)w * cap + m] |= pv.mw(w);
    }
    CS_DEVICE void commit_any(int, uint32_t, const Peg&) {}   // (register stores only)
    CS_DEVICE uint64_t fit_mask_lean(int, const Peg&) const { return ~0ull; }   // (register stores only)
    // some simulated node m < M carries one of the PEG's own NEED bits (block & mark & polarity): a partner of the series is placed already
    CS_DEVICE bool any_node_has_own_need(const Peg& pv, int M) const {
        bool any = false;
        const int lane = cs::lane();
        for (int m = lane; m < M; m += 64)
            for (int w = 0; w < Wx; ++w) any = any || (sexcl[(int64_t)w * cap + m] & pv.bw(w) & pv.mw(w) & pv.pw(w)) != 0;
        return cs::ballot(any) != 0;
    }
    CS_DEVICE void create(int, int m, uint32_t x, const Peg& pv, const Fresh& fn) {
#pragma unroll
        for (int r = 0; r < RMAX_; ++r) if (r < R) sfree[(int64_t)r * cap + m] = fn.free[r] - (int64_t)x * pv.req[r];
        sslots[m] = fn.slots - (int32_t)x;
        snpods[m] = (int32_t)x;
        for (int w = 0; w < Wx; ++w) sexcl[(int64_t)w * cap + m] = fn.ex(w) | (x > 0 ? pv.mw(w) : 0ull);
    }
    CS_DEVICE uint32_t get_c(int, int m) const { return (uint32_t)sctmp[m]; }
    CS_DEVICE void set_c(int, int m, uint32_t v) { sctmp[m] = (int32_t)v; }
    CS_DEVICE int32_t npods(int, int m) const { return snpods[m]; }
    // the newest node lm (wave-uniform): every lane may ask, one lane (`mine`) writes
    CS_DEVICE uint32_t capacity_newest(int lm, const Peg& pv, uint32_t clampk, bool selfx, bool mine) const { return mine ? capacity(0, lm, pv, clampk, selfx) : 0u; }
    CS_DEVICE void commit_newest(int lm, uint32_t x, const Peg& pv, bool mine) { if (mine) commit(0, lm, x, pv); }
    CS_DEVICE int32_t npods_newest(int lm, bool mine) const { return mine ? snpods[lm] : 0; }
};

// ---- node store: int32 state in VGPRs (fast path) ---------------------------------------------------
// L_ = int64_t (round 4, R_ == 2 only): the SAME store on the boundary's own type, for batches whose lanes do not narrow to 32 bits
// (byte-granular, co-prime memory requests beyond 2^31 after the gcd) — register pairs instead of registers, 64-byte records carrying the
// int64 requests, quotients by the same f64 estimate + exact fix-up (exact below 2^53, a real division above).  Until round 4 such a
// batch fell back on the LDS store's generic packer, 3.4 x slower on the headline batch.
template <int R_, int NPT_, int WX_ = 0, class L_ = int32_t>
struct RegStore {
    using Lane = L_;
    static constexpr bool k64 = sizeof(L_) == 8;
    static_assert(!k64 || R_ == 2, "the int64 register store keeps two lanes (64-byte records)");
    static constexpr int kNPT = NPT_;
    static constexpr int kRMax = R_;
    static constexpr bool kHasExcl = WX_ > 0;   // WX_ = 1 or 2 words of node-local exclusion bits (host ports, hostname anti-affinity)
    static constexpr bool X_ = WX_ > 0;
    static constexpr bool kHasZone = WX_ > 0;   // the lean instantiation (WX_ = 0) carries no exclusion state at all
    static constexpr int kZoneWords = 2;        // group-wide exclusion words are wave-uniform: up to two, in scalar registers
    static constexpr bool kRecWords = !k64 && R_ == 2 && WX_ > 0;   // the PEG's node-local exclusion words ride with its record (DevResults::rec_xw)
    static constexpr int kRecDw = (k64 || R_ > 2 || kRecWords) ? 16 : 8;   // PEG records: one scalar load of kRecDw dwords per PEG (casim_types.h)
    using Peg = PegView<L_, R_, (WX_ > 0)>;
    using Fresh = FreshNode<L_, R_, (WX_ > 0)>;
    L_ fr[NPT_][R_];
    uint64_t excl[NPT_][WX_ > 0 ? WX_ : 1];
    int wx = 0;   // words the batch really has (<= WX_)
    CS_DEVICE bool blocked(int s, const Peg& pv) const {
        bool b = false;
#pragma unroll
        for (int w = 0; w < WX_; ++w) b = b || (excl[s][w] & pv.xb[w]) != pv.xp[w];   // (NEED bits: blocked while clear)
        return b;
    }
    // some simulated node carries one of the PEG's own NEED bits (slots past M hold zeros)
    CS_DEVICE bool any_node_has_own_need(const Peg& pv, int) const {
        bool any = false;
#pragma unroll
        for (int s = 0; s < NPT_; ++s)
#pragma unroll
            for (int w = 0; w < WX_; ++w) any = any || (w < wx && (excl[s][w] & pv.bw(w) & pv.mw(w) & pv.pw(w)) != 0);
        return cs::ballot(any) != 0;
    }
    int32_t slots[NPT_];
    int32_t fresh_slots;  // pod slots of an empty node: pods on a node = fresh_slots - slots (no per-node counter)

    // `simple`: the common PEG asks for every lane and no request is huge (CASIM_REC_SIMPLE, decided once by order_kernel),
    // so that the slot code carries no per-lane branches (each wave-uniform branch costs scalar issue slots, and this
    // kernel is bound by them more than by the VALU: profiles/r01s_*, r02n_*).
    //
    // fit_mask: the wave mask of "a pod of the PEG fits node (s, lane)" as the AND of single-compare masks (each v_cmp
    // writes its mask register pair).  ONLY this mask crosses the wave-uniform branches of a2: inside them it IS the lane
    // predicate again (cs::lane_pred).  A bool carried across such a branch is materialised in a VGPR and compared back
    // (6 VALU per slot), and so is a compare that two branches share.
    CS_DEVICE uint64_t fit_mask(int s, const Peg& pv, uint32_t pf /* record flags */) const {
        uint64_t fb;
        if ((pf & CASIM_REC_SIMPLE) != 0) {   // (plain here, opaque in capacity_slot: two separate bit tests, no shared lane-mask bool)
            fb = cs::ballot(slots[s] > 0);
            if (X_) fb &= cs::ballot(!blocked(s, pv));
#pragma unroll
            for (int r = 0; r < R_; ++r) fb &= cs::ballot(fr[s][r] >= pv.req[r]);
        } else {
            // (the slot count through an opaque copy: otherwise the compiler shares the compare with the branch above and
            // carries its result across as a 0 / 1 VGPR — two VALU more on the common path)
            fb = cs::ballot(cs::opaque_i32(slots[s]) > 0);
            if (X_) fb &= cs::ballot(!blocked(s, pv));
#pragma unroll
            for (int r = 0; r < R_; ++r) if (pv.req[r] > 0) fb &= cs::ballot(fr[s][r] >= pv.req[r]);   // wave-uniform test
        }
        return fb;
    }
    // the same for a PEG without exclusion words whose requests are all positive (an `idle` step of pack_body): slots and requests only
    CS_DEVICE uint64_t fit_mask_lean(int s, const Peg& pv) const {
        uint64_t fb = cs::ballot(slots[s] > 0);
#pragma unroll
        for (int r = 0; r < R_; ++r) fb &= cs::ballot(fr[s][r] >= pv.req[r]);
        return fb;
    }
    // c_j of slot s for a non-empty fit mask: cheap compares decided who fits, only now the quotient chain
    // (cvt -> f64 mul by the PEG's reciprocal -> cvt -> exact +-1 fix-up) runs.  In the steady state of a scale-up most PEGs
    // fit nowhere or on a few nodes, so most slots stop at the mask.
    CS_DEVICE uint32_t capacity_slot(int s, const Peg& pv, uint32_t clampk, uint32_t pf, uint64_t fb) const {
        const bool fit = cs::lane_pred(fb);
        if (cs::flag_set(pf, CASIM_PEG_SELF_EXCL_NODE)) return fit ? 1u : 0u;   // clampk >= 1, a pod slot is free and every requested lane fits once
        uint32_t k = (uint32_t)slots[s] < clampk ? (uint32_t)slots[s] : clampk;
        if constexpr (k64) {
            // int64 lanes: the quotient estimate in f64 is exact up to +-1 while the free amount is below 2^53 and the quotient below 2^31
            // (an estimate beyond that is clamped: the pod-slot bound k is smaller anyway); larger amounts take the real division
#pragma unroll
            for (int r = 0; r < R_; ++r) {
                const int64_t q = pv.req[r];
                if (q > 0) {   // wave-uniform
                    const int64_t f = fit ? fr[s][r] : 0;
                    uint32_t e;
                    if (f < (1ll << 53)) {
                        const double ed = (double)f * pv.rq[r];
                        e = ed >= 2147483648.0 ? 0x7fffffffu : (uint32_t)ed;
                        const int64_t rem = f - (int64_t)((uint64_t)e * (uint64_t)q);
                        if (e != 0x7fffffffu) e = rem < 0 ? e - 1 : (rem >= q ? e + 1 : e);
                    } else {
                        const uint64_t d = (uint64_t)f / (uint64_t)q;
                        e = d > 0x7fffffffull ? 0x7fffffffu : (uint32_t)d;
                    }
                    k = e < k ? e : k;
                }
            }
            return fit ? k : 0u;
        }
        if (cs::flag_set(pf, CASIM_REC_SIMPLE)) {
            // (lanes that do not fit compute garbage and are masked at the end: no select per operand)
#pragma unroll
            for (int r = 0; r < R_; ++r) {
                const int32_t q = pv.req[r];
                const uint32_t fpos = (uint32_t)fr[s][r];
                uint32_t e = (uint32_t)((double)fpos * pv.rq[r]);
                const int32_t rem = (int32_t)(fpos - e * (uint32_t)q);   // wrapping 32-bit: it lies in (-q, 2q)
                e = e - ((uint32_t)rem >> 31) + (rem >= q ? 1u : 0u);    // exact +-1 fix-up: sign bit, carry-in
                k = e < k ? e : k;
            }
        } else {
#pragma unroll
            for (int r = 0; r < R_; ++r) {
                const int32_t q = pv.req[r];
                if (q > 0) {  // wave-uniform
                    const uint32_t fpos = fit ? (uint32_t)fr[s][r] : 0u;
                    uint32_t e = (uint32_t)((double)fpos * pv.rq[r]);
                    if (q < (1 << 30)) {  // remainder in wrapping 32-bit arithmetic: it lies in (-q, 2q)
                        const int32_t rem = (int32_t)(fpos - e * (uint32_t)q);
                        e = rem < 0 ? e - 1 : (rem >= q ? e + 1 : e);
                    } else {
                        const int64_t rem = (int64_t)fpos - (int64_t)((uint64_t)e * (uint64_t)(uint32_t)q);
                        e = rem < 0 ? e - 1 : (rem >= (int64_t)q ? e + 1 : e);
                    }
                    k = e < k ? e : k;
                }
            }
        }
        return fit ? k : 0u;
    }
    // Pass A for all slots.  Slots beyond M hold zeros (slots == 0): no extra guard.  Returns n1 = nodes taking >= 1 pod.
    // The capacities go to the caller's array: they are dead after a2, and as a member they travelled through every
    // merge point of the PEG loop with the rest of the state.
    // `act`: bit s set when some node of slot s takes a pod — the later passes skip the other slots (measured on C1:
    // 0.6 of the 4 slots per PEG).
    // S = slots in use ((M + 63) >> 6): the 16-slot store skips the empty ones with one scalar compare each (a group whose BOUND is
    // 1024 nodes usually creates far fewer: BenchmarkRunOnceScaleUp 200) — stores of <= 4 slots walk them all (constant bound).
    CS_DEVICE int32_t capacity_all(const Peg& pv, uint32_t clampk, uint32_t pf, uint32_t* c, uint32_t& act, int S) const {
        act = 0;
        int32_t n1 = 0;
#pragma unroll
        for (int s = 0; s < NPT_; ++s) {
            if (NPT_ > 4 && s >= S) { c[s] = 0; continue; }   // wave-uniform
            const uint64_t fb = fit_mask(s, pv, pf);
            uint32_t k = 0;
            if (fb) {  // wave-uniform
                n1 += cs::popc64(fb);
                act |= 1u << s;
                k = capacity_slot(s, pv, clampk, pf, fb);
            }
            c[s] = k;
            if (NPT_ >= 4 && (s & 1) == 1) cs::sched_fence();  // interleave two slots at a time: bounds the live temporaries
        }
        return n1;
    }
    CS_DEVICE void commit(int s, int, uint32_t x, const Peg& pv) {
#pragma unroll
        for (int r = 0; r < R_; ++r) fr[s][r] -= (L_)x * pv.req[r];
        slots[s] -= (int32_t)x;
#pragma unroll
        for (int w = 0; w < WX_; ++w) excl[s][w] |= pv.xm[w];
    }
    // the same for any x >= 0 (x == 0: no change), straight-line
    CS_DEVICE void commit_any(int s, uint32_t x, const Peg& pv) {
#pragma unroll
        for (int r = 0; r < R_; ++r) fr[s][r] -= (L_)x * pv.req[r];
        slots[s] -= (int32_t)x;
#pragma unroll
        for (int w = 0; w < WX_; ++w) excl[s][w] |= x > 0 ? pv.xm[w] : 0ull;
    }
    CS_DEVICE void create(int s, int, uint32_t x, const Peg& pv, const Fresh& fn) {
#pragma unroll
        for (int r = 0; r < R_; ++r) fr[s][r] = fn.free[r] - (L_)x * pv.req[r];
        slots[s] = fn.slots - (int32_t)x;
#pragma unroll
        for (int w = 0; w < WX_; ++w) excl[s][w] = (w < wx ? fn.ex(w) : 0ull) | (x > 0 ? pv.xm[w] : 0ull);   // wx <= WX_ words exist
    }
    CS_DEVICE uint32_t get_c(int, int) const { return 0; }   // (register store: pack_body keeps the capacities itself)
    CS_DEVICE void set_c(int, int, uint32_t) {}
    CS_DEVICE int32_t npods(int s, int) const { return fresh_slots - slots[s]; }  // only asked for created nodes
    // The newest node lm (wave-uniform index; slot lm >> 6 of lane lm & 63).  Straight-line selects over the slots
    // instead of a branch per slot (with branches the compiler carried the whole register state through every merge
    // point), and the select condition is ONE lane compare per slot (s * 64 + lane == lm): only the owner lane sees its
    // node, the others evaluate an empty one.  The caller reads the owner lane.
    CS_DEVICE uint32_t capacity_newest(int lm, const Peg& pv, uint32_t clampk, bool selfx, bool) const {
        const int lane = cs::lane();
        L_ f[R_]; int32_t sl = 0;
        bool b = false;
#pragma unroll
        for (int r = 0; r < R_; ++r) f[r] = 0;
#pragma unroll
        for (int s = 0; s < NPT_; ++s) {
            const bool h = s * 64 + lane == lm;
#pragma unroll
            for (int r = 0; r < R_; ++r) f[r] = h ? fr[s][r] : f[r];
            sl = h ? slots[s] : sl;
            if (X_) b = h ? blocked(s, pv) : b;
        }
        if (X_ && b) return 0;
        uint32_t k = capacity_lanes<Lane, R_>(f, sl, R_, pv, clampk);
        if (selfx && k > 1) k = 1;
        return k;
    }
    CS_DEVICE void commit_newest(int lm, uint32_t x, const Peg& pv, bool) {
        const int lane = cs::lane();
#pragma unroll
        for (int s = 0; s < NPT_; ++s) {
            const bool h = s * 64 + lane == lm;
#pragma unroll
            for (int r = 0; r < R_; ++r) fr[s][r] -= h ? (L_)x * pv.req[r] : (L_)0;
            slots[s] -= h ? (int32_t)x : 0;
#pragma unroll
            for (int w = 0; w < WX_; ++w) excl[s][w] |= h ? pv.xm[w] : 0ull;
        }
    }
    CS_DEVICE int32_t npods_newest(int lm, bool) const {
        const int lane = cs::lane();
        int32_t sl = 0;
#pragma unroll
        for (int s = 0; s < NPT_; ++s) sl = s * 64 + lane == lm ? slots[s] : sl;
        return fresh_slots - sl;
    }
    // (An earlier version kept wave-uniform upper bounds of the free resources to skip the sweep of a PEG that fits
    // nowhere, re-tightened by R+1 wave reductions whenever a sweep came back empty.  Measured on C1 the bounds never
    // pruned a single PEG while the re-tightening cost ~13 VALU instructions per PEG; an empty sweep is 3 compares per
    // slot now, so the bounds are gone.)
};

// slot loops: fully unrolled for the register store (static VGPR indices), runtime for the memory store
template <class Store, class F>
CS_DEVICE void for_slots(int S, F&& f) {
    if constexpr (Store::kNPT > 0) {
#pragma unroll
        for (int s = 0; s < Store::kNPT; ++s)
            if (s < S) f(s);
    } else {
        for (int s = 0; s < S; ++s) f(s);
    }
}
// loops over the (<= 2) exclusion words of a register store with constant indices (the words live in scalar registers: a runtime index
// would send them through memory), over any number of words for the memory store
template <bool kTwoStatic, class F>
CS_DEVICE void for_words(int n, F&& f) {
    if constexpr (kTwoStatic) { if (0 < n) f(0); if (1 < n) f(1); }
    else { for (int w = 0; w < n; ++w) f(w); }
}
// one 64-bit word at a wave-uniform address, written before the kernel started: a scalar load
CS_DEVICE uint64_t uniform_word(const uint64_t* p) {
    const cs::Words<2> v = cs::const_load<2>((const uint32_t*)p);
    return ((uint64_t)v.w[1] << 32) | v.w[0];
}
// The algorithm body, shared by every store.
//   ReqLoader(kk, r) -> request lane r of sorted record kk (int64 original or int32 gcd-scaled)
template <class Store, class ReqLoader>
CS_DEVICE void pack_body(const DevTables& t, const DevResults& res, Store& st, const typename Store::Fresh& fn_in,
                         uint64_t* szone /*[Wz][64] per-lane copies or null*/, ReqLoader load_req, const int64_t* sum_scale,
                         int64_t* prof_out = nullptr) {
    using L = typename Store::Lane;
    constexpr int RM = Store::kRMax;
    constexpr int DW = Store::kRecDw;
    constexpr bool kRecScalar = DW > 0;
    const int ng = cs::bid();
    const int lane = cs::lane();
    const int R = t.R;
    const int Wx = Store::kHasExcl ? t.Wx : 0;
    const int Wz = Store::kHasZone ? t.Wz : 0;
    // (per-group scalars: written by the host or by earlier kernels, read through the scalar cache into scalar registers —
    // as vector loads they lived in VGPRs and every use of them was a VALU instruction)
    auto gload = [&](const void* base, int64_t i) -> int32_t { return (int32_t)cs::const_load<1>((const uint32_t*)base + i).w[0]; };
    const int off = gload(t.peg_off, ng);
    const int Gn = t.peg_cnt ? gload(t.peg_cnt, ng) : gload(t.peg_off, ng + 1) - off;   // (fixed-stride lists carry their length apart)

    const int32_t maxn = gload(t.max_nodes, ng);
    const int32_t E = gload(t.existing, ng);
    const bool fast_last = t.fastpath && res.fast_last[ng];
    // the limiter as one number: 0 = grants nothing (max_nodes < 0), INT32_MAX = no limit (max_nodes == 0)
    const int32_t grant_bound = maxn < 0 ? 0 : (maxn == 0 ? 0x7fffffff : maxn);
    int32_t fast_k = fast_last ? Gn - 1 : -1;   // the PEG that takes tryFastPath, if any (pinned: one compare per PEG, not flag AND compare)
    { uint32_t fk = (uint32_t)fast_k; cs::keep_scalar(fk); fast_k = (int32_t)fk; }
    const bool group_unschedulable = ((uint32_t)gload(t.gflags, ng) & CASIM_NG_UNSCHEDULABLE) != 0;
    const uint64_t* zvalid_p = t.zone_valid + (int64_t)ng * Wz;
    // Register store with exclusion state: <= 2 node-local and <= 2 group-wide words, ALL of them wave-uniform.  What is constant for the
    // group — polarity words, the bits a fresh node starts with, the valid group bits — is fetched ONCE, by scalar loads, into scalar
    // registers, and travels as VALUES (PegView / FreshNode kVal; the accessors below).  (Read through the table pointers where they were
    // needed, the compiler issued a VECTOR load of the same address in every lane and waited for it, several times per PEG step: 87 % of
    // the anti-affinity packer's time on BASELINE config C4, profiles/r15c_pack_phase_profile_c4.txt.)
    constexpr bool kRegX = Store::kNPT > 0 && Store::kHasExcl;
    uint64_t c_xpol[2] = {0, 0}, c_zpol[2] = {0, 0}, c_zvalid[2] = {0, 0};
    typename Store::Fresh fn = fn_in;
    if constexpr (kRegX) {
        for_words<true>(Wx, [&](int w) { c_xpol[w] = uniform_word(t.xpol + w); fn.ve[w] = uniform_word(fn_in.excl + w); });
        for_words<true>(Wz, [&](int w) { c_zpol[w] = uniform_word(t.zpol + w); c_zvalid[w] = uniform_word(zvalid_p + w); });
    }
    auto zpol = [&](int w) -> uint64_t { if constexpr (kRegX) return c_zpol[w]; else return t.zpol[w]; };
    auto zvalid = [&](int w) -> uint64_t { if constexpr (kRegX) return c_zvalid[w]; else return zvalid_p[w]; };
    auto xpolw = [&](int w) -> uint64_t { if constexpr (kRegX) return c_xpol[w]; else return t.xpol[w]; };
    // (register store: words past the batch's Wx / Wz are zeros that change no verdict — the word logic below runs over both words without
    // asking, each `w < Wx` was a scalar compare + branch + the register moves of its merge point in every PEG step)
    const int WxL = kRegX ? 2 : Wx;
    // group-wide exclusion state (anti-affinity on non-hostname keys): identical in every lane
    constexpr int ZR = Store::kZoneWords;
    uint64_t zreg[ZR > 0 ? ZR : 1];
    auto zone_init = [&]() {
        if constexpr (ZR > 0) {
#pragma unroll
            for (int w = 0; w < ZR; ++w) zreg[w] = w < Wz ? t.init_zone[(int64_t)ng * Wz + w] : 0ull;
        } else {
            for (int w = 0; w < Wz; ++w) szone[w * 64 + lane] = t.init_zone[(int64_t)ng * Wz + w];
        }
    };
    auto zone_blocked = [&](auto zb /* w -> block word */) -> bool {
        bool b = false;
        if constexpr (ZR > 0) {
#pragma unroll
            for (int w = 0; w < ZR; ++w) if (w < Wz) b |= ((zreg[w] ^ zpol(w)) & zb(w)) != 0;   // (a NEED bit forbids while clear)
        } else {
            for (int w = 0; w < Wz; ++w) b |= ((szone[w * 64 + lane] ^ zpol(w)) & zb(w)) != 0;
        }
        return b;
    };
    auto zone_mark = [&](auto zm /* w -> mark word */) {
        if constexpr (ZR > 0) {
#pragma unroll
            for (int w = 0; w < ZR; ++w) if (w < Wz) zreg[w] |= zm(w) & zvalid(w);
        } else {
            for (int w = 0; w < Wz; ++w) szone[w * 64 + lane] |= zm(w) & zvalid(w);
        }
    };
    zone_init();

    int32_t M = 0;                         // simulated nodes so far (estimationState.newNodeNameIndex)
    int32_t last_index = gload(t.last_index, ng); // lastIndexOrderMapping.lastIndex
    int32_t granted = 0;                   // limiter.nodes
    // newNodesAvailable, as an all-ones / zero word: wave-uniform flags that live across PEGs are kept as integers and
    // tested with integer arithmetic — as bools the compiler carries them as 64-bit lane masks (s_cselect_b64 / s_and_b64 /
    // s_andn2 with exec / s_cbranch_vcc: 5-6 scalar instructions per test instead of 2-3, in a kernel bound by scalar issue)
    int32_t more_mask = -1;
    int32_t fakes = 0;                     // fastpath fake nodes
    int32_t total_placed = 0;
    // self-affine hostname series (casim_pegs.excl_polarity): 1 = the record is being walked a second time (see peg_step)
    int series_phase = 0;
    bool series_skip_a2 = false, series_repeat = false;
    // register store of 16 slots per lane only: the group may ask for more than its 1024 node slots (its BOUND — limiter cap or
    // pods — exceeds them; most such groups never get there: BenchmarkRunOnceScaleUp is bound by 10 000 and creates 200).  The
    // step that would create node 1025 stops node creation and marks the group for the generic packer's retry launch.
    uint32_t overflow = 0;

    CASIM_PROF_DECL;
    // ONE loop over the PEGs of the group (a nested chunk / record loop made the compiler keep two copies of the node state,
    // one per loop level, and shuffle ~70 registers per PEG).  Memory store: every 64th iteration loads the next chunk of
    // records into the lanes.  Register store: record k + 1 is fetched by a scalar load at the end of PEG k.
    int32_t my_cnt = 0, my_g = 0, my_placed = 0;
    uint32_t my_flags = 0, my_cf = 0;   // my_cf: pods of the record's PEG that fit an EMPTY node (state-independent)
    L my_req[RM];
    double my_rq[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) { my_req[r] = 0; my_rq[r] = 0.0; }
    const uint32_t* recp = nullptr;          // this group's records (register store)
    // The record walk carries ONE 32-bit scalar: the byte offset of the current record inside the group's record array (a buffer
    // resource, cs::rec_load).  The PEG index k = roff >> kRecShift is derived where a step needs it (placements, chunk edges);
    // the common step — a PEG that fits nowhere — only adds the record size and compares with the end offset.
    constexpr uint32_t kRecBytes = 4u * (uint32_t)(DW > 0 ? DW : 1);
    constexpr int kRecShift = DW == 16 ? 6 : 5;
    cs::RecBase rbase;
    uint32_t roff = 0;
    const uint32_t rend = (uint32_t)Gn * kRecBytes;
    cs::Words<(DW > 0 ? DW : 1)> cur;
    cur.w[0] = 0;
    // Register store: one dword of every record of the NEXT chunk is touched by a vector load at each chunk boundary (64 lanes x
    // record stride = the chunk's 2 / 4 KB, 16 / 32 cache lines in flight together) and consumed a chunk later, i.e. never waited
    // for: the scalar loads of the PEG steps then find their lines in this XCD's L2 instead of paying the memory latency line by
    // line — a lone wave (one Estimate per call, BenchmarkRunOnceScaleUp's 10 000 dependent steps) is bound by exactly that latency.
    uint32_t rec_touch = 0;
    auto touch_chunk = [&](int kbase) {
        if constexpr (kRecScalar) {
            cs::consume_u32(rec_touch);
            const int kk = kbase + lane;
            rec_touch = kk < Gn ? recp[(int64_t)kk * DW] : 0u;
        }
    };
    int32_t g_cur = 0;                       // the PEG's index into the mask tables (stores with exclusion state only)
    constexpr bool kNeedG = Store::kHasExcl || Store::kHasZone;
    // a2 is worth entering when the record says so (CASIM_REC_A2_OK) AND the group has simulated nodes AND its template is
    // schedulable: the last two as a mask that is zero until the first node exists
    // (the records of an unschedulable template carry no CASIM_REC_A2_OK: order_group)
    if constexpr (kRecScalar) {
        recp = res.rec + (int64_t)off * DW;
        rbase = cs::rec_base(recp);
        touch_chunk(0);
        // (the record array ends with one spare record: the load of record k + 1 needs no bound)
        cur = cs::rec_load<DW>(rbase, 0u);
        if (kNeedG) g_cur = (int32_t)cs::const_load<1>((const uint32_t*)res.order + off).w[0];
    }
    // results of a finished chunk: placed[] (one coalesced wave-store per 64 PEGs); memory store: also each lane's share of
    // the sum(placed * request) totals, per lane and per chunk instead of two 64-bit scalar multiply-adds per PEG
    int64_t acc0 = 0, acc1 = 0;
    auto flush_chunk = [&](int kbase) {
        if (kbase + lane < Gn) res.placed[off + kbase + lane] = my_placed;
        if constexpr (!kRecScalar) {
            acc0 = cs::wrap_madd_i64(acc0, my_placed, (int64_t)my_req[0]);   // lanes without a record hold my_placed == 0
            acc1 = cs::wrap_madd_i64(acc1, my_placed, (int64_t)(RM > 1 ? my_req[1] : (L)0));
        }
    };
    // register store: the totals grow where a PEG records what it placed — every lane the same product, on the VALU (the
    // scalar unit is the busy one, and as scalar 64-bit multiply-adds these are eight instructions).  Reading the lane's own
    // record back at the chunk flush instead touched every record line a second time, 64 PEGs after the scalar load had
    // dropped it from the caches: the kernel's HBM read traffic was twice its algorithmic bytes (profiles/r03c).
    auto add_totals = [&](int32_t placed, L q0, L q1) {
        const int32_t pv_ = cs::opaque_i32(placed);   // (a VGPR copy: keeps the arithmetic off the scalar unit)
        acc0 = cs::wrap_madd_i64(acc0, pv_, (int64_t)q0);
        acc1 = cs::wrap_madd_i64(acc1, pv_, (int64_t)q1);
    };
    // One PEG.  kDry = the limiter has run dry (newNodesAvailable == false): a3 / a4 can never be entered again, and a PEG that
    // fits no simulated node leaves no trace at all (placed 0, my_placed already 0, lastIndex kept) — in C2 that is 60 % of
    // all steps (profiles/r02s_packer_notes.txt), so the register store runs them in a loop of their own without the a3
    // code and its tests.
    auto peg_step = [&](const int k_arg, auto dry_tag) __attribute__((always_inline)) {
        constexpr bool kDry = decltype(dry_tag)::value;
        int k;
        if constexpr (kRecScalar) k = (int)(roff >> kRecShift); else k = k_arg;   // (scalar records: only the steps that place pods read it)
        const int j = k & 63;
        if (!kDry && j == 0) {   // (the dry loop below walks chunk by chunk and does this itself)
            CASIM_PROF(0);  // chunk load / store, loop overhead
            if (k > 0) flush_chunk(k - 64);
            my_placed = 0;
            touch_chunk(k + 64);
            if constexpr (!kRecScalar) {
                // ---- one coalesced wave-load: PEG record k+lane of this group, in processing order ----
                const int kk = k + lane;
                const bool have = kk < Gn;
                my_cnt = have ? res.s_count[off + kk] : 0;
                my_flags = have ? res.s_flags[off + kk] : 0u;
                my_g = (have && kNeedG) ? res.order[off + kk] : 0;   // (only the mask tables are indexed by it)
#pragma unroll
                for (int r = 0; r < RM; ++r) my_req[r] = (have && r < R) ? load_req(off + kk, r) : (L)0;
                // 1 / req for capacity_lanes' quotient estimate: computed by the record's own lane, i.e. 64 PEGs'
                // worth of f64 divisions per wave instruction instead of one uniform division per PEG
#pragma unroll
                for (int r = 0; r < RM; ++r) my_rq[r] = my_req[r] > 0 ? 1.0 / (double)my_req[r] : 0.0;
                // capacity of a fresh node for the record's PEG, by the record's own lane: once per 64 PEGs instead of a
                // wave-uniform quotient chain in every a3
                typename Store::Peg mine;
#pragma unroll
                for (int r = 0; r < RM; ++r) { mine.req[r] = my_req[r]; mine.rq[r] = my_rq[r]; }
                mine.xblock = nullptr; mine.xmark = nullptr; mine.xb[0] = mine.xb[1] = 0; mine.xm[0] = mine.xm[1] = 0;
                my_cf = have ? capacity_lanes<L, RM>(fn.free, fn.slots, R, mine, 0x7fffffffu) : 0u;
            }
        }
        {
            int32_t cnt; uint32_t pf, cf_rec = 0;
            typename Store::Peg pv;
            int32_t series_carry = 0, series_cnt = 0;   // pods of the record placed by pass 0 / pods of the whole record
            bool series_first = false;                  // this is pass 0 of a series
            if constexpr (kRecScalar) {
                cnt = (int32_t)cur.w[0];
                pf = cur.w[1];
                cf_rec = (cur.w[1] >> CASIM_REC_FRESH_SHIFT) & CASIM_REC_FRESH_MAX;
                if constexpr (sizeof(L) == 8) {   // 64-byte record of the int64 register store: [2..5] two int64 requests, [6..9] their reciprocals
#pragma unroll
                    for (int r = 0; r < RM; ++r) {
                        pv.req[r] = (L)(((uint64_t)cur.w[2 + 2 * r + 1] << 32) | cur.w[2 + 2 * r]);
                        pv.rq[r] = cs::bits_double(((uint64_t)cur.w[2 + 2 * RM + 2 * r + 1] << 32) | cur.w[2 + 2 * RM + 2 * r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < RM; ++r) {
                        pv.req[r] = (L)cur.w[2 + r];   // (lanes past R hold a zero request and a zero reciprocal)
                        pv.rq[r] = cs::bits_double(((uint64_t)cur.w[2 + RM + 2 * r + 1] << 32) | cur.w[2 + RM + 2 * r]);
                    }
                }
            } else {
                cnt = (int32_t)cs::bcast_u32((uint32_t)my_cnt, j);
                pf = cs::bcast_u32(my_flags, j);
#pragma unroll
                for (int r = 0; r < RM; ++r) {
                    if (r < R) {
                        if constexpr (sizeof(L) == 4) pv.req[r] = (L)cs::bcast_u32((uint32_t)my_req[r], j);
                        else pv.req[r] = (L)cs::bcast_u64((uint64_t)my_req[r], j);
                    } else pv.req[r] = 0;
                    pv.rq[r] = cs::bits_double(cs::bcast_u64(cs::double_bits(my_rq[r]), j));
                }
            }
            // Behind a dry limiter, a PEG WITHOUT exclusion words (a record that carries them says so in CASIM_REC_A2_SIMPLE: order_group) in a
            // batch without group-wide words either fits a node by requests and pod slots alone or leaves no trace at all: the lean store's
            // three compares, before any of the word logic below (BASELINE C4: ~70 % of all steps; it cost them ~60 scalar instructions each)
            bool idle = false;
            if constexpr (kDry && Store::kRecWords && Store::kNPT == 1) {
                if ((pf & CASIM_REC_A2_SIMPLE) != 0 && Wz == 0) idle = st.fit_mask_lean(0, pv) == 0ull;
            }
            if (!idle) {
            const bool selfx = (pf & CASIM_PEG_SELF_EXCL_NODE) != 0;
            // (group-wide self-exclusion needs a store with zone state: the pipeline sends such batches to one)
            bool zselfx = Store::kHasZone && (pf & CASIM_PEG_SELF_EXCL_ZONE) != 0;
            const bool static_ok = (pf & CASIM_KFLAG_STATIC_OK) != 0;
            pv.xblock = nullptr; pv.xmark = nullptr; pv.xb[0] = pv.xb[1] = 0; pv.xm[0] = pv.xm[1] = 0;
            const uint64_t *zblock_p = nullptr, *zmark_p = nullptr;
            uint64_t c_zb[2] = {0, 0}, c_zm[2] = {0, 0};   // register store: the PEG's group-wide words, in scalar registers
            auto zblock = [&](int w) -> uint64_t { if constexpr (kRegX) return c_zb[w]; else return zblock_p[w]; };
            auto zmark = [&](int w) -> uint64_t { if constexpr (kRegX) return c_zm[w]; else return zmark_p[w]; };
            if (Wx > 0 || Wz > 0) {
                int g;
                if constexpr (kRecScalar) g = g_cur; else g = (int)cs::bcast_u32((uint32_t)my_g, j);
                if constexpr (kRegX) {
                    if constexpr (Store::kRecWords) {   // (they came with the record: casim_types.h, DevResults::rec_xw)
                        pv.vb[0] = ((uint64_t)cur.w[9] << 32) | cur.w[8];   pv.vb[1] = ((uint64_t)cur.w[11] << 32) | cur.w[10];
                        pv.vm[0] = ((uint64_t)cur.w[13] << 32) | cur.w[12]; pv.vm[1] = ((uint64_t)cur.w[15] << 32) | cur.w[14];
                    } else {
                        for_words<true>(Wx, [&](int w) { pv.vb[w] = uniform_word(t.xblock + (int64_t)g * Wx + w); pv.vm[w] = uniform_word(t.xmark + (int64_t)g * Wx + w); });
                    }
                    pv.vp[0] = c_xpol[0]; pv.vp[1] = c_xpol[1];
                    if (Wz > 0) for_words<true>(Wz, [&](int w) { c_zb[w] = uniform_word(t.zblock + (int64_t)g * Wz + w); c_zm[w] = uniform_word(t.zmark + (int64_t)g * Wz + w); });
                } else {
                    pv.xblock = t.xblock + (int64_t)g * Wx; pv.xmark = t.xmark + (int64_t)g * Wx;
                    zblock_p = t.zblock + (int64_t)g * Wz; zmark_p = t.zmark + (int64_t)g * Wz;
                }
                if constexpr (Store::kHasExcl) {
                    if (Wx > 0) {
                        pv.xpol = t.xpol;
                        // A PEG that needs a NEED bit it marks itself: a self-affine series on the hostname key (casim_pegs.excl_polarity).
                        // While no simulated node carries the bit and a fresh node would not either, NO matching pod exists anywhere (the
                        // encoder only builds the series when the cluster holds none): the first pod passes without it (filtering.go:396-407)
                        // — every pod of the PEG would, one by one, but the moment one is placed the rest has to join ITS node.  So the
                        // record is walked twice: pass 0 places ONE pod with the bits waived, pass 1 the other cnt - 1 with the bits in force.
                        uint64_t own = 0;
                        for_words<kRegX>(WxL, [&](int w) { own |= pv.bw(w) & pv.mw(w) & xpolw(w); });
                        if (own != 0) {   // wave-uniform, rare
                            if (series_phase == 1) { series_carry = 1; cnt -= 1; }
                            else {
                                bool partner = st.any_node_has_own_need(pv, M);
                                for_words<kRegX>(WxL, [&](int w) { partner = partner || (fn.ex(w) & pv.bw(w) & pv.mw(w) & xpolw(w)) != 0; });
                                if (!partner) { series_first = true; series_cnt = cnt; cnt = cnt > 0 ? 1 : 0; pv.waive = true; }
                            }
                        }
                    }
                }
                if (Store::kNPT > 0 && Wx > 0) {   // register store: the words travel in SGPRs
                    pv.xb[0] = pv.block_word(0); pv.xm[0] = pv.mw(0); pv.xp[0] = pv.need_word(0);
                    if (kRegX || Wx > 1) { pv.xb[1] = pv.block_word(1); pv.xm[1] = pv.mw(1); pv.xp[1] = pv.need_word(1); }
                }
            }
            bool zblocked = Wz > 0 && zone_blocked(zblock);
            if (!kRegX || Wz > 0) for_words<kRegX>(kRegX ? 2 : Wz, [&](int w) { zselfx |= (zblock(w) & zmark(w) & zvalid(w) & ~zpol(w)) != 0; });  // the PEG excludes itself group-wide (a NEED bit it sets itself is the opposite)

            CASIM_PROF(1);  // record broadcast + reciprocals
            int32_t placed = 0;
            uint32_t on_last = 0;  // pods of THIS PEG that a2 put on the newest node (self-exclusion has no node bit)

            // ---- a2: closed form of the cyclic first-fit over the simulated nodes ----
            // RunFiltersUntilPassingNode skips Spec.Unschedulable nodes before any Filter runs, tolerated or
            // not (plugin_runner.go:108-110); every simulated node clones the template's flag.
            const uint32_t keff = (uint32_t)(zselfx ? (cnt > 0 ? 1 : 0) : cnt);
            bool a2_go;
            uint64_t fb_first = 0;   // one-slot register store behind a dry limiter: the fit mask IS the gate
            if constexpr (kRecScalar) {
                if constexpr (!kDry) {
                    // (two scalar tests, each its own compare + branch.  As ONE bit test against a gate word that a3 rewrote — round 3 — the gate
                    // travelled through a VGPR phi and the verdict, reused by a3, became a lane mask: 8 scalar instructions at the head of every step)
                    uint32_t m0 = (uint32_t)M; cs::keep_scalar(m0);
                    a2_go = false;
                    if ((int32_t)m0 > 0) a2_go = cs::flag_set(pf, CASIM_REC_A2_OK);
                }
                else if constexpr (Store::kNPT == 1) {
                    // (the dry loop only runs with the gate open.)  The common PEG — template Filters pass, pods > 0, every lane asked for
                    // and small: CASIM_REC_A2_SIMPLE — goes from ONE bit test straight to its three compares; the others test A2_OK
                    // and take the general mask.  A PEG that fits nowhere is done after the mask: 11 scalar + 3 vector instructions.
                    if ((pf & CASIM_REC_A2_SIMPLE) != 0) fb_first = st.fit_mask(0, pv, CASIM_REC_SIMPLE);
                    else if ((pf & CASIM_REC_A2_OK) != 0) fb_first = st.fit_mask(0, pv, 0u);
                    a2_go = fb_first != 0;
                }
                else a2_go = (pf & CASIM_REC_A2_OK) != 0;   // (the dry loop only runs with the gate open: one bit test)
            }
            else a2_go = M > 0 && keff > 0 && static_ok && !group_unschedulable;
            if constexpr (Store::kHasExcl) {
                // (pass 1 of a series whose first pod opened a NEW node: the reference tried the existing nodes for every pod of the PEG before it
                // added that node — tryToScheduleOnExistingNodes runs first, binpacking_estimator.go:136-141 — so the rest reaches it by name, the
                // next-fit on the newest node below, and lastIndex stays where it was)
                if (series_phase == 1 && series_skip_a2) a2_go = false;
                // (a series that also excludes itself per node — a host port: the only node with the bit holds its first pod, which is not
                // remembered by a node bit: nothing more fits anywhere, a3 below still opens the node the reference opens in vain)
                if (series_phase == 1 && selfx) { a2_go = false; on_last = 1u; }
            }
            if (a2_go && !zblocked) {
                // register stores of up to 4 slots walk all of them: slots past M hold zero state (c_j = 0, never a
                // candidate, nothing committed), and a constant bound drops one scalar compare + branch per slot and pass
                const int S = (Store::kNPT > 0 && Store::kNPT <= 4) ? Store::kNPT : (M + 63) >> 6;
                const uint64_t cap1 = (uint64_t)keff + 1;
                // exact wave sum of per-lane values <= cap1: one 32-bit DPP reduction when 64 * cap1 < 2^32
                auto wsum = [&](uint64_t v) -> uint64_t {
                    if (v > cap1) v = cap1;
                    return cap1 <= (1ull << 25) ? (uint64_t)cs::wave_sum_u32((uint32_t)v) : cs::wave_sum_u32_wide((uint32_t)v);
                };
                // pass A: capacities c_j; n1 = nodes that take at least one pod (ballots only, no reduction)
                uint32_t lane_max = 0;
                int32_t n1 = 0;
                uint32_t creg[Store::kNPT > 0 ? Store::kNPT : 1];   // register store: c_j of this PEG, dead after a2
#pragma unroll
                for (int s = 0; s < (Store::kNPT > 0 ? Store::kNPT : 1); ++s) creg[s] = 0;
                uint32_t act = ~0u;   // slots with a non-zero capacity (register store); the memory store walks them all
                auto live = [&](int s) -> bool { return Store::kNPT == 0 || s >= 32 || ((act >> s) & 1u) != 0; };
                auto getc = [&](int s, int m) -> uint32_t {
                    if constexpr (Store::kNPT > 0) return creg[s]; else return st.get_c(s, m);
                };
                // everything after pass A, for n1 > 0 nodes that take a pod (a continuation: the one-slot register store runs
                // it inside the "somebody fits" branch of its only slot instead of re-testing a count after the merge)
                // rotated order starts at list position (lastIndex + 1) % n; positions < E are the pre-existing cluster nodes (never
                // acceptable): first simulated node of the rotated order
                auto rotation_origin = [&]() -> int32_t {
                    const int32_t n = E + M;
                    // (lastIndex + 1 < n unless the caller's lastIndex came from a longer list: the division is the rare path)
                    const uint32_t li1 = (uint32_t)last_index + 1u;
                    const int32_t o = last_index < 0 ? 0 : (int32_t)(li1 < (uint32_t)n ? li1 : li1 % (uint32_t)n);   // (negative: origin 0)
                    return o > E ? o - E : 0;
                };
                auto a2_finish_core = [&](const int32_t new_last) {
                    last_index = new_last;
                    if (Wz > 0) zone_mark(zmark);
                    if constexpr (kDry) {   // placed > 0 here; the tail of the step does not run
                        uint32_t mp = (uint32_t)my_placed; cs::write_lane_u32(mp, (uint32_t)(placed + series_carry), j); my_placed = (int32_t)mp;
                        total_placed += placed;
                        add_totals(placed, pv.req[0], RM > 1 ? pv.req[1] : (L)0);
                    }
                };
                auto a2_finish = [&](const int32_t new_last, const uint32_t x_mine_last) {
                    if constexpr (!kDry) on_last = cs::bcast_u32(x_mine_last, (M - 1) & 63);   // (only a3 asks)
                    a2_finish_core(new_last);
                };
                auto a2_rest = [&](const int32_t n1, const uint64_t fb1 = 0) {
                    CASIM_PROF(2);  // a2 pass A (capacities)
                    if constexpr (Store::kNPT == 1) {
                        // (the count as a 32-bit scalar: as the 64-bit popcount it came from, "n1 == 1" was a VECTOR compare of a scalar pair)
                        uint32_t n1s = (uint32_t)n1; cs::keep_scalar(n1s);
                        // ONE node takes the PEG (57 % of the placing steps of a C2 batch, 2.6 fitting nodes on average over the rest): its capacity
                        // is clamped to the pods of the PEG, so it saturates (tot = cmax = c <= keff), the node is the last one served and
                        // nothing needs a wave reduction — the lane index from the fit mask, the capacity by one v_readlane.
                        if (n1s == 1u) {
                            const int l1 = cs::ffs64(fb1);
                            const uint32_t c1 = cs::bcast_u32(creg[0], l1);
                            placed = (int32_t)c1;
                            st.commit_any(0, creg[0], pv);   // (c == 0 in every other lane: no change there)
                            if constexpr (!kDry) on_last = l1 == ((M - 1) & 63) ? c1 : 0u;
                            a2_finish_core(E + l1);
                            CASIM_PROF(3);
                            return;
                        }
                        // TWO nodes that both saturate (15 % of the placing steps): capacities by two v_readlane, the last node served is the
                        // one with the larger capacity — equal capacities: the later one in rotated order
                        if (n1s == 2u) {
                            const int la = cs::ffs64(fb1), lb = cs::fls64(fb1);
                            const uint32_t ca = cs::bcast_u32(creg[0], la), cb = cs::bcast_u32(creg[0], lb);
                            if (ca + cb <= keff) {   // (both <= keff < 2^31: no wrap)
                                placed = (int32_t)(ca + cb);
                                st.commit_any(0, creg[0], pv);
                                int last;
                                if (ca != cb) last = ca > cb ? la : lb;
                                else { const int32_t m0 = rotation_origin(); last = lb < m0 ? lb : (la < m0 ? la : lb); }
                                if constexpr (!kDry) { const int lm = (M - 1) & 63; on_last = la == lm ? ca : (lb == lm ? cb : 0u); }
                                a2_finish_core(E + last);
                                CASIM_PROF(3);
                                return;
                            }
                        }
                    }
                    uint32_t T, Rr;
                    bool done = false;
                    if ((uint32_t)n1 > keff) {
                        // S(1) = n1 > k: not even one full round — the k pods go to the first k fitting nodes
                        T = 0; Rr = keff; placed = (int32_t)keff;
                    } else {
                        // sum and max of the capacities are only needed when the pods complete >= 1 round.
                        // U = lane-sum type: c_j <= keff, so a register store of <= 16 slots with keff < 2^24 sums in
                        // 32 bits (lane sums < 2^28, clamped wave sums < 2^31) — no 64-bit compares / adds in the lanes.
                        auto rounds = [&](auto tag) {
                            using U = decltype(tag);
                            auto wsumU = [&](U v) -> U {
                                if constexpr (sizeof(U) == 4) return cs::wave_sum_u32(v > (U)cap1 ? (U)cap1 : v);
                                else return (U)wsum((uint64_t)v);
                            };
                            U lsum = 0;
                            for_slots<Store>(S, [&](int s) {
                                if (!live(s)) return;
                                const uint32_t cj = getc(s, s * 64 + lane);
                                lsum += cj;
                                lane_max = cj > lane_max ? cj : lane_max;
                            });
                            U tot; uint32_t cmax;
                            if constexpr (sizeof(U) == 4) {   // sum and max in one interleaved pass (the chains fill each other's DPP wait states)
                                uint32_t t32;
                                cs::wave_sum_max_u32(lsum > (U)cap1 ? (U)cap1 : lsum, lane_max, t32, cmax);
                                tot = (U)t32;
                            } else { tot = wsumU(lsum); cmax = cs::wave_max_u32(lane_max); }
                            if (tot <= (U)keff) {   // every fitting node saturates (28 % of the C2 steps, 2.6 nodes on average)
                                T = cmax; Rr = 0; placed = (int32_t)tot;
                                if constexpr (Store::kNPT > 0) {
                                    // Each node takes its whole capacity, and the last pod of the round-robin lands on the LAST node, in
                                    // rotated order, among those with the largest capacity (they alone are still served in round cmax):
                                    // one lane mask per slot and find-last-bit, no ranks, no second reduction.
                                    const int32_t m0 = rotation_origin();
                                    int32_t last_lo = -1, last_all = -1;   // largest node index below the origin / overall with c == cmax
                                    uint32_t x_mine_last = 0;
                                    for_slots<Store>(S, [&](int s) {
                                        if (!live(s)) return;
                                        const int m = s * 64 + lane;
                                        const uint32_t cj = getc(s, m);
                                        const uint64_t b = cs::ballot(cj == cmax);
                                        if (b) {
                                            last_all = s * 64 + cs::fls64(b);
                                            const uint64_t bl = b & cs::ballot(m < m0);
                                            if (bl) last_lo = s * 64 + cs::fls64(bl);
                                        }
                                        st.commit_any(s, cj, pv);
                                        if (m == M - 1) x_mine_last = cj;
                                    });
                                    a2_finish(E + (last_lo >= 0 ? last_lo : last_all), x_mine_last);
                                    done = true;
                                }
                            } else {
                                uint32_t lo = 1, hi = cmax; U slo = (U)n1;  // S(lo) <= keff < S(hi)
                                while (hi - lo > 1) {
                                    const uint32_t mid = lo + ((hi - lo) >> 1);
                                    U ls = 0;
                                    for_slots<Store>(S, [&](int s) {
                                        if (!live(s)) return;
                                        const uint32_t cj = getc(s, s * 64 + lane);
                                        ls += cj < mid ? cj : mid;
                                    });
                                    const U sm = wsumU(ls);
                                    if (sm <= (U)keff) { lo = mid; slo = sm; } else hi = mid;
                                }
                                T = lo; Rr = keff - (uint32_t)slo; placed = (int32_t)keff;
                            }
                        };
                        if (Store::kNPT > 0 && Store::kNPT <= 16 && keff < (1u << 24)) rounds((uint32_t)0);
                        else rounds((uint64_t)0);
                    }
                    CASIM_PROF(3);  // a2 reductions + bisection
                    if (done) return;
                    const uint32_t Tf = Rr > 0 ? T + 1 : T;  // last round: candidates have c >= Tf
                    const int32_t m0 = rotation_origin();
                    int32_t A = 0, Tot = 0;
                    for_slots<Store>(S, [&](int s) {
                        if (!live(s)) return;
                        const uint64_t bc = cs::ballot(getc(s, s * 64 + lane) >= Tf);
                        Tot += cs::popc64(bc);
                        A += cs::popc64(bc & cs::ballot(s * 64 + lane < m0));   // (a lane compare: the scalar low-mask form cost ~10 SALU per slot)
                    });
                    const int32_t target = Rr > 0 ? (int32_t)Rr - 1 : Tot - 1;
                    int32_t basec = 0, new_last = last_index;
                    uint32_t x_mine_last = 0;
                    for_slots<Store>(S, [&](int s) {
                        if (!live(s)) return;   // c_j == 0 in every lane: no candidate, nothing to commit, x on the newest node 0
                        const int m = s * 64 + lane;
                        const uint32_t cj = getc(s, m);
                        const bool cand = cj >= Tf;
                        const uint64_t b = cs::ballot(cand);
                        const int32_t pex = basec + cs::mbcnt(b);
                        const int32_t rot = m >= m0 ? pex - A : (Tot - A) + pex;
                        uint32_t x = cj < T ? cj : T;
                        if (Rr > 0 && cand && rot < (int32_t)Rr) x += 1;
                        const uint64_t hit = b & cs::ballot(rot == target);
                        if (hit) new_last = E + s * 64 + cs::ffs64(hit);
                        if constexpr (Store::kNPT > 0) st.commit_any(s, x, pv);   // x == 0 changes nothing: no exec-mask branch
                        else { if (x > 0) st.commit(s, m, x, pv); }
                        if (m == M - 1) x_mine_last = x;
                        basec += cs::popc64(b);
                    });
                    a2_finish(new_last, x_mine_last);
                };
                if constexpr (Store::kNPT == 1) {
                    uint64_t fb;
                    if constexpr (kRecScalar && kDry) fb = fb_first; else fb = st.fit_mask(0, pv, pf);
                    if (fb) {  // wave-uniform
                        act = 1u;
                        creg[0] = st.capacity_slot(0, pv, keff, pf, fb);
                        a2_rest(cs::popc64(fb), fb);
                    }
                } else if constexpr (Store::kNPT > 0) {
                    n1 = st.capacity_all(pv, keff, pf, creg, act, S);  // (<= 4 slots: every slot, nodes >= M are all-zero)
                    if (n1 > 0) a2_rest(n1);
                } else {
                    for_slots<Store>(S, [&](int s) {
                        const int m = s * 64 + lane;
                        uint32_t cj = 0;
                        if (m < M) cj = st.capacity(s, m, pv, keff, selfx);
                        st.set_c(s, m, cj);
                        n1 += cs::popc64(cs::ballot(cj > 0));
                    });
                    if (n1 > 0) a2_rest(n1);
                }
            }

            CASIM_PROF(4);  // a2 passes B + C (rotated rank, commit)
            const int32_t placed_a2 = placed;
            // ---- a3 / a4: tryToScheduleOnNewNodes (:190-269) or tryFastPath (:274-324) ----
            int32_t rem = cnt - placed;
            if constexpr (Store::kHasExcl && !kDry) {
                // tryFastPath of a series whose first pod fitted no simulated node (then none of its pods did): the one node it simulates takes
                // the first pod by the exception and every later one finds it there — the whole PEG behaves like one without the bits
                if (series_first && placed == 0 && k == fast_k) { series_first = false; cnt = series_cnt; rem = cnt; }
            }
            if (!kDry && (rem & more_mask) > 0) {   // pods left && newNodesAvailable
                zblocked = Wz > 0 && zone_blocked(zblock);
                // "no node of this group takes the PEG": the template-level Filters fail (CASIM_KFLAG_STATIC_OK clear — tested as a bit
                // where it is needed: a wave-uniform bool costs scalar mask instructions) or the group-wide exclusion state blocks it
                uint32_t zone_stop = zblocked ? 1u : 0u;
                auto blocked = [&]() -> bool { return !cs::flag_set(pf, CASIM_KFLAG_STATIC_OK) || (Store::kHasZone && zone_stop != 0); };
                // capacity of a FRESH node for this PEG
                uint32_t cfresh = 0;
                {
                    bool xb = false;
                    for_words<kRegX>(WxL, [&](int w) { xb |= (fn.ex(w) & pv.block_word(w)) != pv.need_word(w); });   // (a NEED bit the template's own pods do not set: no partner on a fresh node)
                    if (!xb) {
                        uint32_t cf;
                        if constexpr (kRecScalar) cf = cf_rec; else cf = cs::bcast_u32(my_cf, j);
                        cfresh = cf < (uint32_t)rem ? cf : (uint32_t)rem;
                    }
                    if (cs::flag_set(pf, CASIM_PEG_SELF_EXCL_NODE) || zselfx) cfresh = cs::scalar_min_u32(cfresh, 1u);
                }
                // the lane that owns node m writes its fresh state + x pods: node first+i gets
                // min(per, pods_total - i*per) pods
                auto create_nodes = [&](int32_t first, int32_t nadd, uint32_t per, int32_t pods_total) {
                    if constexpr (Store::kNPT == 16) {
                        if (first + nadd > 64 * 16) { overflow = 1u; more_mask = 0; return; }   // (whatever the callers book after this is discarded)
                    }
                    const int s_lo = first >> 6, s_hi = (first + nadd - 1) >> 6;
                    for_slots<Store>(s_hi + 1, [&](int s) {
                        const int32_t m = s * 64 + lane;
                        const uint32_t i = (uint32_t)(m - first);
                        if ((Store::kNPT == 1 || s >= s_lo) && i < (uint32_t)nadd) {   // (one slot: the node bound is <= 64, first is in it)
                            // i * per <= (nadd - 1) * per < pods_total < 2^31 for the nodes being created: 32-bit arithmetic
                            const int32_t left = pods_total - (int32_t)(i * per);
                            const uint32_t x = left <= 0 ? 0u : ((uint32_t)left < per ? (uint32_t)left : per);
                            st.create(s, m, x, pv, fn);
                        }
                    });
                };
                // nodes the limiter would still grant (granted never passes the bound: one subtraction)
                auto permission_left = [&]() -> int64_t { return (int64_t)(grant_bound - granted); };
                bool marked = false;

                if (k == fast_k) {
                    // tryFastPath: one simulated node, the rest by arithmetic
                    if (permission_left() <= 0) more_mask = 0;
                    else {
                        granted++;
                        const uint32_t per = blocked() ? 0u : (cfresh < (uint32_t)rem ? cfresh : (uint32_t)rem);
                        create_nodes(M, 1, per, (int32_t)per);
                        M++;
                        if (per > 0) {
                            marked = true;
                            placed += (int32_t)per;
                            const int32_t size = (int32_t)cs::uniform_div_u32((uint32_t)rem + per - 1u, per);  // scaleUpSize
                            const int64_t left = permission_left();
                            const int32_t want = size - 1;
                            const int32_t nf = want < left ? want : (int32_t)left;
                            const int64_t fp = (int64_t)nf * per;
                            placed += (int32_t)(nf == want ? (int64_t)rem - per : fp);
                            fakes += nf; granted += nf;
                            if (nf < want) more_mask = 0;
                        }
                    }
                } else {
                    // next-fit on the newest node (:198-209).  When a2 ran for this PEG it walked EVERY simulated node, the newest
                    // included, with the same predicate, and left pods over: the newest node is full for this PEG (register store
                    // without group-wide state: a2 ran iff the gate bit was set)
                    // a2_may_have_run: CsFalse where the caller knows a2 did not run for this PEG (then it put nothing on the newest node: on_last is
                    // dead in that instantiation and the packer stops computing it)
                    auto ask_newest = [&](auto a2_may_have_run) {
                        const int lm = M - 1, owner = lm & 63;
                        uint32_t cl = 0;
                        if (!blocked() && !(decltype(a2_may_have_run)::value && selfx && on_last > 0))   // wave-uniform
                            cl = cs::bcast_u32(st.capacity_newest(lm, pv, (uint32_t)rem, selfx || zselfx, lane == owner), owner);
                        if (cl > 0) {
                            st.commit_newest(lm, cl, pv, lane == owner);
                            placed += (int32_t)cl; rem -= (int32_t)cl; marked = true;
                            if (zselfx) zone_stop = 1u;
                        }
                    };
                    // (nested ifs: each is one scalar compare + branch; as one combined bool the tests became lane masks)
                    if constexpr (!(kRecScalar && !Store::kHasZone)) { if (M > 0) ask_newest(CsTrue{}); }
                    // newest node still empty and the pod does not fit it: a new one would not help (:234-236)
                    auto newest_pods = [&]() -> uint32_t {
                        const int lm = M - 1, owner = lm & 63;
                        return cs::bcast_u32((uint32_t)st.npods_newest(lm, lane == owner), owner);
                    };
                    // (without zone state the body runs at most once: a straight line instead of a loop whose
                    // back edge carried the whole register state)
                    auto new_nodes = [&]() -> bool {   // false = done
                        // (records of a store without group-wide state carry 0 pods per empty node when the template-level Filters fail:
                        // order_group — "blocked" is cfresh == 0 there, one test instead of a flag test merged into a lane mask)
                        uint32_t cn;
                        if constexpr (kRecScalar && !Store::kHasZone) cn = cfresh; else cn = blocked() ? 0u : cfresh;
                        if (cn == 0 || zselfx) {
                            if (permission_left() <= 0) { more_mask = 0; return false; }       // :244-246
                            granted++;
                            const uint32_t x = cn < (uint32_t)rem ? cn : (uint32_t)rem;  // 0 or 1
                            create_nodes(M, 1, x, (int32_t)x);
                            M++;
                            if (x == 0) return false;                                   // :257-263 node stays, PEG abandoned
                            placed += (int32_t)x; rem -= (int32_t)x; marked = true;
                            zone_stop = 1u;                                             // zselfx: the group now holds one
                            return Store::kHasZone && rem != 0;
                        } else {
                            // rem <= 2^31 - 1 and cn <= rem: 32-bit unsigned arithmetic is exact (an emulated
                            // 64-bit division here cost more than the whole node creation)
                            // (records carry the pods that fit an empty node in 21 bits: the short form of the division)
                            // need = ceil(rem / cn) <= rem < 2^31, 0 <= left < 2^31, nadd * cn <= need * cn < rem + cn < 2^32: everything in 32 bits
                            // (as int64 the two minima became VECTOR 64-bit compares of scalar values — there is no scalar s_cmp_lt_i64)
                            const int32_t need = (int32_t)(kRecScalar ? cs::uniform_div_u32_small((uint32_t)rem + cn - 1u, cn) : cs::uniform_div_u32((uint32_t)rem + cn - 1u, cn));
                            const int32_t left = grant_bound - granted;
                            const int32_t nadd = need < left ? need : left;
                            const uint32_t fit = (uint32_t)nadd * cn;
                            const int32_t pl = fit < (uint32_t)rem ? (int32_t)fit : rem;
                            if (nadd > 0) {
                                create_nodes(M, nadd, cn, pl);
                                // A run of k identical singleton PEGs merged into this row (CASIM_KFLAG_SINGLETON_RUN, casim_pipeline.h): each
                                // of them runs tryToScheduleOnExistingNodes first, so every pod after the first on a new node got there
                                // through a match that moved lastIndex to that node (plugin_runner.go:138) — the last new node holding >= 2
                                // pods.  (A template that is unschedulable never matches there: its pods arrive by name, as a PEG's do.)
                                if (cs::flag_set(pf, CASIM_KFLAG_SINGLETON_RUN)) { cs::keep_apart(); if (!group_unschedulable && cn >= 2u) {   // (the rare flag first, as a branch of its own)
                                    const int32_t on_last_new = pl - (nadd - 1) * (int32_t)cn;
                                    if (on_last_new >= 2) last_index = E + M + nadd - 1;
                                    else if (nadd >= 2) last_index = E + M + nadd - 2;
                                } }
                                M += nadd; granted += nadd; placed += pl; rem -= pl; marked = true;
                            }
                            if (need > left) more_mask = 0;
                            return false;
                        }
                    };
                    if constexpr (Store::kHasZone) {
                        bool stop = rem == 0;
                        if (!stop && M > 0 && newest_pods() == 0) stop = true;
                        while (!stop && new_nodes()) {}
                    } else if constexpr (kRecScalar) {
                        // ONE test of "a node exists" for both questions (as two tests of M the compiler kept the verdict as a pair of lane masks)
                        uint32_t m1 = (uint32_t)M; cs::keep_scalar(m1);
                        if ((int32_t)m1 > 0) {
                            if (!cs::flag_set(pf, CASIM_REC_A2_OK)) ask_newest(CsFalse{});     // a2 did not run for this PEG
                            if (rem != 0) { if (newest_pods() != 0) new_nodes(); }
                        } else { if (rem != 0) new_nodes(); }   // (no node yet: nothing to compare)
                    } else {
                        if (rem != 0) {
                            uint32_t np = 1u;   // (no node yet: nothing to compare)
                            if (M > 0) np = newest_pods();
                            if (np != 0) new_nodes();
                        }
                    }
                }
                if (marked && Wz > 0) zone_mark(zmark);
            }

            CASIM_PROF(5);  // a3 / a4
            if constexpr (!kDry) {   // (kDry: a2 recorded what it placed, nothing else can place)
                if constexpr (kRecScalar) {
                    uint32_t mp = (uint32_t)my_placed; cs::write_lane_u32(mp, (uint32_t)(placed + series_carry), j); my_placed = (int32_t)mp;
                    add_totals(placed, pv.req[0], RM > 1 ? pv.req[1] : (L)0);
                } else { if (lane == j) my_placed = placed + series_carry; }
                total_placed += placed;
            }
            if constexpr (Store::kHasExcl) {
                // pass 0 of a series placed its pod and the record holds more: the same record again, NEED bits in force
                series_repeat = series_first && placed == 1 && series_cnt > 1;
                series_skip_a2 = placed_a2 == 0;
                series_phase = series_repeat ? 1 : 0;
            }
            }   // (!idle)
        }
        if constexpr (kRecScalar) {   // record k + 1 into the registers record k just left (in flight across the loop edge)
            if (!(Store::kHasExcl && series_repeat)) {
                roff += kRecBytes;
                cur = cs::rec_load<DW>(rbase, roff);
                if (kNeedG) g_cur = (int32_t)cs::const_load<1>((const uint32_t*)res.order + off + k + 1).w[0];
            }
        }
    };
    {
        if constexpr (kRecScalar) {
            // (scalar tests and branches: `roff < rend && more_mask != 0` as one condition became lane-mask arithmetic, 11 scalar
            // instructions of loop overhead per step)
            if (roff < rend) {
                for (;;) {
                    peg_step(0, CsFalse{});
                    if (roff >= rend) break;
                    uint32_t mm = (uint32_t)more_mask; cs::keep_scalar(mm); more_mask = (int32_t)mm;
                    if (mm == 0) break;
                }
            }
            // behind a dry limiter: nothing at all can happen without a node that takes pods (no a3, a2 gated off) — else chunk by
            // chunk, so that the steps themselves carry no chunk test
            constexpr uint32_t kChunkBytes = 64u * kRecBytes;
            if (M > 0 && !group_unschedulable) {   // (a2 can run: a node exists and the template is schedulable)
                while (roff < rend) {
                    if ((roff & (kChunkBytes - 1u)) == 0) { const int k = (int)(roff >> kRecShift); if (k > 0) flush_chunk(k - 64); my_placed = 0; touch_chunk(k + 64); }
                    const uint32_t cnext = (roff | (kChunkBytes - 1u)) + 1u;
                    const uint32_t cend = cnext < rend ? cnext : rend;
                    do peg_step(0, CsTrue{}); while (roff < cend);
                }
            } else {
                // the chunks still have to be flushed (placed[] of the PEGs before the limiter ran dry), nothing else
                int k = (int)(roff >> kRecShift);
                while (k < Gn) {
                    if ((k & 63) == 0) { if (k > 0) flush_chunk(k - 64); my_placed = 0; }
                    k = (k | 63) + 1;
                }
            }
        } else {
            for (int k = 0; k < Gn;) { peg_step(k, CsFalse{}); if (!series_repeat) ++k; }
        }
    }
    if constexpr (kRecScalar) {
        // (the loop's scalars leave it as scalars: with a vector use behind the loop the compiler kept VGPR copies of them
        // and refreshed the copies in every iteration)
        uint32_t a = (uint32_t)M, b = (uint32_t)granted, c = (uint32_t)last_index, d = (uint32_t)total_placed, e = (uint32_t)fakes;
        cs::keep_scalar(a); cs::keep_scalar(b); cs::keep_scalar(c); cs::keep_scalar(d); cs::keep_scalar(e);
        M = (int32_t)a; granted = (int32_t)b; last_index = (int32_t)c; total_placed = (int32_t)d; fakes = (int32_t)e;
    }
    if constexpr (kRecScalar) cs::consume_u32(rec_touch);
    if (Gn > 0) flush_chunk((Gn - 1) & ~63);   // the last (possibly partial) chunk
    int64_t sum0, sum1;
    if constexpr (kRecScalar) { sum0 = (int64_t)cs::bcast_u64((uint64_t)acc0, 0); sum1 = (int64_t)cs::bcast_u64((uint64_t)acc1, 0); }   // (every lane holds the total)
    else { sum0 = (int64_t)cs::wave_sum_u64((uint64_t)acc0); sum1 = (int64_t)cs::wave_sum_u64((uint64_t)acc1); }

    CASIM_PROF(0);
    CASIM_PROF_STORE(prof_out);
    // len(newNodesWithPods) (:160)
    int32_t with_pods = 0;
    for_slots<Store>((M + 63) >> 6, [&](int s) {
        const int m = s * 64 + lane;
        with_pods += cs::popc64(cs::ballot(m < M && st.npods(s, m) > 0));
    });
    if (res.node_pods) {   // newNodesWithPods for the analyser hook: pods of every node this estimate added
        int32_t* np = res.node_pods + res.node_pods_off[ng];
        for_slots<Store>((M + 63) >> 6, [&](int s) {
            const int m = s * 64 + lane;
            if (m < M) np[m] = st.npods(s, m);
        });
    }
    if (lane == 0) {
        res.node_count[ng] = with_pods + fakes;
        res.pods[ng] = total_placed;
        res.nodes_added[ng] = M;
        res.limiter_nodes[ng] = granted;
        res.last_index_out[ng] = last_index;
        res.status[ng] = overflow ? CASIM_NG_RETRY_INTERNAL : CASIM_NG_OK;
        res.cpu_sum[ng] = cs::wrap_madd_i64(0, sum0, sum_scale ? sum_scale[0] : 1);
        res.mem_sum[ng] = cs::wrap_madd_i64(0, sum1, sum_scale ? sum_scale[1] : 1);
    }
}

// a group carrying a PEG outside the encoded predicate subset is delegated (status only)
template <int DW /* dwords of a PEG record, 0 = the three-array form */>
CS_DEVICE bool pack_unsupported(const DevTables& t, const DevResults& res) {
    const int ng = cs::bid(), lane = cs::lane();
    if (t.chain_redo && !t.chain_redo[ng]) return true;   // (a fix-up pass of casim_options.chain_last_index: this group's input did not change)
    const int off = t.peg_off[ng], Gn = t.peg_cnt ? t.peg_cnt[ng] : t.peg_off[ng + 1] - off;
    bool bad = false;
    for (int i = lane; i < Gn; i += 64) bad |= ((DW > 0 ? res.rec[(int64_t)(off + i) * DW + 1] : res.s_flags[off + i]) & CASIM_PEG_UNSUPPORTED) != 0;
    if (!cs::ballot(bad)) return false;
    for (int i = lane; i < Gn; i += 64) res.placed[off + i] = 0;
    if (lane == 0) {
        res.node_count[ng] = 0; res.pods[ng] = 0; res.nodes_added[ng] = 0; res.limiter_nodes[ng] = 0;
        res.last_index_out[ng] = t.last_index[ng]; res.status[ng] = CASIM_NG_UNSUPPORTED;
        res.cpu_sum[ng] = 0; res.mem_sum[ng] = 0;
    }
    return true;
}

// ---- generic kernel: int64 state in LDS (kLds) or an HBM slab ------------------------------------
template <bool kLds, int RMAX_>
CS_GLOBAL CS_LAUNCH_BOUNDS(64, 1) void pack_kernel(DevTables t, DevResults res, PackScratch ps) {
    if (ps.retry_only && res.status[cs::bid()] != CASIM_NG_RETRY_INTERNAL) return;   // (standing by behind the register packer)
    if (pack_unsupported<0>(t, res)) return;
    const int ng = cs::bid();
    const int R = t.R, Wx = t.Wx, Wz = t.Wz;
    MemStore<kLds, RMAX_> st;
    st.R = R; st.Wx = Wx; st.cap = ps.node_cap[ng];
    char* base = kLds ? cs::dyn_smem() : ps.gstate + ps.state_off[ng];
    st.sfree = (int64_t*)base;
    st.sexcl = (uint64_t*)(st.sfree + (int64_t)R * st.cap);
    uint64_t* szone = st.sexcl + (int64_t)Wx * st.cap;
    st.sslots = (int32_t*)(szone + 64 * (Wz > 0 ? Wz : 1));
    st.snpods = st.sslots + st.cap;
    st.sctmp = st.snpods + st.cap;
    FreshNode<int64_t, RMAX_> fn;
    for (int r = 0; r < RMAX_; ++r) fn.free[r] = r < R ? t.alloc[(int64_t)ng * R + r] - t.init_req[(int64_t)ng * R + r] : 0;
    fn.slots = t.allowed[ng] - t.init_pods[ng];
    fn.excl = t.init_excl + (int64_t)ng * Wx;
    const int64_t* s_req = res.s_req;
    pack_body(t, res, st, fn, szone, [=](int idx, int r) -> int64_t { return s_req[(int64_t)idx * R + r]; }, nullptr);
}

// ---- fast kernel: int32 gcd-scaled lanes, node state in VGPRs, no exclusion masks ----------------
// Launched only when the host proved the batch eligible (casim_pipeline.h): either no exclusion state at all (WX_ = 0,
// the bench path) or Wx <= 2 node-local words (two VGPRs each) and Wz <= 2 group-wide words (scalar) with WX_ = 2; R <= R_, every scaled value < 2^31 and every group's node bound <= 64 * NPT_.
// BUILD_: which build of this file the instantiation belongs to — the product library carries the kernels twice (casim_pack_tu.hip:
// 0 = compiled with the experimental structurizer option, 1 = without it), and kernels of two translation units need two names.
template <int R_, int NPT_, int WX_, int BUILD_ = 0>
// launch bounds: a floor of CASIM_FAST_WAVES waves per SIMD for the 256-node instantiation (see the note at the top)
CS_GLOBAL CS_LAUNCH_BOUNDS(64, ((NPT_ == 4 && WX_ == 0) ? CASIM_FAST_WAVES : 1)) void pack_fast_kernel(DevTables t, DevResults res, FastScratch fs) {
    if (pack_unsupported<RegStore<R_, NPT_, WX_>::kRecDw>(t, res)) return;
    const int ng = cs::bid();
    RegStore<R_, NPT_, WX_> st;
#pragma unroll
    for (int s = 0; s < NPT_; ++s) {
#pragma unroll
        for (int r = 0; r < R_; ++r) st.fr[s][r] = 0;
        st.slots[s] = 0;
#pragma unroll
        for (int w = 0; w < WX_; ++w) st.excl[s][w] = 0;
    }
    typename RegStore<R_, NPT_, WX_>::Fresh fn;
#pragma unroll
    for (int r = 0; r < R_; ++r) fn.free[r] = r < t.R ? fs.fresh32[(int64_t)ng * t.R + r] : 0;
    fn.slots = t.allowed[ng] - t.init_pods[ng];
    st.fresh_slots = fn.slots;
    fn.excl = WX_ > 0 ? t.init_excl + (int64_t)ng * t.Wx : nullptr;
    st.wx = t.Wx < WX_ ? t.Wx : WX_;
    pack_body(t, res, st, fn, (uint64_t*)nullptr, [](int, int) -> int32_t { return 0; } /* (requests come with the records) */, fs.scale, fs.prof);
}

// ---- the same on int64 lanes (two of them): no gcd narrowing, requests and free amounts as the boundary carries them -----------------
template <int NPT_, int WX_, int BUILD_ = 0>
CS_GLOBAL CS_LAUNCH_BOUNDS(64, 1) void pack_fast64_kernel(DevTables t, DevResults res, FastScratch fs) {
    if (pack_unsupported<16>(t, res)) return;
    const int ng = cs::bid();
    RegStore<2, NPT_, WX_, int64_t> st;
#pragma unroll
    for (int s = 0; s < NPT_; ++s) {
        st.fr[s][0] = 0; st.fr[s][1] = 0;
        st.slots[s] = 0;
#pragma unroll
        for (int w = 0; w < WX_; ++w) st.excl[s][w] = 0;
    }
    typename RegStore<2, NPT_, WX_, int64_t>::Fresh fn;
#pragma unroll
    for (int r = 0; r < 2; ++r) fn.free[r] = r < t.R ? t.alloc[(int64_t)ng * t.R + r] - t.init_req[(int64_t)ng * t.R + r] : 0;
    fn.slots = t.allowed[ng] - t.init_pods[ng];
    st.fresh_slots = fn.slots;
    fn.excl = WX_ > 0 ? t.init_excl + (int64_t)ng * t.Wx : nullptr;
    st.wx = t.Wx < WX_ ? t.Wx : WX_;
    pack_body(t, res, st, fn, (uint64_t*)nullptr, [](int, int) -> int64_t { return 0; } /* (requests come with the records) */, (const int64_t*)nullptr, fs.prof);
}

}  // namespace casim
