// casim_pipeline.h — host orchestration of one batch (upload, scratch sizing, launch sequence,
// fetch), written once against a small Backend concept:
//
//   struct Backend {
//     void*  alloc(size_t bytes);            // device memory (HBM)
//     void   free(void* p);
//     void   h2d(void* dst, const void* src, size_t bytes);   // async on the backend's stream
//     void   d2h(void* dst, const void* src, size_t bytes);
//     void   zero(void* dst, size_t bytes);
//     void   record_turn_event(); void wait_turn_event(BK& prev); bool turns_enabled();   // the parts of a streamed call taking the link in turn (casim_streams.h)
//     void*  stage(int which, size_t bytes);  // host staging buffer (pinned on the device backend) of >= bytes, owned by the
//                                             // backend and reused by later calls: 0 = uploads, 1 = fetches
//     void   sync();
//     size_t lds_budget() const;             // dynamic LDS bytes one workgroup may ask for
//     template <class K, class... A> void launch(K kernel, int gx, int gy, int block, size_t smem, A... args);
//     void   launch_pack_fast(int lanes, int slots_per_lane, int excl_words, int n_groups, DevTables, DevResults, FastScratch);
//     void   prepare_pack_fast(int build, int lanes, int slots_per_lane, int excl_words);   (before the first launch of an instantiation: init)
//     bool   ok() const;  const char* error() const;
//   };
//
// The product backend (casim_engine.hip) is the HIP runtime on a gfx950 device; tests/emu
// provides a host backend that runs the same kernels under the wave emulator.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <pthread.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "casim_kernels.h"

namespace casim {

// ---- host threads ---------------------------------------------------------------------------------------------------------------------
// A process-wide pool of worker threads, created on first use and never torn down (the threads sleep on a condition variable; the OS takes
// them with the process).  Until round 6 every parallel host loop and every part of a streamed call started threads of its own: ~150 us
// before the four parts of a headline call were running, and per-call spawns made threading the 6 MB columns of a part a loss (0.9 ->
// 1.3 ms), so each part staged its 18 MB on ONE thread — 0.9 of the 3.3 ms of an enter -> return call (profiles/r14c_enter_return_timeline.txt).
// run(n, f) executes f(0) .. f(n - 1), the caller working along with the pool, and returns when all are done.  Tasks are handed out in
// index order and a caller never waits for a task nobody has started (it takes them itself), so nested calls — a part's worker that cuts
// a copy over the pool — and tasks that wait for LOWER-numbered tasks of their own job (the parts' turn order) cannot deadlock.
// CASIM_POOL_THREADS (default 12, 0 = none: every run() is the caller's own loop).
class HostPool {
public:
    static HostPool& get() {
        static std::atomic<HostPool*> inst{nullptr};
        HostPool* p = inst.load(std::memory_order_acquire);
        if (!p) {
            static std::mutex mk;
            std::lock_guard<std::mutex> l(mk);
            p = inst.load(std::memory_order_relaxed);
            if (!p) {
                p = new HostPool();
                inst.store(p, std::memory_order_release);
                // a forked child has none of the threads and maybe a locked mutex: it gets a fresh pool (the old one is left where it is)
                static std::atomic<HostPool*>* slot = &inst;
                pthread_atfork(nullptr, nullptr, [] { slot->store(nullptr, std::memory_order_release); });
            }
        }
        return *p;
    }
    int workers() const { return (int)n_workers_; }
    template <class F>
    void run(int n, F&& f) {
        if (n <= 0) return;
        if (n == 1 || n_workers_ == 0) { for (int i = 0; i < n; ++i) f(i); return; }
        auto job = std::make_shared<Job>();
        job->n = n; job->fn = [&f](int i) { f(i); };
        { std::lock_guard<std::mutex> l(mu_); jobs_.push_back(job); }
        cv_.notify_all();
        work_on(*job);
        // (the tasks still running belong to other threads: wait for them — the function object lives on this stack)
        std::unique_lock<std::mutex> l(job->mu);
        job->cv.wait(l, [&] { return job->done.load(std::memory_order_acquire) >= n; });
    }
private:
    struct Job { std::function<void(int)> fn; int n = 0; std::atomic<int> next{0}, done{0}; std::mutex mu; std::condition_variable cv; };
    static void work_on(Job& j) {
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_acq_rel);
            if (i >= j.n) return;
            j.fn(i);
            if (j.done.fetch_add(1, std::memory_order_acq_rel) + 1 >= j.n) { std::lock_guard<std::mutex> l(j.mu); j.cv.notify_all(); }
        }
    }
    HostPool() {
        const char* e = getenv("CASIM_POOL_THREADS");
        long n = e ? atol(e) : 12;
        const long hw = (long)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw - 1) n = hw - 1;
        if (n < 0) n = 0;
        if (n > 64) n = 64;
        n_workers_ = (size_t)n;
        for (long t = 0; t < n; ++t) std::thread([this] { loop(); }).detach();
    }
    void loop() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> l(mu_);
                for (;;) {
                    while (!jobs_.empty() && jobs_.front()->next.load(std::memory_order_acquire) >= jobs_.front()->n) jobs_.pop_front();   // handed out completely
                    // (a job further back may still have tasks while the front one has none left to give: take the first that has)
                    for (auto& j : jobs_) if (j->next.load(std::memory_order_acquire) < j->n) { job = j; break; }
                    if (job) break;
                    cv_.wait(l);
                }
            }
            work_on(*job);
        }
    }
    std::mutex mu_; std::condition_variable cv_; std::deque<std::shared_ptr<Job>> jobs_; size_t n_workers_ = 0;
};

// Host-side loops over the table columns of a million-PEG batch (the gcd pass, the int32 tables, the copy into pinned staging) are
// cut over a few threads of the pool: an enter -> return call of the headline batch spent 2.1 ms of every part's 4.4 ms of init in them
// (profiles/r05p_init_stages.txt).  f(lo, hi, t) for slice t of at most kHostLoopThreads; short loops run inline.
// CASIM_HOST_THREADS = 1 switches it off.
constexpr int kHostLoopThreads = 4;
inline int host_loop_threads() {
    const char* e = getenv("CASIM_HOST_THREADS");
    const int v = e ? atoi(e) : kHostLoopThreads;
    return v < 1 ? 1 : (v > kHostLoopThreads ? kHostLoopThreads : v);
}
template <class F>
inline void par_for(size_t n, size_t min_per_thread, F f) {
    int T = host_loop_threads();
    if (const char* e = getenv("CASIM_HOST_GRAIN")) { const long v = atol(e); if (v > 0) min_per_thread = (size_t)v; }   // (tests: threads on small tables)
    if (min_per_thread > 0 && n / min_per_thread < (size_t)T) T = (int)(n / min_per_thread);
    if (T <= 1) { f((size_t)0, n, 0); return; }
    HostPool::get().run(T, [&f, n, T](int t) { f(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T, t); });
}
inline void par_memcpy(void* dst, const void* src, size_t bytes) {
    // (slices of >= 1 MiB on the pool's threads: no spawn per call any more — the 6 MB columns of a 1024-simulation part, which copied SLOWER
    // on four freshly started threads, 0.9 -> 1.3 ms per part, are worth cutting now)
    if (bytes < (2u << 20)) { memcpy(dst, src, bytes); return; }
    par_for(bytes, (size_t)1 << 20, [dst, src](size_t lo, size_t hi, int) { memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
}

inline int64_t round_up64(int64_t v) { return (v + 63) & ~63ll; }

// The instantiations of the register packer: lanes R in {2, 4} (int32) or 8 (= two int64 lanes), node slots per lane in {1, 4, 16},
// exclusion words in {0, 2}.
#define CASIM_FAST_DISPATCH(LAUNCH, r, npt, wx)                                                                      \
    do {                                                                                                             \
        if ((r) == 8) { if ((wx) == 2) { if ((npt) == 1) LAUNCH(8, 1, 2); else if ((npt) == 4) LAUNCH(8, 4, 2); else LAUNCH(8, 16, 2); }          \
                        else           { if ((npt) == 1) LAUNCH(8, 1, 0); else if ((npt) == 4) LAUNCH(8, 4, 0); else LAUNCH(8, 16, 0); } }        \
        else if ((r) == 2) { if ((wx) == 2) { if ((npt) == 1) LAUNCH(2, 1, 2); else if ((npt) == 4) LAUNCH(2, 4, 2); else LAUNCH(2, 16, 2); }          \
                        else           { if ((npt) == 1) LAUNCH(2, 1, 0); else if ((npt) == 4) LAUNCH(2, 4, 0); else LAUNCH(2, 16, 0); } }        \
        else          { if ((wx) == 2) { if ((npt) == 1) LAUNCH(4, 1, 2); else if ((npt) == 4) LAUNCH(4, 4, 2); else LAUNCH(4, 16, 2); }          \
                        else           { if ((npt) == 1) LAUNCH(4, 1, 0); else if ((npt) == 4) LAUNCH(4, 4, 0); else LAUNCH(4, 16, 0); } }        \
    } while (0)


// ---- runs of identical singleton PEGs (host) ------------------------------------------------------------------------------------
// Pods without a controller are one PodEquivalenceGroup each (CA/core/scaleup/equivalence/groups.go:69-73): the
// reference's own BenchmarkRunOnceScaleUp hands Estimate 10 000 singleton PEGs of one and the same pod
// (CA/core/bench/benchmark_runonce_test.go:395-418), i.e. 10 000 dependent PEG steps of one lone wave (14.5 ms on the MI355X against
// 7.4 ms of the C restatement on one CPU core, profiles/r05a_bench.json).  Adjacent rows of the PEG table that are identical in every
// column the device reads, hold one pod each and carry no exclusion state are merged on the host into ONE row of k pods before the
// upload: they have equal scores, the orderer keeps equal scores in input order, so they are adjacent in processing order too, and
// scheduling k identical pods one by one is what a PEG of k pods does — tryToScheduleOnExistingNodes is pod by pod anyway
// (binpacking_estimator.go:163-186), and in tryToScheduleOnNewNodes (:190-269) the two differ in ONE observable: a later singleton
// reaches the node its predecessor opened through tryToScheduleOnExistingNodes, whose match moves lastIndex to that node
// (plugin_runner.go:138), where a PEG of k pods fills it with SchedulePod on the node's name and leaves lastIndex alone.  The
// row is flagged CASIM_KFLAG_SINGLETON_RUN and the packer applies that rule; the results are expanded again on the way out
// (placed = 1 for the first `placed` members, 0 for the rest: identical pods tried in order, the ones that find no room are the
// last).  Off with the fastpath (its chooser reads PEG sizes), with batches of simulations (the scan would sit in every
// enter -> return call of a million-PEG batch) and for callers that pass no options (the feasibility entry points index by PEG).
struct SingletonRuns {
    bool active = false;
    std::vector<int32_t> first, len;   // [Gm] first original id and length of merged row m
    casim_pegs p; casim_groups g;      // the merged tables (views into the vectors below)
    std::vector<int64_t> req; std::vector<int32_t> count; std::vector<uint32_t> flags;
    std::vector<uint64_t> tol, sel, xb, xm, zb, zm; std::vector<double> fpc, fpm;
    std::vector<int32_t> lo, hi, off, idx;

    bool build(const casim_pegs* P, const casim_groups* Gp, const casim_options* o) {
        active = false;
        if (!o || o->fastpath || o->no_singleton_merge || o->winners_only || Gp->n_sims > 1) return false;
        const int G = P->n_pegs, NG = Gp->n_groups, R = P->n_res;
        if (G < 2 || !P->count || !P->flags || !P->req) return false;
        bool any = false;
        for (int i = 1; i < G && !any; ++i) any = P->count[i] == 1 && P->count[i - 1] == 1;
        if (!any) return false;
        const int Wt = P->w_taint, Wl = P->w_label, Wx = P->w_excl, Wz = P->w_zone;
        auto zero = [](const uint64_t* m, int64_t at, int w) { for (int k = 0; k < w; ++k) if (m && m[at + k]) return false; return true; };
        auto plain = [&](int i) {
            return P->count[i] == 1 && (P->flags[i] & (CASIM_PEG_SELF_EXCL_NODE | CASIM_PEG_SELF_EXCL_ZONE | CASIM_PEG_UNSUPPORTED | CASIM_KFLAG_SINGLETON_RUN)) == 0 &&
                   zero(P->excl_block, (int64_t)i * Wx, Wx) && zero(P->excl_mark, (int64_t)i * Wx, Wx) && zero(P->zone_block, (int64_t)i * Wz, Wz) && zero(P->zone_mark, (int64_t)i * Wz, Wz);
        };
        auto same = [&](int a, int b) {
            if (P->flags[a] != P->flags[b]) return false;
            for (int r = 0; r < R; ++r) if (P->req[(int64_t)a * R + r] != P->req[(int64_t)b * R + r]) return false;
            for (int k = 0; k < Wt; ++k) if (P->tol_mask[(int64_t)a * Wt + k] != P->tol_mask[(int64_t)b * Wt + k]) return false;
            for (int k = 0; k < Wl; ++k) if (P->sel_mask[(int64_t)a * Wl + k] != P->sel_mask[(int64_t)b * Wl + k]) return false;
            if (P->fp_cpu && P->fp_cpu[a] != P->fp_cpu[b]) return false;
            if (P->fp_mem && P->fp_mem[a] != P->fp_mem[b]) return false;
            return true;
        };
        // a run may not cross the edge of any group's candidate range or of a consecutive stretch of an explicit list
        std::vector<uint8_t> cut((size_t)G + 2, 0);
        if (Gp->peg_offsets && Gp->peg_index) {
            // (offsets that do not run from 0 upwards would walk this scan out of the caller's list: init reports them)
            if (NG > 0 && Gp->peg_offsets[0] != 0) return false;
            for (int i = 0; i < NG; ++i) if (Gp->peg_offsets[i + 1] < Gp->peg_offsets[i]) return false;
            for (int i = 0; i < NG; ++i) {
                const int32_t a = Gp->peg_offsets[i], b = Gp->peg_offsets[i + 1];
                for (int32_t k = a; k < b; ++k) {
                    const int32_t id = Gp->peg_index[k];
                    if (id < 0 || id >= G) return false;   // (init reports it)
                    if (k == a || Gp->peg_index[k - 1] + 1 != id) { cut[(size_t)id] = 1; if (k > a) cut[(size_t)Gp->peg_index[k - 1] + 1] = 1; }
                    if (k == b - 1) cut[(size_t)id + 1] = 1;
                }
            }
        } else if (Gp->peg_lo && Gp->peg_hi) {
            for (int i = 0; i < NG; ++i) {
                if (Gp->peg_lo[i] < 0 || Gp->peg_hi[i] > G || Gp->peg_hi[i] < Gp->peg_lo[i]) return false;
                cut[(size_t)Gp->peg_lo[i]] = 1; cut[(size_t)Gp->peg_hi[i]] = 1;
            }
        }
        std::vector<int32_t> new_of((size_t)G + 1, 0);
        first.clear(); len.clear();
        bool prev_plain = false;
        for (int i = 0; i < G; ++i) {
            const bool pl = plain(i);
            if (i > 0 && pl && prev_plain && !cut[(size_t)i] && same(i, i - 1)) len.back()++;
            else { first.push_back(i); len.push_back(1); }
            new_of[(size_t)i] = (int32_t)first.size() - 1;
            prev_plain = pl;
        }
        const int Gm = (int)first.size();
        new_of[(size_t)G] = Gm;
        if (Gm == G) return false;
        // ---- merged PEG table ----
        auto rows = [&](auto& dst, const auto* src, int w) { dst.clear(); if (!src || w == 0) return; dst.resize((size_t)Gm * (size_t)w); for (int m = 0; m < Gm; ++m) for (int k = 0; k < w; ++k) dst[(size_t)m * w + k] = src[(int64_t)first[(size_t)m] * w + k]; };
        rows(req, P->req, R); rows(count, P->count, 1); rows(flags, P->flags, 1); rows(tol, P->tol_mask, Wt); rows(sel, P->sel_mask, Wl);
        rows(xb, P->excl_block, Wx); rows(xm, P->excl_mark, Wx); rows(zb, P->zone_block, Wz); rows(zm, P->zone_mark, Wz); rows(fpc, P->fp_cpu, 1); rows(fpm, P->fp_mem, 1);
        for (int m = 0; m < Gm; ++m) if (len[(size_t)m] > 1) { count[(size_t)m] = len[(size_t)m]; flags[(size_t)m] |= CASIM_KFLAG_SINGLETON_RUN; }
        p = *P; p.n_pegs = Gm; p.req32 = nullptr; p.req_unit = nullptr;   // (the merged rows carry the int64 requests)
        p.req = req.data(); p.count = count.data(); p.flags = flags.data();
        p.tol_mask = tol.empty() ? nullptr : tol.data(); p.sel_mask = sel.empty() ? nullptr : sel.data();
        p.excl_block = xb.empty() ? nullptr : xb.data(); p.excl_mark = xm.empty() ? nullptr : xm.data();
        p.zone_block = zb.empty() ? nullptr : zb.data(); p.zone_mark = zm.empty() ? nullptr : zm.data();
        p.fp_cpu = fpc.empty() ? nullptr : fpc.data(); p.fp_mem = fpm.empty() ? nullptr : fpm.data();
        // ---- the groups' views of it ----
        g = *Gp;
        if (Gp->peg_offsets && Gp->peg_index) {
            off.assign(1, 0); idx.clear();
            for (int i = 0; i < NG; ++i) {
                for (int32_t k = Gp->peg_offsets[i]; k < Gp->peg_offsets[i + 1]; ++k) {
                    const int32_t m = new_of[(size_t)Gp->peg_index[k]];
                    if (Gp->peg_index[k] == first[(size_t)m]) idx.push_back(m);   // (the cuts made every listed run whole: its head stands for it)
                }
                off.push_back((int32_t)idx.size());
            }
            if (idx.empty()) idx.push_back(0);
            g.peg_offsets = off.data(); g.peg_index = idx.data();
        } else if (Gp->peg_lo && Gp->peg_hi) {
            lo.resize((size_t)NG); hi.resize((size_t)NG);
            for (int i = 0; i < NG; ++i) { lo[(size_t)i] = new_of[(size_t)Gp->peg_lo[i]]; hi[(size_t)i] = new_of[(size_t)Gp->peg_hi[i]]; }
            g.peg_lo = lo.data(); g.peg_hi = hi.data();
        }
        active = true;
        return true;
    }
    // entries of merged lists -> entries of the caller's lists
    int64_t expanded(const int32_t* ids, int64_t n) const { int64_t t = 0; for (int64_t k = 0; k < n; ++k) t += len[(size_t)ids[k]]; return t; }
};

// The parts of a streamed batch upload IN TURN (casim_streams.h).  Four parts that stage and copy at the same time share the link
// equally, so every part's tables arrive at the same late moment and the kernels of all of them start together: upload phase, then
// kernel phase, nothing overlapped (4.2 ms per headline call).  With a turn, part 0's bytes travel first (while it is still staging:
// the early chunks), its kernels run under part 1's copy, and so on: the last part is ready when the link has moved everything
// once, and only ITS kernels are left.  Staging (the host memcpy into pinned memory) stays parallel: it needs no turn.
struct UploadGate {
    std::mutex mu; std::condition_variable cv; int turn = 0;
    bool my_turn(int i) { std::lock_guard<std::mutex> l(mu); return turn >= i; }
    void wait_turn(int i) { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return turn >= i; }); }
    void pass(int i) { { std::lock_guard<std::mutex> l(mu); if (turn == i) turn = i + 1; } cv.notify_all(); }
};

// the calling thread's last chained run (casim_last_chain_info): [0] passes the fixed point is bounded by (groups per simulation - 1),
// [1] passes enqueued, [2] times the host looked at the marks (a wait each), [3] 1 = short chain, enqueued whole
inline int32_t* last_chain_info() { static thread_local int32_t info[4] = {0, 0, 0, 0}; return info; }

template <class BK>
class ProblemT {
public:
    explicit ProblemT(BK& bk) : bk_(bk) {}
    ~ProblemT() { release(); }
    ProblemT(const ProblemT&) = delete;
    ProblemT& operator=(const ProblemT&) = delete;

    // ---- upload -----------------------------------------------------------------------
    int32_t init(const casim_pegs* p, const casim_groups* g, const casim_options* o) {
        if (!p || !g) return fail(CASIM_ERR_INVALID, "null table");
        static const bool timing = getenv("CASIM_INIT_TIMING") != nullptr;
        struct Stage { bool on; const char* what; std::chrono::steady_clock::time_point t0;
                       void mark(const char* next) { if (!on) return; const auto t1 = std::chrono::steady_clock::now();
                           fprintf(stderr, "[init] %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); what = next; t0 = t1; } } stage{timing, "runs + checks", std::chrono::steady_clock::now()};
        // whatever way init() ends, the next part gets its turn — and a FAILED init does not return while a copy out of the caller's own
        // page-locked columns is still in flight (casim.h: "does not return before the copies have finished"; the caller may reuse its
        // tables right after the error return — ADVICE r4)
        struct GatePass { ProblemT* self; ~GatePass() { if (!self->ready_ && self->direct_uploads_ > 0) self->bk_.sync(); self->pass_gate(); } } gate_pass{this};
        if (p->n_pegs < 0 || g->n_groups < 0) return fail(CASIM_ERR_INVALID, "negative size");
        if (p->n_res < 2 || p->n_res > CASIM_KMAX_RES) return fail(CASIM_ERR_INVALID, "n_res must be in [2, 8]");
        if (p->w_taint < 0 || p->w_label < 0 || p->w_excl < 0 || p->w_zone < 0) return fail(CASIM_ERR_INVALID, "negative mask width");
        // the caller's flag words carry CASIM_PEG_* only: the bits above them are the library's own (CASIM_KFLAG_SINGLETON_RUN changes the
        // lastIndex rule of the packer, CASIM_REC_* / CASIM_KFLAG_STATIC_OK are record bits) — a caller that sets one is refused, not obeyed
        if (p->n_pegs > 0 && p->flags) {
            uint32_t acc = 0;
            for (int32_t i = 0; i < p->n_pegs; ++i) acc |= p->flags[i];
            if (acc & ~(uint32_t)CASIM_PEG_FLAG_MASK) return fail(CASIM_ERR_INVALID, "PEG flags: a reserved bit is set (only CASIM_PEG_* bits 0-5 belong to the caller)");
        }
        if (runs_.build(p, g, o)) { p = &runs_.p; g = &runs_.g; }   // adjacent identical singleton PEGs become one row (SingletonRuns)
        G_ = p->n_pegs; NG_ = g->n_groups;
        memset(&dt_, 0, sizeof dt_); memset(&dr_, 0, sizeof dr_); memset(&ps_, 0, sizeof ps_); memset(&os_, 0, sizeof os_);
        dt_.G = G_; dt_.R = p->n_res; dt_.Wt = p->w_taint; dt_.Wl = p->w_label; dt_.Wx = p->w_excl; dt_.Wz = p->w_zone;
        dt_.NG = NG_; dt_.fastpath = o ? o->fastpath : 0;
        winners_only_ = o && o->winners_only != 0; winners_ready_ = false;
        const int R = dt_.R;
        const size_t G = (size_t)G_, NG = (size_t)NG_;
        // casim_pegs.req32 + req_unit (ABI 10): the requests came narrowed by the caller; the int64 table is rebuilt on the device
        const bool narrow_in = G > 0 && p->req32 != nullptr && p->req_unit != nullptr;
        if (G > 0 && ((!p->req && !narrow_in) || !p->count || !p->flags)) return fail(CASIM_ERR_INVALID, "PEG table has null columns");
        if (narrow_in) for (int r = 0; r < R; ++r) if (p->req_unit[r] <= 0) return fail(CASIM_ERR_INVALID, "req_unit must be positive");
        if (NG > 0 && (!g->alloc || !g->init_req || !g->allowed_pods || !g->init_pods || !g->flags || !g->max_nodes ||
                       !g->existing_nodes || !g->last_index))
            return fail(CASIM_ERR_INVALID, "group table has null columns");
        // existing_nodes is a LENGTH (the snapshot's nodes in front of the simulated ones: with the added nodes the modulus of the cyclic
        // search, scheduling_opts.go:54-59) and last_index a list position: a negative length or a position below -1 is a caller's bug, and a
        // modulus of zero or less is not something to hand a kernel
        // Both stay below 2^30, like the bound on the nodes an Estimate can add: E + added nodes and lastIndex + 1 are 32-bit sums in the kernels.
        // Pod counts of a template are counts: negative ones (or an allowed number past 2^30) only make `allowed - init` overflow.
        for (size_t i = 0; i < NG; ++i) {
            // (last_index = -1 is "start at list position 0": lastIndex + 1 is the first position tried; the reference never holds it, the packer's
            // self-check corpus and older callers do)
            if (g->existing_nodes[i] < 0 || g->last_index[i] < -1) return fail(CASIM_ERR_INVALID, "negative existing_nodes / last_index");
            if (g->existing_nodes[i] > 0x3fffffff || g->last_index[i] > 0x3fffffff) return fail(CASIM_ERR_INVALID, "existing_nodes / last_index too large");
            if (g->init_pods[i] < 0 || g->allowed_pods[i] < 0 || g->init_pods[i] > 0x3fffffff || g->allowed_pods[i] > 0x3fffffff)
                return fail(CASIM_ERR_INVALID, "allowed_pods / init_pods out of range");
        }
        if (G > 0 && ((dt_.Wt && !p->tol_mask) || (dt_.Wl && !p->sel_mask) || (dt_.Wx && (!p->excl_block || !p->excl_mark)) ||
                      (dt_.Wz && (!p->zone_block || !p->zone_mark))))
            return fail(CASIM_ERR_INVALID, "PEG mask column missing");
        if (NG > 0 && ((dt_.Wt && !g->taint_mask) || (dt_.Wl && !g->label_mask) || (dt_.Wx && !g->init_excl) || (dt_.Wz && (!g->init_zone || !g->zone_valid))))
            return fail(CASIM_ERR_INVALID, "group mask column missing");
        if (dt_.fastpath && (!p->fp_cpu || !p->fp_mem || !g->cap_cpu || !g->cap_mem))
            return fail(CASIM_ERR_INVALID, "fastpath needs fp_cpu/fp_mem/cap_cpu/cap_mem");

        // ---- ONE host-to-device copy for every table of the batch: the columns (and the small derived arrays below) are
        // packed into the backend's pinned staging buffer and land in one device slab; 30+ separate pageable copies cost
        // 0.17 ms of a 0.5 ms single-simulation call (profiles/r02d_bench.json, configs[].phases_ms.upload_ms)
        {
            const size_t masks = (size_t)(dt_.Wt + dt_.Wl + 2 * dt_.Wx + 2 * dt_.Wz);
            size_t bound = G * (8 * (size_t)R + 8 + 8 * masks + 16 + 4 * (size_t)R) +
                           NG * (16 * (size_t)R + 24 + 8 * masks + 32 + 4 + 8 + 4 + 8 + 8 + 4 * (size_t)R) + 8 * (size_t)R +
                           4 * ((size_t)(g->n_sims > 0 ? g->n_sims : 0) + 1) + 4 * (NG + 1) + 8 * (size_t)dt_.Wz + 64 * 64 + 4096 +
                           (NG <= kFrontMaxGroups ? 8 * NG + 16 : 0);   // (front_kernel's tickets)
            if (g->peg_offsets && NG > 0 && g->peg_offsets[NG] > 0) bound += 4 * (size_t)g->peg_offsets[NG];
            if (getenv("CASIM_TEST_SMALL_UPLOAD_BOUND")) bound = 512;   // (tests: most columns take the fallback copy of their own; the tickets of front_kernel then do not fit and the separate launches run)
            begin_uploads(bound);
        }
        stage.mark("table columns -> staging");
        const int32_t* d_req32_in = nullptr; int64_t* d_req64 = nullptr; int32_t* d_req32_scaled = nullptr;
        if (narrow_in) {
            d_req32_in = up(p->req32, G * R);
            d_req64 = (int64_t*)dalloc(8 * G * (size_t)R);
            if (!d_req32_in || !d_req64) return fail(CASIM_ERR_NOMEM, "no room for the request table");
            dt_.req = d_req64;
        } else dt_.req = up(p->req, G * R);
        dt_.count = up(p->count, G); dt_.pflags = up(p->flags, G);
        dt_.tol = up(p->tol_mask, G * dt_.Wt); dt_.sel = up(p->sel_mask, G * dt_.Wl);
        dt_.xblock = up(p->excl_block, G * dt_.Wx); dt_.xmark = up(p->excl_mark, G * dt_.Wx);
        dt_.zblock = up(p->zone_block, G * dt_.Wz); dt_.zmark = up(p->zone_mark, G * dt_.Wz);
        zpol_host_.assign((size_t)dt_.Wz, 0ull);   // (the kernels always read Wz polarity words: zeros when the caller passed none)
        if (p->zone_polarity) for (int w = 0; w < dt_.Wz; ++w) zpol_host_[(size_t)w] = p->zone_polarity[w];
        dt_.zpol = up(zpol_host_.data(), (size_t)dt_.Wz);
        xpol_host_.assign((size_t)dt_.Wx, 0ull);
        if (p->excl_polarity) for (int w = 0; w < dt_.Wx; ++w) xpol_host_[(size_t)w] = p->excl_polarity[w];
        dt_.xpol = up(xpol_host_.data(), (size_t)dt_.Wx);
        // (the fastpath chooser's columns only travel when the fastpath is on: 16 bytes per PEG, a quarter of a C2 batch's upload)
        dt_.fp_cpu = (dt_.fastpath && p->fp_cpu) ? up(p->fp_cpu, G) : nullptr; dt_.fp_mem = (dt_.fastpath && p->fp_mem) ? up(p->fp_mem, G) : nullptr;
        dt_.alloc = up(g->alloc, NG * R); dt_.init_req = up(g->init_req, NG * R);
        dt_.allowed = up(g->allowed_pods, NG); dt_.init_pods = up(g->init_pods, NG); dt_.gflags = up(g->flags, NG);
        dt_.taint = up(g->taint_mask, NG * dt_.Wt); dt_.label = up(g->label_mask, NG * dt_.Wl);
        dt_.init_excl = up(g->init_excl, NG * dt_.Wx); dt_.init_zone = up(g->init_zone, NG * dt_.Wz); dt_.zone_valid = up(g->zone_valid, NG * dt_.Wz);
        dt_.max_nodes = up(g->max_nodes, NG); dt_.existing = up(g->existing_nodes, NG); dt_.last_index = up(g->last_index, NG);
        dt_.cap_cpu = (dt_.fastpath && g->cap_cpu) ? up(g->cap_cpu, NG) : nullptr; dt_.cap_mem = (dt_.fastpath && g->cap_mem) ? up(g->cap_mem, NG) : nullptr;
        dt_.waste_cpu = g->waste_cpu ? up(g->waste_cpu, NG) : nullptr; dt_.waste_mem = g->waste_mem ? up(g->waste_mem, NG) : nullptr;

        // the parts of a streamed call take the link in turn (CASIM_UPLOAD_FIFO=1): this part's table columns are enqueued NOW, behind the event
        // of the part in front of it, and the next part's follow — nobody holds the turn while it computes its geometry (what is staged later
        // is small and not ordered)
        if (gate_issue_order_ && gate_ && !gate_passed_) {
            stage.mark("wait for the turn");
            gate_->wait_turn(gate_idx_);
            stage.mark("issue the table columns");
            if (up_dev_ && !up_reserved_) issue_uploads();
            pass_gate();
        }
        stage.mark("slabs, ranges, csr buffers");
        // ---- results slab: the per-group scalars and the CSR offsets side by side, so that ONE device-to-host copy fetches them
        {
            const size_t ng = NG > 0 ? NG : 1;
            res_bytes_ = 16 * ng + 24 * ng + 4 * (ng + 1);
            // behind them the answer of ONE expander reduce (winner + count, packed key, key block, surviving set): a call that asks for
            // both gets both with the same copy (best_option_query(q, defer_sync))
            opt_off_ = (res_bytes_ + 7) & ~(size_t)7;
            res_bytes_ = opt_off_ + 16 + 8 + 80 + ((ng + 7) & ~(size_t)7);
            // and, when the lists are short (a single simulation), the PEG order and the pods placed per PEG: the fetch of a call is
            // ONE copy instead of three (each copy is a launch of its own on this runtime)
            {
                int64_t cap = 0;
                if (g->peg_offsets) cap = NG > 0 ? g->peg_offsets[NG] : 0;
                else if (g->peg_lo && g->peg_hi) { for (size_t i = 0; i < NG; ++i) cap += (int64_t)g->peg_hi[i] - g->peg_lo[i]; }
                else cap = (int64_t)NG * G_;
                ord_in_slab_ = cap >= 0 && cap <= 16384 && !winners_only_;   // (winners only: the compacted lists are a copy of their own)
                if (ord_in_slab_) {
                    ord_off_ = (res_bytes_ + 15) & ~(size_t)15;
                    ord_cap_ = (size_t)cap;
                    res_bytes_ = ord_off_ + 4 * (ord_cap_ + 1) + 4 * ord_cap_;
                }
            }
            res_slab_ = (char*)dalloc(res_bytes_);
            dr_.cpu_sum = (int64_t*)res_slab_; dr_.mem_sum = dr_.cpu_sum + ng;
            int32_t* i32 = (int32_t*)(dr_.mem_sum + ng);
            dr_.node_count = i32; dr_.pods = i32 + ng; dr_.nodes_added = i32 + 2 * ng; dr_.limiter_nodes = i32 + 3 * ng;
            dr_.last_index_out = i32 + 4 * ng; dr_.status = i32 + 5 * ng; res_off_ = i32 + 6 * ng;
        }
        // ---- simulations of the batch (expander reduce per simulation) ----
        n_sims_ = 0;
        if (g->n_sims > 0) {
            if (!g->sim_offsets || g->sim_offsets[0] != 0 || g->sim_offsets[g->n_sims] != NG_) return fail(CASIM_ERR_INVALID, "sim_offsets must run from 0 to n_groups");
            for (int32_t i = 0; i < g->n_sims; ++i) {
                if (g->sim_offsets[i + 1] < g->sim_offsets[i]) return fail(CASIM_ERR_INVALID, "sim_offsets not monotone");
                const int32_t len = g->sim_offsets[i + 1] - g->sim_offsets[i];
                max_sim_groups_ = len > max_sim_groups_ ? len : max_sim_groups_;
            }
            n_sims_ = g->n_sims;
            dt_.sim_off = up(g->sim_offsets, (size_t)n_sims_ + 1); dt_.n_sims = n_sims_;
        }
        dt_.global_id = g->global_id ? up(g->global_id, NG) : nullptr;
        // casim_options.chain_last_index: lastIndex handed from group to group inside a simulation (run_pack: fixed-point passes)
        chain_ = o && o->chain_last_index != 0 && NG > 1;
        chain_passes_ = chain_ ? (n_sims_ > 0 ? max_sim_groups_ : NG_) - 1 : 0;
        if (chain_passes_ <= 0) chain_ = false;
        d_chain_redo_ = nullptr; d_chain_marks_ = nullptr;
        if (chain_) {
            d_chain_redo_ = (int32_t*)dalloc(4 * NG);
            d_chain_marks_ = (int32_t*)dalloc(4 * ((size_t)chain_passes_ + 1));
            bk_.zero(d_chain_marks_, 4 * ((size_t)chain_passes_ + 1));
        }
        // ---- schedulable subsets ----
        csr_on_device_ = g->peg_offsets == nullptr;
        dt_.lists_from_feas = csr_on_device_ ? 1 : 0;
        std::vector<int64_t> pods_of_group(NG, 0);  // sum of max(count, 1) over the group's PEGs (node bound)
        std::vector<int32_t> pegs_of_group(NG, 0);
        if (!csr_on_device_) {
            if (NG > 0 && g->peg_offsets[0] != 0) return fail(CASIM_ERR_INVALID, "peg_offsets[0] != 0");
            nnz_cap_ = NG > 0 ? g->peg_offsets[NG] : 0;
            // (the whole offsets column first: peg_index is peg_offsets[NG] entries long, and an offset in the middle that lies past it
            // would walk the scan below out of the caller's list before the next group's offset gave it away)
            for (size_t i = 0; i < NG; ++i) if (g->peg_offsets[i + 1] < g->peg_offsets[i]) return fail(CASIM_ERR_INVALID, "peg_offsets not monotone");
            for (size_t i = 0; i < NG; ++i) {
                const int32_t a = g->peg_offsets[i], b = g->peg_offsets[i + 1];
                pegs_of_group[i] = b - a;
                for (int32_t k = a; k < b; ++k) {
                    const int32_t pg = g->peg_index[k];
                    if (pg < 0 || pg >= G_) return fail(CASIM_ERR_INVALID, "peg_index out of range");
                    if (p->count[pg] < 0) return fail(CASIM_ERR_INVALID, "negative PEG count");
                    pods_of_group[i] += p->count[pg] > 1 ? p->count[pg] : 1;
                }
            }
            h_off_.assign(g->peg_offsets, g->peg_offsets + NG + 1);
            dt_.peg_off = up(g->peg_offsets, NG + 1);
            dt_.peg_idx = up(g->peg_index, (size_t)nnz_cap_);
        } else {
            // candidate range of every group: its simulation's PEGs (peg_lo / peg_hi) or all of them
            if ((g->peg_lo == nullptr) != (g->peg_hi == nullptr)) return fail(CASIM_ERR_INVALID, "peg_lo and peg_hi go together");
            std::vector<int32_t> lo(NG, 0), hi(NG, G_);
            // pods of a group's candidate range = sum of max(count, 1) over it.  Groups of one simulation share their range and ranges of a
            // batch are disjoint or equal, so the sum of the range just seen is remembered; a prefix-sum table over all G PEGs (3.3 MB
            // written and read back per part of a headline call) is only built when ranges overlap in another way
            std::vector<int64_t> pre;
            bool neg = false;
            auto range_sum = [&](int32_t a, int32_t b) { int64_t s = 0; for (int32_t k = a; k < b; ++k) { const int32_t c = p->count[k]; neg = neg || c < 0; s += c > 1 ? c : 1; } return s; };
            int64_t cap = 0; int32_t lmax = 0;
            int32_t seen_lo = -1, seen_hi = -1; int64_t seen_sum = 0; int64_t walked = 0;
            for (size_t i = 0; i < NG; ++i) {
                if (g->peg_lo) { lo[i] = g->peg_lo[i]; hi[i] = g->peg_hi[i]; }
                if (lo[i] < 0 || hi[i] < lo[i] || hi[i] > G_) return fail(CASIM_ERR_INVALID, "peg_lo / peg_hi out of range");
                pegs_of_group[i] = hi[i] - lo[i];
                if (lo[i] != seen_lo || hi[i] != seen_hi) {
                    if (pre.empty() && walked + pegs_of_group[i] > (int64_t)G + 4096) {   // ranges that overlap without being equal: the table after all
                        pre.assign(G + 1, 0);
                        for (size_t k = 0; k < G; ++k) { neg = neg || p->count[k] < 0; pre[k + 1] = pre[k] + (p->count[k] > 1 ? p->count[k] : 1); }
                    }
                    seen_lo = lo[i]; seen_hi = hi[i];
                    if (!pre.empty()) seen_sum = pre[(size_t)hi[i]] - pre[(size_t)lo[i]];
                    else { seen_sum = range_sum(lo[i], hi[i]); walked += pegs_of_group[i]; }
                }
                pods_of_group[i] = seen_sum;
                cap += pegs_of_group[i]; lmax = pegs_of_group[i] > lmax ? pegs_of_group[i] : lmax;
            }
            // (a PEG no group's range covers is never read by a kernel of this problem; the check of every count stays with the explicit lists
            // and with the request passes further down, which walk all G rows)
            if (!neg && walked < (int64_t)G && pre.empty()) for (size_t k = 0; k < G && !neg; ++k) neg = p->count[k] < 0;
            if (neg) return fail(CASIM_ERR_INVALID, "negative PEG count");
            if (cap > 0x7fffffffll) return fail(CASIM_ERR_INVALID, "sum of candidate PEG ranges too large for device-side CSR");
            nnz_cap_ = (int32_t)cap; feas_len_ = lmax;
            // simulation-major feasibility kernel: every group of a simulation shares its PEG range, one word per mask kind
            // Exclusion words of a FRESH node only speak through what the template already holds: (block & init_excl) != (block & polarity)
            // (fits_fresh_node).  A batch whose templates carry no marked pods and whose dictionaries have no NEED bits — pod anti-affinity
            // between PENDING pods only: BASELINE config C4, 240 hostname bits = 4 words — has a vacuous exclusion term: its cells are decided
            // by requests, taints and selectors alone, and the simulation-major kernels (one word per mask kind) run it with Wx = 0.
            // (round 6: the batched C4 row took the dense feas_kernel + scan + fill + order_kernel instead, 35 % of its step)
            excl_vacuous_ = false;
            if (n_sims_ > 0 && dt_.Wx > 1 && g->init_excl && !getenv("CASIM_NO_VACUOUS_EXCL")) {
                bool any = false;
                for (int w = 0; w < dt_.Wx && !any; ++w) any = xpol_host_[(size_t)w] != 0;
                const size_t nw = NG * (size_t)dt_.Wx;
                for (size_t i = 0; i < nw && !any; ++i) any = g->init_excl[i] != 0;
                excl_vacuous_ = !any;
            }
            feas_by_sim_ = n_sims_ > 0 && dt_.Wt <= 1 && dt_.Wl <= 1 && (dt_.Wx <= 1 || excl_vacuous_) && dt_.Wz <= 1 &&
                           (size_t)128 * (size_t)max_sim_groups_ <= 48 * 1024;   // (the kernel stages a simulation's group records in LDS)
            for (int32_t si = 0; si < n_sims_ && feas_by_sim_; ++si)
                for (int32_t i = g->sim_offsets[si] + 1; i < g->sim_offsets[si + 1]; ++i)
                    if (lo[(size_t)i] != lo[(size_t)g->sim_offsets[si]] || hi[(size_t)i] != hi[(size_t)g->sim_offsets[si]]) { feas_by_sim_ = false; break; }
            Wg_ = (lmax + 63) / 64;
            dt_.peg_lo = up(lo.data(), NG); dt_.peg_hi = up(hi.data(), NG);   // (copied into the staging buffer: lo / hi may die)
            d_bits_ = (uint64_t*)dalloc(sizeof(uint64_t) * NG * (size_t)(Wg_ > 0 ? Wg_ : 1));
            d_counts_ = (int32_t*)dalloc(sizeof(int32_t) * (NG + 1));
            d_off_ = res_off_;   // inside the results slab: one copy brings the scalars and the offsets back
            d_block_sums_ = (int32_t*)dalloc(sizeof(int32_t) * (NG / 64 + 2));
            d_off_local_ = (int32_t*)dalloc(sizeof(int32_t) * (NG + 1));
            d_idx_ = (int32_t*)dalloc(sizeof(int32_t) * (size_t)nnz_cap_);
            dt_.peg_off = d_off_; dt_.peg_idx = d_idx_;
            // a single call's chain in one launch (front_kernel): a few hundred groups at most — the blocks wait for each other's counts
            // (a batch of >= 2 simulations takes front_sim_kernel further down: no tickets at all)
            const bool want_strided = feas_by_sim_ && n_sims_ >= 2 && feas_len_ > 0 && feas_len_ <= 1024 && !(o && o->no_front_kernel) && !getenv("CASIM_NO_STRIDED");
            front_ = !want_strided && NG > 0 && NG <= kFrontMaxGroups && feas_len_ > 0 && Wg_ <= 512 && !(o && o->no_front_kernel) && !getenv("CASIM_NO_FRONT");
            if (front_) {
                const uint64_t* dev = nullptr;
                uint64_t* h = up_reserve<uint64_t>(NG, &dev);
                if (h) { memset(h, 0, 8 * NG); d_ticket_ = (uint64_t*)dev; } else front_ = false;
            }
            // batches: fixed-stride lists and ONE launch in front of the packer (front_sim_kernel) — group i owns [soff[i], soff[i] + (hi - lo))
            // of order / placed / records, the bound those arrays are sized for; its length is written to the slab's offset area (peg_cnt)
            strided_ = want_strided;
            strided_one_launch_ = getenv("CASIM_FRONT_SIM") != nullptr && atoi(getenv("CASIM_FRONT_SIM")) != 0 && !excl_vacuous_;   // (the one-launch form: measured slower in the loop)
            // rank once per (simulation, allocatable pair) instead of a sort per group (casim_kernels.h rank_shapes_kernel): when the candidate
            // ranges are long and a pair serves several groups — C3: 64 groups, 5 pairs, 1000 PEGs.  CASIM_RANK_ONCE=0 / 1: never / whenever possible
            rank_once_ = false;
            if (strided_ && !dt_.fastpath && R >= 2) {
                const char* e = getenv("CASIM_RANK_ONCE");
                const int mode = e ? atoi(e) : -1;
                if (mode != 0) {
                    std::vector<int32_t> pair_of(NG), rep;
                    bool few = true;
                    for (int32_t si = 0; si < n_sims_ && few; ++si) {
                        const size_t first_pair = rep.size();
                        for (int32_t i = g->sim_offsets[si]; i < g->sim_offsets[si + 1]; ++i) {
                            const int64_t a0 = g->alloc[(size_t)i * R], a1 = g->alloc[(size_t)i * R + 1];
                            size_t k = first_pair;
                            for (; k < rep.size(); ++k) if (g->alloc[(size_t)rep[k] * R] == a0 && g->alloc[(size_t)rep[k] * R + 1] == a1) break;
                            if (k == rep.size()) { rep.push_back(i); if (rep.size() - first_pair > 32) { few = false; break; } }
                            pair_of[(size_t)i] = (int32_t)k;
                        }
                    }
                    // pairs proportional to their simulation's first one (k x (cpu, memory): instance families) share ITS ranking whenever a linear
                    // check on the device passes (rank_shapes_kernel, second launch): one sort per simulation instead of one per pair
                    std::vector<int32_t> base_of(rep.size()), sorted_list, checked_list;
                    bool all_prop = few && !rep.empty();
                    if (few) {
                        size_t pi = 0;
                        for (int32_t si = 0; si < n_sims_; ++si) {
                            if (g->sim_offsets[si + 1] == g->sim_offsets[si]) continue;
                            const size_t first = pi;
                            while (pi < rep.size() && rep[pi] < g->sim_offsets[si + 1]) {
                                const __int128 a0 = g->alloc[(size_t)rep[first] * R], a1 = g->alloc[(size_t)rep[first] * R + 1];
                                const __int128 b0 = g->alloc[(size_t)rep[pi] * R], b1 = g->alloc[(size_t)rep[pi] * R + 1];
                                // (off by default: one sort per simulation plus checks is less WORK than a sort per pair, but the sorts of all pairs run side by
                                // side in one launch while the checks wait for the base pairs' launch — C3 x 512: 0.69 ms per step shared, 0.66 every pair
                                // sorting, profiles/r12d; CASIM_RANK_SHARE=1 turns it on)
                                const bool share = getenv("CASIM_RANK_SHARE") && atoi(getenv("CASIM_RANK_SHARE")) != 0;
                                const bool prop = pi == first || (share && a0 > 0 && a1 > 0 && b0 > 0 && b1 > 0 && a0 * b1 == a1 * b0);
                                base_of[pi] = (int32_t)(prop ? first : pi);
                                if (pi == first || !prop) sorted_list.push_back((int32_t)pi); else checked_list.push_back((int32_t)pi);
                                all_prop = all_prop && prop;
                                ++pi;
                            }
                        }
                    }
                    // (the headline's shape — 20 groups, 5 proportional pairs, ~110 of 400 PEGs per list — gains nothing even with ONE sort per simulation:
                    // 0.869 ms per step against 0.865 with the per-group register networks, profiles/r12c; the rule stays with long ranges)
                    (void)all_prop;
                    const bool pays = lmax >= 512 && NG >= 3 * rep.size();
                    if (few && !rep.empty() && (mode > 0 || pays)) {
                        rank_once_ = true;
                        n_pairs_ = (int32_t)rep.size();
                        n_sorted_pairs_ = (int32_t)sorted_list.size(); n_checked_pairs_ = (int32_t)checked_list.size();
                        d_pair_base_ = up(base_of.data(), base_of.size());
                        d_sorted_list_ = up(sorted_list.data(), sorted_list.size());
                        d_checked_list_ = checked_list.empty() ? nullptr : up(checked_list.data(), checked_list.size());
                        d_pair_src_ = (int32_t*)dalloc(4 * rep.size());
                        rank_stride_ = (int32_t)round_up64(lmax);
                        int64_t npad = 64; while (npad < lmax) npad <<= 1;
                        rank_smem_ = (size_t)npad * 12;
                        d_pair_of_ = up(pair_of.data(), NG); d_pair_rep_ = up(rep.data(), rep.size());
                        d_ranks_ = (int32_t*)dalloc(4 * (size_t)n_pairs_ * (size_t)rank_stride_);
                        if (!d_pair_of_ || !d_pair_rep_ || !d_ranks_ || !d_pair_base_ || !d_sorted_list_ || !d_pair_src_ || rank_smem_ > bk_.lds_budget()) rank_once_ = false;
                    }
                }
            }
            if (strided_) {
                h_off_static_.assign(NG + 1, 0);
                for (size_t i = 0; i < NG; ++i) h_off_static_[i + 1] = h_off_static_[i] + pegs_of_group[i];
                dt_.peg_off = up(h_off_static_.data(), NG + 1);
                dt_.peg_cnt = res_off_;
                const size_t hdr = ((size_t)max_sim_groups_ * 4 + 15) & ~(size_t)15;
                const int nwv = Wg_;
                const size_t phase_ab = (size_t)max_sim_groups_ * 16 * 8 + (size_t)max_sim_groups_ * (size_t)Wg_ * 8 + (size_t)nwv * front_sim_wave_scratch();
                int64_t npad = 1; while (npad < lmax) npad <<= 1;
                const size_t phase_c = lmax > 256 ? (size_t)npad * 16 + 8 + 8 * (size_t)(64 * nwv) : 0;
                front_sim_smem_ = hdr + (phase_ab > phase_c ? phase_ab : phase_c) + 64;
                if (strided_one_launch_ && front_sim_smem_ > bk_.lds_budget()) strided_one_launch_ = false;
            }
        }

        stage.mark("packer geometry");
        // ---- packer state geometry ----
        std::vector<int32_t> cap(NG);
        std::vector<int64_t> soff(NG);
        int64_t total = 0, worst = 0;
        for (size_t i = 0; i < NG; ++i) {
            int64_t n = g->max_nodes[i] > 0 ? (int64_t)g->max_nodes[i] : (g->max_nodes[i] < 0 ? 0 : pods_of_group[i]);
            if (g->max_nodes[i] > 0 && pods_of_group[i] < n) n = pods_of_group[i];  // never more nodes than pods (+ empty ones)
            n = round_up64(n > 0 ? n : 1);  // both terms bound the nodes ever added (limiter grants / one node per pod)
            if (n > 0x3fffffffll) return fail(CASIM_ERR_INVALID, "node bound too large");
            cap[i] = (int32_t)n;
            const int64_t bytes = casim_pack_state_bytes(R, dt_.Wx, dt_.Wz, n);
            soff[i] = total; total += (bytes + 255) & ~255ll;
            worst = bytes > worst ? bytes : worst;
        }
        pack_smem_ = (size_t)worst;
        pack_lds_ = worst <= (int64_t)bk_.lds_budget();
        // ---- fast packer eligibility: no exclusion masks, <= 4 lanes, <= 1024 nodes per group and every
        // lane value representable as int32 after dividing the lane by the gcd of all its values ----
        stage.mark("gcd scaling + int32 tables");
        fast_npt_ = 0;
        {
            int32_t maxcap = 0;
            for (size_t i = 0; i < NG; ++i) maxcap = cap[i] > maxcap ? cap[i] : maxcap;
            bool ok = o ? o->force_generic_packer != 1 : true;
            // force_generic_packer == 2 (tests, the bench's int64 row): skip the int32 narrowing, take the int64 register store when it applies
            const bool want_i64 = o && o->force_generic_packer == 2;
            fast_i64_ = false;
            pack_build_ = o ? o->pack_build : 0;
            // groups whose node BOUND exceeds the 1024 register slots still start in the register packer: the bound (limiter cap or
            // pods) is rarely reached — BenchmarkRunOnceScaleUp: bound 10 000, 200 nodes created — and a group that does run out
            // is packed again by the generic packer's retry launch (pack_kernel, PackScratch::retry_only)
            fast_retry_ = maxcap > 64 * 16;
            ok = ok && dt_.Wx <= 2 && dt_.Wz <= 2 && R <= 4 && NG > 0;
            // (the PEG record carries the pods that fit an empty node in 21 bits: casim_types.h)
            for (size_t i = 0; i < NG && ok; ++i) ok = (int64_t)g->allowed_pods[i] - (int64_t)g->init_pods[i] <= CASIM_REC_FRESH_MAX;
            bool zone_self = false;   // a PEG that excludes itself group-wide (anti-affinity on a non-hostname key)
            for (size_t i = 0; i < G; ++i) zone_self = zone_self || (p->flags[i] & CASIM_PEG_SELF_EXCL_ZONE) != 0;
            fast_wx_ = (dt_.Wx > 0 || dt_.Wz > 0 || zone_self) ? 2 : 0;   // lean instantiation, or the one with room for both kinds of words
            std::vector<int64_t> scale((size_t)R, 0);
            std::vector<char> lane_has_request((size_t)R, 0);   // (narrow_in: a lane without a single request has no unit to honour)
            auto gcd64 = [](int64_t a, int64_t b) { if (a < 0) a = -a; if (b < 0) b = -b; while (b) { const int64_t x = a % b; a = b; b = x; } return a; };
            // (a million-PEG batch walks these loops in every enter -> return call: 5.3 of 11.8 ms as three passes of 64-bit divisions,
            // profiles/r05p_init_stages.txt.  One modulo per value while the gcd settles — it is almost always final after a few rows —
            // no division in the range check, and the exact quotient by a multiply with the modular inverse of the gcd's odd part)
            auto fold = [&](int64_t& sc, int64_t v) {
                if (sc == 1 || v == 0) return;
                if (sc == 0) { sc = v < 0 ? -v : v; return; }
                // (byte-granular lanes settle on a power of two — MiB multiples — : a mask; milli-cpu lanes fit 32 bits: the short division)
                int64_t m;
                if ((sc & (sc - 1)) == 0) m = v & (sc - 1);
                else if (v >= 0 && v <= 0xffffffffll && sc <= 0xffffffffll) m = (int64_t)((uint32_t)v % (uint32_t)sc);
                else m = v % sc;
                if (m != 0) sc = gcd64(sc, m < 0 ? -m : m);
            };
            // (device form of the pass over the request table: tables of >= kDevGcdMin values whose staged columns may go out now — nothing
            // reserved for later writes — and whose lanes fit the kernel's four; CASIM_DEV_GCD_MIN: tests run it on small tables)
            const char* dgm = getenv("CASIM_DEV_GCD_MIN");
            const long dev_gcd_min = dgm ? atol(dgm) : (long)kDevGcdMin;
            const bool dev_gcd = ok && !narrow_in && !want_i64 && R <= 4 && (long)(G * (size_t)R) >= dev_gcd_min && dt_.req != nullptr && !up_reserved_;
            if (ok) {
                // one pass over the columns, cut over the host threads: per lane the gcd, the largest magnitude (of a request, an
                // allocatable, a preloaded amount, a fresh node's free amount) and "some request is negative"
                struct Part { int64_t sc[CASIM_MAX_RES], amax[CASIM_MAX_RES]; bool neg; };
                Part parts[kHostLoopThreads];
                for (auto& pt : parts) { for (int r = 0; r < CASIM_MAX_RES; ++r) pt.sc[r] = pt.amax[r] = 0; pt.neg = false; }
                auto mag = [](int64_t v) -> int64_t { return v < 0 ? (v == INT64_MIN ? INT64_MAX : -v) : v; };
                // (requests narrowed by the caller: the unit is a common divisor by definition — one pass of compares for sign and magnitude)
                if (narrow_in) par_for(G, 262144, [&](size_t lo, size_t hi, int t) {   // (a part of 1024 C2 simulations: one thread, no spawn)
                    // (smallest and largest value per lane: compares only — a division per value made this pass three times the device
                    // gcd pass it replaces, profiles/r11h)
                    int32_t vmin[CASIM_MAX_RES], vmax[CASIM_MAX_RES];
                    for (int r = 0; r < R; ++r) { vmin[r] = 0; vmax[r] = 0; }
                    if (R == 2) {
                        int32_t a0 = 0, b0 = 0, a1 = 0, b1 = 0;
                        const int32_t* q = p->req32 + lo * 2;
                        for (size_t i = 0, n = hi - lo; i < n; ++i) {
                            const int32_t x = q[2 * i], y = q[2 * i + 1];
                            a0 = x < a0 ? x : a0; b0 = x > b0 ? x : b0; a1 = y < a1 ? y : a1; b1 = y > b1 ? y : b1;
                        }
                        vmin[0] = a0; vmax[0] = b0; vmin[1] = a1; vmax[1] = b1;
                    } else {
                        for (size_t i = lo; i < hi; ++i) for (int r = 0; r < R; ++r) {
                            const int32_t v = p->req32[i * R + r];
                            vmin[r] = v < vmin[r] ? v : vmin[r]; vmax[r] = v > vmax[r] ? v : vmax[r];
                        }
                    }
                    Part& pt = parts[t];
                    for (int r = 0; r < R; ++r) {
                        if (vmin[r] == 0 && vmax[r] == 0) continue;
                        const int64_t u = p->req_unit[r], a = -(int64_t)vmin[r] > (int64_t)vmax[r] ? -(int64_t)vmin[r] : (int64_t)vmax[r];
                        pt.sc[r] = u; pt.neg = pt.neg || vmin[r] < 0;
                        const int64_t m = a > INT64_MAX / u ? INT64_MAX : a * u;
                        if (m > pt.amax[r]) pt.amax[r] = m;
                    }
                });
                else if (!dev_gcd) par_for(G, 65536, [&](size_t lo, size_t hi, int t) {
                    Part& pt = parts[t];
                    for (size_t i = lo; i < hi; ++i) for (int r = 0; r < R; ++r) {
                        const int64_t v = p->req[i * R + r];
                        fold(pt.sc[r], v); pt.neg = pt.neg || v < 0; if (mag(v) > pt.amax[r]) pt.amax[r] = mag(v);
                    }
                });
                // big tables: the fold over the request table runs on the DEVICE, where the table is anyway (gcd_reduce_kernel: one partial per
                // block; the quotients by scale_requests_kernel further down) — the host pass above was skipped
                if (dev_gcd) {
                    flush_uploads_all();
                    if (gate_issue_order_) pass_gate();   // the bulk of this part's tables is enqueued: the next part's wait for the event behind them
                    const int nb = (int)(G / 2048 < 1 ? 1 : (G / 2048 > 1024 ? 1024 : G / 2048));
                    GcdPartial* d_part = (GcdPartial*)dalloc(sizeof(GcdPartial) * (size_t)nb);
                    GcdPartial* h_part = (GcdPartial*)bk_.stage(1, sizeof(GcdPartial) * (size_t)nb);
                    if (!d_part || !h_part) return fail(CASIM_ERR_NOMEM, "no room for the gcd partials");
                    bk_.launch(gcd_reduce_kernel, nb, 1, 256, (size_t)(8 * 9 * 4), (const int64_t*)dt_.req, (int64_t)G, R, d_part);
                    bk_.d2h(h_part, d_part, sizeof(GcdPartial) * (size_t)nb);
                    bk_.sync();
                    pass_gate();   // the bulk of this part's tables is on the device: the next part's turn on the link
                    if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
                    Part& pt = parts[0];
                    for (int b = 0; b < nb; ++b) {
                        for (int r = 0; r < R; ++r) {
                            if (h_part[b].sc[r] != 0) pt.sc[r] = pt.sc[r] == 0 ? (int64_t)h_part[b].sc[r] : gcd64(pt.sc[r], (int64_t)h_part[b].sc[r]);
                            if ((int64_t)h_part[b].amax[r] > pt.amax[r]) pt.amax[r] = (int64_t)h_part[b].amax[r];
                        }
                        pt.neg = pt.neg || h_part[b].neg != 0;
                    }
                }
                Part grp; for (int r = 0; r < CASIM_MAX_RES; ++r) grp.sc[r] = grp.amax[r] = 0; grp.neg = false;
                for (size_t i = 0; i < NG; ++i) for (int r = 0; r < R; ++r) {
                    const int64_t a = g->alloc[i * R + r], b = g->init_req[i * R + r];
                    fold(grp.sc[r], a); fold(grp.sc[r], b);
                    if (a == INT64_MIN || b == INT64_MIN) grp.neg = true;   // (no int32 image either way)
                    int64_t d; const bool ovf = __builtin_sub_overflow(a, b, &d);
                    const int64_t m3 = ovf ? INT64_MAX : mag(d), m2 = mag(a) > mag(b) ? mag(a) : mag(b);
                    const int64_t mm = m3 > m2 ? m3 : m2;
                    if (mm > grp.amax[r]) grp.amax[r] = mm;
                }
                bool neg = false;
                std::vector<int64_t> amax((size_t)R, 0);
                for (int r = 0; r < R; ++r) {
                    int64_t sc = grp.sc[r], am = grp.amax[r];
                    for (auto& pt : parts) { if (pt.sc[r] != 0) { sc = sc == 0 ? pt.sc[r] : gcd64(sc, pt.sc[r]); lane_has_request[(size_t)r] = 1; } if (pt.amax[r] > am) am = pt.amax[r]; }
                    scale[(size_t)r] = sc == 0 ? 1 : sc; amax[(size_t)r] = am;
                }
                neg = grp.neg;
                for (auto& pt : parts) neg = neg || pt.neg;
                // |v / scale| <= 2^31 - 1  <=>  |v| <= (2^31 - 1) * scale   (the product saturates: a scale beyond 2^32 admits every int64)
                bool fits32 = true, fits62 = true;
                for (int r = 0; r < R; ++r) {
                    const int64_t vmax = scale[(size_t)r] > (0x7fffffffffffffffll / 0x7fffffffll) ? 0x7fffffffffffffffll : 0x7fffffffll * scale[(size_t)r];
                    fits32 = fits32 && amax[(size_t)r] <= vmax;
                    fits62 = fits62 && amax[(size_t)r] < (1ll << 62);
                }
                // lanes that do not narrow (byte-granular co-prime amounts beyond 2^31 after the gcd): the same register store on int64 lanes,
                // two of them (RegStore<2, NPT, WX, int64_t>) — requests as they came, no scaling.  More lanes, negative requests or amounts
                // beyond 2^62: the LDS store's generic packer.
                fast_i64_ = !neg && R <= 2 && fits62 && (want_i64 || !fits32);
                ok = ok && !neg && (fits32 || fast_i64_);
            }
            if (ok && fast_i64_) {
                fs_.req32 = nullptr; fs_.fresh32 = nullptr; fs_.scale = nullptr;
                if (getenv("CASIM_PACK_PROF_DUMP")) { fs_.prof = (int64_t*)dalloc(8 * 8 * NG); bk_.zero(fs_.prof, 8 * 8 * NG); }
                fast_npt_ = maxcap <= 64 ? 1 : (maxcap <= 256 ? 4 : 16);
                fast_r_ = 2;
            } else if (ok) {
                // exact division by scale = 2^tz * odd: shift, then multiply by the inverse of `odd` modulo 2^64 (Newton: 5 steps)
                std::vector<uint64_t> inv((size_t)R); std::vector<int> tz((size_t)R);
                for (int r = 0; r < R; ++r) {
                    uint64_t sc = (uint64_t)scale[(size_t)r]; int z = 0;
                    while ((sc & 1ull) == 0) { sc >>= 1; ++z; }
                    uint64_t x = sc;                       // 3 correct bits
                    for (int it = 0; it < 5; ++it) x *= 2ull - sc * x;
                    inv[(size_t)r] = x; tz[(size_t)r] = z;
                }
                auto quot = [&](int64_t v, int r) -> int32_t { return (int32_t)(int64_t)((uint64_t)(v >> tz[(size_t)r]) * inv[(size_t)r]); };   // (v is a multiple of scale: the arithmetic shift is exact)
                // (the big table is written straight into the staging buffer)
                std::vector<int32_t> req32_own, fresh32(NG * (size_t)R);
                const int32_t* req32_dev = nullptr;
                if (narrow_in) {
                    // the caller's table IS the packer's when no node-group amount forces a finer scale than its unit (a lane without a single
                    // request keeps whatever is there: zeros); else req32 * (unit / scale), by the kernel that rebuilds the int64 table
                    bool same = true;
                    for (int r = 0; r < R; ++r) {
                        narrow_factor_[r] = lane_has_request[(size_t)r] ? (int32_t)(p->req_unit[r] / scale[(size_t)r]) : 1;
                        same = same && narrow_factor_[r] == 1;
                    }
                    if (same) req32_dev = d_req32_in;
                    else {
                        d_req32_scaled = (int32_t*)dalloc(4 * G * (size_t)R);
                        if (!d_req32_scaled) return fail(CASIM_ERR_NOMEM, "no room for the int32 request table");
                        req32_dev = d_req32_scaled;
                    }
                } else
                if (dev_gcd) {   // the quotients on the device, from the int64 table that is there already: 8 bytes per PEG less over the link
                    int32_t* d32 = (int32_t*)dalloc(4 * G * (size_t)R);
                    if (!d32) return fail(CASIM_ERR_NOMEM, "no room for the int32 request table");
                    ScaleParams sp; memset(&sp, 0, sizeof sp);
                    for (int r = 0; r < R; ++r) { sp.inv[r] = inv[(size_t)r]; sp.tz[r] = tz[(size_t)r]; }
                    const int64_t nv = (int64_t)(G * (size_t)R);
                    const int nb = (int)(nv / 1024 < 1 ? 1 : (nv / 1024 > 2048 ? 2048 : nv / 1024));
                    bk_.launch(scale_requests_kernel, nb, 1, 256, (size_t)0, (const int64_t*)dt_.req, nv, R, sp, d32);
                    req32_dev = d32;
                } else {
                    int32_t* req32 = up_reserve<int32_t>(G * (size_t)R, &req32_dev);
                    if (!req32) { req32_own.resize(G * (size_t)R); req32 = req32_own.data(); }
                    par_for(G, 65536, [&](size_t lo, size_t hi, int) { for (size_t i = lo; i < hi; ++i) for (int r = 0; r < R; ++r) req32[i * R + r] = quot(p->req[i * R + r], r); });
                }
                for (size_t i = 0; i < NG; ++i) for (int r = 0; r < R; ++r)
                    fresh32[i * R + r] = quot(g->alloc[i * R + r] - g->init_req[i * R + r], r);
                fs_.req32 = req32_dev ? req32_dev : up(req32_own.data(), req32_own.size());
                fs_.fresh32 = up(fresh32.data(), fresh32.size());
                fs_.scale = up(scale.data(), scale.size());   // (all three copied into the staging buffer already)
                if (getenv("CASIM_PACK_PROF_DUMP")) { fs_.prof = (int64_t*)dalloc(8 * 8 * NG); bk_.zero(fs_.prof, 8 * 8 * NG); }
                fast_npt_ = maxcap <= 64 ? 1 : (maxcap <= 256 ? 4 : 16);
                fast_r_ = R <= 2 ? 2 : 4;
            }
            if (fast_npt_ == 0) fast_retry_ = false;
            else bk_.prepare_pack_fast(pack_build_, fast_i64_ ? 8 : fast_r_, fast_npt_, fast_wx_);   // (the instantiation's self-check, first use only: here, not inside the first launch)
        }
        ps_.node_cap = up(cap.data(), NG);
        if (o && o->node_pods) {   // pods per simulated node (estimationAnalyserFunc's newNodesWithPods)
            np_off_.assign(NG + 1, 0);
            for (size_t i = 0; i < NG; ++i) np_off_[i + 1] = np_off_[i] + cap[i];
            d_node_pods_ = (int32_t*)dalloc(4 * (size_t)(np_off_[NG] > 0 ? np_off_[NG] : 1));
            dr_.node_pods = d_node_pods_; dr_.node_pods_off = up(np_off_.data(), NG + 1);
        }
        if (!pack_lds_) {
            ps_.state_off = up(soff.data(), NG);
            ps_.gstate = (char*)dalloc((size_t)total);
        }
        stage.mark("order geometry + result arrays");
        // ---- order scratch geometry ----
        int64_t npad_max = 1;
        std::vector<int64_t> npad_of(NG);
        for (size_t i = 0; i < NG; ++i) {
            int64_t npad = 1;
            while (npad < pegs_of_group[i]) npad <<= 1;
            npad_of[i] = npad;
            npad_max = npad > npad_max ? npad : npad_max;
        }
        // one thread per pair of the bitonic network: npad / 2 threads, 64..kOrderThreads
        order_threads_ = (int)(npad_max / 2 < 64 ? 64 : (npad_max / 2 > kOrderThreads ? kOrderThreads : (npad_max / 2 + 63) / 64 * 64));
        // a batch with thousands of groups fills the chip with blocks anyway: one wave per group then, so that no wave sits idle
        // in the block barriers of a list that is far shorter than its bound (C2 batch: bound 400 PEGs -> 256 threads, actual
        // lists ~110 -> 64 pairs; 0.49 ms of a 2.3 ms step, r02a).  The kernel is latency-bound (dependent gathers, LDS round
        // trips), so its LDS is sized for the lists the one-wave networks take (<= 256 PEGs: 3.5 KB, 8 waves per SIMD) and
        // the few longer lists of such a launch sort in an HBM slab; sized for the BOUND (400 -> 512 entries + the
        // reduction array of 256 threads = 8.2 KB) the launch ran at 4-5 waves per SIMD.
        // (CASIM_TEST_BATCH_GROUPS: the CPU suite reaches the batch geometry with a few dozen groups under the emulator)
        const char* bg = getenv("CASIM_TEST_BATCH_GROUPS");
        const bool batch = NG_ >= (bg && atoi(bg) > 0 ? atoi(bg) : 2048) && npad_max <= 1024;
        if (batch) order_threads_ = 64;
        const int64_t lds_cap = batch && npad_max > 256 ? 256 : 0;
        std::vector<int64_t> ooff(NG, 0);
        int64_t ototal = 0, oworst = 0;
        bool any_slab = false;
        for (size_t i = 0; i < NG; ++i) {
            // (a one-wave block sorts lists of <= 256 PEGs in a network of 64 / 128 / 256 entries: never less than 64)
            const int64_t entries = order_threads_ == 64 && npad_of[i] < 64 ? 64 : npad_of[i];
            const int64_t bytes = entries * 16 + 8 + 8 * order_threads_;   // keys, positions, PEG ids + the fastpath reduction
            if (lds_cap == 0 || npad_of[i] > lds_cap) { ooff[i] = ototal; ototal += (bytes + 255) & ~255ll; any_slab = any_slab || lds_cap > 0; }
            const int64_t in_lds = lds_cap > 0 && npad_of[i] > lds_cap ? lds_cap * 16 + 8 + 8 * order_threads_ : bytes;
            oworst = in_lds > oworst ? in_lds : oworst;
        }
        order_smem_ = (size_t)oworst;
        order_lds_ = oworst <= (int64_t)bk_.lds_budget();
        os_.lds_list_cap = (int32_t)lds_cap;
        if (!order_lds_ || any_slab) {
            os_.off = up(ooff.data(), NG);
            os_.gbuf = (char*)dalloc((size_t)(ototal > 0 ? ototal : 256));
        }
        // ---- results ----
        if (ord_in_slab_ && (size_t)nnz_cap_ <= ord_cap_) { dr_.order = (int32_t*)(res_slab_ + ord_off_); dr_.placed = dr_.order + ord_cap_ + 1; }
        else { ord_in_slab_ = false; dr_.order = (int32_t*)dalloc(4 * ((size_t)nnz_cap_ + 1)); dr_.placed = (int32_t*)dalloc(4 * (size_t)nnz_cap_); }
        dr_.fast_last = (uint8_t*)dalloc(NG);
        if (fast_npt_ > 0) {   // register packer: one record per PEG (casim_types.h) instead of the three arrays
            // two int32 lanes next to exclusion words: the words ride with the record (DevResults::rec_xw — RegStore<2, ., 2>::kRecWords expects them)
            dr_.rec_xw = (fast_r_ == 2 && !fast_i64_ && fast_wx_ > 0) ? 1 : 0;
            dr_.rec_dw = (fast_r_ == 2 && !fast_i64_ && !dr_.rec_xw) ? 8 : 16;
            dr_.rec_i64 = fast_i64_ ? 1 : 0;
            dr_.rec = (uint32_t*)dalloc(4 * ((size_t)nnz_cap_ + 1) * (size_t)dr_.rec_dw);   // + one spare record: the packer loads record k + 1 unconditionally
            dr_.req32 = fs_.req32; dr_.fresh32 = fs_.fresh32;
        }
        if (fast_npt_ == 0 || fast_retry_) {
            dr_.s_count = (int32_t*)dalloc(4 * (size_t)nnz_cap_); dr_.s_flags = (uint32_t*)dalloc(4 * (size_t)nnz_cap_);
            dr_.s_req = (int64_t*)dalloc(8 * (size_t)nnz_cap_ * (size_t)R);
        }
        d_opt_out_ = (int32_t*)(res_slab_ + opt_off_); d_opt_packed_ = (int64_t*)(res_slab_ + opt_off_ + 16);
        d_opt_key_ = (int64_t*)(res_slab_ + opt_off_ + 24); d_opt_set_ = (uint8_t*)(res_slab_ + opt_off_ + 104);
        opt_cap_ = 1;
        stage.mark("H2D copy + sync");
        end_uploads();
        if (narrow_in) {   // behind the uploads on the same stream, in front of every kernel that reads a request
            UnitParams upar; memset(&upar, 0, sizeof upar);
            for (int r = 0; r < R && r < CASIM_KMAX_RES; ++r) { upar.unit[r] = p->req_unit[r]; upar.factor[r] = d_req32_scaled ? narrow_factor_[r] : 1; }
            const int64_t nv = (int64_t)(G * (size_t)R);
            const int nb = (int)(nv / 1024 < 1 ? 1 : (nv / 1024 > 2048 ? 2048 : nv / 1024));
            bk_.launch(expand_requests_kernel, nb, 1, 256, (size_t)0, d_req32_in, nv, R, upar, d_req64, d_req32_scaled);
        }
        // the streaming feasibility kernel's group records (feas_group_records_kernel): once per problem, behind the uploads on the same stream
        d_feas_rec_ = nullptr;
        {
            const bool off = getenv("CASIM_NO_FEAS_STREAM") != nullptr;   // A/B switch: the LDS-staged feas_sim_kernel of round 4
            if (!off && feas_by_sim_ && csr_on_device_ && fast_npt_ > 0 && !fast_i64_ && fs_.fresh32 && NG_ > 0 && dt_.R <= 4) {
                d_feas_rec_ = (uint32_t*)dalloc((size_t)NG_ * CASIM_FEAS_REC_DW * 4);
                // which bits of the taint / selector words the batch uses at all: decided on the device (mask_or_kernel), read back behind
                // the wait a resident problem's init ends with anyway — unused upper halves drop the kHi terms, an unused bit carries
                // NodeUnschedulable (the records are then built a second time, once per problem); a one-shot call does not wait here and
                // runs the general instantiation
                feas_hi_ = true; feas_us_word_ = -1; feas_us_bit_ = 0; h_mask_used_[0] = h_mask_used_[1] = ~0ull;
                feas_smem_ = (size_t)(((max_sim_groups_ + 3) & ~3) + 4) * CASIM_FEAS_REC_DW * 4;   // (rounded up to four records, plus four: see the kernel's look-ahead)
                if (d_feas_rec_) bk_.launch(feas_group_records_kernel, (NG_ + 255) / 256, 1, 256, (size_t)0, feas_tables(), fs_.fresh32, d_feas_rec_, -1, 0);
                if (d_feas_rec_ && !one_shot_) {
                    uint64_t* flag = (uint64_t*)dalloc(16);
                    if (flag) {
                        bk_.zero(flag, 16);
                        const int64_t na = (int64_t)NG_ * dt_.Wt, nb = (int64_t)G_ * dt_.Wl;
                        const int64_t most = na > nb ? na : nb;
                        const int blocks = (int)(most / 2048 > 1024 ? 1024 : (most / 2048 < 1 ? 1 : most / 2048));
                        bk_.launch(mask_or_kernel, blocks, 1, 256, (size_t)0, dt_.taint, na, dt_.sel, nb, flag);
                        bk_.d2h(h_mask_used_, flag, 16);
                    }
                }
            }
        }
        // the staging buffer belongs to the backend: the next problem of this context may reuse it after init() — unless the caller
        // runs and fetches THIS problem before anything else touches the context (one call = one problem: casim_estimate_batch); the
        // copy then drains with the kernels behind it and a single call waits for the device once instead of twice
        if (!one_shot_) {
            bk_.sync();
            if (d_feas_rec_ && (h_mask_used_[0] & h_mask_used_[1]) != ~0ull) {
                const uint64_t ut = h_mask_used_[0], us = h_mask_used_[1];
                feas_hi_ = ((ut | us) >> 32) != 0;
                const int top = feas_hi_ ? 64 : 32;
                feas_us_word_ = -1;
                for (int b = 0; b < top && feas_us_word_ < 0; ++b) if (!((ut >> b) & 1)) { feas_us_word_ = 0; feas_us_bit_ = b; }
                for (int b = 0; b < top && feas_us_word_ < 0; ++b) if (!((us >> b) & 1)) { feas_us_word_ = 1; feas_us_bit_ = b; }
                if (feas_us_word_ >= 0) bk_.launch(feas_group_records_kernel, (NG_ + 255) / 256, 1, 256, (size_t)0, feas_tables(), fs_.fresh32, d_feas_rec_, feas_us_word_, feas_us_bit_);
            }
        }
        pass_gate();
        stage.mark("done");
        if (!bk_.ok()) return fail(CASIM_ERR_HIP, bk_.error());
        ready_ = true;
        return CASIM_OK;
    }

    // ---- launch sequence ----------------------------------------------------------------
    int32_t run_feasibility() {
        if (!csr_on_device_ || NG_ == 0) return CASIM_OK;
        front_ran_ = false;
        if (front_) {
            const int nw = (order_threads_ + 63) / 64;
            const size_t own = 8 * (size_t)Wg_ + 4 * (2 * (size_t)nw + 3) + 8;
            // polls per ticket before a block counts a missing predecessor's row itself (front_kernel); CASIM_FRONT_SPIN=0: never wait,
            // always count (tests)
            uint32_t spin = 256;
            if (const char* e = getenv("CASIM_FRONT_SPIN")) { const long v = atol(e); spin = v < 0 ? 0u : (uint32_t)v; }
            const size_t smem = order_lds_ && order_smem_ > own ? order_smem_ : own;
            ++front_epoch_;
            if (order_lds_) bk_.launch(front_kernel<true>, NG_, 1, order_threads_, smem, dt_, dr_, os_, d_bits_, Wg_, d_off_, d_idx_, d_ticket_, front_epoch_, NG_, spin);
            else bk_.launch(front_kernel<false>, NG_, 1, order_threads_, smem, dt_, dr_, os_, d_bits_, Wg_, d_off_, d_idx_, d_ticket_, front_epoch_, NG_, spin);
            front_ran_ = true;
            return CASIM_OK;
        }
        if (strided_ && strided_one_launch_) {
            bk_.launch(front_sim_kernel<true>, n_sims_, 1, 64 * Wg_, front_sim_smem_, dt_, dr_, os_, d_bits_, Wg_,
                       fast_npt_ > 0 ? fs_.req32 : (const int32_t*)nullptr, fast_npt_ > 0 ? fs_.fresh32 : (const int32_t*)nullptr, res_off_, d_idx_, max_sim_groups_);
            front_ran_ = true;
            return CASIM_OK;
        }
        if (strided_) {   // rows by feas_sim_kernel; run_order() builds and orders the lists (order_strided_kernel): no scan, no fill
            launch_feas_sim((feas_len_ + 255) / 256, n_sims_, 256, (size_t)128 * (size_t)max_sim_groups_, feas_tables(), d_bits_, Wg_,
                       fast_npt_ > 0 ? fs_.req32 : (const int32_t*)nullptr, fast_npt_ > 0 ? fs_.fresh32 : (const int32_t*)nullptr);
            return CASIM_OK;
        }
        if (feas_len_ > 0) {
            if (feas_by_sim_) launch_feas_sim((feas_len_ + 255) / 256, n_sims_, 256, (size_t)128 * (size_t)max_sim_groups_, feas_tables(), d_bits_, Wg_,
                                         fast_npt_ > 0 ? fs_.req32 : (const int32_t*)nullptr, fast_npt_ > 0 ? fs_.fresh32 : (const int32_t*)nullptr);
            else bk_.launch(feas_kernel, (feas_len_ + 255) / 256, NG_, 256, (size_t)0, dt_, d_bits_, Wg_);
        }
        // short rows (a simulation's few hundred PEGs): the row popcount is folded into the scan; long rows get a block each
        const bool fold = Wg_ <= 16;
        if (!fold) bk_.launch(csr_count_kernel, NG_, 1, 256, (size_t)64, (const uint64_t*)d_bits_, Wg_, d_counts_);
        const int scan_threads = NG_ <= 64 ? 64 : (NG_ <= 256 ? 256 : 1024);
        const int nb = (NG_ + scan_threads - 1) / scan_threads;
        // block-local offsets + block totals, then the fill kernel adds the totals in front of a group's block itself and writes the
        // final offsets (no separate fix-up launch)
        bk_.launch(csr_scan_local_kernel, nb, 1, scan_threads, (size_t)(4 * ((scan_threads + 63) / 64)), (const int32_t*)d_counts_,
                   fold ? (const uint64_t*)d_bits_ : (const uint64_t*)nullptr, Wg_, NG_, d_off_local_, d_block_sums_, d_counts_);
        bk_.launch(csr_fill_kernel, NG_, 1, 64, (size_t)0, (const uint64_t*)d_bits_, Wg_, d_off_, d_idx_, dt_.peg_lo,
                   (const int32_t*)d_off_local_, (const int32_t*)d_block_sums_, scan_threads, NG_, (const int32_t*)d_counts_);
        return CASIM_OK;
    }
    int32_t run_order() {
        if (NG_ == 0 || front_ran_) return CASIM_OK;   // (front_kernel ordered the lists it made)
        if (getenv("CASIM_PACK_PROF_DUMP") && !os_.prof) { os_.prof = (int64_t*)dalloc(8 * 4 * (size_t)NG_); bk_.zero(os_.prof, 8 * 4 * (size_t)NG_); }
        if (strided_ && rank_once_) {
            bk_.launch(rank_shapes_kernel, (int)n_sorted_pairs_, 1, 256, rank_smem_, dt_, d_sorted_list_, d_pair_rep_, (const int32_t*)nullptr, d_pair_src_, d_ranks_, (int)rank_stride_);
            if (n_checked_pairs_ > 0)
                bk_.launch(rank_shapes_kernel, (int)n_checked_pairs_, 1, 256, rank_smem_, dt_, d_checked_list_, d_pair_rep_, d_pair_base_, d_pair_src_, d_ranks_, (int)rank_stride_);
            bk_.launch(order_ranked_kernel, NG_, 1, 64, (size_t)(8 * ((Wg_ + 1) & ~1) + 4 * (size_t)rank_stride_), dt_, dr_, os_, (const uint64_t*)d_bits_, Wg_, res_off_,
                       d_pair_of_, (const int32_t*)d_pair_src_, (const int32_t*)d_ranks_, (int)rank_stride_);
        } else if (strided_) {
            const size_t smem = order_smem_ > front_sim_wave_scratch() ? order_smem_ : front_sim_wave_scratch();
            if (order_lds_) bk_.launch(order_strided_kernel<true>, NG_, 1, order_threads_, smem, dt_, dr_, os_, (const uint64_t*)d_bits_, Wg_, res_off_, d_idx_);
            else bk_.launch(order_strided_kernel<false>, NG_, 1, order_threads_, front_sim_wave_scratch(), dt_, dr_, os_, (const uint64_t*)d_bits_, Wg_, res_off_, d_idx_);
        }
        else if (order_lds_) bk_.launch(order_kernel<true>, NG_, 1, order_threads_, order_smem_, dt_, dr_, os_);
        else bk_.launch(order_kernel<false>, NG_, 1, order_threads_, (size_t)0, dt_, dr_, os_);
        if (os_.prof) {  // profiling builds: mean ticks per phase over the groups
            std::vector<int64_t> h((size_t)NG_ * 4);
            bk_.d2h(h.data(), os_.prof, h.size() * 8); bk_.sync();
            double m[4] = {0};
            for (int i = 0; i < NG_; ++i) for (int j = 0; j < 4; ++j) m[j] += (double)h[(size_t)i * 4 + j] / NG_;
            fprintf(stderr, "[order prof] ticks/group: scores %.0f sort %.0f records %.0f (threads %d)\n", m[0], m[1], m[2], order_threads_);
        }
        return CASIM_OK;
    }
    int32_t run_pack() {
        if (NG_ == 0) return CASIM_OK;
        if (chain_) bk_.zero(d_chain_marks_, 4 * ((size_t)chain_passes_ + 1));
        run_pack_pass(dt_);
        if (chain_) {
            // casim_options.chain_last_index (chain_fix_kernel): fix-up passes to the sequential loop's fixed point.
            // (the caller's last_index column sits in the upload slab: device memory of this problem, rewritten in place)
            // SHORT chains — batches of simulations, a few tens of groups each — are enqueued whole and nothing is waited for: the bound is
            // groups per simulation - 1 passes, a pass without marked groups is a row of waves that leave at once.  LONG chains (the Go
            // shim's prefetch: ONE simulation with a group per node group, a few hundred passes of near-empty launches) stop at the
            // fixed point: passes go out in growing blocks and after each block the marks of its LAST pass come back (4 bytes, one wait) —
            // a pass that marked nothing re-estimated nothing, so every later pass would find the same tables and mark nothing either.
            DevTables dc = dt_;
            dc.chain_redo = d_chain_redo_;
            const int n_sims = n_sims_ > 0 ? n_sims_ : 1;
            const char* async_env = getenv("CASIM_CHAIN_ASYNC_MAX");
            const int async_max = async_env ? atoi(async_env) : 24;
            const bool whole = chain_passes_ <= async_max;
            int pass = 0, block = 4, checks = 0;
            while (pass < chain_passes_) {
                const int end = whole ? chain_passes_ : (pass + block < chain_passes_ ? pass + block : chain_passes_);
                for (; pass < end; ++pass) {
                    bk_.launch(chain_fix_kernel, (n_sims + 255) / 256, 1, 256, (size_t)0, dt_, dr_, (int32_t*)dt_.last_index, d_chain_redo_, d_chain_marks_ + pass);
                    run_pack_pass(dc);
                }
                if (pass >= chain_passes_) break;
                int32_t marked = 0;
                bk_.d2h(&marked, d_chain_marks_ + (pass - 1), 4); bk_.sync(); ++checks;
                if (marked == 0) break;
                if (block < 32) block *= 2;
            }
            chain_passes_run_ = pass;
            int32_t* ci = last_chain_info();
            ci[0] = chain_passes_; ci[1] = pass; ci[2] = checks; ci[3] = whole ? 1 : 0;
        }
        return CASIM_OK;
    }
    // marked groups per fix-up pass of the last run (chain mode; tests and the bench's chain row): waits for the device
    std::vector<int32_t> chain_marks() {
        std::vector<int32_t> h((size_t)(chain_ ? chain_passes_ : 0));   // (passes an early stop never enqueued read zero: the marks were cleared)
        if (!h.empty()) { bk_.d2h(h.data(), d_chain_marks_, 4 * h.size()); bk_.sync(); }
        return h;
    }
    int32_t run_pack_pass(const DevTables& dt_) {   // (one launch set of the packer over the groups `dt_` lets through)
        if (fast_npt_ > 0) {
            // register-resident int32 packer: the instantiation (lanes, node slots per lane, exclusion words) is picked by the
            // backend — the product compiles these kernels in their own translation unit (casim_pack_tu.hip)
            bk_.launch_pack_fast(pack_build_, fast_i64_ ? 8 : fast_r_, fast_npt_, fast_wx_, NG_, dt_, dr_, fs_);
            if (fs_.prof) {  // profiling builds: mean ticks per phase over the groups
                std::vector<int64_t> h((size_t)NG_ * 8);
                bk_.d2h(h.data(), fs_.prof, h.size() * 8); bk_.sync();
                double m[8] = {0};
                for (int i = 0; i < NG_; ++i) for (int j = 0; j < 8; ++j) m[j] += (double)h[(size_t)i * 8 + j] / NG_;
                fprintf(stderr, "[pack prof] ticks/group: loop %.0f bcast %.0f passA %.0f reduce %.0f passBC %.0f a3 %.0f\n", m[0], m[1], m[2], m[3], m[4], m[5]);
            }
            if (!fast_retry_) return CASIM_OK;
        }
        ps_.retry_only = fast_npt_ > 0 ? 1 : 0;
        if (dt_.R <= 2) {
            if (pack_lds_) bk_.launch(pack_kernel<true, 2>, NG_, 1, 64, pack_smem_, dt_, dr_, ps_);
            else bk_.launch(pack_kernel<false, 2>, NG_, 1, 64, (size_t)0, dt_, dr_, ps_);
        } else {
            if (pack_lds_) bk_.launch(pack_kernel<true, CASIM_KMAX_RES>, NG_, 1, 64, pack_smem_, dt_, dr_, ps_);
            else bk_.launch(pack_kernel<false, CASIM_KMAX_RES>, NG_, 1, 64, (size_t)0, dt_, dr_, ps_);
        }
        return CASIM_OK;
    }
    int32_t run() {
        if (!ready_) return fail(CASIM_ERR_INVALID, "problem not initialised");
        run_feasibility(); run_order(); run_pack();
        ran_ = true; h_off_fresh_ = false;
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    // the kernels were enqueued phase by phase (timed callers): results may be fetched
    int32_t run_mark() { if (!ready_) return fail(CASIM_ERR_INVALID, "problem not initialised"); ran_ = true; h_off_fresh_ = false; return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error()); }

    int32_t csr(int32_t* nnz_out, int32_t* offsets_out) {
        if (!ready_) return fail(CASIM_ERR_INVALID, "problem not initialised");
        if (csr_on_device_ && NG_ > 0) {
            if (!ran_) return fail(CASIM_ERR_INVALID, "run the problem first");
            if (!h_off_fresh_) {   // (fetch() brings the offsets along with the scalars: no second round trip behind it)
                h_off_.resize((size_t)NG_ + 1);
                bk_.d2h(h_off_.data(), d_off_, 4 * ((size_t)NG_ + 1));
                bk_.sync();
                if (strided_) counts_to_offsets();
                h_off_fresh_ = true;
            }
        }
        if (NG_ == 0) { if (nnz_out) *nnz_out = 0; if (offsets_out) offsets_out[0] = 0; return CASIM_OK; }   // an empty shard
        const std::vector<int32_t>* offs = &h_off_;
        if (runs_.active) {   // the caller counts entries of ITS lists: a merged row stands for `len` of them
            const int32_t rc = expand_offsets();
            if (rc != CASIM_OK) return rc;
            offs = &h_off_exp_;
        }
        if (nnz_out) *nnz_out = NG_ > 0 ? (*offs)[(size_t)NG_] : 0;
        if (offsets_out && NG_ >= 0) for (int i = 0; i <= NG_; ++i) offsets_out[i] = offs->empty() ? 0 : (*offs)[(size_t)i];
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }
    // fixed-stride lists: h_off_ holds the groups' list LENGTHS as the device wrote them -> the compact CSR offsets the caller indexes with
    void counts_to_offsets() {
        int32_t run = 0;
        for (int i = 0; i < NG_; ++i) { const int32_t c = h_off_[(size_t)i]; h_off_[(size_t)i] = run; run += c; }
        h_off_[(size_t)NG_] = run;
    }
    // merged lists -> offsets in the caller's numbering (h_off_ must be current)
    int32_t expand_offsets() {
        const size_t NG = (size_t)NG_, nnz = NG > 0 ? (size_t)h_off_[NG] : 0;
        h_idx_m_.resize(nnz + 1);
        if (nnz > 0) {
            if (csr_on_device_) { bk_.d2h(h_idx_m_.data(), d_idx_, 4 * nnz); bk_.sync(); }
            else memcpy(h_idx_m_.data(), runs_.g.peg_index, 4 * nnz);
        }
        h_off_exp_.assign(NG + 1, 0);
        for (size_t i = 0; i < NG; ++i) {
            const int64_t t = (int64_t)h_off_exp_[i] + runs_.expanded(h_idx_m_.data() + h_off_[i], (int64_t)h_off_[i + 1] - h_off_[i]);
            if (t > 0x7fffffffll) return fail(CASIM_ERR_INVALID, "expanded PEG lists too long");
            h_off_exp_[i + 1] = (int32_t)t;
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    int32_t fetch(casim_results* out) {
        if (!ready_ || !ran_) return fail(CASIM_ERR_INVALID, "nothing to fetch: run the problem first");
        if (!out) return fail(CASIM_ERR_INVALID, "null results");
        if (runs_.active && (out->order || out->placed)) {
            // merged rows on the device, the caller's rows in its arrays: fetch into host copies, then write every run out member by
            // member — identical pods tried in order, so the first `placed` of them were scheduled
            casim_results m = *out;
            int32_t rc = csr(nullptr, nullptr);                       // h_off_ (merged), h_idx_m_, h_off_exp_
            if (rc != CASIM_OK) return rc;
            const size_t NG = (size_t)NG_, nnz_m = NG > 0 ? (size_t)h_off_[NG] : 0;
            std::vector<int32_t> order_m(nnz_m + 1), placed_m(nnz_m + 1);
            m.order = order_m.data(); m.placed = placed_m.data();
            runs_.active = false;                                     // (the plain path below, on the merged numbering)
            rc = fetch(&m);
            runs_.active = true;
            if (rc != CASIM_OK) return rc;
            for (size_t i = 0; i < NG; ++i) {
                int64_t at = h_off_exp_[i];
                for (int32_t k = h_off_[i]; k < h_off_[i + 1]; ++k) {
                    const int32_t row = order_m[(size_t)k], n = runs_.len[(size_t)row], f = runs_.first[(size_t)row], pl = placed_m[(size_t)k];
                    if (n == 1) { if (out->order) out->order[at] = f; if (out->placed) out->placed[at] = pl; ++at; continue; }
                    for (int32_t j = 0; j < n; ++j, ++at) { if (out->order) out->order[at] = f + j; if (out->placed) out->placed[at] = j < pl ? 1 : 0; }
                }
            }
            return CASIM_OK;
        }
        const size_t NG = (size_t)NG_, ng = NG > 0 ? NG : 1;
        const bool winners = winners_only_ && (out->order || out->placed);
        if (winners && !winners_ready_) return fail(CASIM_ERR_INVALID, "winners_only: run the expander query (per simulation, best_out set) before the fetch");
        // copy 1: scalars + offsets (one slab); copies 2, 3: order / placed — enqueued with the first one when their bound is
        // small or the offsets are the caller's, after it (the device-side nnz is in the slab) otherwise
        const bool spec = (!csr_on_device_ || nnz_cap_ <= 16384) && !winners && !strided_;
        const bool compact = strided_ && !winners && (out->order || out->placed);
        if (compact) {   // fixed-stride lists -> compact CSR on the device, behind everything else on the stream (never in the resident loop)
            if (!d_coff_) { d_coff_ = (int32_t*)dalloc(4 * ((size_t)NG_ + 1)); d_corder_ = (int32_t*)dalloc(4 * ((size_t)nnz_cap_ + 1)); d_cplaced_ = (int32_t*)dalloc(4 * ((size_t)nnz_cap_ + 1)); }
            bk_.launch(count_offsets_kernel, 1, 1, 1024, (size_t)(4 * (1024 / 64 + 2)), (const int32_t*)dt_.peg_cnt, NG_, d_coff_);
            bk_.launch(compact_lists_kernel, NG_, 1, 64, (size_t)0, (const int32_t*)dt_.peg_off, (const int32_t*)dt_.peg_cnt, (const int32_t*)d_coff_,
                       (const int32_t*)dr_.order, (const int32_t*)dr_.placed, d_corder_, d_cplaced_, order_id_base_);
        }
        fetch_rebased_ = compact;
        const size_t spec_n = !csr_on_device_ ? (size_t)(NG > 0 ? h_off_[NG] : 0) : (size_t)nnz_cap_;
        char* st = (char*)bk_.stage(1, res_bytes_ + (spec ? 8 * spec_n : 0) + 64);
        if (!st) return fail(CASIM_ERR_NOMEM, "no staging buffer");
        bk_.d2h(st, res_slab_, res_bytes_);
        int32_t wtotal = 0;
        if (winners) bk_.d2h(&wtotal, d_woff_ + winners_s_, 4);
        int32_t* st_order = (int32_t*)(st + ((res_bytes_ + 15) & ~(size_t)15));
        int32_t* st_placed = st_order + spec_n;
        if (ord_in_slab_) { st_order = (int32_t*)(st + ord_off_); st_placed = st_order + ord_cap_ + 1; }   // (they came with the slab)
        else if (spec && spec_n > 0) {
            if (out->order) bk_.d2h(st_order, dr_.order, 4 * spec_n);
            if (out->placed) bk_.d2h(st_placed, dr_.placed, 4 * spec_n);
        }
        bk_.sync();
        if (opt_in_slab_) opt_keep_.assign(st + opt_off_, st + opt_off_ + 104 + ((ng + 7) & ~(size_t)7));   // (the expander's answer came along: best_option_finish)
        const int64_t* h64 = (const int64_t*)st;
        const int32_t* h32 = (const int32_t*)(h64 + 2 * ng);
        if (csr_on_device_ && NG_ > 0) { h_off_.assign(h32 + 6 * ng, h32 + 6 * ng + NG + 1); if (strided_) counts_to_offsets(); h_off_fresh_ = true; }
        const size_t nnz = NG_ > 0 ? (size_t)h_off_[NG] : 0;
        if (out->req_cpu_sum) memcpy(out->req_cpu_sum, h64, 8 * NG);
        if (out->req_mem_sum) memcpy(out->req_mem_sum, h64 + ng, 8 * NG);
        if (out->node_count) memcpy(out->node_count, h32, 4 * NG);
        if (out->pods_scheduled) memcpy(out->pods_scheduled, h32 + ng, 4 * NG);
        if (out->nodes_added) memcpy(out->nodes_added, h32 + 2 * ng, 4 * NG);
        if (out->limiter_nodes) memcpy(out->limiter_nodes, h32 + 3 * ng, 4 * NG);
        if (out->last_index_out) memcpy(out->last_index_out, h32 + 4 * ng, 4 * NG);
        if (out->status) memcpy(out->status, h32 + 5 * ng, 4 * NG);
        if (spec) {
            if (out->order && nnz) memcpy(out->order, st_order, 4 * nnz);
            if (out->placed && nnz) memcpy(out->placed, st_placed, 4 * nnz);
        }
        if (winners) {            // the compacted lists of the simulations' winning groups: sum(len(winner)) entries
            if (wtotal > 0) {
                if (out->order) bk_.d2h(out->order, d_worder_, 4 * (size_t)wtotal);
                if (out->placed) bk_.d2h(out->placed, d_wplaced_, 4 * (size_t)wtotal);
                bk_.sync();
            }
            winners_total_ = wtotal;
        } else if (compact) {
            if (nnz > 0) {
                if (out->order) bk_.d2h(out->order, d_corder_, 4 * nnz);
                if (out->placed) bk_.d2h(out->placed, d_cplaced_, 4 * nnz);
                bk_.sync();
            }
        } else if (!spec && nnz > 0) {   // a big batch: straight into the caller's arrays (through the pinned staging buffer in two pieces it was
                                  // no faster: 6.4-6.5 ms against 6.2-6.3 per headline call, r07n)
            if (out->order) bk_.d2h(out->order, dr_.order, 4 * nnz);
            if (out->placed) bk_.d2h(out->placed, dr_.placed, 4 * nnz);
            bk_.sync();
        }
        if (d_node_pods_ && out->node_pods && out->node_pods_offsets) {
            // padded per-group slices on the device -> compact lists (nodes_added[i] entries each) for the caller
            std::vector<int32_t> padded((size_t)np_off_[NG] + 1);
            bk_.d2h(padded.data(), d_node_pods_, 4 * (size_t)np_off_[NG]);
            bk_.sync();
            int64_t at = 0;
            out->node_pods_offsets[0] = 0;
            for (size_t i = 0; i < NG; ++i) {
                int64_t n = h32[2 * ng + i];   // nodes_added
                if (h32[5 * ng + i] != CASIM_NG_OK) n = 0;
                if (at + n > out->node_pods_capacity) n = out->node_pods_capacity - at > 0 ? out->node_pods_capacity - at : 0;
                if (n > 0) memcpy(out->node_pods + at, padded.data() + np_off_[i], 4 * (size_t)n);
                at += n;
                out->node_pods_offsets[i + 1] = (int32_t)at;
            }
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    // overwrite one group's result (a group estimated elsewhere joins the expander reduce)
    int32_t set_group_result(int32_t ng, const casim_cluster_estimate_result* r) {
        if (!ready_ || !ran_) return fail(CASIM_ERR_INVALID, "run the problem first");
        if (!r || ng < 0 || ng >= NG_) return fail(CASIM_ERR_INVALID, "bad group index");
        const int32_t ok = CASIM_NG_OK;
        bk_.h2d(dr_.node_count + ng, &r->node_count, 4); bk_.h2d(dr_.pods + ng, &r->pods_scheduled, 4);
        bk_.h2d(dr_.nodes_added + ng, &r->nodes_added, 4); bk_.h2d(dr_.limiter_nodes + ng, &r->limiter_nodes, 4);
        bk_.h2d(dr_.last_index_out + ng, &r->last_index_out, 4); bk_.h2d(dr_.status + ng, &ok, 4);
        bk_.h2d(dr_.cpu_sum + ng, &r->req_cpu_sum, 8); bk_.h2d(dr_.mem_sum + ng, &r->req_mem_sum, 8);
        bk_.sync();
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    // SchedulingError codes [NG][L] (L = longest candidate range): synchronous
    int32_t reasons(const uint64_t* port_block_host, uint16_t* out_codes) {
        if (!ready_ || !csr_on_device_) return fail(CASIM_ERR_INVALID, "reasons need device-side subsets (peg_offsets == NULL)");
        if (!out_codes) return fail(CASIM_ERR_INVALID, "null output");
        if (NG_ == 0 || feas_len_ == 0) return CASIM_OK;
        const size_t n = (size_t)NG_ * (size_t)feas_len_;
        uint16_t* d = (uint16_t*)dalloc(2 * n);
        const uint64_t* pb = (port_block_host && dt_.Wx > 0) ? up(port_block_host, (size_t)G_ * dt_.Wx) : nullptr;
        bk_.launch(reason_kernel, (feas_len_ + 255) / 256, NG_, 256, (size_t)0, dt_, pb, d, feas_len_);
        bk_.d2h(out_codes, d, 2 * n);
        bk_.sync();
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }
    int feas_len() const { return feas_len_; }

    // bit-matrix [NG][ceil(G/64)] of the last run_feasibility()
    int32_t fetch_bits(uint64_t* out_bits) {
        if (!csr_on_device_) return fail(CASIM_ERR_INVALID, "feasibility was not computed on the device");
        bk_.d2h(out_bits, d_bits_, sizeof(uint64_t) * (size_t)NG_ * (size_t)Wg_);
        bk_.sync();
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    int32_t best_option(const int32_t* kinds, int32_t n_kinds, int32_t group_id_base, int32_t* best_ng_out, int32_t* n_best_out,
                        uint8_t* best_set_out, int64_t* key_out, void* dev_key_out) {
        casim_option_query q; memset(&q, 0, sizeof q);
        q.kinds = kinds; q.n_kinds = n_kinds; q.group_id_base = group_id_base; q.best_out = best_ng_out; q.n_best_out = n_best_out;
        q.best_set_out = best_set_out; q.key_out = key_out; q.dev_key_out = dev_key_out;
        return best_option_query(&q);
    }

    // The expander chain over the options of the last run: one reduce over every group, or (per_sim) one per simulation
    // of the batch — one workgroup each, every simulation's winner packed into ONE int64 so that a single
    // all-reduce(min) over [S] keys settles all of them across GPUs.
    int32_t best_option_query(const casim_option_query* q, bool defer_sync = false) {
        if (!ready_ || !ran_) return fail(CASIM_ERR_INVALID, "run the problem first");
        if (!q) return fail(CASIM_ERR_INVALID, "null query");
        const int32_t n_kinds = q->n_kinds;
        if (n_kinds < 0 || n_kinds > 8 || (n_kinds > 0 && !q->kinds)) return fail(CASIM_ERR_INVALID, "bad expander chain");
        const bool per_sim = q->per_sim != 0 && n_sims_ > 0;
        const int S = per_sim ? n_sims_ : 1;
        OptionArgs a; memset(&a, 0, sizeof a);
        a.node_count = dr_.node_count; a.pods = dr_.pods; a.status = dr_.status; a.cpu_sum = dr_.cpu_sum; a.mem_sum = dr_.mem_sum;
        a.waste_cpu = dt_.waste_cpu; a.waste_mem = dt_.waste_mem; a.NG = NG_; a.n_kinds = n_kinds; a.group_id_base = q->group_id_base;
        a.global_id = dt_.global_id; a.sim_off = per_sim ? dt_.sim_off : nullptr;
        for (int i = 0; i < n_kinds; ++i) {
            a.kinds[i] = q->kinds[i];
            if (q->kinds[i] < 0 || q->kinds[i] > 2) return fail(CASIM_ERR_INVALID, "unknown expander kind");
            if (q->kinds[i] == CASIM_EXPANDER_LEAST_WASTE && NG_ > 0 && (!dt_.waste_cpu || !dt_.waste_mem)) return fail(CASIM_ERR_INVALID, "least-waste needs waste_cpu/waste_mem");
        }
        if (q->valid) {
            if (!d_opt_valid_) d_opt_valid_ = (uint8_t*)dalloc((size_t)NG_);
            bk_.h2d(d_opt_valid_, q->valid, (size_t)NG_);
            a.valid = d_opt_valid_;
        }
        if ((size_t)S > opt_cap_) {   // per-simulation result blocks (allocated on first use, kept)
            d_opt_out_ = (int32_t*)dalloc(8 * (size_t)S); d_opt_key_ = (int64_t*)dalloc(80 * (size_t)S); d_opt_packed_ = (int64_t*)dalloc(8 * (size_t)S);
            opt_cap_ = (size_t)S;
        }
        a.best_set = d_opt_set_; a.out = d_opt_out_; a.key_out = q->dev_key_out ? (int64_t*)q->dev_key_out : d_opt_key_;
        a.packed_out = q->dev_packed_out ? (int64_t*)q->dev_packed_out : d_opt_packed_;
        // one block walks the groups of one simulation: with thousands of them in ONE simulation (batched C1) 1024 threads —
        // the kernel was 0.106 ms of a 1.0 ms step at NG = 16384 (profiles/r01u_rocpd_summary.txt); a simulation of a
        // batch has tens of groups: one wave
        const int span = per_sim ? max_sim_groups_ : NG_;
        const int opt_threads = span > 1024 ? 1024 : (span > 64 ? 256 : 64);
        bk_.launch(option_kernel, S, 1, opt_threads, (size_t)(8 * opt_threads), a);
        if (winners_only_) {
            // the winners' lists, compacted on the device right behind the reduce: what fetch() copies back (casim_options.winners_only)
            if (!d_woff_ || (size_t)S > woff_cap_) { d_woff_ = (int32_t*)dalloc(4 * ((size_t)S + 1)); woff_cap_ = (size_t)S; }
            if (!d_worder_) { d_worder_ = (int32_t*)dalloc(4 * ((size_t)nnz_cap_ + 1)); d_wplaced_ = (int32_t*)dalloc(4 * ((size_t)nnz_cap_ + 1)); }
            const int wt = S > 256 ? 1024 : 256;
            bk_.launch(winner_offsets_kernel, 1, 1, wt, (size_t)(4 * ((wt + 63) / 64 + 2)), (const int32_t*)d_opt_out_, (const int32_t*)dt_.peg_off, (const int32_t*)dt_.peg_cnt, S, d_woff_);
            bk_.launch(gather_winners_kernel, S, 1, 64, (size_t)0, (const int32_t*)d_opt_out_, (const int32_t*)dt_.peg_off, (const int32_t*)dt_.peg_cnt, (const int32_t*)d_woff_,
                       (const int32_t*)dr_.order, (const int32_t*)dr_.placed, d_worder_, d_wplaced_);
            winners_ready_ = true; winners_s_ = S;
        }
        if (q->best_out || q->n_best_out || q->best_set_out || q->key_out || q->packed_out) {
            // deferred: the answers land in the (pinned) upload staging buffer — a copy into the caller's pageable arrays would wait for
            // the stream by itself — and move to the caller's arrays once the fetch has waited; the upload that used the buffer is
            // ahead of these copies on the same stream
            const size_t b_out = 8 * (size_t)S, b_set = q->best_set_out ? ((size_t)NG_ + 7) & ~(size_t)7 : 0, b_key = q->key_out ? 80 * (size_t)S : 0,
                         b_pk = q->packed_out ? 8 * (size_t)S : 0;
            if (defer_sync && S == 1 && opt_cap_ == 1 && !q->dev_key_out && !q->dev_packed_out) {
                // the one-simulation answer sits inside the results slab: the fetch that follows brings it along, no copy of its own
                opt_pending_q_ = q; opt_pending_s_ = 1; opt_stage_ = nullptr; opt_in_slab_ = true; opt_keep_.clear();
                return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
            }
            char* st = defer_sync ? (char*)bk_.stage_if_fits(0, b_out + b_set + b_key + b_pk) : nullptr;
            if (st) {
                bk_.d2h(st, d_opt_out_, b_out);
                if (q->best_set_out) bk_.d2h(st + b_out, d_opt_set_, (size_t)NG_);
                if (q->key_out) bk_.d2h(st + b_out + b_set, a.key_out, b_key);
                if (q->packed_out) bk_.d2h(st + b_out + b_set + b_key, a.packed_out, b_pk);
                opt_pending_q_ = q; opt_pending_s_ = S; opt_stage_ = st;
                return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
            }
            opt_host_.resize(2 * (size_t)S);
            bk_.d2h(opt_host_.data(), d_opt_out_, b_out);
            if (q->best_set_out) bk_.d2h(q->best_set_out, d_opt_set_, (size_t)NG_);
            if (q->key_out) bk_.d2h(q->key_out, a.key_out, b_key);
            if (q->packed_out) bk_.d2h(q->packed_out, a.packed_out, b_pk);
            bk_.sync();
            for (int i = 0; i < S; ++i) {
                if (q->best_out) q->best_out[i] = opt_host_[2 * (size_t)i];
                if (q->n_best_out) q->n_best_out[i] = opt_host_[2 * (size_t)i + 1];
            }
        }
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }
    // the host side of a query whose copies were left in flight (best_option_query(q, defer_sync = true)); synced = the stream has been
    // waited for since
    int32_t best_option_finish(bool synced) {
        const casim_option_query* q = opt_pending_q_;
        if (!q) return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
        if (opt_in_slab_) {
            opt_in_slab_ = false; opt_pending_q_ = nullptr;
            if (!synced || opt_keep_.empty()) return fail(CASIM_ERR_INVALID, "expander answer deferred without a fetch");
            const char* b = opt_keep_.data();
            const int32_t* o = (const int32_t*)b;
            if (q->best_out) q->best_out[0] = o[0];
            if (q->n_best_out) q->n_best_out[0] = o[1];
            if (q->packed_out) memcpy(q->packed_out, b + 16, 8);
            if (q->key_out) memcpy(q->key_out, b + 24, 80);
            if (q->best_set_out) memcpy(q->best_set_out, b + 104, (size_t)NG_);
            return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
        }
        if (!synced) bk_.sync();
        const size_t S = (size_t)opt_pending_s_;
        const size_t b_out = 8 * S, b_set = q->best_set_out ? ((size_t)NG_ + 7) & ~(size_t)7 : 0, b_key = q->key_out ? 80 * S : 0;
        const int32_t* o = (const int32_t*)opt_stage_;
        for (size_t i = 0; i < S; ++i) {
            if (q->best_out) q->best_out[i] = o[2 * i];
            if (q->n_best_out) q->n_best_out[i] = o[2 * i + 1];
        }
        if (q->best_set_out) memcpy(q->best_set_out, opt_stage_ + b_out, (size_t)NG_);
        if (q->key_out) memcpy(q->key_out, opt_stage_ + b_out + b_set, b_key);
        if (q->packed_out) memcpy(q->packed_out, opt_stage_ + b_out + b_set + b_key, 8 * S);
        opt_pending_q_ = nullptr; opt_stage_ = nullptr;
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }
    int sims() const { return n_sims_; }
    bool winners_only() const { return winners_only_; }
    // entries of the compacted winners' lists (after the expander query; waits for the device): a streamed batch needs every part's
    // share before it knows where the next part's lists start in the caller's arrays
    int32_t winners_total(int32_t* n_out) {
        if (!winners_only_ || !winners_ready_) return fail(CASIM_ERR_INVALID, "winners_only: no expander query ran");
        int32_t t = 0;
        bk_.d2h(&t, d_woff_ + winners_s_, 4);
        bk_.sync();
        *n_out = t;
        return bk_.ok() ? CASIM_OK : fail(CASIM_ERR_HIP, bk_.error());
    }

    const std::string& error() const { return err_; }
    BK& backend() { return bk_; }
    const DevTables& tables() const { return dt_; }
    int groups() const { return NG_; }
    int pegs() const { return G_; }
    bool csr_on_device() const { return csr_on_device_; }
    void set_one_shot(bool v) { one_shot_ = v; }   // before init(): see the end of init()
    // a part of a cut batch: what its PEG ids lack to be the whole batch's (casim_streams.h); fetch adds it on the device where it compacts
    // the lists anyway and says so (last_fetch_rebased), the other fetch paths leave it to the caller
    void set_order_id_base(int32_t b) { order_id_base_ = b; }
    bool last_fetch_rebased() const { return fetch_rebased_; }
    // before init(): parts of a streamed batch.  issue_order: the parts' big uploads are ordered on the device by an event chain, so a part hands the
    // turn on as soon as its tables are IN the queue — no wait for the device in between
    // `prev`: the backend of the part in front (null for part 0): this part's first big copy waits, on the device, for the event that part recorded
    // behind its last one
    void set_upload_gate(UploadGate* g, int index, bool issue_order = false, BK* prev = nullptr) {
        gate_ = g; gate_idx_ = index; gate_passed_ = false; gate_issue_order_ = issue_order; turn_prev_ = prev; turn_waited_ = false;
    }
    void pass_gate() {
        if (gate_ && !gate_passed_) {
            if (gate_issue_order_) bk_.record_turn_event();   // (behind this part's last big copy; the next part's first one waits for it on the device)
            gate_passed_ = true; gate_->pass(gate_idx_);
        }
    }
    // [0] the streaming feasibility kernel serves this problem, [1] lean, [2] mask31, [3] its workgroups
    void feasibility_info(int32_t out[4]) const {
        const bool stream = d_feas_rec_ != nullptr && fast_npt_ > 0 && (strided_ || feas_by_sim_) && !front_ && !(strided_ && strided_one_launch_);
        out[0] = stream ? 1 : 0; out[1] = stream && dt_.Wx == 0 && dt_.Wz == 0 && dt_.R <= 2; out[2] = stream ? ((feas_hi_ ? 0 : 1) | (feas_us_word_ >= 0 ? 2 : 0)) : 0;
        out[3] = stream ? ((n_sims_ + 7) / 8) * 8 * ((feas_len_ + 255) / 256) : 0;
    }
    bool uses_front() const { return front_; }
    bool uses_strided_lists() const { return strided_; }
    bool uses_rank_once() const { return strided_ && rank_once_ && !strided_one_launch_; }
    bool pack_in_lds() const { return pack_lds_; }
    int fast_npt() const { return fast_npt_; }
    int fast_lanes() const { return fast_npt_ > 0 ? (fast_i64_ ? 8 : fast_r_) : 0; }   // > 0: the register-resident packer handles this batch (2 / 4 int32 lanes; 8 = two int64 lanes)
    static constexpr int kOrderThreads = 256;

private:
    // the lean instantiation for batches without exclusion words whose (at most two) lanes are narrowed to int32: the headline's shape
    void launch_feas_sim(int gx, int gy, int block, size_t smem, const DevTables& t, uint64_t* bits, int wg, const int32_t* req32, const int32_t* fresh32) {
        const bool lean = t.Wx == 0 && t.Wz == 0 && t.R <= 2 && req32 != nullptr;
        if (d_feas_rec_ && req32 != nullptr) {   // the streaming form (round 5): group records through LDS, one accumulated word per cell, XCD-aware 1-D grid
            const int n_sims = gy, blocks = ((n_sims + 7) / 8) * 8 * gx;
            const bool term = feas_us_word_ < 0;
            if (getenv("CASIM_FEAS_TRACE")) fprintf(stderr, "[feas] feas_stream_kernel<%s, %s, %s> %d blocks x %d, %d simulations\n", lean ? "lean" : "full", feas_hi_ ? "hi" : "lo",
                                                    term ? "term" : "bit", blocks, block, n_sims);
            const uint32_t* rec = (const uint32_t*)d_feas_rec_;
#define CASIM_FEAS_LAUNCH(L, H, T) bk_.launch(feas_stream_kernel<L, H, T>, blocks, 1, block, feas_smem_, t, bits, wg, req32, rec, gx, n_sims, feas_us_word_, feas_us_bit_)
            if (lean) { if (feas_hi_) { if (term) CASIM_FEAS_LAUNCH(true, true, true); else CASIM_FEAS_LAUNCH(true, true, false); }
                        else { if (term) CASIM_FEAS_LAUNCH(true, false, true); else CASIM_FEAS_LAUNCH(true, false, false); } }
            else { if (feas_hi_) { if (term) CASIM_FEAS_LAUNCH(false, true, true); else CASIM_FEAS_LAUNCH(false, true, false); }
                   else { if (term) CASIM_FEAS_LAUNCH(false, false, true); else CASIM_FEAS_LAUNCH(false, false, false); } }
#undef CASIM_FEAS_LAUNCH
            return;
        }
        if (lean) bk_.launch(feas_sim_kernel<true>, gx, gy, block, smem, t, bits, wg, req32, fresh32);
        else bk_.launch(feas_sim_kernel<false>, gx, gy, block, smem, t, bits, wg, req32, fresh32);
    }
    // Uploads between begin_uploads() and end_uploads() are packed: the bytes go to the backend's staging buffer right away
    // (the caller's array may die), the device pointer is a slice of ONE slab, and end_uploads() issues the one copy.
    void begin_uploads(size_t bound) {
        up_host_ = (char*)bk_.stage(0, bound);
        up_dev_ = up_host_ ? (char*)dalloc(bound) : nullptr;
        up_cap_ = up_dev_ ? bound : 0; up_used_ = 0; up_flushed_ = 0; up_reserved_ = false;
        up_segs_.clear(); up_seg_bytes_ = 0;
    }
    // What is ready to travel and has not been handed to the backend yet: `up_segs_` (in slab order: staged ranges and page-locked columns
    // of the caller that go out from where they lie) followed by the staged range [up_flushed_, up_used_).  When the parts of a streamed
    // call take the link in turn (set_upload_gate, issue order) the first copy of a part waits, on the device, for the event the part in
    // front of it recorded behind its last big copy.
    void issue_uploads() {
        if (gate_issue_order_ && !gate_passed_ && turn_prev_ && !turn_waited_) { bk_.wait_turn_event(*turn_prev_); turn_waited_ = true; }
        for (const UpSeg& sg : up_segs_) bk_.h2d(up_dev_ + sg.at, sg.src ? sg.src : (const void*)(up_host_ + sg.at), sg.bytes);
        up_segs_.clear(); up_seg_bytes_ = 0;
        if (up_used_ > up_flushed_) { bk_.h2d(up_dev_ + up_flushed_, up_host_ + up_flushed_, up_used_ - up_flushed_); up_flushed_ = up_used_; }
    }
    void end_uploads() {
        if (gate_ && !gate_passed_) gate_->wait_turn(gate_idx_);
        if (up_dev_) issue_uploads();
        if (gate_issue_order_) pass_gate();   // (everything of this part is in its queue: the next part's tables wait for the event behind it)
        up_dev_ = up_host_ = nullptr; up_cap_ = 0;
    }
    // A big batch does not wait for its last column before the first one travels: every kUploadChunk bytes of staged columns go out
    // as a copy of their own (same slab, same stream), so the link works while the host stages the next columns and runs the gcd
    // pass.  Only columns that are final when up() returns: once somebody has RESERVED room to write in place later (up_reserve), the
    // rest goes out with end_uploads().  A single simulation stays one copy.
    static constexpr size_t kUploadChunk = (size_t)4 << 20;
    static constexpr size_t kDirectUploadMin = (size_t)1 << 20;   // columns below this are cheaper to stage than to ask about (hipPointerGetAttributes)
    static size_t direct_upload_min() {   // (CASIM_TEST_DIRECT_MIN: tests send small page-locked columns from where they lie)
        const char* e = getenv("CASIM_TEST_DIRECT_MIN");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)v : kDirectUploadMin;
    }
    static size_t upload_chunk() {   // (CASIM_TEST_UPLOAD_CHUNK: tests send small tables in many pieces)
        const char* e = getenv("CASIM_TEST_UPLOAD_CHUNK");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)v : kUploadChunk;
    }
    void flush_uploads_all() {   // everything staged so far goes out now (a kernel is about to read it)
        if (gate_ && !gate_passed_) gate_->wait_turn(gate_idx_);
        if (up_reserved_ || !up_dev_) return;
        issue_uploads();
    }
    void flush_uploads_early() {
        if (up_reserved_ || !up_dev_ || up_seg_bytes_ + (up_used_ - up_flushed_) < upload_chunk()) return;
        if (gate_ && !gate_passed_ && !gate_->my_turn(gate_idx_)) return;   // (not this part's turn on the link yet: keep staging)
        issue_uploads();
    }
    template <class T>
    const T* up(const T* src, size_t n) {
        if (n == 0 || !src) return nullptr;
        const size_t bytes = sizeof(T) * n, at = (up_used_ + 15) & ~(size_t)15;
        if (up_dev_ && at + bytes <= up_cap_) {
            // a big column in page-locked memory travels from where it lies: what is staged in front of it is listed first (the staged range
            // must not cover the column's slice of the slab: its bytes in the staging buffer are garbage), then the column itself — both go
            // out at the next flush that finds the link free for this part (a streamed call's parts take it in turn)
            if (bytes >= direct_upload_min() && !up_reserved_ && bk_.pinned(src)) {
                if (up_used_ > up_flushed_) { up_segs_.push_back({up_flushed_, nullptr, up_used_ - up_flushed_}); up_seg_bytes_ += up_used_ - up_flushed_; }
                up_segs_.push_back({at, (const void*)src, bytes}); up_seg_bytes_ += bytes;
                up_used_ = up_flushed_ = at + bytes;
                ++direct_uploads_;
                flush_uploads_early();
                return (const T*)(up_dev_ + at);
            }
            par_memcpy(up_host_ + at, src, bytes);
            up_used_ = at + bytes;
            flush_uploads_early();
            return (const T*)(up_dev_ + at);
        }
        T* d = (T*)dalloc(bytes);   // outside a packed section (or a bound that was too small): its own copy
        if (d) bk_.h2d(d, src, bytes);
        return d;
    }
    // room for n values inside the packed section, to be written in place (host pointer) instead of built in a vector and copied;
    // null when there is no packed section or no room (the caller falls back on up())
    template <class T>
    T* up_reserve(size_t n, const T** dev_out) {
        const size_t bytes = sizeof(T) * n, at = (up_used_ + 15) & ~(size_t)15;
        if (n == 0 || !up_dev_ || at + bytes > up_cap_) return nullptr;
        up_reserved_ = true;
        up_used_ = at + bytes;
        *dev_out = (const T*)(up_dev_ + at);
        return (T*)(up_host_ + at);
    }
    void* dalloc(size_t bytes) {
        if (bytes == 0) bytes = 8;
        void* p = bk_.alloc(bytes);
        if (p) allocs_.push_back(p);
        return p;
    }
    void release() {
        for (void* p : allocs_) bk_.free(p);
        allocs_.clear();
    }
    int32_t fail(int32_t code, const char* msg) { err_ = msg ? msg : ""; return code; }

    BK& bk_;
    DevTables dt_; DevResults dr_; PackScratch ps_; OrderScratch os_ = {nullptr, nullptr, nullptr}; FastScratch fs_ = {nullptr, nullptr, nullptr, nullptr};
    int G_ = 0, NG_ = 0, Wg_ = 0, fast_npt_ = 0, fast_r_ = 0;
    int32_t nnz_cap_ = 0;
    bool csr_on_device_ = false, pack_lds_ = true, order_lds_ = true, ready_ = false, ran_ = false;
    int fast_wx_ = 0;
    int32_t narrow_factor_[CASIM_KMAX_RES] = {1, 1, 1, 1, 1, 1, 1, 1};   // casim_pegs.req_unit / the packer's scale, per lane
    bool fast_i64_ = false;   // the register store on int64 lanes (lanes that do not narrow to 32 bits, R <= 2)
    bool fast_retry_ = false;
    SingletonRuns runs_;
    std::vector<int32_t> h_off_exp_, h_idx_m_;
    int pack_build_ = 0;      // casim_options.pack_build
    size_t pack_smem_ = 0, order_smem_ = 0;
    int order_threads_ = kOrderThreads;
    uint64_t* d_bits_ = nullptr; int32_t* d_counts_ = nullptr; int32_t* d_off_ = nullptr; int32_t* d_idx_ = nullptr; int32_t* d_block_sums_ = nullptr; int32_t* d_off_local_ = nullptr;
    uint8_t* d_opt_set_ = nullptr; int32_t* d_opt_out_ = nullptr; int64_t* d_opt_key_ = nullptr; int64_t* d_opt_packed_ = nullptr;
    uint8_t* d_opt_valid_ = nullptr; size_t opt_cap_ = 0;
    int n_sims_ = 0, max_sim_groups_ = 0, feas_len_ = 0;
    // feas_stream_kernel: [NG][16] group records; bits the batch's taint / selector words use at all (mask_or_kernel); upper halves in use;
    // where the NodeUnschedulable bit rides (-1: a term of its own); dynamic LDS of the launch
    uint64_t h_mask_used_[2] = {~0ull, ~0ull};
    uint32_t* d_feas_rec_ = nullptr; bool feas_hi_ = true; int feas_us_word_ = -1, feas_us_bit_ = 0; size_t feas_smem_ = 0;
    bool chain_ = false; int chain_passes_ = 0; int chain_passes_run_ = 0; int32_t* d_chain_redo_ = nullptr; int32_t* d_chain_marks_ = nullptr;   // casim_options.chain_last_index
    bool feas_by_sim_ = false;
    bool excl_vacuous_ = false;   // Wx > 1, but no template holds a marked pod and no bit has NEED polarity: the simulation-major feasibility kernels run with Wx = 0
    DevTables feas_tables() const { DevTables t = dt_; if (excl_vacuous_) t.Wx = 0; return t; }
    bool one_shot_ = false;
    UploadGate* gate_ = nullptr; int gate_idx_ = 0; bool gate_passed_ = false, gate_issue_order_ = false;
    BK* turn_prev_ = nullptr; bool turn_waited_ = false;
    bool ord_in_slab_ = false; size_t ord_off_ = 0, ord_cap_ = 0;   // order / placed inside the results slab (short lists)
    std::vector<int32_t> opt_host_; const casim_option_query* opt_pending_q_ = nullptr; int opt_pending_s_ = 0; const char* opt_stage_ = nullptr;
    bool opt_in_slab_ = false; size_t opt_off_ = 0; std::vector<char> opt_keep_;   // (the expander's answer as the last fetch() brought it)
    bool winners_only_ = false, winners_ready_ = false; int winners_s_ = 0; int32_t winners_total_ = 0;   // casim_options.winners_only
    int32_t* d_woff_ = nullptr; int32_t* d_worder_ = nullptr; int32_t* d_wplaced_ = nullptr; size_t woff_cap_ = 0;
    bool front_ = false, front_ran_ = false;   // feas + offsets + lists + order in ONE launch (front_kernel)
    bool rank_once_ = false; int32_t n_pairs_ = 0, rank_stride_ = 0; size_t rank_smem_ = 0;   // the orderer's ranks per (simulation, allocatable pair)
    const int32_t* d_pair_of_ = nullptr; const int32_t* d_pair_rep_ = nullptr; int32_t* d_ranks_ = nullptr;
    int32_t n_sorted_pairs_ = 0, n_checked_pairs_ = 0;   // pairs that sort / pairs proportional to their simulation's base pair (check, sort only on failure)
    const int32_t* d_pair_base_ = nullptr; const int32_t* d_sorted_list_ = nullptr; const int32_t* d_checked_list_ = nullptr; int32_t* d_pair_src_ = nullptr;
    bool strided_ = false, strided_one_launch_ = false; size_t front_sim_smem_ = 0; std::vector<int32_t> h_off_static_;   // batches: fixed-stride lists, front_sim_kernel
    int32_t* d_coff_ = nullptr; int32_t* d_corder_ = nullptr; int32_t* d_cplaced_ = nullptr;   // ... compacted at fetch time
    uint64_t* d_ticket_ = nullptr; uint32_t front_epoch_ = 0;
    static constexpr size_t kFrontMaxGroups = 1024;
    static constexpr size_t kDevGcdMin = (size_t)1 << 17;   // request values from which the gcd / int32 pass runs on the device
    std::vector<int32_t> h_off_;
    bool h_off_fresh_ = false;
    char* up_dev_ = nullptr; char* up_host_ = nullptr; size_t up_cap_ = 0, up_used_ = 0; size_t up_flushed_ = 0; bool up_reserved_ = false;
    struct UpSeg { size_t at; const void* src /* null: staged bytes of up_host_ */; size_t bytes; };
    std::vector<UpSeg> up_segs_; size_t up_seg_bytes_ = 0;   // listed for upload, not handed to the backend yet (issue_uploads)
    int direct_uploads_ = 0;   // columns copied straight from the caller's page-locked arrays
    int32_t order_id_base_ = 0; bool fetch_rebased_ = false;
    std::vector<uint64_t> zpol_host_, xpol_host_;
    char* res_slab_ = nullptr; size_t res_bytes_ = 0; int32_t* res_off_ = nullptr;
    int32_t* d_node_pods_ = nullptr; std::vector<int64_t> np_off_;
    std::vector<void*> allocs_;
    std::string err_;
};

}  // namespace casim

#include "casim_sched.h"
#include "casim_estimate.h"
