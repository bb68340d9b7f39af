// casim_streams.h — a batch of independent simulations as sub-batches on HIP streams of their own, INSIDE one casim_problem
// (casim_options.n_streams; VERDICT r2 next #2: the overlap used to live in Python, kubernetes_autoscaler_amd/streams.py).
//
// Why: the feasibility / ordering kernels of the scale-up path wait on memory, the packer on instruction issue; run back to
// back on one stream they leave each other's resource idle and every launch ends in a tail.  Cut by simulation, each part a
// ProblemT on a "lane" of the context (a backend with its own stream, memory pool and pinned staging buffers on the SAME
// device), the parts overlap (1.41 -> 1.06 ms per 4096 C2 simulations).
//
// Ordering contract towards the caller (who only knows the context's stream):
//   run()                fork: every lane waits for what the context's stream holds at that moment, then the parts are enqueued;
//   best_option_query()  with DEVICE outputs: join — the context's stream waits for every lane, so that work the caller enqueues
//                        there (an RCCL all-reduce on the keys) sees them; with host outputs the call synchronises the lanes;
//   fetch() / csr()      synchronise the lanes.
// A caller that leaves the context's stream alone between steps (resident tables, keys read at the end) gets steps that
// overlap each other as well: the forks then wait for nothing.
//
// The parts see VIEWS of the caller's tables (pointer offsets, no copies) — only peg_lo / peg_hi / sim_offsets are re-based
// into small arrays — and write their results straight into the caller's arrays at the part's offsets; PEG ids of `order`
// are shifted back to the numbering of the whole batch on the host.  Results are identical to the unstreamed call
// (tests/test_streams_emu.py, tests/test_gpu_round3.py).
//
// Backend concept additions:  void mark();  void wait_mark(BK& other);  bool idle();  void make_wait(void* raw_stream);  (record an
// event on the own stream / make the own stream wait for the other's last mark / nothing is pending on the own stream / make a stream of
// the CALLER's wait for the own last mark; trivial for the synchronous emulator).
#pragma once
#include <memory>
#include <thread>

#include "casim_pipeline.h"

namespace casim {

// CASIM_CALL_TIMELINE=1: where the wall time of an enter -> return call goes — every worker stamps its milestones (microseconds since the call
// entered estimate()), printed to stderr when the call returns.  Diagnostics only (tests/tools/enter_return_timeline.py).
struct CallTimeline {
    bool on = false;
    std::chrono::steady_clock::time_point t0;
    std::mutex mu;
    std::vector<std::string> lines;
    void start() { static const bool env = getenv("CASIM_CALL_TIMELINE") && atoi(getenv("CASIM_CALL_TIMELINE")) != 0; on = env; if (on) { t0 = std::chrono::steady_clock::now(); lines.clear(); } }
    void stamp(int part, const char* what) {
        if (!on) return;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        char buf[160]; snprintf(buf, sizeof buf, "[timeline] part %d %-34s %9.1f us", part, what, us);
        std::lock_guard<std::mutex> l(mu); lines.emplace_back(buf);
    }
    void dump() { if (!on) return; for (auto& l : lines) fprintf(stderr, "%s\n", l.c_str()); fprintf(stderr, "[timeline] --\n"); }
};

template <class BK>
class StreamedProblemT {
public:
    StreamedProblemT(BK& primary, std::vector<BK*> lanes) : primary_(primary), lanes_(std::move(lanes)) {}

    // a batch is cut when it holds at least two simulations, the schedulable subsets are derived on the device from per-group
    // candidate ranges, and the per-node pod lists are not asked for (their compaction runs over the whole batch)
    static bool eligible(const casim_pegs* p, const casim_groups* g, const casim_options* o) {
        return p && g && o && o->n_streams > 1 && g->n_sims >= 2 && g->sim_offsets && !g->peg_offsets && g->peg_lo && g->peg_hi && !o->node_pods &&
               g->n_groups > 0;
    }

    struct Part {
        int s0 = 0, s1 = 0, g0 = 0, g1 = 0, p0 = 0, p1 = 0;
        casim_pegs pv; casim_groups gv;
        std::vector<int32_t> lo, hi, so;
        std::unique_ptr<ProblemT<BK>> prob;
        int32_t rc = CASIM_OK;
    };

    // cut + upload.  `threads`: every part is prepared (table packing into its lane's pinned buffer, H2D) by a thread of its own
    int32_t init(const casim_pegs* p, const casim_groups* g, const casim_options* o, bool threads) {
        if (!eligible(p, g, o)) return fail(CASIM_ERR_INVALID, "batch cannot be cut into streamed parts");
        const int S = g->n_sims;
        if (g->sim_offsets[0] != 0 || g->sim_offsets[S] != g->n_groups) return fail(CASIM_ERR_INVALID, "sim_offsets must run from 0 to n_groups");
        for (int s = 0; s < S; ++s) if (g->sim_offsets[s + 1] < g->sim_offsets[s]) return fail(CASIM_ERR_INVALID, "sim_offsets not monotone");
        for (int i = 0; i < g->n_groups; ++i)
            if (g->peg_lo[i] < 0 || g->peg_hi[i] < g->peg_lo[i] || g->peg_hi[i] > p->n_pegs) return fail(CASIM_ERR_INVALID, "peg_lo / peg_hi out of range");
        int K = (int)lanes_.size();
        if (K > o->n_streams) K = o->n_streams;
        if (K > S) K = S;
        if (K < 1) return fail(CASIM_ERR_INVALID, "no lane");
        opts_ = *o; opts_.n_streams = 0;
        n_groups_ = g->n_groups; n_sims_ = S; has_gid_ = g->global_id != nullptr;
        parts_.clear(); parts_.resize((size_t)K);
        for (int i = 0; i < K; ++i) cut(parts_[(size_t)i], p, g, (int)((int64_t)S * i / K), (int)((int64_t)S * (i + 1) / K));
        // The parts take the link IN TURN (round 6): part i's big uploads wait — on the device, by an event — for the last big upload of part
        // i - 1, and the host only keeps the ISSUE order (UploadGate in issue order: part i enqueues its first copy when part i - 1 has
        // enqueued its last one and recorded the event; nobody waits for the device).  Part 0's tables arrive first at the full rate of the
        // link and its kernels run under the uploads behind it; with the copies of four streams sharing the link every part's tables
        // arrive at the same late moment.  Taking turns by WAITING for each part's copies on the host (round 4's gate, CASIM_UPLOAD_GATE=1
        // with CASIM_UPLOAD_FIFO=0) measured neutral to 9 % slower (profiles/r09h_upload_gate_ab.txt).  CASIM_UPLOAD_GATE=0 / 1 forces the gate.
        UploadGate gate;
        const bool fifo = pipeline_ && K > 1 && lanes_[0]->turns_enabled();
        const int gate_env = getenv("CASIM_UPLOAD_GATE") ? atoi(getenv("CASIM_UPLOAD_GATE")) : -1;
        const bool use_gate = gate_env >= 0 ? gate_env != 0 : fifo;
        auto work = [&](int i) {
            Part& pt = parts_[(size_t)i];
            tl_.stamp(i, "worker starts");
            lanes_[(size_t)i]->bind();
            pt.prob.reset(new ProblemT<BK>(*lanes_[(size_t)i]));
            if (use_gate) pt.prob->set_upload_gate(&gate, i, /*issue_order=*/fifo, fifo && i > 0 ? lanes_[(size_t)i - 1] : nullptr);
            pt.prob->set_one_shot(pipeline_q_ != nullptr || pipeline_);
            pt.prob->set_order_id_base(pt.p0);
            pt.rc = pt.prob->init(&pt.pv, &pt.gv, &opts_);
            tl_.stamp(i, "init returned (tables enqueued)");
            // enter -> return in one call (estimate()): the part's kernels and its expander reduce are enqueued by the part's own worker
            // as soon as ITS tables are up — under the uploads of the parts behind it in the turn order
            if (pipeline_ && pt.rc == CASIM_OK) {
                if (pipeline_fork_) lanes_[(size_t)i]->wait_mark(primary_);
                pt.rc = pt.prob->run();
                tl_.stamp(i, "kernels enqueued");
                if (pt.rc == CASIM_OK && pipeline_q_) pt.rc = part_query((size_t)i, pipeline_q_);
                tl_.stamp(i, "expander reduce returned");
            }
            // ... and fetched by it as well: a part's results travel while the parts behind it still compute.  The lists of part i start where
            // the lists of parts 0 .. i - 1 end, so every worker publishes its total as soon as it knows it and waits for its predecessors'
            // (a joined fetch — totals part by part, then a second round of workers — left a tail of ~0.6 ms of small copies and waits behind
            // the last kernel of a headline call: profiles/r14d_enter_return_trace.txt)
            if (pipeline_ && pipeline_out_) {
                int32_t nnz = 0;
                if (pt.rc == CASIM_OK) {
                    const bool winners = opts_.winners_only != 0 && (pipeline_out_->order || pipeline_out_->placed);
                    pt.rc = winners ? pt.prob->winners_total(&nnz) : pt.prob->csr(&nnz, nullptr);
                }
                int64_t base = 0;
                {
                    std::unique_lock<std::mutex> l(tot_mu_);
                    totals_[(size_t)i] = pt.rc == CASIM_OK ? nnz : 0; total_known_[(size_t)i] = 1;   // (a failed part still reports: nobody waits for ever)
                    tot_cv_.notify_all();
                    tot_cv_.wait(l, [&] { for (int k = 0; k < i; ++k) if (!total_known_[(size_t)k]) return false; return true; });
                    for (int k = 0; k < i; ++k) base += totals_[(size_t)k];
                }
                tl_.stamp(i, "fetch starts");
                if (pt.rc == CASIM_OK) {
                    if (base + nnz > 0x7fffffffll) pt.rc = CASIM_ERR_INVALID;
                    else fetch_part(i, pipeline_out_, (int32_t)base, (int32_t)(base + nnz));
                }
                tl_.stamp(i, "fetch done");
            }
        };
        if (pipeline_ && pipeline_out_) { totals_.assign((size_t)K, 0); total_known_.assign((size_t)K, 0); }
        each(work, threads);
        for (auto& pt : parts_) if (pt.rc != CASIM_OK) return fail(pt.rc, pt.prob->error().c_str());
        ready_ = true;
        return CASIM_OK;
    }

    int32_t run() {
        if (!ready_) return fail(CASIM_ERR_INVALID, "problem not initialised");
        // fork — unless the context's stream holds nothing (a stream query, no device work): an event wait in front of every
        // step cost the resident loop 15 % (1.22 vs 1.06 ms per 4096 C2 simulations, profiles/r03a_*)
        primary_.bind();
        static const int fork_mode = getenv("CASIM_FORK_MODE") ? atoi(getenv("CASIM_FORK_MODE")) : 1;   // (experiments: 0 never, 1 when busy, 2 always)
        const bool need_fork = fork_mode == 2 || (fork_mode == 1 && !primary_.idle());
        n_forks_ += need_fork ? 1 : 0;
        if (need_fork) primary_.mark();
        for (size_t i = 0; i < parts_.size(); ++i) {
            if (need_fork) lanes_[i]->wait_mark(primary_);
            const int32_t rc = parts_[i].prob->run();
            if (rc != CASIM_OK) return fail(rc, parts_[i].prob->error().c_str());
        }
        ran_ = true;
        return CASIM_OK;
    }
    // fork only (timed callers enqueue the phases of a part themselves)
    void fork() { primary_.bind(); if (primary_.idle()) return; primary_.mark(); for (size_t i = 0; i < parts_.size(); ++i) lanes_[i]->wait_mark(primary_); }
    void join() { for (size_t i = 0; i < parts_.size(); ++i) { lanes_[i]->mark(); primary_.wait_mark(*lanes_[i]); } }
    void sync_all() { for (size_t i = 0; i < parts_.size(); ++i) lanes_[i]->sync(); }
    void set_ran() { ran_ = true; }

    // whole-batch CSR offsets of order / placed
    int32_t csr(int32_t* nnz_out, int32_t* offsets_out) {
        if (!ready_ || !ran_) return fail(CASIM_ERR_INVALID, "run the problem first");
        int32_t base = 0;
        std::vector<int32_t> loc;
        if (offsets_out) offsets_out[0] = 0;
        for (auto& pt : parts_) {
            const int n = pt.g1 - pt.g0;
            loc.assign((size_t)n + 1, 0);
            int32_t nnz = 0;
            const int32_t rc = pt.prob->csr(&nnz, loc.data());
            if (rc != CASIM_OK) return fail(rc, pt.prob->error().c_str());
            if (offsets_out) for (int k = 1; k <= n; ++k) offsets_out[pt.g0 + k] = base + loc[(size_t)k];
            base += nnz;
        }
        if (nnz_out) *nnz_out = base;
        return CASIM_OK;
    }

    int32_t fetch(casim_results* out, bool threads) {
        if (!ready_ || !ran_) return fail(CASIM_ERR_INVALID, "nothing to fetch: run the problem first");
        if (!out) return fail(CASIM_ERR_INVALID, "null results");
        // the parts' shares of order / placed: their nnz (device-side CSR) first
        std::vector<int32_t> base(parts_.size() + 1, 0);
        const bool winners = opts_.winners_only != 0 && (out->order || out->placed);   // the parts' compacted winners' lists, back to back
        for (size_t i = 0; i < parts_.size(); ++i) {
            int32_t nnz = 0;
            const int32_t rc = winners ? parts_[i].prob->winners_total(&nnz) : parts_[i].prob->csr(&nnz, nullptr);
            if (rc != CASIM_OK) return fail(rc, parts_[i].prob->error().c_str());
            base[i + 1] = base[i] + nnz;
        }
        tl_.stamp(-1, "list totals known");
        auto work = [&](int i) { tl_.stamp(i, "fetch starts"); fetch_part(i, out, base[(size_t)i], base[(size_t)i + 1]); tl_.stamp(i, "fetch done"); };
        each(work, threads);
        for (auto& pt : parts_) if (pt.rc != CASIM_OK) return fail(pt.rc, pt.prob->error().c_str());
        return CASIM_OK;
    }

    // part i's results into its slice of the caller's arrays: groups [g0, g1), list entries [base, base_end)
    void fetch_part(int i, casim_results* out, int32_t base, int32_t base_end) {
        Part& pt = parts_[(size_t)i];
        lanes_[(size_t)i]->bind();
        casim_results r; memset(&r, 0, sizeof r);
        auto at = [&](auto* ptr, int64_t off) { return ptr ? ptr + off : ptr; };
        r.node_count = at(out->node_count, pt.g0); r.pods_scheduled = at(out->pods_scheduled, pt.g0); r.nodes_added = at(out->nodes_added, pt.g0);
        r.limiter_nodes = at(out->limiter_nodes, pt.g0); r.last_index_out = at(out->last_index_out, pt.g0); r.status = at(out->status, pt.g0);
        r.req_cpu_sum = at(out->req_cpu_sum, pt.g0); r.req_mem_sum = at(out->req_mem_sum, pt.g0);
        r.order = at(out->order, base); r.placed = at(out->placed, base);
        pt.rc = pt.prob->fetch(&r);
        if (pt.rc == CASIM_OK && r.order && pt.p0 != 0 && !pt.prob->last_fetch_rebased()) {   // PEG ids back in the numbering of the whole batch (host loop: small fetches only)
            const int32_t n = base_end - base, add = pt.p0;
            for (int32_t k = 0; k < n; ++k) r.order[k] += add;
        }
    }

    // The expander chain, one reduce per simulation; every part fills its slice of the caller's outputs.
    int32_t best_option_query(const casim_option_query* q) {
        if (!ready_ || !ran_) return fail(CASIM_ERR_INVALID, "run the problem first");
        if (!q) return fail(CASIM_ERR_INVALID, "null query");
        if (!q->per_sim) return fail(CASIM_ERR_INVALID, "a streamed batch reduces per simulation (per_sim = 1)");
        for (size_t i = 0; i < parts_.size(); ++i) {
            const int32_t rc = part_query(i, q);
            if (rc != CASIM_OK) return fail(rc, parts_[i].prob->error().c_str());
        }
        if (q->dev_key_out || q->dev_packed_out) {
            if (q->join_stream) { for (size_t i = 0; i < parts_.size(); ++i) { lanes_[i]->mark(); lanes_[i]->make_wait(q->join_stream); } }   // the caller's stream waits, not ours
            else join();
        }
        return CASIM_OK;
    }
    int32_t part_query(size_t i, const casim_option_query* q) {
        Part& pt = parts_[i];
        lanes_[i]->bind();
        casim_option_query lq = *q;
        lq.per_sim = 1;
        lq.group_id_base = q->group_id_base + (has_gid_ ? 0 : pt.g0);
        if (q->valid) lq.valid = q->valid + pt.g0;
        if (q->best_out) lq.best_out = q->best_out + pt.s0;
        if (q->n_best_out) lq.n_best_out = q->n_best_out + pt.s0;
        if (q->best_set_out) lq.best_set_out = q->best_set_out + pt.g0;
        if (q->key_out) lq.key_out = q->key_out + 10 * (int64_t)pt.s0;
        if (q->packed_out) lq.packed_out = q->packed_out + pt.s0;
        if (q->dev_key_out) lq.dev_key_out = (char*)q->dev_key_out + 80 * (int64_t)pt.s0;
        if (q->dev_packed_out) lq.dev_packed_out = (char*)q->dev_packed_out + 8 * (int64_t)pt.s0;
        const int32_t rc = pt.prob->best_option_query(&lq);
        if (rc != CASIM_OK) return rc;   // (the caller reads the part's message: several parts may be in here at once)
        if (q->best_out) for (int s = pt.s0; s < pt.s1; ++s) if (q->best_out[s] >= 0) q->best_out[s] += pt.g0;   // index inside the whole batch
        return CASIM_OK;
    }

    // enter -> return in one go: every part is uploaded, run, reduced and fetched by its own worker, so that the upload of one
    // part overlaps the kernels of another and the result copies of a third (wall = enter -> return)
    int32_t estimate(const casim_pegs* p, const casim_groups* g, const casim_options* o, casim_results* out, const casim_option_query* q, bool threads) {
        if (q && !q->per_sim) return fail(CASIM_ERR_INVALID, "a streamed batch reduces per simulation (per_sim = 1)");
        // fork once, up front: every lane waits for what the context's stream holds NOW (nothing, for a caller that leaves it alone)
        tl_.start();
        primary_.bind();
        pipeline_fork_ = !primary_.idle();
        if (pipeline_fork_) { primary_.mark(); ++n_forks_; }
        // (device outputs of the expander join into the caller's stream below: those calls fetch after the join, the old way)
        const bool fetch_in_workers = out != nullptr && !(q && (q->dev_key_out || q->dev_packed_out)) && !getenv("CASIM_JOINED_FETCH");
        pipeline_ = true; pipeline_q_ = q; pipeline_out_ = fetch_in_workers ? out : nullptr;
        int32_t rc = init(p, g, o, threads);   // upload, run, reduce and fetch every part, part by part (see init)
        pipeline_ = false; pipeline_q_ = nullptr; pipeline_out_ = nullptr;
        if (rc != CASIM_OK) return rc;
        ran_ = true;
        if (fetch_in_workers) { tl_.stamp(-1, "parts joined (fetched by their workers)"); tl_.dump(); return CASIM_OK; }
        if (q && (q->dev_key_out || q->dev_packed_out)) {
            if (q->join_stream) { for (size_t i = 0; i < parts_.size(); ++i) { lanes_[i]->mark(); lanes_[i]->make_wait(q->join_stream); } }
            else join();
        }
        // (measured, round 4: fetching every part from its upload worker as soon as the parts in front of it have reported their list lengths —
        // D2H under the uploads of the parts behind — changes nothing: 5.3-5.7 ms either way for the every-list form of the headline call)
        tl_.stamp(-1, "parts joined");
        if (out) rc = fetch(out, threads);
        tl_.stamp(-1, "fetch returned");
        tl_.dump();
        return rc;
    }

    int32_t set_group_result(int32_t ng, const casim_cluster_estimate_result* r) {
        for (auto& pt : parts_) if (ng >= pt.g0 && ng < pt.g1) {
            const int32_t rc = pt.prob->set_group_result(ng - pt.g0, r);
            return rc == CASIM_OK ? rc : fail(rc, pt.prob->error().c_str());
        }
        return fail(CASIM_ERR_INVALID, "bad group index");
    }

    size_t n_parts() const { return parts_.size(); }
    Part& part(size_t i) { return parts_[i]; }
    BK& lane(size_t i) { return *lanes_[i]; }
    int groups() const { return n_groups_; }
    int sims() const { return n_sims_; }
    int64_t forks() const { return n_forks_; }
    const std::string& error() const { return err_; }

private:
    template <class T> static const T* off(const T* p, int64_t n) { return p ? p + n : p; }
    void cut(Part& pt, const casim_pegs* p, const casim_groups* g, int s0, int s1) {
        pt.s0 = s0; pt.s1 = s1; pt.g0 = g->sim_offsets[s0]; pt.g1 = g->sim_offsets[s1];
        int lo = p->n_pegs, hi = 0;
        for (int i = pt.g0; i < pt.g1; ++i) { lo = g->peg_lo[i] < lo ? g->peg_lo[i] : lo; hi = g->peg_hi[i] > hi ? g->peg_hi[i] : hi; }
        if (hi < lo) { lo = 0; hi = 0; }
        pt.p0 = lo; pt.p1 = hi;
        const int R = p->n_res;
        casim_pegs& v = pt.pv; v = *p;
        v.n_pegs = hi - lo;
        v.req = off(p->req, (int64_t)lo * R); v.req32 = off(p->req32, (int64_t)lo * R); v.count = off(p->count, lo); v.flags = off(p->flags, lo);
        v.tol_mask = off(p->tol_mask, (int64_t)lo * p->w_taint); v.sel_mask = off(p->sel_mask, (int64_t)lo * p->w_label);
        v.excl_block = off(p->excl_block, (int64_t)lo * p->w_excl); v.excl_mark = off(p->excl_mark, (int64_t)lo * p->w_excl);
        v.zone_block = off(p->zone_block, (int64_t)lo * p->w_zone); v.zone_mark = off(p->zone_mark, (int64_t)lo * p->w_zone);
        v.fp_cpu = off(p->fp_cpu, lo); v.fp_mem = off(p->fp_mem, lo);
        casim_groups& w = pt.gv; w = *g;
        const int64_t a = pt.g0;
        w.n_groups = pt.g1 - pt.g0;
        w.alloc = off(g->alloc, a * R); w.init_req = off(g->init_req, a * R); w.allowed_pods = off(g->allowed_pods, a); w.init_pods = off(g->init_pods, a);
        w.flags = off(g->flags, a); w.taint_mask = off(g->taint_mask, a * p->w_taint); w.label_mask = off(g->label_mask, a * p->w_label);
        w.init_excl = off(g->init_excl, a * p->w_excl); w.init_zone = off(g->init_zone, a * p->w_zone); w.zone_valid = off(g->zone_valid, a * p->w_zone);
        w.max_nodes = off(g->max_nodes, a); w.existing_nodes = off(g->existing_nodes, a); w.last_index = off(g->last_index, a);
        w.cap_cpu = off(g->cap_cpu, a); w.cap_mem = off(g->cap_mem, a); w.waste_cpu = off(g->waste_cpu, a); w.waste_mem = off(g->waste_mem, a);
        w.global_id = off(g->global_id, a);
        w.peg_offsets = nullptr; w.peg_index = nullptr;
        pt.lo.resize((size_t)w.n_groups); pt.hi.resize((size_t)w.n_groups);
        for (int i = 0; i < w.n_groups; ++i) { pt.lo[(size_t)i] = g->peg_lo[pt.g0 + i] - lo; pt.hi[(size_t)i] = g->peg_hi[pt.g0 + i] - lo; }
        pt.so.resize((size_t)(s1 - s0) + 1);
        for (int s = s0; s <= s1; ++s) pt.so[(size_t)(s - s0)] = g->sim_offsets[s] - pt.g0;
        static const int32_t z32[1] = {0};
        w.peg_lo = w.n_groups ? pt.lo.data() : z32; w.peg_hi = w.n_groups ? pt.hi.data() : z32;
        w.n_sims = s1 - s0; w.sim_offsets = pt.so.data();
    }
    // the parts' workers: tasks of the process-wide pool (casim_pipeline.h: HostPool), handed out in part order — a worker that waits for the
    // part in front of it (upload turn, list base) waits for a task that is running
    template <class F>
    void each(F&& f, bool threads) {
        const int K = (int)parts_.size();
        if (!threads || K == 1) { for (int i = 0; i < K; ++i) f(i); return; }
        HostPool::get().run(K, f);
    }
    int32_t fail(int32_t code, const char* msg) { err_ = msg ? msg : ""; return code; }

    BK& primary_;
    std::vector<BK*> lanes_;
    std::vector<Part> parts_;
    casim_options opts_;
    int n_groups_ = 0, n_sims_ = 0;
    bool has_gid_ = false, ready_ = false, ran_ = false;
    bool pipeline_ = false, pipeline_fork_ = false; const casim_option_query* pipeline_q_ = nullptr;   // estimate(): parts run from their init workers
    casim_results* pipeline_out_ = nullptr;                                                              // ... and are fetched by them
    std::mutex tot_mu_; std::condition_variable tot_cv_; std::vector<int32_t> totals_; std::vector<char> total_known_;   // list totals, part by part

    int64_t n_forks_ = 0;
    std::string err_;
    CallTimeline tl_;
};

}  // namespace casim
