// casim_emu_api.cpp — host backend for casim_pipeline.h running the product kernels under the
// wave emulator (TEST INFRASTRUCTURE ONLY; built by tests/emu/Makefile into
// tests/emu/libcasim_emu.so; never part of libcasim.so).
#define CASIM_HOST_EMU 1
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "casim_emu.h"
#include "../../kubernetes_autoscaler_amd/csrc/casim_pipeline.h"
#include "../../kubernetes_autoscaler_amd/csrc/casim_multi.h"
#include "../../kubernetes_autoscaler_amd/csrc/casim_streams.h"

namespace {
struct EmuBackend {
    size_t lds = 160 * 1024;
    // Device memory is NOT zero when a kernel first sees it (the product's backend hands out blocks of its pool again, with whatever the last
    // call left in them): 0xA5 bytes here, as in the emulated LDS, so that a kernel that reads what nobody wrote computes something else
    // than the oracle.  CASIM_EMU_ZERO_ALLOC=1: zeroed blocks (to tell such a read from other differences).
    static bool zero_alloc() { static const bool z = getenv("CASIM_EMU_ZERO_ALLOC") && atoi(getenv("CASIM_EMU_ZERO_ALLOC")) != 0; return z; }
    void* alloc(size_t b) { void* p = malloc(b ? b : 1); if (p && b) memset(p, zero_alloc() ? 0 : 0xA5, b); return p; }
    void free(void* p) { ::free(p); }
    void h2d(void* d, const void* s, size_t n) { memcpy(d, s, n); }
    void record_turn_event() { ++turn_records; }
    void wait_turn_event(EmuBackend& prev) { if (prev.turn_records > 0) ++turn_waits; }
    int turn_records = 0, turn_waits = 0;
    // CASIM_EMU_FIFO=1: the parts of a streamed call take the link in turn, in issue order (the device backend's CASIM_UPLOAD_FIFO=1: an event
    // chain on the lanes' streams); CASIM_EMU_PINNED=1: every column counts as page-locked (the direct-upload path of ProblemT::up)
    bool turns_enabled() const { const char* e = getenv("CASIM_EMU_FIFO"); return e && atoi(e) != 0; }
    void d2h(void* d, const void* s, size_t n) { memcpy(d, s, n); }
    void zero(void* d, size_t n) { memset(d, 0, n); }
    void fill8(void* d, int v, size_t n) { memset(d, v, n); }
    void sync() {}
    void bind() {}
    void mark() {}
    bool idle() { return true; }
    void wait_mark(EmuBackend&) {}
    void make_wait(void*) {}
    std::vector<char> staging[2];
    void* stage_if_fits(int which, size_t bytes) { return staging[which & 1].size() >= bytes ? staging[which & 1].data() : nullptr; }
    void* stage(int which, size_t bytes) { if (staging[which & 1].size() < bytes) staging[which & 1].resize(bytes); return staging[which & 1].data(); }
    size_t lds_budget() const { return lds; }
    bool pinned(const void*) const { const char* e = getenv("CASIM_EMU_PINNED"); return e && atoi(e) != 0; }
    bool ok() const { return true; }
    const char* error() const { return ""; }
    template <class K, class... A>
    void launch(K kernel, int gx, int gy, int block, size_t smem, A... args) {
        casim_emu::launch(gx, gy, block, smem, [&]() { kernel(args...); });
    }
    void prepare_pack_fast(int, int, int, int) {}
    void launch_pack_fast(int /*build: one build under the emulator*/, int lanes, int slots_per_lane, int excl_words, int n_groups, const DevTables& t, const DevResults& res, const FastScratch& fs) {
#define CASIM_EMU_FAST(R, N, X) do { if constexpr ((R) == 8) launch(casim::pack_fast64_kernel<N, X>, n_groups, 1, 64, (size_t)0, t, res, fs); \
                                     else launch(casim::pack_fast_kernel<((R) == 8 ? 2 : (R)), N, X>, n_groups, 1, 64, (size_t)0, t, res, fs); } while (0)
        CASIM_FAST_DISPATCH(CASIM_EMU_FAST, lanes, slots_per_lane, excl_words);
#undef CASIM_EMU_FAST
    }
};
thread_local std::string g_err;
}  // namespace

#define EMU_API __attribute__((visibility("default")))
extern "C" {

static int32_t g_last_front = 0, g_last_lanes = 0;
EMU_API const char* emu_last_error() { return g_err.c_str(); }

// lds_budget_bytes <= 0 keeps the default (160 KiB); a tiny value forces the HBM-scratch variants.
EMU_API int32_t emu_estimate_batch(const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                           casim_results* out, int64_t lds_budget_bytes, int32_t* nnz_out, int32_t* offsets_out,
                           const int32_t* kinds, int32_t n_kinds, int32_t group_id_base, int32_t* best_out /*[2]*/,
                           uint8_t* best_set_out, int64_t* key_out /*[10]*/) {
    EmuBackend bk;
    if (lds_budget_bytes > 0) bk.lds = (size_t)lds_budget_bytes;
    casim::ProblemT<EmuBackend> p(bk);
    int32_t rc = p.init(pegs, groups, opts);
    if (rc == CASIM_OK) rc = p.run();
    if (rc == CASIM_OK) rc = p.fetch(out);
    if (rc == CASIM_OK && (nnz_out || offsets_out)) rc = p.csr(nnz_out, offsets_out);
    if (rc == CASIM_OK && n_kinds >= 0 && best_out)
        rc = p.best_option(kinds, n_kinds, group_id_base, &best_out[0], &best_out[1], best_set_out, key_out, nullptr);
    if (rc != CASIM_OK) g_err = p.error();
    g_last_front = p.uses_front() ? 1 : (p.uses_strided_lists() ? 2 : 0);
    g_last_lanes = p.fast_lanes() * 100 + p.fast_npt();
    return rc;
}

// Same with the general expander query (validity mask, one reduce per simulation of the batch).
EMU_API int32_t emu_estimate_batch_query(const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                                 casim_results* out, int64_t lds_budget_bytes, int32_t* nnz_out, int32_t* offsets_out,
                                 const casim_option_query* q) {
    EmuBackend bk;
    if (lds_budget_bytes > 0) bk.lds = (size_t)lds_budget_bytes;
    casim::ProblemT<EmuBackend> p(bk);
    // the sequence of casim_estimate_batch_query (csrc/casim_engine.hip): one-shot problem, the expander's answer left in flight until
    // the fetch has waited, offsets from the fetch
    // (CASIM_EMU_RESIDENT=1: the resident form instead — casim_problem_create + run: init ends with a wait, which is where a problem learns
    // facts about its tables from the device, e.g. feas_stream_kernel's narrow-dictionary instantiation)
    p.set_one_shot(getenv("CASIM_EMU_RESIDENT") == nullptr);
    int32_t rc = p.init(pegs, groups, opts);
    if (rc == CASIM_OK) rc = p.run();
    const bool one_wait = q && out;
    if (rc == CASIM_OK && q) rc = p.best_option_query(q, /*defer_sync=*/one_wait);
    if (rc == CASIM_OK) rc = p.fetch(out);
    if (one_wait) { const int32_t rc2 = p.best_option_finish(/*synced=*/rc == CASIM_OK); if (rc == CASIM_OK) rc = rc2; }
    if (rc == CASIM_OK && (nnz_out || offsets_out)) rc = p.csr(nnz_out, offsets_out);
    if (rc != CASIM_OK) g_err = p.error();
    g_last_front = p.uses_front() ? 1 : (p.uses_strided_lists() ? 2 : 0);
    g_last_lanes = p.fast_lanes() * 100 + p.fast_npt();
    return rc;
}
// The host pool (casim_pipeline.h: HostPool) under the patterns the library uses it in, `rounds` times from `callers` threads at once: tasks that wait for
// the task in front of them (the parts' turn order), each cutting a loop over the pool from inside (a part that stages its tables).  Returns 0, or
// the number of wrong sums; a deadlock shows as the test's timeout.
EMU_API int32_t emu_pool_selftest(int32_t rounds, int32_t callers, int32_t tasks) {
    std::atomic<int> bad{0};
    auto one_caller = [&](int c) {
        std::vector<int64_t> data((size_t)1 << 16);
        for (size_t i = 0; i < data.size(); ++i) data[i] = (int64_t)(i % 97) + c;
        int64_t want = 0;
        for (int64_t v : data) want += v;
        for (int r = 0; r < rounds; ++r) {
            std::vector<std::atomic<int>> turn((size_t)tasks);
            for (auto& t : turn) t.store(0);
            std::vector<int64_t> sums((size_t)tasks, 0);
            casim::HostPool::get().run(tasks, [&](int i) {
                if (i > 0) while (turn[(size_t)i - 1].load(std::memory_order_acquire) == 0) std::this_thread::yield();   // the part in front passes its turn
                int64_t parts[casim::kHostLoopThreads] = {0, 0, 0, 0};
                casim::par_for(data.size(), 1024, [&](size_t lo, size_t hi, int t) { int64_t a = 0; for (size_t k = lo; k < hi; ++k) a += data[k]; parts[t] += a; });
                turn[(size_t)i].store(1, std::memory_order_release);
                casim::par_for(data.size(), 4096, [&](size_t lo, size_t hi, int t) { int64_t a = 0; for (size_t k = lo; k < hi; ++k) a += data[k]; parts[t] += a; });
                sums[(size_t)i] = parts[0] + parts[1] + parts[2] + parts[3];
            });
            for (int i = 0; i < tasks; ++i) if (sums[(size_t)i] != 2 * want) bad.fetch_add(1);
        }
    };
    std::vector<std::thread> th;
    for (int c = 1; c < callers; ++c) th.emplace_back(one_caller, c);
    one_caller(0);
    for (auto& t : th) t.join();
    return bad.load() + (casim::HostPool::get().workers() < 0 ? 1 : 0);
}
EMU_API int32_t emu_pool_workers() { return casim::HostPool::get().workers(); }

// ProblemT::init alone, `iters` times (host-side cost of an enter -> return call's table preparation: CASIM_INIT_TIMING=1 prints the stages;
// the few small kernels init launches run under the emulator, the stage that holds them says so)
EMU_API int32_t emu_init_only(const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts, int32_t iters) {
    int32_t rc = CASIM_OK;
    for (int32_t i = 0; i < iters && rc == CASIM_OK; ++i) {
        EmuBackend bk;
        casim::ProblemT<EmuBackend> p(bk);
        p.set_one_shot(true);
        rc = p.init(pegs, groups, opts);
        if (rc != CASIM_OK) g_err = p.error();
    }
    return rc;
}
// 1 when the last emu_estimate_batch_query ran feasibility / offsets / lists / order as ONE launch (front_kernel), 2: front_sim_kernel with fixed-stride lists
EMU_API int32_t emu_last_front() { return g_last_front; }
// packer of the last emu_estimate_batch(_query): lanes * 100 + node slots per lane (lanes 2 / 4: int32 register store, 8: two int64 lanes, 0: LDS store)
EMU_API int32_t emu_last_packer() { return g_last_lanes; }

// The batch cut into sub-batches on the lanes of one context (casim_streams.h; the emulator runs the parts one after the other by default):
// how casim_options.n_streams cuts the tables and puts the results back together.  parts_out: how many parts ran (1 = not cut).
EMU_API int32_t emu_estimate_batch_streams(const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts, casim_results* out,
                                           int32_t* nnz_out, int32_t* offsets_out, const casim_option_query* q, int32_t* parts_out) {
    typedef casim::StreamedProblemT<EmuBackend> SP;
    EmuBackend primary;
    if (!SP::eligible(pegs, groups, opts)) {
        if (parts_out) *parts_out = 1;
        return emu_estimate_batch_query(pegs, groups, opts, out, 0, nnz_out, offsets_out, q);
    }
    std::vector<EmuBackend> bks((size_t)opts->n_streams);
    std::vector<EmuBackend*> lanes;
    for (auto& b : bks) lanes.push_back(&b);
    SP sp(primary, lanes);
    // CASIM_EMU_THREADS=1: the parts run as tasks of the host pool, as on the device (upload turns, list bases handed from part to part, every
    // part fetched by its own worker) — the emulator keeps its running block per thread; default: one part after the other
    const char* thr = getenv("CASIM_EMU_THREADS");
    int32_t rc = sp.estimate(pegs, groups, opts, out, q, /*threads=*/thr && atoi(thr) != 0);
    if (rc == CASIM_OK && (nnz_out || offsets_out)) rc = sp.csr(nnz_out, offsets_out);
    if (rc != CASIM_OK) g_err = sp.error();
    if (parts_out) *parts_out = (int32_t)sp.n_parts();
    return rc;
}

// The batch over n_devices emulated devices (casim_multi.h), cross-device reduce by the host-side hook or on the host.
EMU_API int32_t emu_estimate_batch_multi(int32_t n_devices, int32_t use_reduce_hook, const casim_pegs* pegs, const casim_groups* groups,
                                         const casim_options* opts, casim_results* out, int32_t* offsets_out, const casim_option_query* q,
                                         int32_t* info_out /*[1 + n_devices]: reduced by hook, groups per device*/) {
    std::vector<EmuBackend> bks((size_t)n_devices);
    std::vector<EmuBackend*> ptrs;
    for (auto& b : bks) ptrs.push_back(&b);
    casim::MultiProblemT<EmuBackend> mp(ptrs);
    casim::MultiProblemT<EmuBackend>::ReduceFn hook = [](const std::vector<int64_t*>& keys, int S) {
        for (int s = 0; s < S; ++s) {
            int64_t m = keys[0][s];
            for (size_t d = 1; d < keys.size(); ++d) m = keys[d][s] < m ? keys[d][s] : m;
            for (size_t d = 0; d < keys.size(); ++d) keys[d][s] = m;
        }
        return true;
    };
    const int32_t rc = mp.run(pegs, groups, opts, out, offsets_out, q, use_reduce_hook ? hook : casim::MultiProblemT<EmuBackend>::ReduceFn());
    if (rc != CASIM_OK) g_err = mp.error();
    if (info_out) { info_out[0] = mp.reduced_by(); for (size_t d = 0; d < mp.groups_per_device().size(); ++d) info_out[1 + d] = mp.groups_per_device()[d]; }
    return rc;
}

// ---- resident cluster under the emulator: a handle owns its backend -------------------------------------------
struct EmuCluster { EmuBackend bk; casim::ClusterT<EmuBackend>* c = nullptr; };
EMU_API void* emu_cluster_create(const casim_pegs* classes, const casim_groups* nodes, int64_t lds_budget_bytes) {
    EmuCluster* h = new EmuCluster();
    if (lds_budget_bytes > 0) h->bk.lds = (size_t)lds_budget_bytes;
    h->c = new casim::ClusterT<EmuBackend>(h->bk);
    if (h->c->init(classes, nodes) != CASIM_OK) { g_err = h->c->error(); delete h->c; delete h; return nullptr; }
    return h;
}
EMU_API void emu_cluster_destroy(void* p) { EmuCluster* h = (EmuCluster*)p; if (h) { delete h->c; delete h; } }
EMU_API int32_t emu_cluster_update_nodes(void* p, int32_t n, const int32_t* idx, const casim_groups* rows) {
    EmuCluster* h = (EmuCluster*)p; const int32_t rc = h->c->update_nodes(n, idx, rows); if (rc < 0) g_err = h->c->error(); return rc;
}
EMU_API int32_t emu_cluster_try_schedule_pods(void* p, const casim_pod_sequence* seq, int32_t commit, int32_t* node_out, int32_t* li, int32_t* ns) {
    EmuCluster* h = (EmuCluster*)p; const int32_t rc = h->c->try_schedule(seq, commit, node_out, li, ns); if (rc < 0) g_err = h->c->error(); return rc;
}
EMU_API int32_t emu_cluster_simulate_node_removals(void* p, const casim_removal_candidates* cand, casim_removal_results* out) {
    EmuCluster* h = (EmuCluster*)p; const int32_t rc = h->c->simulate_removals(cand, out); if (rc < 0) g_err = h->c->error(); return rc;
}
EMU_API int32_t emu_cluster_fetch_nodes(void* p, int64_t* init_req, int32_t* init_pods, uint64_t* init_excl) {
    EmuCluster* h = (EmuCluster*)p; return h->c->fetch_nodes(init_req, init_pods, init_excl);
}
EMU_API int32_t emu_cluster_stats(void* p, int64_t out[4]) { ((EmuCluster*)p)->c->stats(out); return 0; }
EMU_API int32_t emu_cluster_forget_commits(void* p) { ((EmuCluster*)p)->c->forget_commits(); return 0; }

EMU_API int32_t emu_feasibility_reasons(const casim_pegs* pegs, const casim_groups* groups, const uint64_t* port_block, uint16_t* out_codes) {
    EmuBackend bk;
    casim::ProblemT<EmuBackend> p(bk);
    casim_groups g = *groups;
    g.peg_offsets = nullptr; g.peg_index = nullptr;
    int32_t rc = p.init(pegs, &g, nullptr);
    if (rc == CASIM_OK) rc = p.reasons(port_block, out_codes);
    if (rc != CASIM_OK) g_err = p.error();
    return rc;
}

EMU_API int32_t emu_feasibility(const casim_pegs* pegs, const casim_groups* groups, uint64_t* out_bits) {
    EmuBackend bk;
    casim::ProblemT<EmuBackend> p(bk);
    casim_groups g = *groups;
    g.peg_offsets = nullptr; g.peg_index = nullptr;
    int32_t rc = p.init(pegs, &g, nullptr);
    if (rc == CASIM_OK) rc = p.run_feasibility();
    if (rc == CASIM_OK) rc = p.fetch_bits(out_bits);
    if (rc != CASIM_OK) g_err = p.error();
    return rc;
}

// lds_budget_bytes: as above (a tiny value forces the HBM-slab variant of K_sched)
EMU_API int32_t emu_try_schedule_pods(const casim_pegs* classes, const casim_groups* nodes, const casim_pod_sequence* seq,
                                      int64_t lds_budget_bytes, int32_t* node_out, int32_t* last_index_out, int32_t* n_scheduled_out,
                                      int32_t* info_out /*[2]: runs, state in LDS*/) {
    EmuBackend bk;
    if (lds_budget_bytes > 0) bk.lds = (size_t)lds_budget_bytes;
    casim::SchedulerT<EmuBackend> s(bk);
    int32_t rc = s.init(classes, nodes, seq);
    if (rc == CASIM_OK) rc = s.run();
    if (rc == CASIM_OK) rc = s.fetch(node_out, last_index_out, n_scheduled_out);
    if (info_out) { info_out[0] = s.runs(); info_out[1] = s.in_lds() ? 1 : 0; }
    if (rc < 0) g_err = s.error();
    return rc;
}

EMU_API int32_t emu_simulate_node_removals(const casim_pegs* classes, const casim_groups* nodes, const casim_removal_candidates* cand,
                                           int64_t lds_budget_bytes, casim_removal_results* out) {
    EmuBackend bk;
    if (lds_budget_bytes > 0) bk.lds = (size_t)lds_budget_bytes;
    casim::SchedulerT<EmuBackend> s(bk);
    int32_t rc = s.init_removals(classes, nodes, cand);
    if (rc == CASIM_OK) rc = s.run();
    if (rc == CASIM_OK) rc = s.fetch_removals(out);
    if (rc < 0) g_err = s.error();
    return rc;
}

EMU_API int32_t emu_last_removals_info(int32_t info_out[4]) {
    const int32_t* li = casim::last_removals_info();
    for (int i = 0; i < 4; ++i) info_out[i] = li[i];
    return 0;
}

EMU_API int32_t emu_last_chain_info(int32_t info_out[4]) {
    const int32_t* ci = casim::last_chain_info();
    for (int i = 0; i < 4; ++i) info_out[i] = ci[i];
    return 0;
}

EMU_API int32_t emu_estimate_on_cluster(const casim_pegs* classes, const casim_groups* nodes, const casim_cluster_estimate* params,
                                        int64_t lds_budget_bytes, casim_cluster_estimate_result* out) {
    EmuBackend bk;
    if (lds_budget_bytes > 0) bk.lds = (size_t)lds_budget_bytes;
    casim::ClusterEstimatorT<EmuBackend> s(bk);
    int32_t rc = s.init(classes, nodes, params);
    if (rc == CASIM_OK) rc = s.run();
    if (rc == CASIM_OK) rc = s.fetch(out);
    if (rc < 0) g_err = s.error();
    return rc;
}

}  // extern "C"
