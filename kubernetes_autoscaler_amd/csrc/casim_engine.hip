// casim_engine.hip — the product runtime: HIP backend for casim_pipeline.h + the C ABI of
// include/casim.h (casim_ctx_*, casim_problem_*, ...).  gfx950 only; no CPU fallback: every
// entry point fails with CASIM_ERR_NO_DEVICE when no HIP device is visible.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <atomic>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "casim_pipeline.h"
#include "casim_multi.h"
#include "casim_streams.h"

namespace casim {
// casim_pack_tu.hip: the register packer, compiled in its own translation unit (see there)
int hip_launch_pack_fast(int lanes, int slots_per_lane, int excl_words, int n_groups, void* stream, DevTables t, DevResults res, FastScratch fs);
int hip_launch_pack_fast_plain(int lanes, int slots_per_lane, int excl_words, int n_groups, void* stream, DevTables t, DevResults res, FastScratch fs);
}

namespace {

thread_local std::string g_err;
int32_t set_err(int32_t code, const std::string& m) { g_err = m; return code; }

// which build of the register packer an AUTO launch of instantiation (lanes, slots per lane, exclusion words) takes on `device`: runs
// the self-check of THAT instantiation on first use (resolve below)
bool casim_pack_use_plain(int device, size_t lds, int lanes, int slots_per_lane, int excl_words);

struct HipBackend {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    size_t lds = 64 * 1024;
    hipError_t last = hipSuccess;
    std::string msg;

    void check(hipError_t e, const char* what) {
        if (e != hipSuccess && last == hipSuccess) { last = e; msg = std::string(what) + ": " + hipGetErrorString(e); }
    }
    void bind() { check(hipSetDevice(device), "hipSetDevice"); }
    // Device memory comes from a per-context pool of power-of-two blocks: a call that comes back every loop iteration
    // (casim_estimate_batch allocates ~25 scratch / result arrays) finds its blocks in the free lists instead of paying
    // hipMalloc / hipFree each time.  At most kPoolKeepBytes stay cached; everything is released with the context.
    static constexpr size_t kPoolKeepBytes = (size_t)8 << 30;
    std::map<size_t, std::vector<void*>> pool;     // block size -> free blocks
    std::map<void*, size_t> live;                  // block -> its size
    size_t pooled_bytes = 0;
    static size_t block_size(size_t b) { size_t s = 256; while (s < b) s <<= 1; return s; }
    void* alloc(size_t b) {
        const size_t s = block_size(b);
        auto it = pool.find(s);
        void* p = nullptr;
        if (it != pool.end() && !it->second.empty()) { p = it->second.back(); it->second.pop_back(); pooled_bytes -= s; }
        else check(hipMalloc(&p, s), "hipMalloc");
        if (p) live[p] = s;
        // CASIM_POISON_ALLOC=1 (tests): every block starts as 0xA5 bytes — fresh HBM pages tend to be zero and a pooled block holds the last
        // call's similar data, so a kernel that reads what nobody wrote can go unnoticed; poisoned, it computes something else than the oracle
        static const bool poison = getenv("CASIM_POISON_ALLOC") && atoi(getenv("CASIM_POISON_ALLOC")) != 0;
        if (p && poison) { check(hipMemsetAsync(p, 0xA5, s, stream), "hipMemsetAsync (poison)"); check(hipStreamSynchronize(stream), "hipStreamSynchronize (poison)"); }   // (done before any stream uses the block)
        return p;
    }
    void free(void* p) {
        if (!p) return;
        auto it = live.find(p);
        if (it == live.end()) { (void)hipFree(p); return; }
        const size_t s = it->second;
        live.erase(it);
        if (pooled_bytes + s <= kPoolKeepBytes) { pool[s].push_back(p); pooled_bytes += s; }
        else (void)hipFree(p);
    }
    void release_pool() {
        for (auto& kv : pool) for (void* p : kv.second) (void)hipFree(p);
        pool.clear(); pooled_bytes = 0;
        for (int i = 0; i < 2; ++i) { if (staging[i]) (void)hipHostFree(staging[i]); staging[i] = nullptr; staging_cap[i] = 0; }
        if (mark_ev) { (void)hipEventDestroy(mark_ev); mark_ev = nullptr; }
        if (turn_ev) { (void)hipEventDestroy(turn_ev); turn_ev = nullptr; }
    }
    // pinned host staging: 0 = uploads, 1 = fetches
    void* staging[2] = {nullptr, nullptr}; size_t staging_cap[2] = {0, 0};
    void* stage(int which, size_t bytes) {
        which &= 1;
        if (staging_cap[which] < bytes) {
            if (staging[which]) (void)hipHostFree(staging[which]);
            size_t cap = 1 << 16; while (cap < bytes) cap <<= 1;
            staging[which] = nullptr; staging_cap[which] = 0;
            check(hipHostMalloc(&staging[which], cap, hipHostMallocDefault), "hipHostMalloc");
            if (staging[which]) staging_cap[which] = cap;
        }
        return staging[which];
    }
    void* stage_if_fits(int which, size_t bytes) { which &= 1; return staging_cap[which] >= bytes ? staging[which] : nullptr; }   // (never re-allocates)
    void h2d(void* d, const void* s, size_t n) { if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream), "hipMemcpyAsync H2D"); }
    // Big table uploads of a streamed call.  Copies that four lanes issue on four streams share the link, so every part's tables arrive at the
    // same late moment and no kernel runs under an upload (profiles/r14d_enter_return_trace.txt: 60 MB in 1.2 ms, THEN 1.1 ms of kernels).  The
    // parts therefore take the link IN TURN, on the device's own clock: part i's first big copy waits for the event part i - 1 recorded behind
    // its last one (wait_turn_event / record_turn_event; the host only makes sure the record is issued before the wait: UploadGate in issue
    // order).  Part 0's tables arrive first, at the full rate of the link, and its kernels run under the uploads of the parts behind it.
    // (A stream of its own for all uploads was tried first: the runtime served it with an SDMA engine at ~31 GB/s where the lanes' own
    // streams copy with blit kernels at ~55 — profiles/r14f_upload_stream_trace.txt — and every cell of the matrix got slower.)
    hipEvent_t turn_ev = nullptr;
    // OFF by default (CASIM_UPLOAD_FIFO=1 turns it on): measured neutral to slower in every cell of tables x requests x parts
    // (profiles/r14i_enter_return_matrix.txt) — a part behind the turn event finds its later small copies waiting on the host until its
    // stream has drained, and calls with small tables (PEG rows shared between simulations) only lose the stagger
    bool turns_enabled() const { static const bool on = getenv("CASIM_UPLOAD_FIFO") && atoi(getenv("CASIM_UPLOAD_FIFO")) != 0; return on; }
    void record_turn_event() {
        if (!turn_ev) check(hipEventCreateWithFlags(&turn_ev, hipEventDisableTiming), "hipEventCreate");
        if (turn_ev) check(hipEventRecord(turn_ev, stream), "hipEventRecord");
    }
    void wait_turn_event(HipBackend& prev) { if (prev.turn_ev) check(hipStreamWaitEvent(stream, prev.turn_ev, 0), "hipStreamWaitEvent"); }
    // the caller's array is page-locked (casim_host_alloc, hipHostRegister, a pinned torch tensor): the DMA engine can read it where it lies
    bool pinned(const void* p) const {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // (plain pageable memory: "invalid value", and sticky)
        return a.type == hipMemoryTypeHost;
    }
    void d2h(void* d, const void* s, size_t n) { if (n) check(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream), "hipMemcpyAsync D2H"); }
    void zero(void* d, size_t n) { if (n) check(hipMemsetAsync(d, 0, n, stream), "hipMemsetAsync"); }
    void fill8(void* d, int v, size_t n) { if (n) check(hipMemsetAsync(d, v, n, stream), "hipMemsetAsync"); }
    void sync() { check(hipStreamSynchronize(stream), "hipStreamSynchronize"); }
    // fork / join between the streams of one context (casim_streams.h): mark() records an event on the own stream,
    // wait_mark(o) makes the own stream wait for o's last mark
    hipEvent_t mark_ev = nullptr;
    void mark() {
        if (!mark_ev) check(hipEventCreateWithFlags(&mark_ev, hipEventDisableTiming), "hipEventCreate");
        if (mark_ev) check(hipEventRecord(mark_ev, stream), "hipEventRecord");
    }
    bool idle() { const hipError_t e = hipStreamQuery(stream); if (e == hipErrorNotReady) return false; check(e, "hipStreamQuery"); return true; }
    void wait_mark(HipBackend& o) { if (o.mark_ev) check(hipStreamWaitEvent(stream, o.mark_ev, 0), "hipStreamWaitEvent"); }
    void make_wait(void* raw_stream) { if (mark_ev) check(hipStreamWaitEvent((hipStream_t)raw_stream, mark_ev, 0), "hipStreamWaitEvent"); }
    size_t lds_budget() const { return lds; }
    bool ok() const { return last == hipSuccess; }
    const char* error() const { return msg.c_str(); }
    void clear() { last = hipSuccess; msg.clear(); (void)hipGetLastError(); }  // also drop HIP's sticky last error

    // The register packer exists twice in the library (casim_pack_tu.hip): build 0 compiled with the experimental structurizer
    // option, build 1 without.  `want`: casim_options.pack_build of the problem (CASIM_PACK_BUILD_*); AUTO takes the build the
    // self-check left standing for this device (pack_plain).
    bool pack_plain = false;   // (CASIM_PACK_BUILD=plain / a failed check: the process-wide verdict, copied when the context is created)
    // Verdicts this backend already holds, one bit per instantiation (pack_bit): the hot path reads them without the process-wide lock.
    // prepare_pack_fast() — called from ProblemT::init, where the instantiation is known and no kernel of the problem is in flight yet —
    // runs the instantiation's self-check on first use (it used to run inside the first launch, ~15 ms of streams, pool frees and
    // 32-36 simulations in the middle of an enter -> return call, under a mutex every pack launch of the process took).
    uint32_t pack_known_mask = 0, pack_plain_mask = 0;
    static uint32_t pack_bit(int lanes, int slots_per_lane, int excl_words) {
        const int lanes4 = lanes == 8 ? 2 : (lanes > 2 ? 1 : 0), sc = slots_per_lane <= 1 ? 0 : (slots_per_lane <= 4 ? 1 : 2), excl = excl_words > 0 ? 1 : 0;
        return 1u << (lanes4 * 6 + sc * 2 + excl);
    }
    void prepare_pack_fast(int want, int lanes, int slots_per_lane, int excl_words) {
        if (want == CASIM_PACK_BUILD_PLAIN || want == CASIM_PACK_BUILD_OPTION || pack_plain) return;
        const uint32_t bit = pack_bit(lanes, slots_per_lane, excl_words);
        if (pack_known_mask & bit) return;
        if (casim_pack_use_plain(device, lds, lanes, slots_per_lane, excl_words)) { pack_plain = true; pack_plain_mask |= bit; }
        pack_known_mask |= bit;
    }
    void launch_pack_fast(int want, int lanes, int slots_per_lane, int excl_words, int n_groups, const DevTables& t, const DevResults& res, const FastScratch& fs) {
        // AUTO: the instantiation about to run has been through the self-check (prepare_pack_fast at init; here only for a caller that skipped it)
        prepare_pack_fast(want, lanes, slots_per_lane, excl_words);
        const bool plain = want == CASIM_PACK_BUILD_PLAIN || (want != CASIM_PACK_BUILD_OPTION && pack_plain);
        if (plain) check((hipError_t)casim::hip_launch_pack_fast_plain(lanes, slots_per_lane, excl_words, n_groups, (void*)stream, t, res, fs), "pack_fast_kernel launch (plain build)");
        else check((hipError_t)casim::hip_launch_pack_fast(lanes, slots_per_lane, excl_words, n_groups, (void*)stream, t, res, fs), "pack_fast_kernel launch");
    }
    template <class K, class... A>
    void launch(K kernel, int gx, int gy, int block, size_t smem, A... args) {
        if (gx <= 0 || gy <= 0) return;
        if (smem > 64 * 1024) {
            // MI355X: 160 KiB LDS per CU; anything above the 64 KiB default needs the opt-in
            check(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem), "hipFuncSetAttribute(LDS)");
        }
        hipLaunchKernelGGL(kernel, dim3((unsigned)gx, (unsigned)gy, 1), dim3((unsigned)block, 1, 1), smem, stream, args...);
        check(hipGetLastError(), "kernel launch");
    }
};

typedef casim::ProblemT<HipBackend> HipProblem;
typedef casim::StreamedProblemT<HipBackend> HipStreamed;

}  // namespace

// The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and reads the variable once, at its
// first API call.  A streamed batch (casim_options.n_streams) wants a queue per lane next to the caller's own streams, i.e.
// GPU_MAX_HW_QUEUES=8 in the process environment BEFORE the first HIP call.  That is the HOST's decision: a
// library loaded into the autoscaler process does not edit its environment by default.  The host exports the variable itself
// (one line in the deployment, or os.Setenv in main before the shim is used) — or opts IN to the library
// doing it at load time with CASIM_SET_HW_QUEUES=1 (never overrides a value the process already set).  Without either the lane
// probe of casim_ctx::get_lanes makes the best of the queues there are (typically 3 concurrent lanes instead of 4).
__attribute__((constructor)) static void casim_default_hw_queues() {
    const char* want = getenv("CASIM_SET_HW_QUEUES");
    if (!want || atoi(want) == 0) return;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

namespace casim {
// one wave that spins for `ticks` of the constant-rate wall clock (100 MHz) and reports when it started and ended on that clock: the
// lane-concurrency probe of casim_ctx::get_lanes
__global__ __launch_bounds__(64) void lane_probe_kernel(uint64_t ticks, uint64_t* stamps /*[2], device-visible host memory, or null*/) {
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (stamps && threadIdx.x == 0) { stamps[0] = t0; stamps[1] = wall_clock64(); }
}
}  // namespace casim

struct casim_ctx {
    HipBackend bk;
    // lanes: backends on the SAME device with a stream, a memory pool and pinned staging buffers of their own — the parts of a
    // streamed batch (casim_options.n_streams) run on them; created on first use, kept for the life of the context
    std::vector<HipBackend*> lanes;
    // Streams are not hardware queues: the HIP runtime multiplexes them onto a few HSA queues (GPU_MAX_HW_QUEUES, 4 by default), and two
    // streams that share one run back to back.  Measured on the MI355X (profiles/r04e_stream_queue_probe.txt): four lanes created in
    // a row gave the resident loop NOTHING (1.21 ms per 4096 C2 simulations, the one-stream figure), three gave 1.07, four on
    // distinct queues 1.05.  So every candidate lane proves that it runs CONCURRENTLY with the lanes already taken — a spin kernel of
    // ~150 us on each of them at once must take the time of one — and a stream that does not is parked (kept until the context goes:
    // destroying it would hand the same queue to the next candidate).  CASIM_LANE_PROBE=0 takes the streams as they come.
    std::vector<hipStream_t> parked;
    bool lanes_capped = false;       // the runtime has no further queue to offer: later calls stop asking
    uint64_t spin_ticks = 15000;     // of the 100 MHz wall clock
    uint64_t* stamps = nullptr;      // pinned, device-visible: [2] per probed stream
    // The verdict comes from the DEVICE's own clock (host wall time around launch + sync parked good streams on a busy
    // host, and the number of parts then varied from run to run): every probe kernel stamps its start and end; two streams on two
    // queues overlap for most of the spin, two streams on one queue run back to back (the second starts when the first has ended).
    // The candidate is launched LAST; it runs beside the lanes iff it started before the first of them ended, by half a spin at least.
    bool runs_beside_the_lanes(hipStream_t cand) {
        static const bool probe = !(getenv("CASIM_LANE_PROBE") && atoi(getenv("CASIM_LANE_PROBE")) == 0);
        if (!probe || lanes.empty()) return true;
        std::vector<hipStream_t> all;
        for (HipBackend* l : lanes) all.push_back(l->stream);
        all.push_back(cand);
        if (!stamps && hipHostMalloc((void**)&stamps, sizeof(uint64_t) * 2 * 64, hipHostMallocDefault) != hipSuccess) { stamps = nullptr; (void)hipGetLastError(); return true; }
        if (all.size() > 64) return true;
        for (int attempt = 0; attempt < 2; ++attempt) {   // (the first launch on a new stream pays its queue's creation: look twice)
            for (hipStream_t st : all) (void)hipStreamSynchronize(st);
            memset(stamps, 0, sizeof(uint64_t) * 2 * all.size());
            for (size_t i = 0; i < all.size(); ++i) hipLaunchKernelGGL(casim::lane_probe_kernel, dim3(1), dim3(64), 0, all[i], spin_ticks, stamps + 2 * i);
            for (hipStream_t st : all) (void)hipStreamSynchronize(st);
            uint64_t first_end = ~0ull;
            for (size_t i = 0; i + 1 < all.size(); ++i) first_end = stamps[2 * i + 1] < first_end ? stamps[2 * i + 1] : first_end;
            const uint64_t cand_start = stamps[2 * (all.size() - 1)];
            if (cand_start != 0 && first_end != ~0ull && cand_start + spin_ticks / 2 <= first_end) return true;
        }
        return false;
    }
    // Once per process: a caller asked for more concurrent lanes than the runtime's hardware queues carry.  Until round 5 the library
    // set GPU_MAX_HW_QUEUES=8 by itself at load time (opt-out CASIM_KEEP_ENV); it no longer touches the environment, so a host that
    // relied on the old default now runs ~3 lanes instead of 4 — say so instead of being slower in silence (CASIM_QUIET=1 mutes).
    void note_parked_lanes(int asked) {
        static std::atomic<bool> said{false};
        if (said.exchange(true) || (getenv("CASIM_QUIET") && atoi(getenv("CASIM_QUIET")) != 0)) return;
        const char* q = getenv("GPU_MAX_HW_QUEUES");
        fprintf(stderr, "libcasim: %d of the %d stream lanes asked for run concurrently (%zu streams shared a hardware queue with a lane and were parked); "
                        "GPU_MAX_HW_QUEUES=%s.  Export GPU_MAX_HW_QUEUES=8 before the process's first HIP call, or CASIM_SET_HW_QUEUES=1 to let libcasim "
                        "set it when it is loaded (the library stopped editing the environment by default: INTEGRATION.md section 4; CASIM_KEEP_ENV "
                        "is gone).  CASIM_QUIET=1 silences this note.\n",
                (int)lanes.size(), asked, parked.size(), q ? q : "(unset: the runtime's default, 4)");
    }
    std::vector<HipBackend*> get_lanes(int k) {
        bk.bind();
        int tries = 0;
        while ((int)lanes.size() < k && !lanes_capped) {
            HipBackend* l = new (std::nothrow) HipBackend();
            if (!l) break;
            l->device = bk.device; l->lds = bk.lds; l->own_stream = true; l->pack_plain = bk.pack_plain;
            l->bind();
            l->check(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking), "hipStreamCreate");
            if (!l->ok()) { delete l; break; }
            if (runs_beside_the_lanes(l->stream)) { lanes.push_back(l); continue; }
            parked.push_back(l->stream);
            delete l;
            if (++tries >= 12) lanes_capped = true;
        }
        (void)hipGetLastError();
        if ((int)lanes.size() < k && !parked.empty()) note_parked_lanes(k);
        return std::vector<HipBackend*>(lanes.begin(), lanes.begin() + ((int)lanes.size() < k ? (int)lanes.size() : k));
    }
};
// ---- RCCL, loaded at run time (no link-time dependency: a process that already carries an RCCL — torch ships one — keeps
// using that copy; a plain C / Go host gets /opt/rocm/lib/librccl.so).  Only what the expander exchange needs.
namespace {
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    static constexpr int kInt64 = 4, kMin = 3;   // ncclInt64, ncclMin (nccl.h enums)
    bool load(std::string& err) {
        if (handle) return true;
        // 1. an RCCL the process already carries (a host that links one, torch's bundled copy): never load a second one;
        // 2. CASIM_RCCL_PATH; 3. the system library.
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) { handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (handle) break; }
        if (!handle) { const char* env = getenv("CASIM_RCCL_PATH"); if (env && *env) handle = dlopen(env, RTLD_NOW | RTLD_LOCAL); }
        if (!handle) for (const char* n : names) { handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (handle) break; }
        if (!handle) { err = "librccl.so not found (dlopen)"; return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        AllReduce = (decltype(AllReduce))dlsym(handle, "ncclAllReduce"); GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd"); GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd) { err = "librccl.so misses ncclCommInitAll / ncclAllReduce"; return false; }
        return true;
    }
};
}  // namespace

struct casim_mctx {
    std::vector<casim_ctx*> ctxs;
    Rccl rccl;
    std::vector<Rccl::comm_t> comms;   // one per device when RCCL is in use
    bool use_rccl = false;
    int32_t last_reduced_by = 0;
    std::vector<int32_t> last_groups;
};

struct casim_cluster {
    casim_ctx* ctx;
    casim::ClusterT<HipBackend>* c;
};

struct casim_problem {
    casim_ctx* ctx;
    HipProblem* prob;                // the batch as ONE part on the context's stream, or
    HipStreamed* sp = nullptr;       // ... cut into parts on the context's lanes (casim_options.n_streams); prob = part 0 then
    std::vector<hipEvent_t> marks;   // casim_problem_run_marked: 4 events per kept run (created on first use)
    int32_t n_marked = 0;
    HipBackend& bk0() { return sp ? sp->lane(0) : ctx->bk; }   // where part 0's kernels run (timing helpers)
    const std::string& error() const { return sp && !sp->error().empty() ? sp->error() : prob->error(); }
};

// ---- which build of the register packer serves a device (casim_pack_tu.hip) -------------------------------------------------------
// Build 0 of the packer is compiled with an experimental LLVM option that was caught miscompiling a sibling kernel; build 1 is the
// same source without it.  Before the first context of a process on a device is handed out, a built-in corpus goes through both:
// 12 batches — one per kernel instantiation (2 / 4 resource lanes x 1 / 4 / 16 node slots per lane x without / with exclusion
// words) — of 48 node groups over 160 PEGs with taints, selectors, limits, existing nodes, self-excluding PEGs and host-port
// bits, schedulable subsets derived on the device.  Every output array must be identical; any difference (or an error of build 0
// alone) retires build 0 for the process and says so once on stderr.  ~40 ms, once.  CASIM_PACK_BUILD=plain|option skips the
// check and forces a build; CASIM_PACK_SELFCHECK_FAULT=1 makes the comparison see a flipped word (how the fallback is tested).
namespace {
struct PackBuildState { int checked = 0; int plain = 0; int batches = 0; int differing = 0; int forced = 0; int skipped = 0; double ms = 0; uint32_t checked_mask = 0; };
PackBuildState g_pack_build[64];
std::mutex g_pack_build_mu;

struct SelfCheckRng { uint64_t s; uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); } uint32_t below(uint32_t n) { return n ? next() % n : 0; } };

// one batch of the corpus through one build: every result array into `out`.
// variant bits (VERDICT r3 next #10 / ADVICE r3: the first corpus had no zone words, no NEED polarity, no singleton runs, no fastpath, no
// caller lists, no PEG outside the "simple" shape):
//   1  tryFastPath on (fp_cpu / fp_mem / cap_cpu / cap_mem, CASIM_PEG_FASTPATH_OK; switches the singleton merge off by itself)
//   2  stretches of adjacent identical one-pod PEGs (merged on the host into CASIM_KFLAG_SINGLETON_RUN rows: the lastIndex rule)
//   4  group-wide exclusion words: anti-affinity bits, bits of NEED polarity (blocked until a partner marks them), zone self-exclusion,
//      partly invalid keys (instantiations with exclusion words only)
//   8  the caller's own PEG lists (template-level Filters then run inside the orderer) instead of device-derived ones
//  16  PEGs outside the simple shape: a zero request lane, a request >= 2^30 after gcd scaling (4-lane instantiations)
//  32  node bits of NEED polarity (hostname-level pod affinity, ABI 8): PEGs that wait for a partner on the node, partners, self-affine series
//      that enter by the first-pod exception (the record walked twice), templates that carry the partner (instantiations with exclusion words only)
int32_t self_check_run(HipBackend& bk, int build, int lanes4, int slot_class, int excl, int variant, uint32_t seed, std::vector<int64_t>& out) {
    SelfCheckRng rng{0x9E3779B97F4A7C15ull ^ (uint64_t)(lanes4 * 131 + slot_class * 17 + excl) ^ ((uint64_t)seed << 20) ^ ((uint64_t)variant << 40)};
    const bool v_fast = (variant & 1) != 0, v_runs = (variant & 2) != 0, v_zone = excl && (variant & 4) != 0, v_lists = (variant & 8) != 0, v_odd = (variant & 16) != 0, v_need = excl && (variant & 32) != 0;
    const int G = 160, NG = 48, R = lanes4 == 1 ? 3 : 2;   // lanes4: 0 = two int32 lanes, 1 = four, 2 = two int64 lanes (the same tables as 0, not narrowed)
    std::vector<int64_t> req((size_t)G * R), alloc((size_t)NG * R), ireq((size_t)NG * R);
    std::vector<int32_t> count(G), allowed(NG), ipods(NG), maxn(NG), existing(NG), lastidx(NG);
    std::vector<uint32_t> pflags(G), gflags(NG);
    std::vector<uint64_t> tol(G), sel(G), xb(G), xm(G), zb(G, 0), zm(G, 0), taint(NG), label(NG), iexcl(NG), izone(NG, 0), zvalid(NG, ~0ull);
    std::vector<double> fpc(G), fpm(G), capc(NG), capm(NG);
    const uint64_t zpol = 0x00ff0000ull;   // bits 16-23 of the zone word have NEED polarity
    const uint64_t xpol = 0xffull << 40;   // bits 40-47 of the node word have NEED polarity (the plain bits above use 0-39)
    for (int g = 0; g < G; ++g) {
        req[(size_t)g * R] = 50 + 50 * (int64_t)rng.below(60);
        req[(size_t)g * R + 1] = ((int64_t)64 + 64 * rng.below(96)) << 20;
        if (R > 2) req[(size_t)g * R + 2] = rng.below(4) == 0 ? 1 + rng.below(2) : 0;
        count[g] = 1 + (int32_t)rng.below(slot_class == 2 ? 90 : 40);
        pflags[g] = rng.below(16) == 0 ? CASIM_PEG_TOLERATES_UNSCHEDULABLE : 0u;
        tol[g] = rng.below(3) ? ~0ull : (uint64_t)rng.next();          // most PEGs tolerate everything
        sel[g] = rng.below(4) ? 0ull : (1ull << rng.below(6));          // a few need one label requirement
        xb[g] = xm[g] = 0;
        if (excl) {
            const uint32_t k = rng.below(8);
            if (k == 0) { pflags[g] |= CASIM_PEG_SELF_EXCL_NODE; xb[g] = xm[g] = 1ull << rng.below(40); }   // self anti-affinity / own host port
            else if (k == 1) { xb[g] = 1ull << rng.below(40); }                                              // blocked by somebody's bit
            else if (k == 2) { xm[g] = 1ull << rng.below(40); }                                              // marks a bit others avoid
        }
        if (v_need) {
            const uint32_t k = rng.below(8);
            const uint64_t bit = 1ull << (40 + rng.below(8));
            if (k == 0) { xb[g] |= bit; xm[g] |= bit; }   // a series: needs the bit it marks itself
            else if (k == 1) xb[g] |= bit;                  // waits for a partner on the node
            else if (k == 2 || k == 3) xm[g] |= bit;        // a partner
        }
        if (v_zone) {
            const uint32_t k = rng.below(10);
            if (k == 0) { pflags[g] |= CASIM_PEG_SELF_EXCL_ZONE; zb[g] = zm[g] = 1ull << rng.below(16); }   // one pod of the PEG per group
            else if (k == 1) zb[g] = 1ull << rng.below(16);                                                  // blocked once somebody marked the bit
            else if (k == 2) zm[g] = 1ull << rng.below(16);
            else if (k == 3) zb[g] = 1ull << (16 + rng.below(8));                                            // waits for a partner (NEED polarity)
            else if (k == 4) zm[g] = 1ull << (16 + rng.below(8));                                            // ... the partner
            else if (k == 5) { const uint64_t bit = 1ull << (16 + rng.below(8)); zb[g] = bit; zm[g] = bit; } // needs a partner and is one (not the first-pod case: stays blocked)
        }
        if (v_odd) {
            const uint32_t k = rng.below(6);
            if (k == 0) req[(size_t)g * R] = 0;                                     // no cpu request: the lane test is skipped (fit.go:699)
            else if (k == 1) req[(size_t)g * R + 1] = 0;
            else if (k == 2) { req[(size_t)g * R] = 0; req[(size_t)g * R + 1] = 0; if (R > 2) req[(size_t)g * R + 2] = 0; }   // requests nothing at all: pod slots only
            else if (k == 3 && R > 2) req[(size_t)g * R + 2] = 1100000001ll + 2 * (int64_t)rng.below(1000);   // > 2^30 and co-prime with the others: the wide remainder path
        }
        fpc[g] = (double)req[(size_t)g * R] / 1000.0; fpm[g] = (double)req[(size_t)g * R + 1];
        if (v_fast && rng.below(5) != 0) { pflags[g] |= CASIM_PEG_FASTPATH_OK; if ((pflags[g] & CASIM_PEG_SELF_EXCL_NODE) && rng.below(2)) pflags[g] |= CASIM_PEG_FASTPATH_AA_SELF; }
    }
    if (v_runs) {   // 3 stretches of 5-20 adjacent copies of one controller-less pod
        for (int k = 0; k < 3; ++k) {
            const int at = 8 + 50 * k + (int)rng.below(10), len = 5 + (int)rng.below(16);
            for (int j = 0; j < len && at + j < G; ++j) {
                for (int r = 0; r < R; ++r) req[(size_t)(at + j) * R + r] = req[(size_t)at * R + r];
                count[at + j] = 1; pflags[at + j] = pflags[at] & ~(uint32_t)(CASIM_PEG_SELF_EXCL_NODE | CASIM_PEG_SELF_EXCL_ZONE);
                tol[at + j] = tol[at]; sel[at + j] = sel[at]; xb[at + j] = xm[at + j] = 0; zb[at + j] = zm[at + j] = 0; fpc[at + j] = fpc[at]; fpm[at + j] = fpm[at];
            }
        }
    }
    for (int i = 0; i < NG; ++i) {
        const int32_t pods_cap = slot_class == 2 ? 4 + (int32_t)rng.below(8) : (slot_class == 1 ? 8 + (int32_t)rng.below(24) : 20 + (int32_t)rng.below(90));
        alloc[(size_t)i * R] = 2000 + 1000 * (int64_t)rng.below(slot_class == 0 ? 62 : 14);
        alloc[(size_t)i * R + 1] = ((int64_t)8 + 8 * rng.below(slot_class == 0 ? 32 : 8)) << 30;
        if (R > 2) alloc[(size_t)i * R + 2] = v_odd ? 2100000003ll : (rng.below(3) ? 8 : 0);
        ipods[i] = (int32_t)rng.below(3);
        ireq[(size_t)i * R] = 100 * ipods[i]; ireq[(size_t)i * R + 1] = ((int64_t)128 * ipods[i]) << 20;
        if (R > 2) ireq[(size_t)i * R + 2] = 0;
        allowed[i] = pods_cap + ipods[i];
        gflags[i] = rng.below(12) == 0 ? CASIM_NG_UNSCHEDULABLE : 0u;
        taint[i] = rng.below(3) ? 0ull : (1ull << rng.below(8));
        label[i] = (uint64_t)rng.next() & 0x3full;
        iexcl[i] = excl && rng.below(4) == 0 ? 1ull << rng.below(40) : 0ull;
        if (v_need && rng.below(5) == 0) iexcl[i] |= 1ull << (40 + rng.below(8));   // the template's own pods are the partner
        if (v_zone) {
            izone[i] = (rng.below(4) == 0 ? 1ull << rng.below(16) : 0ull) | (rng.below(4) == 0 ? 1ull << (16 + rng.below(8)) : 0ull);   // pods of the existing cluster: a conflict, a partner
            zvalid[i] = rng.below(5) == 0 ? ~(0x0f0full) : ~0ull;                                                                     // a template without some of the keys
        }
        capc[i] = (double)alloc[(size_t)i * R] / 1000.0; capm[i] = (double)alloc[(size_t)i * R + 1];
        // the node bound decides the instantiation: <= 64 -> 1 slot per lane, <= 256 -> 4, <= 1024 -> 16
        maxn[i] = slot_class == 0 ? 1 + (int32_t)rng.below(60) : (slot_class == 1 ? 70 + (int32_t)rng.below(180) : 300 + (int32_t)rng.below(700));
        if (rng.below(10) == 0) maxn[i] = -1;
        existing[i] = (int32_t)rng.below(6);
        lastidx[i] = (int32_t)rng.below((uint32_t)existing[i] + 4) - 1;
    }
    // group 0 always has the launch's largest node bound (the instantiation is decided by the maximum)
    maxn[0] = slot_class == 0 ? 60 : (slot_class == 1 ? 249 : 999);
    std::vector<int32_t> loff, lidx;
    if (v_lists) {   // ascending subsets, as SchedulablePodGroups would hand them over — some PEGs in them fail the template's Filters
        loff.push_back(0);
        for (int i = 0; i < NG; ++i) {
            const uint32_t keep = 2 + rng.below(3);
            for (int g = 0; g < G; ++g) if (rng.below(4) < keep) lidx.push_back(g);
            loff.push_back((int32_t)lidx.size());
        }
    }
    casim_pegs p; memset(&p, 0, sizeof p);
    p.n_pegs = G; p.n_res = R; p.w_taint = 1; p.w_label = 1; p.w_excl = excl ? 1 : 0; p.w_zone = v_zone ? 1 : 0;
    p.req = req.data(); p.count = count.data(); p.flags = pflags.data(); p.tol_mask = tol.data(); p.sel_mask = sel.data();
    p.excl_block = excl ? xb.data() : nullptr; p.excl_mark = excl ? xm.data() : nullptr;
    if (v_need) p.excl_polarity = &xpol;
    if (v_zone) { p.zone_block = zb.data(); p.zone_mark = zm.data(); p.zone_polarity = &zpol; }
    if (v_fast) { p.fp_cpu = fpc.data(); p.fp_mem = fpm.data(); }
    casim_groups g; memset(&g, 0, sizeof g);
    g.n_groups = NG; g.alloc = alloc.data(); g.init_req = ireq.data(); g.allowed_pods = allowed.data(); g.init_pods = ipods.data(); g.flags = gflags.data();
    g.taint_mask = taint.data(); g.label_mask = label.data(); g.init_excl = excl ? iexcl.data() : nullptr;
    if (v_zone) { g.init_zone = izone.data(); g.zone_valid = zvalid.data(); }
    if (v_fast) { g.cap_cpu = capc.data(); g.cap_mem = capm.data(); }
    if (v_lists) { g.peg_offsets = loff.data(); g.peg_index = lidx.data(); }
    g.max_nodes = maxn.data(); g.existing_nodes = existing.data(); g.last_index = lastidx.data();
    casim_options o; memset(&o, 0, sizeof o);
    o.pack_build = build; o.fastpath = v_fast ? 1 : 0;
    if (lanes4 == 2) o.force_generic_packer = 2;
    bk.clear();
    HipProblem prob(bk);
    int32_t rc = prob.init(&p, &g, &o);
    if (rc != CASIM_OK) return rc;
    if (prob.fast_npt() != (slot_class == 0 ? 1 : (slot_class == 1 ? 4 : 16)) || prob.fast_lanes() != (lanes4 == 2 ? 8 : (lanes4 ? 4 : 2))) return CASIM_ERR_INVALID;   // (the corpus no longer reaches the instantiation it is meant for)
    rc = prob.run();
    if (rc != CASIM_OK) return rc;
    int32_t nnz = 0;
    std::vector<int32_t> off((size_t)NG + 1);
    rc = prob.csr(&nnz, off.data());
    if (rc != CASIM_OK) return rc;
    std::vector<int32_t> a((size_t)NG * 6), order((size_t)nnz + 1), placed((size_t)nnz + 1);
    std::vector<int64_t> sums((size_t)NG * 2);
    casim_results r; memset(&r, 0, sizeof r);
    r.node_count = a.data(); r.pods_scheduled = a.data() + NG; r.nodes_added = a.data() + 2 * NG; r.limiter_nodes = a.data() + 3 * NG;
    r.last_index_out = a.data() + 4 * NG; r.status = a.data() + 5 * NG; r.req_cpu_sum = sums.data(); r.req_mem_sum = sums.data() + NG;
    r.order = order.data(); r.placed = placed.data();
    rc = prob.fetch(&r);
    if (rc != CASIM_OK) return rc;
    out.clear();
    out.push_back(nnz);
    for (int32_t v : off) out.push_back(v);
    for (int32_t v : a) out.push_back(v);
    for (int64_t v : sums) out.push_back(v);
    for (int32_t k = 0; k < nnz; ++k) { out.push_back(order[(size_t)k]); out.push_back(placed[(size_t)k]); }
    return CASIM_OK;
}

// The corpus: every instantiation (2 / 4 int32 lanes or 2 int64 lanes x 1 / 4 / 16 slots x without / with exclusion words) x 18 batches — the 12 plain batches the
// check started with, then every feature bit alone, in pairs, and all together, each on data of its own.  324 batches, 648 runs.
struct SelfCheckCase { int variant; uint32_t seed; };
const SelfCheckCase kSelfCheckCases[] = {{0, 0}, {1, 1}, {2, 2}, {4, 3}, {8, 4}, {16, 5}, {3, 6}, {6, 7}, {12, 8}, {24, 9}, {17, 10}, {10, 11}, {20, 12}, {30, 13}, {31, 14}, {0, 15}, {32, 16}, {44, 17}};

// The check of ONE instantiation (lanes4, slot class, exclusion words) on `device`: its 18 case families through both builds.
// Caller holds g_pack_build_mu.
void self_check_instantiation(PackBuildState& st, int device, size_t lds, int lanes4, int sc, int excl) {
    const bool fault = getenv("CASIM_PACK_SELFCHECK_FAULT") && atoi(getenv("CASIM_PACK_SELFCHECK_FAULT")) != 0;
    HipBackend tmp;
    tmp.device = device; tmp.lds = lds; tmp.own_stream = true;
    tmp.bind();
    tmp.check(hipStreamCreateWithFlags(&tmp.stream, hipStreamNonBlocking), "hipStreamCreate");
    std::vector<int64_t> a, b;
    const auto t0 = std::chrono::steady_clock::now();
    int n_cases = (int)(sizeof kSelfCheckCases / sizeof kSelfCheckCases[0]);
    if (const char* e = getenv("CASIM_PACK_SELFCHECK_CASES")) { const int v = atoi(e); if (v >= 1 && v < n_cases) n_cases = v; }   // (1 = the one batch per instantiation of round 3)
    for (int ci = 0; ci < n_cases && tmp.stream; ++ci) {
        const SelfCheckCase& cse = kSelfCheckCases[ci];
        if (!excl && (cse.variant == 4 || cse.variant == 32)) continue;   // (zone words / node NEED bits alone need an instantiation that carries them: nothing new to run)
        const int32_t rb = self_check_run(tmp, CASIM_PACK_BUILD_PLAIN, lanes4, sc, excl, cse.variant, cse.seed, b);
        if (rb != CASIM_OK) { st.skipped++; continue; }   // (the reference build itself cannot run this batch: nothing to compare)
        const int32_t ra = self_check_run(tmp, CASIM_PACK_BUILD_OPTION, lanes4, sc, excl, cse.variant, cse.seed, a);
        st.batches++;
        if (fault && st.batches == 5 && !a.empty()) a[a.size() / 2] ^= 1;
        if (ra != CASIM_OK || a != b) st.differing++;
    }
    if (tmp.stream) { (void)hipStreamSynchronize(tmp.stream); }
    tmp.release_pool();
    if (tmp.stream) (void)hipStreamDestroy(tmp.stream);
    (void)hipGetLastError();
    st.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    st.checked_mask |= 1u << (lanes4 * 6 + sc * 2 + excl);
    if (st.differing > 0 && !st.plain) {
        st.plain = 1;
        fprintf(stderr, "libcasim: the register packer's self-check found %d of %d batches differing between its two builds on device %d: "
                        "the build with -structurizecfg-skip-uniform-regions is retired for this process (plain build in use)\n",
                st.differing, st.batches, device);
    }
    if (getenv("CASIM_PACK_SELFCHECK_VERBOSE")) fprintf(stderr, "libcasim: packer self-check: %d batches compared, %d differing, %d not runnable, %.1f ms\n", st.batches, st.differing, st.skipped, st.ms);
}

// The verdict for the device of `bk` when a context is created: a forced build (CASIM_PACK_BUILD), or — CASIM_PACK_SELFCHECK=eager — every
// instantiation checked up front (324 - 18 batches, ~300 ms).  Default: LAZY — nothing here; an instantiation is checked right before its first
// AUTO launch (casim_pack_use_plain: 16-18 batches, ~15 ms, paid by the first problem that needs it), so that start-up costs what the process uses.
void resolve_pack_build(HipBackend& bk) {
    if (bk.device < 0 || bk.device >= 64) return;
    std::lock_guard<std::mutex> lock(g_pack_build_mu);
    PackBuildState& st = g_pack_build[bk.device];
    if (!st.checked) {
        st.checked = 1;
        const char* force = getenv("CASIM_PACK_BUILD");
        if (force && !strcmp(force, "plain")) { st.plain = 1; st.forced = 1; }
        else if (force && !strcmp(force, "option")) { st.plain = 0; st.forced = 1; }
        else if (const char* mode = getenv("CASIM_PACK_SELFCHECK")) {
            if (!strcmp(mode, "eager"))
                for (int lanes4 = 0; lanes4 < 3; ++lanes4) for (int sc = 0; sc < 3; ++sc) for (int excl = 0; excl < 2; ++excl) self_check_instantiation(st, bk.device, bk.lds, lanes4, sc, excl);
        }
    }
    bk.pack_plain = st.plain != 0;
}

bool casim_pack_use_plain(int device, size_t lds, int lanes, int slots_per_lane, int excl_words) {
    if (device < 0 || device >= 64) return false;
    const int lanes4 = lanes == 8 ? 2 : (lanes > 2 ? 1 : 0), sc = slots_per_lane <= 1 ? 0 : (slots_per_lane <= 4 ? 1 : 2), excl = excl_words > 0 ? 1 : 0;
    const uint32_t bit = 1u << (lanes4 * 6 + sc * 2 + excl);
    std::lock_guard<std::mutex> lock(g_pack_build_mu);
    PackBuildState& st = g_pack_build[device];
    if (st.forced || st.plain) return st.plain != 0;
    if (!(st.checked_mask & bit)) self_check_instantiation(st, device, lds, lanes4, sc, excl);
    return st.plain != 0;
}
}  // namespace

namespace casim {

__global__ void copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// Known-byte-count read streams for calibrating the profiler's FETCH_SIZE on this access width (MI355X_MICROARCH.md: the
// x2 rule is established for 16 B / lane streams only): every lane reads W bytes per step, one value per wave is written.
template <int W>
__global__ __launch_bounds__(256) void stream_probe_kernel(const uint32_t* __restrict__ src, int64_t n_words, uint32_t* __restrict__ sink) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    if (W == 4) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) acc ^= src[i];
    } else {
        const uint4* s4 = (const uint4*)src;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words / 4; i += stride) { const uint4 v = s4[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;   // (never true for the zero-filled source: keeps the loads alive)
}

// The same for SCALAR loads (the register packer's record fetch): every wave walks its own contiguous region with one
// s_load_dwordx8 (32 B) per step through the constant cache — the access pattern of pack_fast_kernel's PEG records.
__global__ __launch_bounds__(64) void stream_probe_scalar_kernel(const uint32_t* __restrict__ src, int64_t words_per_wave, uint32_t* __restrict__ sink) {
    const uint32_t* p = src + (int64_t)blockIdx.x * words_per_wave;
    uint32_t acc = 0;
    for (int64_t i = 0; i < words_per_wave; i += 8) {
        const cs::Words<8> w = cs::const_load<8>(p + i);
        acc ^= w.w[0] ^ w.w[1] ^ w.w[2] ^ w.w[3] ^ w.w[4] ^ w.w[5] ^ w.w[6] ^ w.w[7];
    }
    if (acc == 0x12345678u) sink[blockIdx.x & 4095] = acc;
}

}  // namespace casim

extern "C" {

int32_t casim_abi_version(void) { return CASIM_ABI_VERSION; }
void* casim_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void casim_host_free(void* p) { if (p) (void)hipHostFree(p); }
const char* casim_last_error(void) { return g_err.c_str(); }

int32_t casim_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

casim_ctx* casim_ctx_create(int32_t device, void* stream) {
    g_err.clear();
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { set_err(CASIM_ERR_NO_DEVICE, "no HIP device visible: libcasim has no CPU path"); return nullptr; }
    if (device < 0 || device >= n) { set_err(CASIM_ERR_INVALID, "device index out of range"); return nullptr; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { set_err(CASIM_ERR_HIP, "hipGetDeviceProperties failed"); return nullptr; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(CASIM_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", libcasim is built for gfx950 only");
        return nullptr;
    }
    casim_ctx* c = new (std::nothrow) casim_ctx();
    if (!c) { set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
    c->bk.device = device;
    c->bk.bind();
    if (stream) c->bk.stream = (hipStream_t)stream;
    else { c->bk.check(hipStreamCreateWithFlags(&c->bk.stream, hipStreamNonBlocking), "hipStreamCreate"); c->bk.own_stream = true; }
    // usable dynamic LDS per workgroup: 160 KiB on MI355X (keep a little headroom)
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && optin > 0) c->bk.lds = (size_t)optin;
    if (c->bk.lds > 160 * 1024) c->bk.lds = 160 * 1024;
    if (!c->bk.ok()) { set_err(CASIM_ERR_HIP, c->bk.msg); delete c; return nullptr; }
    resolve_pack_build(c->bk);
    c->bk.clear();
    return c;
}

int32_t casim_pack_build_info(int32_t device, int32_t out[4]) {
    g_err.clear();
    if (!out || device < 0 || device >= 64) return set_err(CASIM_ERR_INVALID, "bad argument");
    std::lock_guard<std::mutex> lock(g_pack_build_mu);
    const PackBuildState& st = g_pack_build[device];
    // (_OPTION only once an instantiation has been through the comparison — or the build is forced; until then: AUTO = "unchecked")
    out[0] = st.plain ? CASIM_PACK_BUILD_PLAIN : ((st.forced || st.checked_mask) ? CASIM_PACK_BUILD_OPTION : CASIM_PACK_BUILD_AUTO);
    out[1] = st.batches; out[2] = st.differing; out[3] = st.forced;
    return CASIM_OK;
}
void casim_ctx_destroy(casim_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->bk.device);
    if (ctx->bk.stream) (void)hipStreamSynchronize(ctx->bk.stream);
    for (HipBackend* l : ctx->lanes) {
        if (l->stream) (void)hipStreamSynchronize(l->stream);
        l->release_pool();
        if (l->stream) (void)hipStreamDestroy(l->stream);
        delete l;
    }
    ctx->lanes.clear();
    for (hipStream_t st : ctx->parked) (void)hipStreamDestroy(st);
    ctx->parked.clear();
    if (ctx->stamps) { (void)hipHostFree(ctx->stamps); ctx->stamps = nullptr; }
    ctx->bk.release_pool();
    if (ctx->bk.own_stream && ctx->bk.stream) (void)hipStreamDestroy(ctx->bk.stream);
    delete ctx;
}

static bool wants_streams(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts, std::vector<HipBackend*>* lanes) {
    if (!HipStreamed::eligible(pegs, groups, opts)) return false;
    int k = opts->n_streams > 16 ? 16 : opts->n_streams;
    if (k > groups->n_sims) k = groups->n_sims;
    *lanes = ctx->get_lanes(k);
    return lanes->size() >= 2;
}

static casim_problem* problem_create(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts, bool one_shot);
casim_problem* casim_problem_create(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts) {
    return problem_create(ctx, pegs, groups, opts, /*one_shot=*/false);
}
// one_shot: the caller runs, fetches and destroys the problem before the context sees anything else (casim_estimate_batch*): the
// upload is not waited for on its own (ProblemT::init)
static casim_problem* problem_create(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts, bool one_shot) {
    g_err.clear();
    if (!ctx) { set_err(CASIM_ERR_INVALID, "null context"); return nullptr; }
    ctx->bk.bind(); ctx->bk.clear();
    casim_problem* p = new (std::nothrow) casim_problem();
    if (!p) { set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
    p->ctx = ctx;
    std::vector<HipBackend*> lanes;
    if (wants_streams(ctx, pegs, groups, opts, &lanes)) {
        for (HipBackend* l : lanes) l->clear();
        p->sp = new (std::nothrow) HipStreamed(ctx->bk, lanes);
        if (!p->sp) { delete p; set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
        const int32_t rc = p->sp->init(pegs, groups, opts, /*threads=*/true);
        if (rc != CASIM_OK) { set_err(rc, p->sp->error()); delete p->sp; delete p; return nullptr; }
        p->prob = p->sp->part(0).prob.get();
        return p;
    }
    p->prob = new (std::nothrow) HipProblem(ctx->bk);
    if (!p->prob) { delete p; set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
    p->prob->set_one_shot(one_shot);
    const int32_t rc = p->prob->init(pegs, groups, opts);
    if (rc != CASIM_OK) { set_err(rc, p->prob->error()); delete p->prob; delete p; return nullptr; }
    return p;
}
void casim_problem_destroy(casim_problem* p) {
    if (!p) return;
    p->ctx->bk.bind();
    (void)hipStreamSynchronize(p->ctx->bk.stream);
    if (p->sp) p->sp->sync_all();
    for (hipEvent_t e : p->marks) (void)hipEventDestroy(e);
    if (p->sp) delete p->sp; else delete p->prob;
    delete p;
}

#define PROB_ENTER(p)                                                            \
    g_err.clear();                                                               \
    if (!(p) || !(p)->prob) return set_err(CASIM_ERR_INVALID, "null problem");   \
    (p)->ctx->bk.bind(); (p)->ctx->bk.clear()
#define PROB_RET(p, rc) do { const int32_t _rc = (rc); if (_rc != CASIM_OK) set_err(_rc, (p)->error()); return _rc; } while (0)
static void clear_lanes(casim_problem* p) { if (p->sp) for (size_t i = 0; i < p->sp->n_parts(); ++i) p->sp->lane(i).clear(); }
static int32_t lanes_ok(casim_problem* p) {
    if (p->sp) for (size_t i = 0; i < p->sp->n_parts(); ++i) if (!p->sp->lane(i).ok()) return set_err(CASIM_ERR_HIP, p->sp->lane(i).msg);
    return CASIM_OK;
}

int32_t casim_problem_run(casim_problem* p) {
    PROB_ENTER(p);
    if (p->sp) { clear_lanes(p); const int32_t rc = p->sp->run(); if (rc != CASIM_OK) return set_err(rc, p->sp->error()); return lanes_ok(p); }
    PROB_RET(p, p->prob->run());
}
int32_t casim_problem_fetch(casim_problem* p, casim_results* out) {
    PROB_ENTER(p);
    if (p->sp) { clear_lanes(p); const int32_t rc = p->sp->fetch(out, /*threads=*/p->sp->groups() >= 4096); if (rc != CASIM_OK) return set_err(rc, p->sp->error()); return lanes_ok(p); }
    PROB_RET(p, p->prob->fetch(out));
}
int32_t casim_problem_info(casim_problem* p, int32_t info_out[8]) {
    g_err.clear();
    if (!p || !p->prob || !info_out) return set_err(CASIM_ERR_INVALID, "null argument");
    for (int i = 0; i < 8; ++i) info_out[i] = 0;
    info_out[0] = p->prob->fast_npt(); info_out[1] = p->prob->fast_lanes();
    info_out[2] = p->prob->pack_in_lds() ? 1 : 0; info_out[3] = p->prob->csr_on_device() ? 1 : 0;
    info_out[4] = p->sp ? (int32_t)p->sp->n_parts() : 1;
    info_out[5] = p->sp ? (int32_t)(p->sp->forks() & 0x7fffffff) : 0;   // forks from the context's stream so far (diagnostic)
    info_out[6] = p->ctx ? (int32_t)p->ctx->parked.size() : 0;
    info_out[7] = (p->prob->uses_front() ? 1 : 0) | (p->prob->uses_rank_once() ? 2 : 0);
    return CASIM_OK;
}
int32_t casim_problem_set_group_result(casim_problem* p, int32_t ng, const casim_cluster_estimate_result* r) {
    PROB_ENTER(p);
    if (p->sp) { const int32_t rc = p->sp->set_group_result(ng, r); if (rc != CASIM_OK) return set_err(rc, p->sp->error()); return CASIM_OK; }
    PROB_RET(p, p->prob->set_group_result(ng, r));
}
int32_t casim_problem_csr(casim_problem* p, int32_t* nnz_out, int32_t* offsets_out) {
    PROB_ENTER(p);
    if (p->sp) { const int32_t rc = p->sp->csr(nnz_out, offsets_out); if (rc != CASIM_OK) return set_err(rc, p->sp->error()); return lanes_ok(p); }
    PROB_RET(p, p->prob->csr(nnz_out, offsets_out));
}

// enter -> return of a streamed batch: every part is uploaded, run, reduced and fetched on its own lane
static int32_t estimate_streamed(casim_ctx* ctx, std::vector<HipBackend*>& lanes, const casim_pegs* pegs, const casim_groups* groups,
                                 const casim_options* opts, casim_results* out, const casim_option_query* q) {
    ctx->bk.bind(); ctx->bk.clear();
    for (HipBackend* l : lanes) l->clear();
    HipStreamed sp(ctx->bk, lanes);
    const int32_t rc = sp.estimate(pegs, groups, opts, out, q, /*threads=*/true);
    sp.sync_all();
    if (rc != CASIM_OK) return set_err(rc, sp.error());
    for (HipBackend* l : lanes) if (!l->ok()) return set_err(CASIM_ERR_HIP, l->msg);
    return CASIM_OK;
}

int32_t casim_estimate_batch(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts, casim_results* out) {
    g_err.clear();
    if (!ctx) return set_err(CASIM_ERR_INVALID, "null context");
    {
        std::vector<HipBackend*> lanes;
        if (wants_streams(ctx, pegs, groups, opts, &lanes)) return estimate_streamed(ctx, lanes, pegs, groups, opts, out, nullptr);
    }
    casim_problem* p = problem_create(ctx, pegs, groups, opts, /*one_shot=*/true);
    if (!p) return g_err.empty() ? CASIM_ERR_INVALID : (casim_device_count() > 0 ? CASIM_ERR_INVALID : CASIM_ERR_NO_DEVICE);
    int32_t rc = casim_problem_run(p);
    if (rc == CASIM_OK) rc = casim_problem_fetch(p, out);
    const std::string keep = g_err;
    casim_problem_destroy(p);
    g_err = keep;
    return rc;
}

int32_t casim_estimate_batch_query(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                                   casim_results* out, int32_t* offsets_out, const casim_option_query* q) {
    g_err.clear();
    if (!ctx) return set_err(CASIM_ERR_INVALID, "null context");
    {
        std::vector<HipBackend*> lanes;
        if (wants_streams(ctx, pegs, groups, opts, &lanes)) {
            ctx->bk.bind(); ctx->bk.clear();
            for (HipBackend* l : lanes) l->clear();
            HipStreamed sp(ctx->bk, lanes);
            int32_t rc = sp.estimate(pegs, groups, opts, out, q, /*threads=*/true);
            if (rc == CASIM_OK && offsets_out) rc = sp.csr(nullptr, offsets_out);
            sp.sync_all();
            if (rc != CASIM_OK) return set_err(rc, sp.error());
            for (HipBackend* l : lanes) if (!l->ok()) return set_err(CASIM_ERR_HIP, l->msg);
            return CASIM_OK;
        }
    }
    casim_problem* p = problem_create(ctx, pegs, groups, opts, /*one_shot=*/true);
    if (!p) return g_err.empty() ? CASIM_ERR_INVALID : (casim_device_count() > 0 ? CASIM_ERR_INVALID : CASIM_ERR_NO_DEVICE);
    int32_t rc = casim_problem_run(p);
    // one wait for the device serves the expander's answer, the results and the offsets: the query's copies stay in flight until the
    // fetch has waited (a single simulation spent a third of its call in three separate round trips)
    const bool one_wait = q && out && !p->sp && !(q->join_stream && (q->dev_key_out || q->dev_packed_out));
    if (rc == CASIM_OK && q) {
        if (one_wait) { rc = p->prob->best_option_query(q, /*defer_sync=*/true); if (rc != CASIM_OK) set_err(rc, p->prob->error()); }
        else rc = casim_best_option_sims(p, q);
    }
    if (rc == CASIM_OK && out) rc = casim_problem_fetch(p, out);
    if (one_wait) { const int32_t rc2 = p->prob->best_option_finish(/*synced=*/rc == CASIM_OK); if (rc == CASIM_OK && rc2 != CASIM_OK) rc = set_err(rc2, p->prob->error()); }
    if (rc == CASIM_OK && offsets_out) rc = casim_problem_csr(p, nullptr, offsets_out);
    const std::string keep = g_err;
    casim_problem_destroy(p);
    g_err = keep;
    return rc;
}

int32_t casim_estimate_batch_timed(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                                   casim_results* out, const casim_option_query* q, double phase_ms_out[8]) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (!phase_ms_out) return set_err(CASIM_ERR_INVALID, "null phase_ms_out");
    for (int i = 0; i < 8; ++i) phase_ms_out[i] = 0.0;
    const auto t0 = clk::now();
    casim_options one; memset(&one, 0, sizeof one);
    if (opts) one = *opts;
    one.n_streams = 0;   // (phase by phase on ONE stream: the streamed form has no phases to drain between)
    casim_problem* p = casim_problem_create(ctx, pegs, groups, &one);   // init() ends with a stream sync
    if (!p) return g_err.empty() ? CASIM_ERR_INVALID : (casim_device_count() > 0 ? CASIM_ERR_INVALID : CASIM_ERR_NO_DEVICE);
    HipBackend& bk = ctx->bk;
    const auto t1 = clk::now();
    p->prob->run_feasibility(); bk.sync();
    const auto t2 = clk::now();
    p->prob->run_order(); bk.sync();
    const auto t3 = clk::now();
    p->prob->run_pack(); bk.sync();
    const auto t4 = clk::now();
    int32_t rc = p->prob->run_mark();
    if (rc == CASIM_OK && q) { rc = p->prob->best_option_query(q); bk.sync(); }
    const auto t5 = clk::now();
    if (rc == CASIM_OK && out) rc = p->prob->fetch(out);
    const auto t6 = clk::now();
    if (rc != CASIM_OK) set_err(rc, p->prob->error());
    const std::string keep = g_err;
    casim_problem_destroy(p);
    g_err = keep;
    const auto t7 = clk::now();
    phase_ms_out[0] = ms(t0, t1); phase_ms_out[1] = ms(t1, t2); phase_ms_out[2] = ms(t2, t3); phase_ms_out[3] = ms(t3, t4);
    phase_ms_out[4] = ms(t4, t5); phase_ms_out[5] = ms(t5, t6); phase_ms_out[6] = ms(t0, t7);
    return rc;
}

int32_t casim_feasibility(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, uint64_t* out_bits) {
    g_err.clear();
    if (!ctx || !groups || !out_bits) return set_err(CASIM_ERR_INVALID, "null argument");
    casim_groups g = *groups;
    g.peg_offsets = nullptr; g.peg_index = nullptr;
    casim_problem* p = problem_create(ctx, pegs, &g, nullptr, /*one_shot=*/true);
    if (!p) return CASIM_ERR_INVALID;
    int32_t rc = p->prob->run_feasibility();
    if (rc == CASIM_OK) rc = p->prob->fetch_bits(out_bits);
    if (rc != CASIM_OK) set_err(rc, p->prob->error());
    const std::string keep = g_err;
    casim_problem_destroy(p);
    g_err = keep;
    return rc;
}

int32_t casim_feasibility_reasons(casim_ctx* ctx, const casim_pegs* pegs, const casim_groups* groups, const uint64_t* port_block,
                                  uint16_t* out_codes) {
    g_err.clear();
    if (!ctx || !groups || !out_codes) return set_err(CASIM_ERR_INVALID, "null argument");
    casim_groups g = *groups;
    g.peg_offsets = nullptr; g.peg_index = nullptr;
    casim_problem* p = problem_create(ctx, pegs, &g, nullptr, /*one_shot=*/true);
    if (!p) return CASIM_ERR_INVALID;
    const int32_t rc = p->prob->reasons(port_block, out_codes);
    if (rc != CASIM_OK) set_err(rc, p->prob->error());
    const std::string keep = g_err;
    casim_problem_destroy(p);
    g_err = keep;
    return rc;
}

int32_t casim_best_option(casim_problem* p, const int32_t* kinds, int32_t n_kinds, int32_t group_id_base, int32_t* best_ng_out,
                          int32_t* n_best_out, uint8_t* best_set_out, int64_t* key_out, void* dev_key_out) {
    PROB_ENTER(p);
    if (p->sp) return set_err(CASIM_ERR_INVALID, "a streamed batch reduces per simulation: use casim_best_option_sims with per_sim = 1");
    PROB_RET(p, p->prob->best_option(kinds, n_kinds, group_id_base, best_ng_out, n_best_out, best_set_out, key_out, dev_key_out));
}

int32_t casim_best_option_sims(casim_problem* p, const casim_option_query* q) {
    PROB_ENTER(p);
    if (p->sp) { clear_lanes(p); const int32_t rc = p->sp->best_option_query(q); if (rc != CASIM_OK) return set_err(rc, p->sp->error()); return lanes_ok(p); }
    const int32_t rc = p->prob->best_option_query(q);
    if (rc == CASIM_OK && q && q->join_stream && (q->dev_key_out || q->dev_packed_out) && q->join_stream != (void*)p->ctx->bk.stream) {
        p->ctx->bk.mark(); p->ctx->bk.make_wait(q->join_stream);   // an uncut batch ran on the context's stream: the caller's stream waits for it
    }
    PROB_RET(p, rc);
}

// ---- one process, several devices (the caller is one goroutine) -----------------------------------------
casim_mctx* casim_mctx_create(const int32_t* devices, int32_t n_devices, int32_t use_rccl) {
    g_err.clear();
    if (!devices || n_devices <= 0 || n_devices > 64) { set_err(CASIM_ERR_INVALID, "bad device list"); return nullptr; }
    casim_mctx* m = new (std::nothrow) casim_mctx();
    if (!m) { set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
    for (int i = 0; i < n_devices; ++i) {
        casim_ctx* c = casim_ctx_create(devices[i], nullptr);
        if (!c) { const std::string keep = g_err; casim_mctx_destroy(m); g_err = keep; return nullptr; }
        m->ctxs.push_back(c);
    }
    if (use_rccl) {
        // RCCL wants distinct devices in one communicator; a list that names one device twice (tests on a 1-GPU box) reduces on the host
        bool distinct = true;
        for (int i = 0; i < n_devices; ++i) for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
        std::string err;
        if (distinct && m->rccl.load(err)) {
            m->comms.assign((size_t)n_devices, nullptr);
            const int rc = m->rccl.CommInitAll(m->comms.data(), n_devices, devices);
            if (rc != 0) { set_err(CASIM_ERR_HIP, std::string("ncclCommInitAll: ") + (m->rccl.GetErrorString ? m->rccl.GetErrorString(rc) : "failed")); casim_mctx_destroy(m); return nullptr; }
            m->use_rccl = true;
        } else if (distinct) { set_err(CASIM_ERR_HIP, err); casim_mctx_destroy(m); return nullptr; }
    }
    return m;
}
void casim_mctx_destroy(casim_mctx* m) {
    if (!m) return;
    for (size_t i = 0; i < m->comms.size(); ++i) if (m->comms[i]) { (void)hipSetDevice(m->ctxs[i]->bk.device); (void)m->rccl.CommDestroy(m->comms[i]); }
    for (casim_ctx* c : m->ctxs) casim_ctx_destroy(c);
    delete m;
}
int32_t casim_mctx_info(const casim_mctx* m, int32_t* n_devices_out, int32_t* uses_rccl_out, int32_t* last_reduced_by_rccl_out, int32_t* groups_per_device_out) {
    if (!m) return CASIM_ERR_INVALID;
    if (n_devices_out) *n_devices_out = (int32_t)m->ctxs.size();
    if (uses_rccl_out) *uses_rccl_out = m->use_rccl ? 1 : 0;
    if (last_reduced_by_rccl_out) *last_reduced_by_rccl_out = m->last_reduced_by;
    if (groups_per_device_out) for (size_t d = 0; d < m->last_groups.size(); ++d) groups_per_device_out[d] = m->last_groups[d];
    return CASIM_OK;
}
int32_t casim_estimate_batch_multi(casim_mctx* m, const casim_pegs* pegs, const casim_groups* groups, const casim_options* opts,
                                   casim_results* out, int32_t* offsets_out, const casim_option_query* q) {
    g_err.clear();
    if (!m) return set_err(CASIM_ERR_INVALID, "null multi-device context");
    if (opts && opts->chain_last_index && m->ctxs.size() > 1)
        return set_err(CASIM_ERR_INVALID, "chain_last_index needs every group of a simulation in one problem: not with node groups sharded over devices");
    std::vector<HipBackend*> bks;
    for (casim_ctx* c : m->ctxs) { c->bk.bind(); c->bk.clear(); bks.push_back(&c->bk); }
    casim::MultiProblemT<HipBackend> mp(bks);
    casim::MultiProblemT<HipBackend>::ReduceFn hook;
    if (m->use_rccl) hook = [m](const std::vector<int64_t*>& keys, int S) {
        // ONE collective over xGMI: all-reduce(min) of the per-simulation packed keys, in place, every device's own stream
        if (m->rccl.GroupStart() != 0) return false;
        bool ok = true;
        for (size_t d = 0; d < keys.size(); ++d) {
            m->ctxs[d]->bk.bind();
            ok = ok && m->rccl.AllReduce(keys[d], keys[d], (size_t)S, Rccl::kInt64, Rccl::kMin, m->comms[d], m->ctxs[d]->bk.stream) == 0;
        }
        return m->rccl.GroupEnd() == 0 && ok;
    };
    const int32_t rc = mp.run(pegs, groups, opts, out, offsets_out, q, hook);
    m->last_reduced_by = mp.reduced_by(); m->last_groups = mp.groups_per_device();
    if (rc != CASIM_OK) return set_err(rc, mp.error());
    for (casim_ctx* c : m->ctxs) if (!c->bk.ok()) return set_err(CASIM_ERR_HIP, c->bk.msg);
    return CASIM_OK;
}

// ---- measurement -----------------------------------------------------------------------------
// (a streamed batch: part 0 on its lane — one launch of each kernel class with the device to itself)
int32_t casim_problem_time(casim_problem* p, int32_t iters, float* total_ms_out, float* kernel_ms_out) {
    PROB_ENTER(p);
    if (iters <= 0) return set_err(CASIM_ERR_INVALID, "iters must be > 0");
    HipBackend& bk = p->bk0();
    if (p->sp) { p->sp->sync_all(); bk.clear(); }
    hipEvent_t ev[4];
    for (auto& e : ev) bk.check(hipEventCreate(&e), "hipEventCreate");
    double tot = 0, kf = 0, ko = 0, kp = 0;
    for (int i = 0; i < iters; ++i) {
        bk.check(hipEventRecord(ev[0], bk.stream), "hipEventRecord");
        p->prob->run_feasibility();
        bk.check(hipEventRecord(ev[1], bk.stream), "hipEventRecord");
        p->prob->run_order();
        bk.check(hipEventRecord(ev[2], bk.stream), "hipEventRecord");
        p->prob->run_pack();
        bk.check(hipEventRecord(ev[3], bk.stream), "hipEventRecord");
        bk.check(hipEventSynchronize(ev[3]), "hipEventSynchronize");
        float a = 0, b = 0, c = 0, d = 0;
        (void)hipEventElapsedTime(&a, ev[0], ev[3]); (void)hipEventElapsedTime(&b, ev[0], ev[1]);
        (void)hipEventElapsedTime(&c, ev[1], ev[2]); (void)hipEventElapsedTime(&d, ev[2], ev[3]);
        tot += a; kf += b; ko += c; kp += d;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (total_ms_out) *total_ms_out = (float)(tot / iters);
    if (kernel_ms_out) { kernel_ms_out[0] = (float)(kf / iters); kernel_ms_out[1] = (float)(ko / iters); kernel_ms_out[2] = (float)(kp / iters); }
    // mark as run so that fetch works after a timing loop
    const int32_t rc = p->sp ? p->sp->run() : p->prob->run();
    PROB_RET(p, rc != CASIM_OK ? rc : (bk.ok() ? CASIM_OK : CASIM_ERR_HIP));
}

// The feasibility launch of a problem alone, `iters` times back to back between two HIP events on the launch stream: the average includes
// the gap between two launches (conservative for a roofline fraction).  info_out (may be NULL): [0] 1 = feas_stream_kernel (round 5),
// 0 = another form, [1] 1 = its lean instantiation, [2] bit 0: no upper-half terms, bit 1: NodeUnschedulable on a spare mask bit, [3] workgroups of the launch.
int32_t casim_problem_time_feasibility(casim_problem* p, int32_t iters, float* ms_per_launch_out, int32_t info_out[4]) {
    PROB_ENTER(p);
    if (iters <= 0 || !ms_per_launch_out) return set_err(CASIM_ERR_INVALID, "iters must be > 0");
    HipBackend& bk = p->bk0();
    if (p->sp) { p->sp->sync_all(); bk.clear(); }
    hipEvent_t ev[2];
    for (auto& e : ev) bk.check(hipEventCreate(&e), "hipEventCreate");
    p->prob->run_feasibility();   // (warm: code object, caches)
    bk.check(hipEventRecord(ev[0], bk.stream), "hipEventRecord");
    for (int i = 0; i < iters; ++i) p->prob->run_feasibility();
    bk.check(hipEventRecord(ev[1], bk.stream), "hipEventRecord");
    bk.check(hipEventSynchronize(ev[1]), "hipEventSynchronize");
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
    for (auto& e : ev) (void)hipEventDestroy(e);
    *ms_per_launch_out = ms / (float)iters;
    if (info_out) p->prob->feasibility_info(info_out);
    PROB_RET(p, bk.ok() ? CASIM_OK : CASIM_ERR_HIP);
}

static const int kMarkedRuns = 64;
// (a streamed batch: the events go around the kernels of part 0 while the other parts run beside it, as in an unmarked run)
int32_t casim_problem_run_marked(casim_problem* p) {
    PROB_ENTER(p);
    HipBackend& bk = p->bk0();
    if (p->sp) { clear_lanes(p); p->sp->fork(); }
    if (p->marks.empty()) {
        p->marks.resize((size_t)kMarkedRuns * 4);
        for (auto& e : p->marks) bk.check(hipEventCreate(&e), "hipEventCreate");
    }
    hipEvent_t* ev = p->marks.data() + (size_t)(p->n_marked % kMarkedRuns) * 4;
    bk.check(hipEventRecord(ev[0], bk.stream), "hipEventRecord");
    p->prob->run_feasibility();
    bk.check(hipEventRecord(ev[1], bk.stream), "hipEventRecord");
    p->prob->run_order();
    bk.check(hipEventRecord(ev[2], bk.stream), "hipEventRecord");
    p->prob->run_pack();
    bk.check(hipEventRecord(ev[3], bk.stream), "hipEventRecord");
    p->prob->run_mark();
    if (p->sp) {
        for (size_t i = 1; i < p->sp->n_parts(); ++i) { const int32_t rc = p->sp->part(i).prob->run(); if (rc != CASIM_OK) return set_err(rc, p->sp->part(i).prob->error()); }
        p->sp->set_ran();
    }
    p->n_marked++;
    if (p->sp) return lanes_ok(p);
    PROB_RET(p, bk.ok() ? CASIM_OK : CASIM_ERR_HIP);
}

int32_t casim_problem_marked_ms(casim_problem* p, float* total_ms_out, float* kernel_ms_out, int32_t* n_runs_out) {
    PROB_ENTER(p);
    HipBackend& bk = p->bk0();
    bk.sync();
    const int n = p->n_marked < kMarkedRuns ? p->n_marked : kMarkedRuns;
    double tot = 0, k[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        hipEvent_t* ev = p->marks.data() + (size_t)i * 4;
        float a = 0, b = 0, c = 0, d = 0;
        (void)hipEventElapsedTime(&a, ev[0], ev[3]); (void)hipEventElapsedTime(&b, ev[0], ev[1]);
        (void)hipEventElapsedTime(&c, ev[1], ev[2]); (void)hipEventElapsedTime(&d, ev[2], ev[3]);
        tot += a; k[0] += b; k[1] += c; k[2] += d;
    }
    if (total_ms_out) *total_ms_out = n ? (float)(tot / n) : 0.f;
    if (kernel_ms_out) for (int i = 0; i < 3; ++i) kernel_ms_out[i] = n ? (float)(k[i] / n) : 0.f;
    if (n_runs_out) *n_runs_out = n;
    p->n_marked = 0;
    PROB_RET(p, bk.ok() ? CASIM_OK : CASIM_ERR_HIP);
}

// ---- filter-out-schedulable ---------------------------------------------------
int32_t casim_try_schedule_pods(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes, const casim_pod_sequence* seq,
                                int32_t* node_out, int32_t* last_index_out, int32_t* n_scheduled_out) {
    g_err.clear();
    if (!ctx) return set_err(CASIM_ERR_INVALID, "null context");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    casim::SchedulerT<HipBackend> s(bk);
    int32_t rc = s.init(classes, nodes, seq);
    if (rc == CASIM_OK) rc = s.run();
    if (rc == CASIM_OK) rc = s.fetch(node_out, last_index_out, n_scheduled_out);
    if (rc < 0) set_err(rc, s.error());
    return rc;
}

// ---- Estimate on the whole snapshot ------------------------------------------------
int32_t casim_estimate_on_cluster(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                  const casim_cluster_estimate* params, casim_cluster_estimate_result* out) {
    g_err.clear();
    if (!ctx) return set_err(CASIM_ERR_INVALID, "null context");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    casim::ClusterEstimatorT<HipBackend> s(bk);
    int32_t rc = s.init(classes, nodes, params);
    if (rc == CASIM_OK) rc = s.run();
    if (rc == CASIM_OK) rc = s.fetch(out);
    if (rc < 0) set_err(rc, s.error());
    return rc;
}

// ---- resident cluster (second half) ------------------------------------------------------
casim_cluster* casim_cluster_create(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes) {
    g_err.clear();
    if (!ctx) { set_err(CASIM_ERR_INVALID, "null context"); return nullptr; }
    ctx->bk.bind(); ctx->bk.clear();
    casim_cluster* h = new (std::nothrow) casim_cluster();
    if (!h) { set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
    h->ctx = ctx; h->c = new (std::nothrow) casim::ClusterT<HipBackend>(ctx->bk);
    if (!h->c) { delete h; set_err(CASIM_ERR_NOMEM, "out of memory"); return nullptr; }
    const int32_t rc = h->c->init(classes, nodes);
    if (rc != CASIM_OK) { set_err(rc, h->c->error()); delete h->c; delete h; return nullptr; }
    return h;
}
void casim_cluster_destroy(casim_cluster* h) {
    if (!h) return;
    h->ctx->bk.bind(); (void)hipStreamSynchronize(h->ctx->bk.stream);
    delete h->c; delete h;
}
#define CLUSTER_ENTER(h) g_err.clear(); if (!(h) || !(h)->c) return set_err(CASIM_ERR_INVALID, "null cluster"); (h)->ctx->bk.bind(); (h)->ctx->bk.clear()
int32_t casim_cluster_update_nodes(casim_cluster* h, int32_t n, const int32_t* node_index, const casim_groups* rows) {
    CLUSTER_ENTER(h);
    const int32_t rc = h->c->update_nodes(n, node_index, rows);
    if (rc < 0) set_err(rc, h->c->error());
    return rc;
}
int32_t casim_cluster_try_schedule_pods(casim_cluster* h, const casim_pod_sequence* seq, int32_t commit, int32_t* node_out, int32_t* last_index_out,
                                        int32_t* n_scheduled_out) {
    CLUSTER_ENTER(h);
    const int32_t rc = h->c->try_schedule(seq, commit, node_out, last_index_out, n_scheduled_out);
    if (rc < 0) set_err(rc, h->c->error());
    return rc;
}
int32_t casim_cluster_simulate_node_removals(casim_cluster* h, const casim_removal_candidates* cand, casim_removal_results* out) {
    CLUSTER_ENTER(h);
    const int32_t rc = h->c->simulate_removals(cand, out);
    if (rc < 0) set_err(rc, h->c->error());
    return rc;
}
int32_t casim_cluster_fetch_nodes(casim_cluster* h, int64_t* init_req_out, int32_t* init_pods_out, uint64_t* init_excl_out) {
    CLUSTER_ENTER(h);
    const int32_t rc = h->c->fetch_nodes(init_req_out, init_pods_out, init_excl_out);
    if (rc < 0) set_err(rc, h->c->error());
    return rc;
}
int32_t casim_cluster_stats(const casim_cluster* h, int64_t out[4]) {
    if (!h || !h->c || !out) return CASIM_ERR_INVALID;
    h->c->stats(out);
    return CASIM_OK;
}
int32_t casim_cluster_forget_commits(casim_cluster* h) {
    g_err.clear();
    if (!h || !h->c) return set_err(CASIM_ERR_INVALID, "null cluster");
    h->c->forget_commits();
    return CASIM_OK;
}

// ---- scale-down removal simulation ----------------------------------------------
int32_t casim_simulate_node_removals(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                     const casim_removal_candidates* cand, casim_removal_results* out) {
    g_err.clear();
    if (!ctx) return set_err(CASIM_ERR_INVALID, "null context");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    casim::SchedulerT<HipBackend> s(bk);
    int32_t rc = s.init_removals(classes, nodes, cand);
    if (rc == CASIM_OK) rc = s.run();
    if (rc == CASIM_OK) rc = s.fetch_removals(out);
    if (rc < 0) set_err(rc, s.error());
    return rc;
}
int32_t casim_last_removals_info(int32_t info_out[4]) {
    if (!info_out) return CASIM_ERR_INVALID;
    const int32_t* li = casim::last_removals_info();
    for (int i = 0; i < 4; ++i) info_out[i] = li[i];
    return CASIM_OK;
}
int32_t casim_last_chain_info(int32_t info_out[4]) {
    if (!info_out) return CASIM_ERR_INVALID;
    const int32_t* ci = casim::last_chain_info();
    for (int i = 0; i < 4; ++i) info_out[i] = ci[i];
    return CASIM_OK;
}
int32_t casim_time_node_removals(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                 const casim_removal_candidates* cand, int32_t iters, float* ms_out) {
    g_err.clear();
    if (!ctx || iters <= 0 || !ms_out) return set_err(CASIM_ERR_INVALID, "bad argument");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    casim::SchedulerT<HipBackend> s(bk);
    int32_t rc = s.init_removals(classes, nodes, cand);
    if (rc != CASIM_OK) { if (rc < 0) set_err(rc, s.error()); return rc; }
    rc = s.run();
    if (rc == CASIM_OK) rc = s.confirm_kernel();   // (a one-wave kernel whose LDS log is smaller than the worst case may have given up: time what answers)
    hipEvent_t e0, e1;
    bk.check(hipEventCreate(&e0), "hipEventCreate"); bk.check(hipEventCreate(&e1), "hipEventCreate");
    bk.check(hipEventRecord(e0, bk.stream), "hipEventRecord");
    for (int i = 0; i < iters && rc == CASIM_OK; ++i) rc = s.run();
    bk.check(hipEventRecord(e1, bk.stream), "hipEventRecord");
    bk.check(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_out = ms;
    if (rc < 0) return set_err(rc, s.error());
    return bk.ok() ? CASIM_OK : set_err(CASIM_ERR_HIP, bk.msg);
}

int32_t casim_time_try_schedule_pods(casim_ctx* ctx, const casim_pegs* classes, const casim_groups* nodes,
                                     const casim_pod_sequence* seq, int32_t iters, float* ms_out) {
    g_err.clear();
    if (!ctx || iters <= 0 || !ms_out) return set_err(CASIM_ERR_INVALID, "bad argument");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    casim::SchedulerT<HipBackend> s(bk);
    int32_t rc = s.init(classes, nodes, seq);
    if (rc != CASIM_OK) { if (rc < 0) set_err(rc, s.error()); return rc; }
    rc = s.run();  // warm-up (also the LDS opt-in)
    hipEvent_t e0, e1;
    bk.check(hipEventCreate(&e0), "hipEventCreate"); bk.check(hipEventCreate(&e1), "hipEventCreate");
    bk.check(hipEventRecord(e0, bk.stream), "hipEventRecord");
    for (int i = 0; i < iters && rc == CASIM_OK; ++i) rc = s.run();
    bk.check(hipEventRecord(e1, bk.stream), "hipEventRecord");
    bk.check(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_out = ms;
    if (rc < 0) return set_err(rc, s.error());
    return bk.ok() ? CASIM_OK : set_err(CASIM_ERR_HIP, bk.msg);
}

int32_t casim_copy_bandwidth(casim_ctx* ctx, int64_t bytes, int32_t iters, double* gbps_out) {
    g_err.clear();
    if (!ctx || bytes < 4096 || iters <= 0 || !gbps_out) return set_err(CASIM_ERR_INVALID, "bad argument");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    const int64_t n = bytes / 16;
    uint4* a = (uint4*)bk.alloc((size_t)n * 16); uint4* b = (uint4*)bk.alloc((size_t)n * 16);
    if (!bk.ok()) { bk.free(a); bk.free(b); return set_err(CASIM_ERR_HIP, bk.msg); }
    bk.zero(a, (size_t)n * 16);
    hipEvent_t e0, e1;
    bk.check(hipEventCreate(&e0), "hipEventCreate"); bk.check(hipEventCreate(&e1), "hipEventCreate");
    bk.launch(casim::copy_probe_kernel, 2048, 1, 256, (size_t)0, (const uint4*)a, b, n);
    bk.check(hipEventRecord(e0, bk.stream), "hipEventRecord");
    for (int i = 0; i < iters; ++i) bk.launch(casim::copy_probe_kernel, 2048, 1, 256, (size_t)0, (const uint4*)a, b, n);
    bk.check(hipEventRecord(e1, bk.stream), "hipEventRecord");
    bk.check(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    bk.free(a); bk.free(b);
    *gbps_out = ms > 0 ? (2.0 * (double)n * 16.0 * iters) / (ms * 1e-3) / 1e9 : 0.0;
    return bk.ok() ? CASIM_OK : set_err(CASIM_ERR_HIP, bk.msg);
}

int32_t casim_stream_probe(casim_ctx* ctx, int64_t bytes, int32_t lane_bytes, int32_t iters, double* gbps_out) {
    g_err.clear();
    if (!ctx || bytes < 4096 || iters <= 0 || !gbps_out || (lane_bytes != 4 && lane_bytes != 16 && lane_bytes != 0)) return set_err(CASIM_ERR_INVALID, "bad argument");
    HipBackend& bk = ctx->bk; bk.bind(); bk.clear();
    const int n_waves = 65536;   // scalar form: one wave per block, 65536 regions
    const int64_t n_words = lane_bytes == 0 ? bytes / 4 / n_waves / 8 * 8 * n_waves : bytes / 16 * 4;
    uint32_t* a = (uint32_t*)bk.alloc((size_t)n_words * 4); uint32_t* sink = (uint32_t*)bk.alloc(4096 * 4);
    if (!bk.ok()) { bk.free(a); bk.free(sink); return set_err(CASIM_ERR_HIP, bk.msg); }
    bk.zero(a, (size_t)n_words * 4);
    hipEvent_t e0, e1;
    bk.check(hipEventCreate(&e0), "hipEventCreate"); bk.check(hipEventCreate(&e1), "hipEventCreate");
    auto go = [&]() {
        if (lane_bytes == 0) bk.launch(casim::stream_probe_scalar_kernel, n_waves, 1, 64, (size_t)0, (const uint32_t*)a, n_words / n_waves, sink);
        else if (lane_bytes == 4) bk.launch(casim::stream_probe_kernel<4>, 4096, 1, 256, (size_t)0, (const uint32_t*)a, n_words, sink);
        else bk.launch(casim::stream_probe_kernel<16>, 4096, 1, 256, (size_t)0, (const uint32_t*)a, n_words, sink);
    };
    go();
    bk.check(hipEventRecord(e0, bk.stream), "hipEventRecord");
    for (int i = 0; i < iters; ++i) go();
    bk.check(hipEventRecord(e1, bk.stream), "hipEventRecord");
    bk.check(hipEventSynchronize(e1), "hipEventSynchronize");
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    bk.free(a); bk.free(sink);
    *gbps_out = ms > 0 ? ((double)n_words * 4.0 * iters) / (ms * 1e-3) / 1e9 : 0.0;
    return bk.ok() ? CASIM_OK : set_err(CASIM_ERR_HIP, bk.msg);
}

}  // extern "C"
