// casim_emu.cpp — cooperative-fiber wave64 emulator (TEST INFRASTRUCTURE ONLY). See casim_emu.h.
#include "casim_emu.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// Fiber switch.  glibc's swapcontext saves and restores the signal mask — one rt_sigprocmask system call per switch, and a kernel under
// the emulator switches at every ballot / reduction / barrier of every lane: a third of the CPU suite's time was spent in the kernel.
// x86-64: the callee-saved registers, the stack pointer and the two floating-point control words, ~20 instructions in user space.
// Anything else (or -DCASIM_EMU_UCONTEXT): ucontext as before.
#if defined(__x86_64__) && !defined(CASIM_EMU_UCONTEXT)
#define CASIM_EMU_ASM_SWITCH 1
extern "C" void casim_emu_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl casim_emu_switch
    .hidden casim_emu_switch
    .type casim_emu_switch,@function
casim_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size casim_emu_switch, .-casim_emu_switch
)");
#else
#include <ucontext.h>
#endif

// ThreadSanitizer (tests/tools/sanitize_cpu.sh, pass 3) has to be told about every switch of stacks: it keeps a shadow stack per fiber
#if defined(__SANITIZE_THREAD__)
#define CASIM_EMU_TSAN 1
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
}
#endif

namespace casim_emu {
namespace {

constexpr size_t kStack = 256 * 1024;

#ifdef CASIM_EMU_ASM_SWITCH
struct Context { void* sp = nullptr; };
inline void switch_to(Context& from, Context& to) { casim_emu_switch(&from.sp, to.sp); }
// a fresh stack whose first switch_to "returns" into entry() (which never returns): the frame casim_emu_switch pops, top down —
// a null return address for entry (keeps its stack pointer at 8 mod 16, as after a call), entry, rbp, rbx, r12-r15, the control words
inline void make_context(Context& c, char* stack, size_t size, void (*entry)()) {
    uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
    uint64_t* f = (uint64_t*)top;
    f[-1] = 0;
    f[-2] = (uint64_t)(uintptr_t)entry;
    for (int i = 3; i <= 8; ++i) f[-i] = 0;
    f[-9] = 0x1F80ull | (0x037Full << 32);   // MXCSR, x87 control word: the defaults
    c.sp = (void*)(f - 9);
}
#else
struct Context { ucontext_t uc; };
inline void switch_to(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
inline void make_context(Context& c, char* stack, size_t size, void (*entry)()) {
    getcontext(&c.uc);
    c.uc.uc_stack.ss_sp = stack;
    c.uc.uc_stack.ss_size = size;
    c.uc.uc_link = nullptr;
    makecontext(&c.uc, entry, 0);
}
#endif

struct Fiber {
    Context uc;
    FiberCtx ctx;
    char* stack = nullptr;
    bool done = false;
#ifdef CASIM_EMU_TSAN
    void* tsan = nullptr;
#endif
};

struct WaveState {
    int size = 0;        // lanes in this wave
    int arrived = 0;
    uint64_t gen = 0;
    uint64_t slot[64];
    uint64_t result = 0;
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    int n = 0;
    int cur = 0;
    int arrived = 0;
    uint64_t gen = 0;
    int live = 0;
    uint64_t events = 0;  // bumped by every completed collective / finished fiber (deadlock detection)
    Context sched;
#ifdef CASIM_EMU_TSAN
    void* tsan_sched = nullptr;
#endif
    const std::function<void()>* body = nullptr;
    std::vector<char> smem;
};

// The block a thread is running: one per THREAD — the parts of a streamed call launch their kernels from the host pool's workers
// (CASIM_EMU_THREADS=1: casim_streams.h with threads on, as the product runs it on the device).
thread_local Block* g_blk = nullptr;

inline void to_sched(Block& b, Fiber& f) {
#ifdef CASIM_EMU_TSAN
    __tsan_switch_to_fiber(b.tsan_sched, 0);
#endif
    switch_to(f.uc, b.sched);
}

void yield_fiber() {
    Block& b = *g_blk;
    Fiber& f = b.fibers[b.cur];
    to_sched(b, f);
}

void trampoline() {
    Block& b = *g_blk;
    (*b.body)();
    b.fibers[b.cur].done = true;
    b.live--;
    b.events++;
    to_sched(b, b.fibers[b.cur]);
}

WaveState& my_wave() { return g_blk->waves[g_blk->fibers[g_blk->cur].ctx.tid >> 6]; }
int my_lane() { return g_blk->fibers[g_blk->cur].ctx.tid & 63; }

// two-phase wave collective: everyone deposits, last arriver combines, everyone reads
template <class Combine>
uint64_t wave_collective(uint64_t v, Combine combine) {
    WaveState& w = my_wave();
    const int lane = my_lane();
    w.slot[lane] = v;
    const uint64_t g = w.gen;
    if (++w.arrived == w.size) {
        w.result = combine(w);
        w.arrived = 0;
        w.gen++;
        g_blk->events++;
    } else {
        while (w.gen == g) yield_fiber();
    }
    return w.result;
}

}  // namespace

FiberCtx& cur() { return g_blk->fibers[g_blk->cur].ctx; }
char* dyn_smem() { return g_blk->smem.data(); }

void block_sync() {
    Block& b = *g_blk;
    const uint64_t g = b.gen;
    if (++b.arrived == b.n) {
        b.arrived = 0;
        b.gen++;
        b.events++;
    } else {
        while (b.gen == g) yield_fiber();
    }
}

uint64_t wave_ballot(bool p) {
    return wave_collective(p ? 1 : 0, [](WaveState& w) {
        uint64_t m = 0;
        for (int i = 0; i < w.size; ++i) m |= (w.slot[i] & 1ull) << i;
        return m;
    });
}

uint64_t wave_xchg_u64(uint64_t v, int src_lane) {
    // the exchanged vector must survive until every lane has read its source: snapshot per generation
    WaveState& w = my_wave();
    const int lane = my_lane();
    static thread_local uint64_t snap[64][64];  // [wave % 64][lane]
    const int widx = (g_blk->fibers[g_blk->cur].ctx.tid >> 6) & 63;
    w.slot[lane] = v;
    const uint64_t g = w.gen;
    if (++w.arrived == w.size) {
        for (int i = 0; i < 64; ++i) snap[widx][i] = i < w.size ? w.slot[i] : 0;
        w.arrived = 0;
        w.gen++;
        g_blk->events++;
    } else {
        while (w.gen == g) yield_fiber();
    }
    // second rendezvous so that nobody overwrites snap before all lanes have read it
    const uint64_t out = snap[widx][src_lane & 63];
    const uint64_t g2 = w.gen;
    if (++w.arrived == w.size) {
        w.arrived = 0;
        w.gen++;
        g_blk->events++;
    } else {
        while (w.gen == g2) yield_fiber();
    }
    return out;
}

uint64_t wave_sum_u64(uint64_t v) {
    return wave_collective(v, [](WaveState& w) {
        uint64_t s = 0;
        for (int i = 0; i < w.size; ++i) s += w.slot[i];
        return s;
    });
}
uint64_t wave_max_u64(uint64_t v) {
    return wave_collective(v, [](WaveState& w) {
        uint64_t s = 0;
        for (int i = 0; i < w.size; ++i) s = w.slot[i] > s ? w.slot[i] : s;
        return s;
    });
}

void launch(int gx, int gy, int block, size_t smem, const std::function<void()>& body) {
    Block blk;
    blk.n = block;
    blk.fibers.resize((size_t)block);
    blk.waves.resize((size_t)(block + 63) / 64);
    blk.smem.assign(smem + 64, 0);
    blk.body = &body;
    for (int i = 0; i < block; ++i) blk.fibers[(size_t)i].stack = (char*)malloc(kStack);
    Block* prev = g_blk;
    g_blk = &blk;
#ifdef CASIM_EMU_TSAN
    blk.tsan_sched = __tsan_get_current_fiber();
#endif
    for (int by = 0; by < gy; ++by) {
        for (int bx = 0; bx < gx; ++bx) {
            blk.arrived = 0; blk.gen = 0; blk.live = block;
            for (size_t w = 0; w < blk.waves.size(); ++w) {
                blk.waves[w].arrived = 0; blk.waves[w].gen = 0;
                const int left = block - (int)w * 64;
                blk.waves[w].size = left < 64 ? left : 64;
            }
            // LDS content is undefined at kernel start: poison it so that reads of unwritten LDS show up
            memset(blk.smem.data(), 0xA5, blk.smem.size());
            for (int i = 0; i < block; ++i) {
                Fiber& f = blk.fibers[(size_t)i];
                f.ctx = FiberCtx{i, bx, by, block, gx};
                f.done = false;
                make_context(f.uc, f.stack, kStack, trampoline);
#ifdef CASIM_EMU_TSAN
                if (f.tsan) __tsan_destroy_fiber(f.tsan);
                f.tsan = __tsan_create_fiber(0);
#endif
            }
            long spins = 0;
            while (blk.live > 0) {
                bool progressed = false;
                for (int i = 0; i < block; ++i) {
                    if (blk.fibers[(size_t)i].done) continue;
                    blk.cur = i;
                    const uint64_t ev_before = blk.events;
#ifdef CASIM_EMU_TSAN
                    __tsan_switch_to_fiber(blk.fibers[(size_t)i].tsan, 0);
#endif
                    switch_to(blk.sched, blk.fibers[(size_t)i].uc);
                    progressed |= blk.events != ev_before;
                }
                // a block whose threads wait forever (divergent collective) would spin here
                if (!progressed && ++spins > 1000) {
                    fprintf(stderr, "casim_emu: deadlock in block (%d,%d): a collective was not reached by all lanes\n", bx, by);
                    abort();
                }
                if (progressed) spins = 0;
            }
        }
    }
    g_blk = prev;
#ifdef CASIM_EMU_TSAN
    for (int i = 0; i < block; ++i) if (blk.fibers[(size_t)i].tsan) __tsan_destroy_fiber(blk.fibers[(size_t)i].tsan);
#endif
    for (int i = 0; i < block; ++i) free(blk.fibers[(size_t)i].stack);
}

}  // namespace casim_emu
