// casim_emu.h — cooperative-fiber wave64 emulator (TEST INFRASTRUCTURE ONLY).
//
// Runs the kernel bodies of kubernetes_autoscaler_amd/csrc/casim_kernels.h on the host so that
// their index / prefix / rank logic can be checked against the oracle without a GPU.  One
// workgroup at a time; every "thread" is a ucontext fiber; a collective (ballot, wave exchange,
// wave reduction, block barrier) yields until every participant has arrived.  Deterministic
// (single OS thread, round-robin).  Never linked into libcasim.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <functional>

namespace casim_emu {
struct FiberCtx { int tid, bid, bidy, nthreads, nblocks; };
FiberCtx& cur();
char* dyn_smem();
void block_sync();
uint64_t wave_ballot(bool p);
uint64_t wave_xchg_u64(uint64_t v, int src_lane);
uint64_t wave_sum_u64(uint64_t v);
uint64_t wave_max_u64(uint64_t v);

// launch grid (gx, gy) of `block` threads with `smem` bytes of dynamic shared memory
void launch(int gx, int gy, int block, size_t smem, const std::function<void()>& body);
}  // namespace casim_emu
