// casim_device.h — thin device abstraction used by every kernel body in casim_kernels.h.
//
// Product build (hipcc, gfx950): the functions below are the CDNA4 wave64 primitives
// (ballot, DPP/bpermute shuffles, mbcnt, LDS, s_barrier).
//
// CASIM_HOST_EMU build (g++, tests/ only): the SAME kernel source is compiled for the host and
// run by a cooperative-fiber wave emulator (tests/emu/), so the kernels' index / rank /
// prefix logic can be checked against the oracle on a machine without a GPU.  The emulator is
// test infrastructure: it is never linked into libcasim.so and the product path fails loudly
// when no HIP device is present.
#pragma once
#include <stdint.h>

#if defined(CASIM_HOST_EMU)
// ------------------------------------------------------------------------------------------
// host emulation
// ------------------------------------------------------------------------------------------
#include <string.h>
#define CS_GLOBAL
#define CS_DEVICE inline
#define CS_RESTRICT
#define CS_LAUNCH_BOUNDS(t, w)
#include "casim_emu.h"  // tests/emu (add -Itests/emu); test infrastructure only
namespace cs {
CS_DEVICE int tid() { return casim_emu::cur().tid; }
CS_DEVICE int bid() { return casim_emu::cur().bid; }
CS_DEVICE int nthreads() { return casim_emu::cur().nthreads; }
CS_DEVICE int nblocks() { return casim_emu::cur().nblocks; }
CS_DEVICE int bid_y() { return casim_emu::cur().bidy; }
CS_DEVICE char* dyn_smem() { return casim_emu::dyn_smem(); }
CS_DEVICE void sync() { casim_emu::block_sync(); }
CS_DEVICE uint64_t ballot(bool p) { return casim_emu::wave_ballot(p); }
CS_DEVICE uint64_t readlane_u64(uint64_t v, int l) { return casim_emu::wave_xchg_u64(v, l); }
CS_DEVICE uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)casim_emu::wave_sum_u64(v); }
CS_DEVICE uint64_t wave_sum_u64(uint64_t v) { return casim_emu::wave_sum_u64(v); }
CS_DEVICE uint32_t wave_max_u32(uint32_t v) { return (uint32_t)casim_emu::wave_max_u64(v); }
CS_DEVICE int popc64(uint64_t v) { return __builtin_popcountll(v); }
CS_DEVICE int ffs64(uint64_t v) { return v ? __builtin_ctzll(v) : -1; }
CS_DEVICE int fls64(uint64_t v) { return v ? 63 - __builtin_clzll(v) : -1; }
CS_DEVICE uint64_t double_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
}  // namespace cs

#else
// ------------------------------------------------------------------------------------------
// gfx950 device build
// ------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define CS_GLOBAL __global__
#define CS_DEVICE __device__ __forceinline__
#define CS_RESTRICT __restrict__
#define CS_LAUNCH_BOUNDS(t, w) __launch_bounds__(t, w)
namespace cs {
CS_DEVICE int tid() { return (int)threadIdx.x; }
CS_DEVICE int bid() { return (int)blockIdx.x; }
CS_DEVICE int bid_y() { return (int)blockIdx.y; }
CS_DEVICE int nthreads() { return (int)blockDim.x; }
CS_DEVICE int nblocks() { return (int)gridDim.x; }
CS_DEVICE char* dyn_smem() {
    // Guideline 17: dynamic LDS base must stay 16-byte aligned; no static __shared__ anywhere.
    extern __shared__ __attribute__((aligned(16))) char casim_smem[];
    return casim_smem;
}
CS_DEVICE void sync() { __syncthreads(); }
CS_DEVICE uint64_t ballot(bool p) { return __ballot(p); }  // 64-bit on wave64
CS_DEVICE uint64_t readlane_u64(uint64_t v, int l) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl((int)lo, l, 64);
    hi = (uint32_t)__shfl((int)hi, l, 64);
    return ((uint64_t)hi << 32) | lo;
}
CS_DEVICE uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
    return v;
}
CS_DEVICE uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64);
        uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
CS_DEVICE uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
CS_DEVICE int popc64(uint64_t v) { return __popcll(v); }
CS_DEVICE int ffs64(uint64_t v) { return v ? (int)__builtin_ctzll(v) : -1; }
CS_DEVICE int fls64(uint64_t v) { return v ? 63 - (int)__builtin_clzll(v) : -1; }
CS_DEVICE uint64_t double_bits(double d) { return (uint64_t)__double_as_longlong(d); }
}  // namespace cs
#endif

namespace cs {
CS_DEVICE int lane() { return tid() & 63; }
// number of set bits of `mask` strictly below my lane (v_mbcnt on device)
CS_DEVICE int mbcnt(uint64_t mask) {
    int l = lane();
    uint64_t below = l == 0 ? 0ull : (~0ull >> (64 - l));
    return popc64(mask & below);
}
// bits [0, n) set; n may be <= 0 or >= 64
CS_DEVICE uint64_t low_mask(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1)); }
}  // namespace cs
