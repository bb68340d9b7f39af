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
#define CS_HOST_DEVICE inline
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
// the lanes of ONE wave rendezvous (LDS written by some lanes, read by others of the same wave): a ballot is the emulator's wave-level meeting point
CS_DEVICE void wave_sync() { (void)casim_emu::wave_ballot(true); }
CS_DEVICE void lds_order() { (void)casim_emu::wave_ballot(true); }   // (lanes are fibers here: the rendezvous IS the order)
CS_DEVICE void sched_fence() {}
CS_DEVICE int32_t load_relaxed_i32(const int32_t* p) { return *p; }
CS_DEVICE void atomic_add_i32(int32_t* p, int32_t v) { *p += v; }  // fibers of one block never run concurrently
// a word one block publishes for the blocks behind it (front_kernel): the emulator runs the blocks of a launch one after the other in
// ascending order, so a block only ever waits for words that are already there
CS_DEVICE void publish_u64(uint64_t* p, uint64_t v) { *p = v; }
CS_DEVICE uint64_t poll_u64(const uint64_t* p) { return *p; }
CS_DEVICE void atomic_add_i64(int64_t* p, int64_t v) { *p += v; }
CS_DEVICE void atomic_or_u64(uint64_t* p, uint64_t v) { *p |= v; }
CS_DEVICE void lds_or_u64(uint64_t* p, uint64_t v) { *p |= v; }
CS_DEVICE void lds_and_u64(uint64_t* p, uint64_t v) { *p &= v; }
CS_DEVICE void lds_sub_u32(uint32_t* p, uint32_t v) { *p -= v; }
CS_DEVICE uint64_t ballot(bool p) { return casim_emu::wave_ballot(p); }
CS_DEVICE uint64_t readlane_u64(uint64_t v, int l) { return casim_emu::wave_xchg_u64(v, l); }
CS_DEVICE uint32_t shfl_u32(uint32_t v, int l) { return (uint32_t)casim_emu::wave_xchg_u64(v, l); }
CS_DEVICE uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)casim_emu::wave_sum_u64(v); }
CS_DEVICE uint64_t wave_sum_u64(uint64_t v) { return casim_emu::wave_sum_u64(v); }
CS_DEVICE uint32_t wave_max_u32(uint32_t v) { return (uint32_t)casim_emu::wave_max_u64(v); }
CS_DEVICE void wave_sum_max_u32(uint32_t s, uint32_t m, uint32_t& sum, uint32_t& mx) { sum = wave_sum_u32(s); mx = wave_max_u32(m); }
CS_DEVICE uint32_t bcast_u32(uint32_t v, int uniform_lane) { return (uint32_t)casim_emu::wave_xchg_u64(v, uniform_lane); }
CS_DEVICE uint32_t uniform_u32(uint32_t v) { return v; }
template <int N> struct Words { uint32_t w[N]; };
template <int N> CS_DEVICE Words<N> const_load(const uint32_t* p) { Words<N> r; memcpy(r.w, p, 4 * N); return r; }
CS_DEVICE Words<4> load4(const uint32_t* p) { Words<4> r; memcpy(r.w, p, 16); return r; }
CS_DEVICE void store4(uint32_t* p, const Words<4>& v) { memcpy(p, v.w, 16); }
struct RecBase { const char* p; };
CS_DEVICE RecBase rec_base(const uint32_t* p) { return RecBase{(const char*)p}; }
template <int N> CS_DEVICE Words<N> rec_load(const RecBase& b, uint32_t byte_off) { Words<N> r; memcpy(r.w, b.p + byte_off, 4 * N); return r; }
template <int N> CS_DEVICE Words<N> rec_load_all(const RecBase& b, uint32_t byte_off) { return rec_load<N>(b, byte_off); }
CS_DEVICE bool lane_pred(uint64_t mask) { return ((mask >> (casim_emu::cur().tid & 63)) & 1ull) != 0; }
CS_DEVICE void keep_scalar(uint32_t&) {}
CS_DEVICE void keep_apart() {}
CS_DEVICE double estimate_rcp_f64(double x) { return 1.0 / x; }
template <class T> CS_DEVICE const T& kernarg_view(const T& param, size_t) { return param; }
CS_DEVICE int32_t opaque_i32(int32_t v) { return v; }
CS_DEVICE void consume_u32(uint32_t) {}
CS_DEVICE bool flag_set(uint32_t word, uint32_t bit) { return (word & bit) != 0; }
CS_DEVICE uint32_t uniform_div_u32(uint32_t a, uint32_t b) { return a / b; }
CS_DEVICE uint32_t uniform_div_u32_small(uint32_t a, uint32_t b) { return a / b; }
CS_DEVICE uint32_t scalar_min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
CS_DEVICE void write_lane_u32(uint32_t& v, uint32_t uniform_value, int uniform_lane) { if ((casim_emu::cur().tid & 63) == uniform_lane) v = uniform_value; }
CS_DEVICE uint32_t and_or_u32(uint32_t uniform_a, uint32_t b, uint32_t c) { return (uniform_a & b) | c; }
CS_DEVICE int uniform_i32(int v) { return v; }
CS_DEVICE uint32_t and_or_vvv(uint32_t a, uint32_t b, uint32_t c) { return (a & b) | c; }
CS_DEVICE void write_lane2_u32(uint32_t& lo, uint32_t& hi, uint64_t uniform_value, uint32_t uniform_lane) {
    if ((uint32_t)(casim_emu::cur().tid & 63) == uniform_lane) { lo = (uint32_t)uniform_value; hi = (uint32_t)(uniform_value >> 32); }
}
CS_DEVICE int popc64(uint64_t v) { return __builtin_popcountll(v); }
CS_DEVICE int ffs64(uint64_t v) { return v ? __builtin_ctzll(v) : -1; }
CS_DEVICE int fls64(uint64_t v) { return v ? 63 - __builtin_clzll(v) : -1; }
CS_DEVICE uint64_t double_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
CS_DEVICE double bits_double(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
}  // namespace cs

#else
// ------------------------------------------------------------------------------------------
// gfx950 device build
// ------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define CS_GLOBAL __global__
#define CS_DEVICE __device__ __forceinline__
#define CS_HOST_DEVICE __host__ __device__ inline
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
// LDS written by some lanes of a wave, read by other lanes of the SAME wave (waves of a block working on different groups: no s_barrier,
// which would need the same trip count in every wave): the wave's own LDS operations complete in order — wait for them, and keep the
// compiler from moving accesses across
CS_DEVICE void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// LDS written by one lane of a ONE-WAVE block and read by the others later in program order (or the other way round): the wave's LDS
// operations execute in the order they were issued, so nothing has to be waited for — only the compiler may not move accesses across
CS_DEVICE void lds_order() { __builtin_amdgcn_wave_barrier(); }
// instruction-scheduling fence: the machine scheduler may not move anything across (bounds live ranges of unrolled slot code)
CS_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// device-scope relaxed load / add of a counter shared by the waves of a block (served by L2, never a stale L1 line)
CS_DEVICE int32_t load_relaxed_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CS_DEVICE void atomic_add_i32(int32_t* p, int32_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a word one block publishes for the blocks behind it (front_kernel): device-scope release / acquire — the eight XCDs have an L2 each,
// a plain store would sit in the writer's
CS_DEVICE void publish_u64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
CS_DEVICE uint64_t poll_u64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
CS_DEVICE void atomic_add_i64(int64_t* p, int64_t v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CS_DEVICE void atomic_or_u64(uint64_t* p, uint64_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// bits set by several threads of the block in one LDS word (ds_or_b64)
CS_DEVICE void lds_or_u64(uint64_t* p, uint64_t v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the same without a value coming back (ds_and_b64 / ds_sub_u32: nothing to wait for)
CS_DEVICE void lds_and_u64(uint64_t* p, uint64_t v) { (void)__hip_atomic_fetch_and(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
CS_DEVICE void lds_sub_u32(uint32_t* p, uint32_t v) { (void)__hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
CS_DEVICE uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // the lane mask itself (__ballot goes through an int: two more VALU ops per call)
CS_DEVICE uint64_t readlane_u64(uint64_t v, int l) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__shfl((int)lo, l, 64);
    hi = (uint32_t)__shfl((int)hi, l, 64);
    return ((uint64_t)hi << 32) | lo;
}
// value of lane l (per-lane source): ds_bpermute_b32 — the LDS crossbar, no LDS memory
CS_DEVICE uint32_t shfl_u32(uint32_t v, int l) { return (uint32_t)__shfl((int)v, l, 64); }
// Wave64 reductions on the VALU with DPP (data-parallel primitives): 6 dependent VALU ops instead of
// 6 ds_bpermute round trips through the LDS crossbar (each ~100 cycles of latency on the packer's
// critical path — r01a profile).  Pattern: butterfly inside each row of 16 lanes (quad_perm, row_ror),
// then row_bcast:15 into rows 1/3 and row_bcast:31 into rows 2/3; lane 63 holds the result.
template <int CTRL, int ROW_MASK>
CS_DEVICE uint32_t dpp_u32(uint32_t old, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
CS_DEVICE uint32_t wave_sum_u32(uint32_t v) {
    v += dpp_u32<0xB1, 0xF>(0u, v);    // quad_perm [1,0,3,2]
    v += dpp_u32<0x4E, 0xF>(0u, v);    // quad_perm [2,3,0,1]
    v += dpp_u32<0x124, 0xF>(0u, v);   // row_ror:4
    v += dpp_u32<0x128, 0xF>(0u, v);   // row_ror:8  -> every lane holds its row's sum
    v += dpp_u32<0x142, 0xA>(0u, v);   // row_bcast:15 -> rows 1 and 3
    v += dpp_u32<0x143, 0xC>(0u, v);   // row_bcast:31 -> rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
CS_DEVICE uint32_t wave_max_u32(uint32_t v) {
    uint32_t o;
    o = dpp_u32<0xB1, 0xF>(0u, v); v = o > v ? o : v;
    o = dpp_u32<0x4E, 0xF>(0u, v); v = o > v ? o : v;
    o = dpp_u32<0x124, 0xF>(0u, v); v = o > v ? o : v;
    o = dpp_u32<0x128, 0xF>(0u, v); v = o > v ? o : v;
    o = dpp_u32<0x142, 0xA>(0u, v); v = o > v ? o : v;
    o = dpp_u32<0x143, 0xC>(0u, v); v = o > v ? o : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// sum of s and max of m over the wave in ONE pass: the two DPP chains interleaved step by step, so that each fills the other's wait states
// (a DPP read of a VGPR needs two wait states behind the VALU write: alone, every step of a chain is followed by an s_nop that takes the
// wave's issue slot — 15 % of the packer's static scalar instructions were such s_nops)
CS_DEVICE void wave_sum_max_u32(uint32_t s, uint32_t m, uint32_t& sum, uint32_t& mx) {
    uint32_t o;
    s += dpp_u32<0xB1, 0xF>(0u, s);  o = dpp_u32<0xB1, 0xF>(0u, m);  m = o > m ? o : m;
    s += dpp_u32<0x4E, 0xF>(0u, s);  o = dpp_u32<0x4E, 0xF>(0u, m);  m = o > m ? o : m;
    s += dpp_u32<0x124, 0xF>(0u, s); o = dpp_u32<0x124, 0xF>(0u, m); m = o > m ? o : m;
    s += dpp_u32<0x128, 0xF>(0u, s); o = dpp_u32<0x128, 0xF>(0u, m); m = o > m ? o : m;
    s += dpp_u32<0x142, 0xA>(0u, s); o = dpp_u32<0x142, 0xA>(0u, m); m = o > m ? o : m;
    s += dpp_u32<0x143, 0xC>(0u, s); o = dpp_u32<0x143, 0xC>(0u, m); m = o > m ? o : m;
    sum = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
    mx = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
}
// exact 64-bit sum of 32-bit lane values: two 32-bit DPP reductions on the 16-bit halves
CS_DEVICE uint64_t wave_sum_u64(uint64_t v) {
    const uint32_t lo = wave_sum_u32((uint32_t)v & 0xffffu);
    const uint32_t mid = wave_sum_u32(((uint32_t)v >> 16) & 0xffffu);
    const uint32_t h0 = wave_sum_u32((uint32_t)(v >> 32) & 0xffffu);
    const uint32_t h1 = wave_sum_u32((uint32_t)(v >> 48));
    return (uint64_t)lo + ((uint64_t)mid << 16) + ((uint64_t)h0 << 32) + ((uint64_t)h1 << 48);
}
// value of a wave-UNIFORM lane index: v_readlane_b32 (scalar result, no LDS crossbar)
CS_DEVICE uint32_t bcast_u32(uint32_t v, int uniform_lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, uniform_lane); }
// a value every lane holds identically (e.g. an LDS word read at a wave-uniform address): move it to a scalar register
CS_DEVICE uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// N (8 or 16) consecutive dwords at a wave-UNIFORM address, written by an EARLIER kernel: ONE scalar load (s_load_dwordxN
// through the constant cache) straight into scalar registers — no VALU, no LDS, and it can be issued a loop iteration
// ahead.  The data must not be written by the kernel that reads it this way (constant address space semantics).
template <int N> struct Words { uint32_t w[N]; };
template <int N> CS_DEVICE Words<N> const_load(const uint32_t* p) {
    Words<N> r;
    if constexpr (N == 1) r.w[0] = *(const __attribute__((address_space(4))) uint32_t*)p;
    else {
        typedef uint32_t vec_t __attribute__((ext_vector_type(N)));
        const vec_t v = *(const __attribute__((address_space(4))) vec_t*)p;
#pragma unroll
        for (int i = 0; i < N; ++i) r.w[i] = v[i];
    }
    return r;
}
// 16 bytes at a 16-byte-aligned address as ONE access (global_load_dwordx4 / ds_read_b128 / ds_write_b128, whichever memory p points into)
CS_DEVICE Words<4> load4(const uint32_t* p) {
    typedef uint32_t vec4_t __attribute__((ext_vector_type(4)));
    const vec4_t v = *(const vec4_t*)p;
    Words<4> r; r.w[0] = v[0]; r.w[1] = v[1]; r.w[2] = v[2]; r.w[3] = v[3];
    return r;
}
CS_DEVICE void store4(uint32_t* p, const Words<4>& q) {
    typedef uint32_t vec4_t __attribute__((ext_vector_type(4)));
    vec4_t v; v[0] = q.w[0]; v[1] = q.w[1]; v[2] = q.w[2]; v[3] = q.w[3];
    *(vec4_t*)p = v;
}
// The same through a BUFFER resource and a 32-bit byte offset: s_buffer_load_dwordxN sdst, s[rsrc:rsrc+3], s_off.  A loop that
// walks records then carries ONE 32-bit scalar (offset += record size; compare with the end offset) instead of a 64-bit
// pointer plus a counter: s_add_u32 / s_addc_u32 / s_add_i32 / s_cmp / copy of the pointer pair became s_add_i32 / s_cmp — three
// scalar instructions less per PEG step in a kernel bound by scalar issue (the compiler widens a 32-bit offset added to a
// pointer back into 64-bit arithmetic, measured; the intrinsic keeps the offset operand as it is and still tracks lgkmcnt).
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x16_t __attribute__((ext_vector_type(16)));
extern "C" __device__ u32x8_t casim_llvm_s_buffer_load_v8(rsrc_t, uint32_t, uint32_t) __asm("llvm.amdgcn.s.buffer.load.v8i32");
extern "C" __device__ u32x16_t casim_llvm_s_buffer_load_v16(rsrc_t, uint32_t, uint32_t) __asm("llvm.amdgcn.s.buffer.load.v16i32");
struct RecBase { rsrc_t r; };
// raw buffer over [p, p + 4 GiB): stride 0, num_records 0xffffffff (bytes), gfx9 untyped 32-bit data format (word 3 = 0x00020000)
CS_DEVICE RecBase rec_base(const uint32_t* p) {
    const uint64_t a = (uint64_t)p;
    RecBase b;
    b.r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a); b.r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)); b.r[2] = -1; b.r[3] = 0x00020000;
    return b;
}
template <int N> CS_DEVICE Words<N> rec_load(const RecBase& b, uint32_t byte_off) {
    static_assert(N == 8 || N == 16, "records are 8 or 16 dwords");
    Words<N> r;
    if constexpr (N == 8) { const u32x8_t v = casim_llvm_s_buffer_load_v8(b.r, byte_off, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) r.w[i] = v[i]; }
    else { const u32x16_t v = casim_llvm_s_buffer_load_v16(b.r, byte_off, 0);
        // (the 64-byte record uses 14 of its 16 dwords; the optimiser trims a load to the elements somebody reads and the backend cannot
        // select the resulting <14 x i32>: the two spare words are "read" by an empty statement)
        asm volatile("" : : "s"(v[14]), "s"(v[15]));
#pragma unroll
        for (int i = 0; i < 16; ++i) r.w[i] = v[i]; }
    return r;
}
// the same for a caller that reads only SOME words of the record: every element is "read" by an empty statement, so the optimiser cannot
// trim the load to a vector width the backend has no s_buffer_load for (<6 x i32>, <14 x i32>: "Cannot select")
template <int N> CS_DEVICE Words<N> rec_load_all(const RecBase& b, uint32_t byte_off) {
    static_assert(N == 8 || N == 16, "records are 8 or 16 dwords");
    Words<N> r;
    if constexpr (N == 8) { const u32x8_t v = casim_llvm_s_buffer_load_v8(b.r, byte_off, 0);
        asm volatile("" : : "s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]));
#pragma unroll
        for (int i = 0; i < 8; ++i) r.w[i] = v[i]; }
    else { const u32x16_t v = casim_llvm_s_buffer_load_v16(b.r, byte_off, 0);
        asm volatile("" : : "s"(v[0]), "s"(v[1]), "s"(v[2]), "s"(v[3]), "s"(v[4]), "s"(v[5]), "s"(v[6]), "s"(v[7]));
        asm volatile("" : : "s"(v[8]), "s"(v[9]), "s"(v[10]), "s"(v[11]), "s"(v[12]), "s"(v[13]), "s"(v[14]), "s"(v[15]));
#pragma unroll
        for (int i = 0; i < 16; ++i) r.w[i] = v[i]; }
    return r;
}
// my lane's bit of a wave-uniform lane mask as a predicate: the mask register pair IS the condition (no VALU)
CS_DEVICE bool lane_pred(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
// pin a wave-uniform value that lives across loop iterations to a scalar register (the register allocator otherwise may
// give the loop-carried copy a VGPR, and every test of it becomes VALU work)
CS_DEVICE void keep_scalar(uint32_t& v) { v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v); asm volatile("" : "+s"(v)); }
// a copy of a lane value the optimiser cannot see through (no instruction): keeps it from merging two computations
CS_DEVICE int32_t opaque_i32(int32_t v) { asm volatile("" : "+v"(v)); return v; }
// a use of a lane value that emits nothing: keeps a load alive (and its destination register reserved) until here
CS_DEVICE void consume_u32(uint32_t v) { asm volatile("" : : "v"(v)); }
// test of one bit of a wave-UNIFORM word, meant to sit directly in an `if`: the word goes through an opaque scalar copy so
// that every test is its own s_bitcmp + s_cbranch_scc.  A flag tested in several places as one bool is kept by the
// compiler as a 64-bit lane mask (s_cselect_b64, then s_and_b64 with exec + s_cbranch_vcc at every use).
// an empty side effect: a branch that contains it is not folded into its neighbours' conditions (SimplifyCFG merges side-effect-free
// nested tests into one lane-mask expression: s_cselect_b64 + s_or_b64 + s_andn2_b64 + s_cbranch_vcc per merged test)
CS_DEVICE void keep_apart() { asm volatile(""); }
CS_DEVICE bool flag_set(uint32_t word, uint32_t bit) { asm volatile("" : "+s"(word)); return (word & bit) != 0; }
// a / b for wave-UNIFORM 32-bit values (b > 0) on the VECTOR unit: f64 reciprocal estimate + exact +-1 fix-up (the operands are
// below 2^32, the estimate is within one of the quotient), ~10 VALU.  The compiler's expansion of a uniform division is ~17
// scalar instructions around a v_rcp_f32 — and the scalar unit is the packer's bottleneck.
CS_DEVICE double estimate_rcp_f64(double x);
CS_DEVICE uint32_t uniform_div_u32(uint32_t a, uint32_t b) {
    uint32_t va = a, vb = b;
    asm volatile("" : "+v"(va), "+v"(vb));   // vector copies: keeps the arithmetic below off the scalar unit
    // (rcp + one Newton step: the raw v_rcp_f64 is only good to ~2^-24 relative, and the +-1 fix-up needs quotient x error < 1 — ADVICE r4)
    uint32_t q = (uint32_t)((double)va * estimate_rcp_f64((double)vb));
    const uint64_t wide = (uint64_t)q * vb;   // (b may exceed 2^31: the fix-up compares the 64-bit product)
    q = wide > va ? q - 1 : ((uint64_t)va - wide >= vb ? q + 1 : q);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
}
// A kernel's by-value struct parameter, read again from the KERNARG SEGMENT (scalar loads through the constant cache) instead of from the
// scalar registers the prologue loaded it into.  Why: DevTables + DevResults are ~70 pointers; a kernel that needs a third of them at its
// END (order_group's record emission) keeps them live across everything in front of it, and the register allocator parks what does not
// fit in the lanes of a VGPR — order_strided_kernel re-read 28 parked pointers with 2 500 static v_readlane_b32, a quarter of its
// straight-line vector instructions.  A fresh view (the pointer is laundered, so nothing is shared with earlier loads) gives every use a
// short-lived s_load.  `byte_offset` = the parameter's offset in the explicit kernel arguments (declaration order, natural alignment).
template <class T> CS_DEVICE const T& kernarg_view(const T&, size_t byte_offset) {
    const uint64_t a = (uint64_t)(const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    // (through readfirstlane: inside a loop the compiler takes for divergent a plain "+s" copy of the pointer is an illegal VGPR -> SGPR move)
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)(((uint64_t)hi << 32) | lo);
    return *(const T*)(const char*)(p + byte_offset);
}
// 1 / x as a quotient ESTIMATE (relative error ~2^-50: v_rcp_f64 + one Newton step, 3 instructions) — for the +-1 fix-up forms of
// capacity_of / the packer, whose proof asks for 2^-51.  An IEEE division is ~15 instructions (v_div_scale x 2, v_rcp, 5 FMAs,
// v_div_fmas, v_div_fixup) and nothing downstream needs the correctly rounded reciprocal.  The emulator divides.
CS_DEVICE double estimate_rcp_f64(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// the same for a divisor below 2^30 (the packer's: pods that fit an empty node, < 2^21 by eligibility): the remainder of the estimate lies
// in (-b, 2b), so it is exact in WRAPPING 32-bit arithmetic and the fix-up is sign bit + one compare — straight-line code; the general
// form above compares 64-bit products under two exec-mask regions (18 instructions, 4 of them scalar).
// The estimate is within one of the quotient for EVERY a < 2^32 because the reciprocal carries a Newton step (relative error ~2^-50:
// quotient x error < 2^-18); the raw v_rcp_f64 (~2^-24) would be off by more than one from a / b >= 2^24 on — cn = 1 next to a PEG of 2^24
// pods, ADVICE r4 — and the emulator, which divides, could not have told.
CS_DEVICE uint32_t uniform_div_u32_small(uint32_t a, uint32_t b) {
    uint32_t va = a, vb = b;
    asm volatile("" : "+v"(va), "+v"(vb));
    uint32_t q = (uint32_t)((double)va * estimate_rcp_f64((double)vb));
    const int32_t rem = (int32_t)(va - q * vb);
    q = q - ((uint32_t)rem >> 31) + (rem >= (int32_t)vb ? 1u : 0u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
}
// min of two wave-uniform values as ONE s_min_u32 (min(x, 1) written in C++ becomes "x != 0" as a lane mask, a v_cndmask and a
// v_readfirstlane)
CS_DEVICE uint32_t scalar_min_u32(uint32_t a, uint32_t b) {
    a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a); b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);   // (no-ops for scalar operands)
    uint32_t r;
    asm volatile("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}
// v[uniform_lane] = uniform_value: one v_writelane_b32 instead of lane-compare + select + move
CS_DEVICE void write_lane_u32(uint32_t& v, uint32_t uniform_value, int uniform_lane) {
    // (one scalar register per VALU instruction on gfx9: the lane select travels in M0; readfirstlane is a no-op for values
    // the compiler already holds in scalar registers and makes the operands legal when it does not)
    uniform_value = (uint32_t)__builtin_amdgcn_readfirstlane((int)uniform_value);
    uniform_lane = __builtin_amdgcn_readfirstlane(uniform_lane);
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(uniform_value), "s"(uniform_lane) : "m0");
}
// (wave-uniform a & lane b) | lane c as ONE v_and_or_b32 — the accumulation step of feas_stream_kernel's static-Filter word.  Left to itself the
// compiler balances an OR of k AND terms into k v_and + (k - 1) / 2 v_or3 (a tree: more instruction-level parallelism, which a wave that
// issues one instruction per ~4 cycles cannot use) instead of the chain of k fused instructions
CS_DEVICE uint32_t and_or_u32(uint32_t uniform_a, uint32_t b, uint32_t c) {
    uint32_t r;
    uniform_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)uniform_a);   // (a no-op for a value the compiler holds in a scalar register)
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "s"(uniform_a), "v"(b), "v"(c));
    return r;
}
// both halves of a wave-uniform 64-bit word into one lane of two VGPRs: ONE scalar instruction loads M0, both v_writelane share it.
// (A first version derived the lane from a byte offset with s_lshr_b32 m0, x, 6 and did not declare that s_lshr writes SCC: the compiler
// kept a loop's s_cmp result in SCC ACROSS the statement and the group loop of feas_stream_kernel left after its first trip — groups 2.. of
// every simulation came back empty on the MI355X, and the emulator could not see it.  tests/tools/writelane_probe.hip is the micro-probe
// that found it; s_mov_b32 leaves SCC alone.)
CS_DEVICE void write_lane2_u32(uint32_t& lo, uint32_t& hi, uint64_t uniform_value, uint32_t uniform_lane) {
    const uint32_t vl = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)uniform_value), vh = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uniform_value >> 32));
    uniform_lane = (uint32_t)__builtin_amdgcn_readfirstlane((int)uniform_lane);
    asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(lo), "+v"(hi) : "s"(vl), "s"(vh), "s"(uniform_lane) : "m0");
}
// (lane a & lane b) | lane c as ONE v_and_or_b32: see and_or_u32
CS_DEVICE uint32_t and_or_vvv(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a wave-uniform int the compiler cannot prove uniform (a global load indexed by the block id): to a scalar register
CS_DEVICE int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
CS_DEVICE int popc64(uint64_t v) { return __popcll(v); }
CS_DEVICE int ffs64(uint64_t v) { return v ? (int)__builtin_ctzll(v) : -1; }
CS_DEVICE int fls64(uint64_t v) { return v ? 63 - (int)__builtin_clzll(v) : -1; }
CS_DEVICE uint64_t double_bits(double d) { return (uint64_t)__double_as_longlong(d); }
CS_DEVICE double bits_double(uint64_t u) { return __longlong_as_double((long long)u); }
}  // namespace cs
#endif

namespace cs {
CS_DEVICE int lane() { return tid() & 63; }
// number of set bits of `mask` strictly below my lane (v_mbcnt_lo / v_mbcnt_hi on device)
CS_DEVICE int mbcnt(uint64_t mask) {
#if defined(CASIM_HOST_EMU)
    int l = lane();
    uint64_t below = l == 0 ? 0ull : (~0ull >> (64 - l));
    return popc64(mask & below);
#else
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#endif
}
// exact sum of 64 uint32 lane values as a uint64: two 32-bit reductions on the 16-bit halves
CS_DEVICE uint64_t wave_sum_u32_wide(uint32_t v) {
    const uint32_t lo = wave_sum_u32(v & 0xffffu);
    const uint32_t hi = wave_sum_u32(v >> 16);
    return (uint64_t)lo + ((uint64_t)hi << 16);
}
CS_DEVICE uint64_t bcast_u64(uint64_t v, int uniform_lane) {
    return ((uint64_t)bcast_u32((uint32_t)(v >> 32), uniform_lane) << 32) | bcast_u32((uint32_t)v, uniform_lane);
}
// acc + k * q as two's-complement arithmetic: the request totals of an Estimate wrap past INT64_MAX (tables nobody runs, but the
// oracle's totals wrap the same way and signed overflow itself is undefined: tests/tools/sanitize_cpu.sh found these sums)
CS_DEVICE int64_t wrap_madd_i64(int64_t acc, int64_t k, int64_t q) { return (int64_t)((uint64_t)acc + (uint64_t)k * (uint64_t)q); }
// bits [0, n) set; n may be <= 0 or >= 64
CS_DEVICE uint64_t low_mask(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1)); }
}  // namespace cs
