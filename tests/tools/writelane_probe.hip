// micro-probe (round 5 bring-up, TEST INFRASTRUCTURE): which forms of v_writelane_b32 with a run-time lane select work on gfx950?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tests/tools/writelane_probe tests/tools/writelane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../kubernetes_autoscaler_amd/csrc/casim_device.h"

__global__ void k_m0_mov(unsigned* out, unsigned base, int n) {   // s_mov_b32 m0, j ; v_writelane v, s, m0   (cs::write_lane_u32 of round 4)
    unsigned v = 0;
    for (int j = 0; j < n; ++j) cs::write_lane_u32(v, base + (unsigned)j, j);
    out[threadIdx.x] = v;
}
__global__ void k_m0_lshr(unsigned* out, unsigned base, int n) {  // cs::write_lane2_u32 (round 5): one M0 load, two v_writelane
    unsigned lo = 0, hi = 0;
    for (int j = 0; j < n; ++j) cs::write_lane2_u32(lo, hi, ((unsigned long long)(base + 5000u + (unsigned)j) << 32) | (base + (unsigned)j), (unsigned)j);
    out[threadIdx.x] = lo; out[64 + threadIdx.x] = hi;
}
__global__ void k_m0_two_values(unsigned* out, unsigned base, int n) {  // two DIFFERENT uniform values written with one M0 — but each from a VALU-independent SGPR
    unsigned lo = 0, hi = 0;
    for (int j = 0; j < n; ++j) {
        unsigned a = __builtin_amdgcn_readfirstlane(base + (unsigned)j), b = __builtin_amdgcn_readfirstlane(base + 5000u + (unsigned)j), l = __builtin_amdgcn_readfirstlane(j);
        asm volatile("s_mov_b32 m0, %4\n\ts_nop 4\n\tv_writelane_b32 %0, %2, m0\n\ts_nop 4\n\tv_writelane_b32 %1, %3, m0\n\ts_nop 4" : "+v"(lo), "+v"(hi) : "s"(a), "s"(b), "s"(l) : "m0");
    }
    out[threadIdx.x] = lo; out[64 + threadIdx.x] = hi;
}
template <int J> __device__ __forceinline__ void put(unsigned& v, unsigned val) { asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(val), "n"(J)); }
template <int J> __device__ __forceinline__ void put_all(unsigned& v, unsigned base, int n) { if constexpr (J < 64) { if (J < n) { put<J>(v, __builtin_amdgcn_readfirstlane(base + (unsigned)J)); put_all<J + 1>(v, base, n); } } }
__global__ void k_const(unsigned* out, unsigned base, int n) {    // inline-constant lane select, fully unrolled
    unsigned v = 0;
    put_all<0>(v, base, n);
    out[threadIdx.x] = v;
}

static int check(const char* what, const unsigned* h, unsigned base, int n) {
    int bad = 0, first = -1;
    for (int j = 0; j < 64; ++j) { const unsigned want = j < n ? base + (unsigned)j : 0u; if (h[j] != want) { if (first < 0) first = j; ++bad; } }
    printf("%-44s n = %2d: %2d lanes wrong", what, n, bad);
    if (first >= 0) printf(" (first: lane %d holds %u, want %u)", first, h[first], first < n ? base + (unsigned)first : 0u);
    printf("\n");
    return bad;
}
int main() {
    unsigned* d = nullptr; unsigned h[128];
    if (hipMalloc(&d, sizeof h) != hipSuccess) { printf("no device\n"); return 1; }
    for (int n : {1, 2, 3, 5, 20, 64}) {
        hipLaunchKernelGGL(k_m0_mov, dim3(1), dim3(64), 0, 0, d, 1000u, n); (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); check("s_mov m0 + v_writelane (round-4 helper)", h, 1000u, n);
        hipLaunchKernelGGL(k_m0_lshr, dim3(1), dim3(64), 0, 0, d, 2000u, n); (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); check("write_lane2_u32: low words", h, 2000u, n); check("                            high words", h + 64, 7000u, n);
        hipLaunchKernelGGL(k_m0_two_values, dim3(1), dim3(64), 0, 0, d, 3000u, n); (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); check("s_mov m0 + nops + 2 x v_writelane: low", h, 3000u, n); check("                             high", h + 64, 8000u, n);
        hipLaunchKernelGGL(k_const, dim3(1), dim3(64), 0, 0, d, 4000u, n); (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); check("inline-constant lane select", h, 4000u, n);
    }
    return 0;
}
