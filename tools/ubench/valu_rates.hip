// valu_rates — issue-rate probe for the integer instructions a dense pods x nodes predicate kernel is made of (gfx950).
// Each wave runs `iters` rounds of an unrolled block; rate = wave-instructions / s / SIMD.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

// (a) v_cmp_le_i32 (VOPC, writes VCC) + v_addc_co_u32 acc = acc + acc + VCC  -> 2 VALU per bit
__global__ void k_cmp_addc(int* out, int iters, int s0) {
    int a = threadIdx.x, acc = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP32("v_cmp_le_i32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n") : "+v"(acc) : "s"(s0 + i), "v"(a) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// (b) two compares ANDed through SALU: v_cmp -> vcc, v_cmp -> s[pair], s_and_b64, v_addc: 3 VALU + 1 SALU per bit
__global__ void k_cmp2_addc(int* out, int iters, int s0, int s1) {
    int a = threadIdx.x, b = threadIdx.x * 3, acc = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP32("v_cmp_le_i32 vcc, %1, %3\n v_cmp_le_i32 s[20:21], %2, %4\n s_and_b64 vcc, vcc, s[20:21]\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n")
                     : "+v"(acc) : "s"(s0 + i), "s"(s1 - i), "v"(a), "v"(b) : "vcc", "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// (c) packed 16-bit: v_pk_sub_u16 clamp (saturating req - free per half) ; v_cmp_eq_u32 0 ; v_addc   -> 3 VALU per bit, 2 lanes of resources
__global__ void k_pk_addc(int* out, int iters, int s0) {
    int a = threadIdx.x * 65537, acc = 0, t;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP32("v_pk_sub_u16 %1, %3, %2 clamp\n v_cmp_eq_u32 vcc, 0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n")
                     : "+v"(acc), "=&v"(t) : "s"(s0 + i), "v"(a) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// (d) plain v_add_u32 chain (dependent) and (e) independent v_and_b32 pairs as the reference rate
__global__ void k_add(int* out, int iters, int s0) {
    int a = threadIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
// (f) v_cmp only, alternating destination pairs (pure VOPC issue rate)
__global__ void k_cmp_only(int* out, int iters, int s0) {
    int a = threadIdx.x; unsigned long long m = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP32("v_cmp_le_i32 vcc, %1, %2\n") : "+s"(m) : "s"(s0 + i), "v"(a) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)m;
}
// (g) VOP3 compare into an SGPR pair + v_writelane x2 (word of pod p parked in lane p): 3 VALU per 64 checks (one resource)
__global__ void k_cmp_writelane(int* out, int iters, int s0) {
    int a = threadIdx.x, lo = 0, hi = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP32("v_cmp_le_i32 s[20:21], %2, %3\n v_writelane_b32 %0, s20, 5\n v_writelane_b32 %1, s21, 5\n")
                     : "+v"(lo), "+v"(hi) : "s"(s0 + i), "v"(a) : "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = lo ^ hi;
}

template <class K, class... A>
static void run(const char* name, int valu_per_block, int blocks_per_iter, K k, int* d, int iters, A... args) {
    const int grid = 256 * 8, block = 256;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 10, args...);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, iters, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * block / 64, insts = waves * iters * (double)blocks_per_iter * valu_per_block;
    const double per_simd = insts / (ms * 1e-3) / 1024.0;
    printf("%-18s %8.3f ms  %.3e VALU wave-inst/s  = %.3f G/s/SIMD  -> %.2f cycles per VALU inst at 2.4 GHz\n", name, ms, insts / (ms * 1e-3), per_simd / 1e9, 2.4e9 / per_simd);
}

int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 4000;
    run("add (4 indep)", 4, 8, k_add, d, iters, 3);
    run("cmp only", 1, 32, k_cmp_only, d, iters, 7);
    run("cmp+addc", 2, 32, k_cmp_addc, d, iters, 7);
    run("cmp,cmp,sand,addc", 3, 32, k_cmp2_addc, d, iters, 7, 100000);
    run("pk_sub,cmp,addc", 3, 32, k_pk_addc, d, iters, 7);
    run("cmp,writelane x2", 3, 32, k_cmp_writelane, d, iters, 7);
    hipFree(d);
    return 0;
}
