// op_rates — cycles per wave-instruction of the operations the packer's PEG step is made of (gfx950): quarter-rate integer
// multiplies, f64 / f32 conversions, cross-lane reads, DPP adds, LDS broadcast reads, scalar loads.
// 8 waves per SIMD, 4 independent chains per wave.  Build: hipcc --offload-arch=gfx950 -O3 -o op_rates op_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x

#define KERNEL4(name, body)                                                                                      \
    __global__ void name(int* out, int iters, int s0) {                                                          \
        int a = threadIdx.x + 1, b = threadIdx.x + 2, c = threadIdx.x + 3, d = threadIdx.x + 4;                  \
        for (int i = 0; i < iters; ++i) { asm volatile(REP8(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0 | 1)); } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                              \
    }
KERNEL4(k_add, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
KERNEL4(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
KERNEL4(k_mul_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n")
KERNEL4(k_mad_u24, "v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n")
KERNEL4(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n")
KERNEL4(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n")
KERNEL4(k_mul_f32, "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")
KERNEL4(k_rcp_f32, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
KERNEL4(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
// selects: condition in VCC (set once), condition in an SGPR pair (VOP3), and the compare + select pair the compiler emits
__global__ void k_cndmask_vcc(int* out, int iters, int s0) {
    int a = threadIdx.x + 1, b = threadIdx.x + 2, c = threadIdx.x + 3, d = threadIdx.x + 4, e = threadIdx.x * 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_mov_b64 vcc, 0x5555\n" REP8("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void k_cndmask_sgpr(int* out, int iters, int s0) {
    int a = threadIdx.x + 1, b = threadIdx.x + 2, c = threadIdx.x + 3, d = threadIdx.x + 4, e = threadIdx.x * 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile("s_mov_b64 s[20:21], 0x5555\n" REP8("v_cndmask_b32 %0, %0, %4, s[20:21]\n v_cndmask_b32 %1, %1, %4, s[20:21]\n v_cndmask_b32 %2, %2, %4, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void k_cmp_cndmask(int* out, int iters, int s0) {
    int a = threadIdx.x + 1, b = threadIdx.x + 2, e = threadIdx.x * 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_cmp_lt_i32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_lt_i32 s[20:21], %1, %2\n v_cndmask_b32 %1, %1, %2, s[20:21]\n")
                     : "+v"(a), "+v"(b) : "v"(e) : "vcc", "s20", "s21");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b;
}
KERNEL4(k_dpp_add, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n")
KERNEL4(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_hi_u32_b32 %3, %4, %3\n")
KERNEL4(k_min, "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n")

// f64: two registers per value
__global__ void k_f64(int* out, int iters, int s0) {
    double a = threadIdx.x + 1.0, b = threadIdx.x + 2.0; unsigned u = threadIdx.x, v = threadIdx.x + 9;
    const double m = 1.0000001;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8("v_cvt_f64_u32 %0, %2\n v_mul_f64 %0, %0, %4\n v_cvt_u32_f64 %2, %0\n v_cvt_f64_u32 %1, %3\n v_mul_f64 %1, %1, %4\n v_cvt_u32_f64 %3, %1\n")
                     : "+v"(a), "+v"(b), "+v"(u), "+v"(v) : "v"(m));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(u + v);
}
__global__ void k_cvt_f64_u32(int* out, int iters, int s0) {
    double a = 0, b = 0; unsigned u = threadIdx.x, v = threadIdx.x + 9;
    for (int i = 0; i < iters; ++i) { asm volatile(REP8("v_cvt_f64_u32 %0, %2\n v_cvt_f64_u32 %1, %3\n v_cvt_f64_u32 %0, %3\n v_cvt_f64_u32 %1, %2\n") : "+v"(a), "+v"(b) : "v"(u), "v"(v)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a + b);
}
__global__ void k_mul_f64(int* out, int iters, int s0) {
    double a = threadIdx.x + 1.0, b = threadIdx.x + 2.0, c = 3.0 + threadIdx.x, d = 4.0 + threadIdx.x; const double m = 1.0000001;
    for (int i = 0; i < iters; ++i) { asm volatile(REP8("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a + b + c + d);
}
__global__ void k_cvt_u32_f64(int* out, int iters, int s0) {
    double a = threadIdx.x + 1.0, b = threadIdx.x + 2.0; unsigned u = 0, v = 0;
    for (int i = 0; i < iters; ++i) { asm volatile(REP8("v_cvt_u32_f64 %0, %2\n v_cvt_u32_f64 %1, %3\n v_cvt_u32_f64 %0, %3\n v_cvt_u32_f64 %1, %2\n") : "+v"(u), "+v"(v) : "v"(a), "v"(b)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(u + v);
}
__global__ void k_mad_u64_u32(int* out, int iters, int s0) {
    unsigned long long a = threadIdx.x, b = threadIdx.x + 1; unsigned u = threadIdx.x | 1, v = 77;
    for (int i = 0; i < iters; ++i) { asm volatile(REP8("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %3, %2, %0\n v_mad_u64_u32 %1, vcc, %3, %2, %1\n") : "+v"(a), "+v"(b) : "v"(u), "v"(v) : "vcc"); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(a + b);
}
// cross-lane reads into SGPRs
__global__ void k_readlane(int* out, int iters, int s0) {
    int a = threadIdx.x, b = threadIdx.x * 3; int acc = 0;
    for (int i = 0; i < iters; ++i) {
        int t0, t1, t2, t3;
        asm volatile(REP8("v_readlane_b32 %0, %4, %6\n v_readlane_b32 %1, %5, %6\n v_readfirstlane_b32 %2, %4\n v_readfirstlane_b32 %3, %5\n")
                     : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3) : "v"(a), "v"(b), "s"((s0 + i) & 63));
        acc += t0 + t1 + t2 + t3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// LDS broadcast read (uniform address) + readfirstlane: the packer's record fetch today, 8 dwords per "record"
__global__ void k_lds_record(int* out, int iters, int s0) {
    __shared__ unsigned buf[4][640];
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x & 63; i < 640; i += 64) buf[w][i] = i * 2654435761u;
    __syncthreads();
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = (s0 + i * 8 + u) & 63;
            unsigned r[8];
#pragma unroll
            for (int f = 0; f < 8; ++f) r[f] = __builtin_amdgcn_readfirstlane(buf[w][f * 64 + j]);
            acc += (r[0] ^ r[1]) + (r[2] ^ r[3]) + (r[4] ^ r[5]) + (r[6] ^ r[7]);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// the same record as ONE scalar load of 32 bytes at a wave-uniform address (each wave streams its own records)
__global__ void k_smem_record(int* out, int iters, int s0, const uint4* __restrict__ recs, int per_wave) {
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint4* p = recs + (size_t)wave * per_wave * 2;
    unsigned acc = 0;
    for (int i = 0; i < iters * 8; ++i) {
        const int j = i & 255;   // (per_wave == 256)
        const uint4 x = p[2 * j], y = p[2 * j + 1];   // s_load_dwordx8 (uniform address)
        acc += (x.x ^ x.y) + (x.z ^ x.w) + (y.x ^ y.y) + (y.z ^ y.w);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <class K, class... A>
static void run(const char* name, double inst_per_iter, K k, int* d, int iters, A... args) {
    const int grid = 256 * 8, block = 256;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, 10, args...);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, d, iters, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * block / 64, insts = waves * iters * inst_per_iter;
    const double per_simd = insts / (ms * 1e-3) / 1024.0;
    printf("%-22s %8.3f ms  %.3f G inst/s/SIMD -> %6.2f cycles per wave-inst at 2.4 GHz\n", name, ms, per_simd / 1e9, 2.4e9 / per_simd);
}

int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 2000;
    run("v_add_u32", 32, k_add, d, iters, 3);
    run("v_min_u32", 32, k_min, d, iters, 3);
    run("v_cndmask_b32 (chain)", 32, k_cndmask, d, iters, 3);
    run("v_cndmask vcc", 32, k_cndmask_vcc, d, iters, 3);
    run("v_cndmask sgpr pair", 32, k_cndmask_sgpr, d, iters, 3);
    run("v_cmp + v_cndmask", 32, k_cmp_cndmask, d, iters, 3);
    run("v_mul_lo_u32", 32, k_mul_lo, d, iters, 3);
    run("v_mul_u32_u24", 32, k_mul_u24, d, iters, 3);
    run("v_mad_u32_u24", 32, k_mad_u24, d, iters, 3);
    run("v_mad_u64_u32", 32, k_mad_u64_u32, d, iters, 3);
    run("v_cvt_f32_u32", 32, k_cvt_f32_u32, d, iters, 3);
    run("v_cvt_u32_f32", 32, k_cvt_u32_f32, d, iters, 3);
    run("v_mul_f32", 32, k_mul_f32, d, iters, 3);
    run("v_rcp_f32", 32, k_rcp_f32, d, iters, 3);
    run("v_cvt_f64_u32", 32, k_cvt_f64_u32, d, iters, 3);
    run("v_mul_f64", 32, k_mul_f64, d, iters, 3);
    run("v_cvt_u32_f64", 32, k_cvt_u32_f64, d, iters, 3);
    run("cvt,mul,cvt f64 chain", 48, k_f64, d, iters, 3);
    run("dpp add", 32, k_dpp_add, d, iters, 3);
    run("v_mbcnt", 32, k_mbcnt, d, iters, 3);
    run("readlane/firstlane", 32, k_readlane, d, iters, 3);
    run("LDS record (8 dw)", 8, k_lds_record, d, iters, 3);
    const int per_wave = 256; uint4* recs; hipMalloc(&recs, (size_t)8192 * per_wave * 32); hipMemset(recs, 1, (size_t)8192 * per_wave * 32);
    run("SMEM record (8 dw)", 8, k_smem_record, d, iters, 3, (const uint4*)recs, per_wave);
    hipFree(d);
    return 0;
}
