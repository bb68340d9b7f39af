// Standalone probe of feas_stream_kernel on the device (round 5 bring-up; TEST INFRASTRUCTURE): tiny tables built here, the group records
// and the bit rows compared with a host loop, for the product kernel and for variants of its two inline-asm pieces.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o tests/tools/feas_stream_probe tests/tools/feas_stream_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../kubernetes_autoscaler_amd/csrc/casim_kernels.h"

using namespace casim;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* d = nullptr; if (hipMalloc(&d, v.size() * sizeof(T) + 64) != hipSuccess) return nullptr; (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

int run_variant(int kVar, const DevTables& t, uint64_t* d_bits, int Wg, const int32_t* d_req32, const uint32_t* d_rec, int gx, int n_sims, size_t words, const std::vector<uint64_t>& want, const char* what) {
    (void)hipMemset(d_bits, 0xff, words * 8);
    const int blocks = ((n_sims + 7) / 8) * 8 * gx;
    hipLaunchKernelGGL((feas_stream_kernel<true, true, true>), dim3(blocks), dim3(256), (size_t)(((7 + 3) & ~3) + 4) * 64, 0, t, d_bits, Wg, d_req32, d_rec, gx, n_sims, -1, 0);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> got(words);
    CK(hipMemcpy(got.data(), d_bits, words * 8, hipMemcpyDeviceToHost));
    int bad = 0, first = -1;
    for (size_t i = 0; i < words; ++i) if (got[i] != want[i]) { if (first < 0) first = (int)i; ++bad; }
    printf("variant %d (%s): %d of %zu words differ", kVar, what, bad, words);
    if (first >= 0) printf("; first at row %d word %d: got %016llx want %016llx", first / Wg, first % Wg, (unsigned long long)got[first], (unsigned long long)want[first]);
    printf("\n");
    return bad;
}

int main() {
    const int n_sims = 5, groups_per_sim = 7, pegs_per_sim = 100, R = 2;
    const int NG = n_sims * groups_per_sim, G = n_sims * pegs_per_sim, Wg = (pegs_per_sim + 63) / 64;
    std::vector<int32_t> req32((size_t)G * R), fresh32((size_t)NG * R), allowed(NG, 110), init_pods(NG, 0), peg_lo(NG), peg_hi(NG), sim_off(n_sims + 1);
    std::vector<uint32_t> pflags(G, 0), gflags(NG, 0);
    std::vector<uint64_t> tol(G), sel(G), taint(NG), label(NG);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    for (int g = 0; g < G; ++g) { req32[g * R] = rnd() % 5 ? 1 + rnd() % 40 : 0; req32[g * R + 1] = 1 + rnd() % 60; tol[g] = rnd() % 3 ? ~0ull : rnd(); sel[g] = rnd() % 4 ? 0 : 1ull << (rnd() % 40); pflags[g] = rnd() % 8 == 0; }
    for (int i = 0; i < NG; ++i) { fresh32[i * R] = 10 + rnd() % 60; fresh32[i * R + 1] = 20 + rnd() % 80; taint[i] = rnd() % 2 ? 0 : 1ull << (rnd() % 40); label[i] = ((uint64_t)rnd() << 32) | rnd();
                                   gflags[i] = rnd() % 6 == 0 ? CASIM_NG_UNSCHEDULABLE : 0; if (rnd() % 7 == 0) init_pods[i] = 110; peg_lo[i] = (i / groups_per_sim) * pegs_per_sim; peg_hi[i] = peg_lo[i] + pegs_per_sim; }
    for (int k = 0; k <= n_sims; ++k) sim_off[k] = k * groups_per_sim;
    DevTables t; memset(&t, 0, sizeof t);
    t.G = G; t.R = R; t.Wt = 1; t.Wl = 1; t.NG = NG; t.n_sims = n_sims;
    t.pflags = up(pflags); t.tol = up(tol); t.sel = up(sel); t.allowed = up(allowed); t.init_pods = up(init_pods); t.gflags = up(gflags); t.taint = up(taint); t.label = up(label);
    t.peg_lo = up(peg_lo); t.peg_hi = up(peg_hi); t.sim_off = up(sim_off);
    const int32_t* d_req32 = up(req32); const int32_t* d_fresh32 = up(fresh32);
    uint32_t* d_rec = nullptr; CK(hipMalloc(&d_rec, (size_t)(NG + 2) * 64));
    uint64_t* d_bits = nullptr; const size_t words = (size_t)NG * Wg; CK(hipMalloc(&d_bits, words * 8));
    hipLaunchKernelGGL(feas_group_records_kernel, dim3((NG + 255) / 256), dim3(256), 0, 0, t, d_fresh32, d_rec, -1, 0);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> rec((size_t)(NG + 2) * 16);
    CK(hipMemcpy(rec.data(), d_rec, rec.size() * 4, hipMemcpyDeviceToHost));
    int rec_bad = 0;
    for (int i = 0; i < NG; ++i) {
        const uint32_t* r = rec.data() + (size_t)i * 16;
        const uint64_t nl = ~label[i];
        const int32_t f0 = allowed[i] - init_pods[i] <= 0 ? (int32_t)0x80000000 : fresh32[i * R];
        const bool ok = r[0] == (uint32_t)taint[i] && r[4] == (uint32_t)(taint[i] >> 32) && r[1] == (uint32_t)nl && r[5] == (uint32_t)(nl >> 32) && (int32_t)r[2] == f0 &&
                        (int32_t)r[3] == fresh32[i * R + 1] && r[6] == ((gflags[i] & CASIM_NG_UNSCHEDULABLE) ? 1u : 0u);
        if (!ok && rec_bad++ < 4) printf("record %d wrong: %08x %08x %08x %08x %08x %08x %08x\n", i, r[0], r[1], r[2], r[3], r[4], r[5], r[6]);
    }
    printf("group records: %d of %d wrong\n", rec_bad, NG);
    std::vector<uint64_t> want(words, 0);
    for (int i = 0; i < NG; ++i) for (int k = 0; k < pegs_per_sim; ++k) {
        const int g = peg_lo[i] + k;
        bool ok = (taint[i] & ~tol[g]) == 0 && (sel[g] & ~label[i]) == 0 && !((gflags[i] & CASIM_NG_UNSCHEDULABLE) && !(pflags[g] & CASIM_PEG_TOLERATES_UNSCHEDULABLE)) && allowed[i] - init_pods[i] > 0;
        for (int r = 0; r < R; ++r) ok = ok && (req32[g * R + r] <= 0 || req32[g * R + r] <= fresh32[i * R + r]);
        if (ok) want[(size_t)i * Wg + (k >> 6)] |= 1ull << (k & 63);
    }
    const int gx = (pegs_per_sim + 255) / 256;
    int bad = 0;
    bad += run_variant(0, t, d_bits, Wg, d_req32, d_rec, gx, n_sims, words, want, "feas_stream_kernel<lean, hi, term>");
    printf(bad ? "PROBE: product variant WRONG\n" : "PROBE: product variant ok\n");
    return 0;
}
