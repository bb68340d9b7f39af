// casim_pack_tu.hip — the register-resident packer kernels (pack_fast_kernel<R, NPT, WX, BUILD>) in their OWN translation unit,
// compiled TWICE into the product library (csrc/Makefile):
//   BUILD 0  with `-mllvm -structurizecfg-skip-uniform-regions` (wave-uniform branches keep their CFG instead of going through the
//            structurizer's flow blocks: 58 -> 49 VGPRs, 185 -> 95 v_mov, +19 % on the C1 batch);
//   BUILD 1  (-DCASIM_PACK_BUILD=1) the same source without the option.
// Why both: the option is experimental in LLVM, and round 2 caught it MISCOMPILING another kernel — feas_kernel with a per-group
// PEG range lost the node-local exclusion test for one lane (GPU said "fits", emulator / oracle / the same source without the option
// said NodePorts conflict; profiles/r02c_structurizer_flag_miscompile.txt).  Everything else is compiled without it; the packer
// keeps it, and the library decides at run time which build serves a device: the first problem of a process runs a built-in corpus
// through BOTH builds (casim_engine.hip, pack_self_check) and any difference retires build 0 for the process
// (casim_pack_build_info; casim_options.pack_build / CASIM_PACK_BUILD force either).  tests/test_gpu_ab_structurizer.py
// runs ~900 batches through both builds and requires identical results.
#include <hip/hip_runtime.h>

#include "../../include/casim.h"
#include "casim_pack.h"

#ifndef CASIM_PACK_BUILD
#define CASIM_PACK_BUILD 0
#endif

namespace casim {

// returns the HIP error of the launch (hipSuccess = 0)
#if CASIM_PACK_BUILD == 0
int hip_launch_pack_fast(int lanes, int slots_per_lane, int excl_words, int n_groups, void* stream, DevTables t, DevResults res, FastScratch fs) {
#else
int hip_launch_pack_fast_plain(int lanes, int slots_per_lane, int excl_words, int n_groups, void* stream, DevTables t, DevResults res, FastScratch fs) {
#endif
    if (n_groups <= 0) return 0;
#define CASIM_TU_LAUNCH(R, N, X) pack_fast_kernel<R, N, X, CASIM_PACK_BUILD><<<dim3((unsigned)n_groups, 1, 1), dim3(64, 1, 1), (size_t)0, (hipStream_t)stream>>>(t, res, fs)
#define CASIM_TU_PICK(R, X) do { if (slots_per_lane == 1) CASIM_TU_LAUNCH(R, 1, X); else if (slots_per_lane == 4) CASIM_TU_LAUNCH(R, 4, X); else CASIM_TU_LAUNCH(R, 16, X); } while (0)
    if (lanes == 8) {   // two int64 lanes (pack_fast64_kernel)
#define CASIM_TU_LAUNCH64(N, X) pack_fast64_kernel<N, X, CASIM_PACK_BUILD><<<dim3((unsigned)n_groups, 1, 1), dim3(64, 1, 1), (size_t)0, (hipStream_t)stream>>>(t, res, fs)
#define CASIM_TU_PICK64(X) do { if (slots_per_lane == 1) CASIM_TU_LAUNCH64(1, X); else if (slots_per_lane == 4) CASIM_TU_LAUNCH64(4, X); else CASIM_TU_LAUNCH64(16, X); } while (0)
        if (excl_words == 2) CASIM_TU_PICK64(2); else CASIM_TU_PICK64(0);
#undef CASIM_TU_PICK64
#undef CASIM_TU_LAUNCH64
    }
    else if (lanes == 2) { if (excl_words == 2) CASIM_TU_PICK(2, 2); else CASIM_TU_PICK(2, 0); }
    else            { if (excl_words == 2) CASIM_TU_PICK(4, 2); else CASIM_TU_PICK(4, 0); }
#undef CASIM_TU_PICK
#undef CASIM_TU_LAUNCH
    return (int)hipGetLastError();
}

}  // namespace casim
