#!/bin/bash
# Sanitizer tier of the CPU suite (TEST INFRASTRUCTURE; the reference's counterpart is `go test -race`, CA/Makefile:82).
# GPU AddressSanitizer is not available on the pool, so the device code is checked where it also runs on the CPU: the wave emulator
# compiles the product's kernel headers for the host, and an out-of-bounds read of a table there is the same out-of-bounds read on the MI355X.
#
#   pass 1  emulator (the kernels + the host pipeline) and oracle built with gcc -fsanitize=address,undefined; the whole `-m "not gpu"` tier
#           under them (CASIM_EMU_LIB / CASIM_ORACLE_LIB, gcc's libasan preloaded into python)
#   pass 2  libcasim.so itself with the HOST side instrumented (hipcc -fsanitize=address,undefined: clang ignores the option for gfx950 and
#           says so; the encoder, the C ABI's argument checks, the prefetch cache and the host pool are host code) — CASIM_LIB_PATH, clang's
#           runtime preloaded; the tests that drive the encoder and the ABI
#   pass 3  gcc -fsanitize=thread: the host pool's self-test (tasks in index order, nested loops, several callers) and tests/test_streams_emu.py
#           with the parts of a streamed call as tasks of the pool (CASIM_EMU_THREADS=1: upload turns, list bases under a mutex, fetch workers)
#   pass 4  libcasim.so with its host side under hipcc -fsanitize=thread: the encoder's own threads (finalize over nodes / running pods, grouping)
#
# AddressSanitizer / ThreadSanitizer reports go to $OUT/{asan,asan_host,tsan}.<pid> (one file per process that had something to say; an
# AddressSanitizer report also ends its process, i.e. fails the run); UndefinedBehaviorSanitizer writes "runtime error" lines to stderr, so the
# passes run with -s (no capture: pytest-xdist workers inherit stderr) into $OUT/pass{1,2}.log.  The script ends with the counts.
# usage: [PASSES="1 2 3 4"] tests/tools/sanitize_cpu.sh [out_dir]      (34 minutes on 8 cores for all four: profiles/r16d_sanitize_cpu.txt)
# Do not rebuild tests/emu or oracle while a pass runs: the passes load whatever library is there.
set -u
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
OUT="${1:-/tmp/casim_sanitize}"
B="$OUT/build_asan"; T="$OUT/build_tsan"   # (outside the repo: nothing of this travels to the GPU box)
mkdir -p "$OUT" "$B/obj" "$T"
GCC_ASAN="$(gcc -print-file-name=libasan.so)"; GCC_TSAN="$(gcc -print-file-name=libtsan.so)"
CLANG_ASAN="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
EMU_SRC="casim_emu.cpp casim_emu_api.cpp"
EMU_FLAGS="-O1 -g -fPIC -fvisibility=hidden -Wl,-Bsymbolic -std=c++17 -ffp-contract=off -I. -I../../include -fno-omit-frame-pointer -shared"
cd "$ROOT"
PASSES="${PASSES:-1 2 3 4}"
has() { [[ " $PASSES " == *" $1 "* ]]; }

if has 1; then
echo "== build: emulator + oracle, gcc address + undefined =="
(cd tests/emu && g++ $EMU_FLAGS -fsanitize=address,undefined -o "$B/libcasim_emu.so" $EMU_SRC) || exit 1
(cd oracle && gcc -O1 -g -fPIC -std=c11 -D_GNU_SOURCE -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o "$B/libcasim_oracle.so" casim_oracle.c -lm) || exit 1
echo "== pass 1: the CPU tier under them =="
CASIM_EMU_LIB="$B/libcasim_emu.so" CASIM_ORACLE_LIB="$B/libcasim_oracle.so" LD_PRELOAD="$GCC_ASAN" \
  ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:log_path=$OUT/asan" UBSAN_OPTIONS="print_stacktrace=0:halt_on_error=0" \
  python -m pytest tests -q -s -m "not gpu" -n 7 -p no:cacheprovider > "$OUT/pass1.log" 2>&1
tail -1 "$OUT/pass1.log"
fi

if has 2; then
echo "== build: libcasim.so, host side address + undefined (hipcc) =="
make -s -C kubernetes_autoscaler_amd/csrc OUT="$B/libcasim.so" OBJDIR="$B/obj" \
  EXTRA="-fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -g -Wno-option-ignored -Wno-inline-asm" > "$OUT/build_libcasim.txt" 2>&1 || { tail -5 "$OUT/build_libcasim.txt"; exit 1; }
echo "== pass 2: encoder / ABI / prefetch / host-logic tests on it =="
CASIM_LIB_PATH="$B/libcasim.so" LD_PRELOAD="$CLANG_ASAN" \
  ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:log_path=$OUT/asan_host" UBSAN_OPTIONS="print_stacktrace=0:halt_on_error=0" \
  python -m pytest tests -q -s -m "not gpu" -n 7 -p no:cacheprovider > "$OUT/pass2.log" 2>&1
tail -1 "$OUT/pass2.log"
fi

if has 3; then
echo "== build: emulator, gcc thread =="
(cd tests/emu && g++ $EMU_FLAGS -fsanitize=thread -o "$T/libcasim_emu.so" $EMU_SRC) || exit 1
echo "== pass 3: host pool self-test =="
for w in 0 1 2 7 32; do
  CASIM_POOL_THREADS=$w CASIM_EMU_LIB="$T/libcasim_emu.so" LD_PRELOAD="$GCC_TSAN" TSAN_OPTIONS="log_path=$OUT/tsan:halt_on_error=0" \
    python -c "import ctypes as C, os; L = C.CDLL(os.environ['CASIM_EMU_LIB']); L.emu_pool_selftest.argtypes = [C.c_int32] * 3; r = L.emu_pool_selftest(20, 3, 8); print('workers', L.emu_pool_workers(), 'self-test', r)"
done 2>&1 | tee "$OUT/pass3.txt"
echo "== pass 3: the streamed call with its parts on the pool's workers, host loops cut over the pool from 16 elements up =="
# (the cases with parts as pool tasks; the emulator tells ThreadSanitizer about its fiber switches.  test_simulations_of_very_different_sizes...
#  launches blocks of 1 024 fibers by the thousand, a ThreadSanitizer thread state each: left out, it takes an hour)
CASIM_HOST_GRAIN=16 CASIM_EMU_LIB="$T/libcasim_emu.so" LD_PRELOAD="$GCC_TSAN" TSAN_OPTIONS="log_path=$OUT/tsan:halt_on_error=0:report_signal_unsafe=0" \
  python -m pytest tests/test_streams_emu.py -q -n 7 -p no:cacheprovider \
  -k "parts-on-the-pool and (streamed_parts or validity or cannot_be_cut or take_the_link)" 2>&1 | tail -1 | tee -a "$OUT/pass3.txt"
fi

if has 4; then
echo "== build: libcasim.so, host side thread (hipcc) =="
CLANG_TSAN="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)"
mkdir -p "$T/obj"
make -s -C kubernetes_autoscaler_amd/csrc OUT="$T/libcasim.so" OBJDIR="$T/obj" \
  EXTRA="-fsanitize=thread -fno-omit-frame-pointer -shared-libsan -g -Wno-option-ignored -Wno-inline-asm" > "$OUT/build_libcasim_tsan.txt" 2>&1 || { tail -5 "$OUT/build_libcasim_tsan.txt"; exit 1; }
echo "== pass 4: the encoder's threaded loops (casim_enc_finalize, casim_enc_group_pods) from 4 elements up =="
# (CASIM_POOL_THREADS=0: the emulator these tests also load is NOT instrumented here — gcc's and clang's runtimes do not mix — and a
#  pool running in it would be reported on the interceptors' view alone; pass 3 watches the pool with everything instrumented)
CASIM_POOL_THREADS=0 CASIM_HOST_GRAIN=4 CASIM_LIB_PATH="$T/libcasim.so" LD_PRELOAD="$CLANG_TSAN" TSAN_OPTIONS="log_path=$OUT/tsan_host:halt_on_error=0:report_signal_unsafe=0" \
  python -m pytest tests/test_incremental_encode.py tests/test_bulk_pods.py tests/test_named_lanes.py tests/test_grouping_native.py tests/test_resident_cluster_emu.py \
  tests/test_sched_emu.py tests/test_removal_emu.py -q -n 6 -p no:cacheprovider 2>&1 | tail -1 | tee "$OUT/pass4.txt"
fi

echo "== reports =="
n=$(ls "$OUT"/asan.* "$OUT"/asan_host.* "$OUT"/tsan.* "$OUT"/tsan_host.* 2>/dev/null | wc -l)
echo "AddressSanitizer / ThreadSanitizer report files: $n"
echo "UndefinedBehaviorSanitizer lines: pass 1 $(grep -c 'runtime error' "$OUT/pass1.log" 2>/dev/null), pass 2 $(grep -c 'runtime error' "$OUT/pass2.log" 2>/dev/null)"
grep -h -o "[a-z_]*\.[a-z]*:[0-9]*:[0-9]*: runtime error: [a-z ]*" "$OUT/pass1.log" "$OUT/pass2.log" 2>/dev/null | sort | uniq -c | sort -rn | head -20
grep -h "ERROR: AddressSanitizer\|WARNING: ThreadSanitizer" "$OUT"/asan.* "$OUT"/asan_host.* "$OUT"/tsan.* "$OUT"/tsan_host.* 2>/dev/null | sed 's/^==[0-9]*==//' | sort | uniq -c | sort -rn | head -20
