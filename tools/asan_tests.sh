#!/bin/bash
# The C-ABI host layer under AddressSanitizer: builds libcss_mi355_asan.so (host side
# instrumented, device code as shipped) and runs the given pytest selection against it.
#   tools/asan_tests.sh                                   # CPU: the C-ABI error paths (tests/test_cabi.py -m "not gpu")
#   tools/asan_tests.sh -m gpu tests/test_cabi.py tests/test_hip_schedules.py     # on the GPU box: queue / schedule paths too
set -e
cd "$(dirname "$0")/.."
[ -n "$ASAN_NO_BUILD" ] || make -C notsofar1-challenge_amd/csrc -j8 asan > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(-m "not gpu" tests/test_cabi.py)
# (python itself is not instrumented: leaks of the interpreter are not ours to report; ROCm's runtime allocates at exit)
# (under the sanitizer's dlopen interceptor torch no longer finds its own lazily loaded libraries by RPATH)
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))" 2>/dev/null || true)
LD_LIBRARY_PATH="$TL:$LD_LIBRARY_PATH" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:protect_shadow_gap=0 \
  CSS_MI355_LIBRARY="$PWD/notsofar1-challenge_amd/libcss_mi355_asan.so" python -m pytest -x -q "${ARGS[@]}"
