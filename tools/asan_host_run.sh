#!/bin/bash
# tools/asan_host_run.sh OUTDIR [pytest args...]   (on the GPU box, from the repo root; builds csrc/build/libcaf_hip_asan.so there if missing: ~2 min)
# The -m gpu tests of the host-facing parts of the library with its HOST code under AddressSanitizer (libcaf_hip_asan.so: the
# device code is the product's, GPU ASan does not exist on this pool).  Python is not instrumented, so the sanitizer runtime
# is preloaded; leak checking is off (the interpreter and the HIP runtime never free everything), and the shadow gap is left
# unprotected because the HSA runtime maps its apertures wherever it likes.  An ASan report ends the process with status 99.
set -u
O=${1:?output dir}; shift
mkdir -p "$O"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
# (under the sanitizer's dlopen interceptor torch's lazily loaded libraries are no longer found through its RUNPATH)
TL=$(python3 -c 'import importlib.util, os; print(os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib"))')
export LD_LIBRARY_PATH="$TL:${LD_LIBRARY_PATH:-}"
export CAF_HIP_LIB="$PWD/caf_cookoff_amd/csrc/build/libcaf_hip_asan.so"
export CAF_HIP_MEASURE_LIB="$PWD/caf_cookoff_amd/csrc/build/libcaf_hip_asan_measure.so"
{ [ -f "$CAF_HIP_LIB" ] && [ -f "$CAF_HIP_MEASURE_LIB" ]; } || make -j2 -C caf_cookoff_amd/csrc asan asan-measure > "$O/asan_build.log" 2>&1 || { echo "asan build failed"; exit 2; }
export ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:exitcode=99:abort_on_error=0:halt_on_error=1:log_path=$O/asan"
TESTS=("$@")
[ ${#TESTS[@]} -eq 0 ] && TESTS=(tests/test_gpu_host_api.py tests/test_gpu_multi.py tests/test_gpu_timeout.py tests/test_gpu_abi_errors.py tests/test_gpu_stream.py tests/test_gpu_generic_xcor.py)
# (deselected: the test that starts torchrun as a child -- a second instrumented interpreter whose RCCL start-up runs out of
#  the sanitizer allocator's address space; it exercises torch.distributed, not this library's host code)
LD_PRELOAD="$RT" timeout -k 10 900 python -m pytest "${TESTS[@]}" -m gpu -q -p no:cacheprovider \
    --deselect tests/test_gpu_multi.py::test_peak_reduction_through_rccl_single_rank > "$O/pytest.log" 2>&1
rc=$?
echo "asan host run rc=$rc :: $(grep -E 'passed|failed|error' "$O/pytest.log" | tail -1)" | tee "$O/summary.txt"
ls "$O"/asan.* 2>/dev/null | head -5
exit $rc
