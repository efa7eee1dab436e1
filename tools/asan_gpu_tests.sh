#!/bin/bash
# The GPU tests under AddressSanitizer, DEVICE code included (global and LDS accesses of every kernel; host code of the C ABI too).
# ROCm's sanitizer runtime for the GPU is not installed in this image, so a device-side report could not be printed (the process
# would end with "Hostcall: no handler found for service ID 4"): the `asan` build defines the device report functions itself
# (kernels.hip, RMCL_ASAN_LOG) -- the first bad access is recorded, the wave ends, tests/conftest.py fails the test with the
# record.  Checked with a deliberate one-past-the-end store to global memory and to LDS; clean kernels run through.
#   here:       make -C rmcl_amd/csrc asan        (cross-compiles asan_build/librmclhip.so, ~2 min)
#   on the box: /usr/local/graft/bin/gpurun --timeout 1200 -- 'ASAN_TIMEOUT=900 bash tools/asan_gpu_tests.sh [pytest args]'
# (a kernel that trips the sanitizer can also HANG instead of aborting: keep ASAN_TIMEOUT short and run the files one by one)
# The box's copy of the repo is disposable: the product library is REPLACED there by the instrumented build.
set -u
cd "$GRAFT_REPO_ROOT"
cp asan_build/librmclhip.so rmcl_amd/librmclhip.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export RMCLHIP_ASAN_LOG=1 HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:allocator_may_return_null=1
mkdir -p gpurun_out
if [ $# -eq 0 ]; then set -- tests -m gpu; fi
# (the HSA / HIP runtimes are preloaded too: the sanitizer resolves the hsa_* functions it intercepts when IT is loaded)
LD_PRELOAD="$RT:/opt/rocm/lib/libhsa-runtime64.so.1:/opt/rocm/lib/libamdhip64.so" timeout ${ASAN_TIMEOUT:-900} python -m pytest "$@" -v -p no:cacheprovider -p no:faulthandler > gpurun_out/asan_tests.log 2> gpurun_out/asan_tests.err
echo "rc=$?" >> gpurun_out/asan_tests.log
grep -c "PASSED" gpurun_out/asan_tests.log
grep -n "Hostcall\|AddressSanitizer\|ERROR\|FAILED\|rc=" gpurun_out/asan_tests.log gpurun_out/asan_tests.err | head -20
tail -5 gpurun_out/asan_tests.log; tail -20 gpurun_out/asan_tests.err | cut -c1-300
