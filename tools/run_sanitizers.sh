#!/bin/bash
# AddressSanitizer / UBSan on the CPU-side code (GPU ASan is not available on this pool): the C++ host mirror
# (grid, handler, connectivity, sparsity, flatten) and the host half of the C ABI (validation + repacking,
# through pdh_check_problem; the kernel launchers are stubbed).  Usage: tools/run_sanitizers.sh
set -e
cd "$(dirname "$0")/.."
OUT=build/asan
mkdir -p $OUT
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer tools/sanitize/host_asan.cpp -o $OUT/host_asan
$OUT/host_asan
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -x hip --offload-arch=gfx950 -c polydeal_amd/csrc/pdh_capi.cpp -o $OUT/capi.o
$HIPCC -std=c++17 -O1 -g -fsanitize=address -x hip --offload-arch=gfx950 -c tools/sanitize/stubs.cpp -o $OUT/stubs.o
$HIPCC -std=c++17 -O1 -g -fsanitize=address -c tools/sanitize/capi_asan.cpp -o $OUT/driver.o
$HIPCC -fsanitize=address $OUT/driver.o $OUT/capi.o $OUT/stubs.o -o $OUT/capi_asan
ASAN_OPTIONS=detect_leaks=0 $OUT/capi_asan
echo "sanitizers: clean"
