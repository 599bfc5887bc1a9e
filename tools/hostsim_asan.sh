#!/bin/bash
# tools/hostsim_asan.sh -- the search kernels' sources on the workgroup simulator (tests/hostsim) under
# AddressSanitizer: every load and store of the kernels is bounds-checked against the "device" buffers (host
# allocations with red zones), with output capacities cut down to what each golden needs.  Run before taking a new
# search kernel to the GPU: an out-of-bounds access that is silent there (or kills the process with a memory fault)
# is a one-line report here.   usage: tools/hostsim_asan.sh [ubsan]   (ubsan: UndefinedBehaviorSanitizer instead --
# signed overflow, shifts, misaligned accesses: arithmetic whose host and device results could differ)
set -e
cd "$(dirname "$0")/.."
SAN=address; LIBSAN=libasan.so
if [ "${1:-}" = "ubsan" ]; then SAN=undefined; LIBSAN=libubsan.so; fi
OUT=/tmp/psgpu_$SAN; mkdir -p $OUT
g++ -x c++ -std=c++17 -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer -fPIC -shared -ffp-contract=off -w \
    -Itests/hostsim -Iinclude -Ipocketsphinx_amd/csrc -o $OUT/libpsgpu_hostsim.so \
    tests/hostsim/hipsim.cc pocketsphinx_amd/csrc/psgpu_search.hip pocketsphinx_amd/csrc/psgpu_lm.hip pocketsphinx_amd/csrc/psgpu_flat.hip
export PYTHONPATH=$PWD/tests:$PWD ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:verify_asan_link_order=0
LD_PRELOAD=$(gcc -print-file-name=$LIBSAN) PSGPU_SIM_LIB=$OUT/libpsgpu_hostsim.so python tests/hostsim/asan_cases.py
