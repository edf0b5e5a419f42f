#!/bin/bash
# The host layer (parsers, importers, BVH builder, packer, event queue, RayThread's host side) under AddressSanitizer +
# UndefinedBehaviorSanitizer on the CPU (and the oracle with it, the checker): the whole `-m "not gpu"` suite against the sanitizer build, then hostile input
# (tools/fuzz_loader.py).  GPU sanitizers are not available on this pool; this covers what runs on the host.
#     tools/sanitize_host.sh [fuzz seconds] [seed]
set -e
trap 'rm -rf cobbletrace_amd/host/_san oracle/_san' EXIT     # (built artefacts travel to the GPU box; these have no business there)
cd "$(dirname "$0")/.."
make -s -C cobbletrace_amd/host _san/libcthost.so
make -s -C oracle _san/libctoracle.so
export CTHOST_LIBRARY=$PWD/cobbletrace_amd/host/_san/libcthost.so CTORACLE_LIBRARY=$PWD/oracle/_san/libctoracle.so
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
python3 -m pytest tests -x -q -m "not gpu"
python3 tools/fuzz_loader.py "${1:-120}" "${2:-1}"
