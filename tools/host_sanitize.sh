#!/usr/bin/env bash
# Build the host library sources together with their C++ self-test under
# AddressSanitizer + UndefinedBehaviorSanitizer and run it.  Host code only:
# the GPU code is checked by its numerics tests instead.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
out="${TMPDIR:-/tmp}/nanofed_host_selftest"
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -static-libasan \
    -fno-sanitize-recover=undefined \
    "$root"/csrc/host/crypto.cpp "$root"/csrc/host/tensor_json.cpp \
    "$root"/csrc/host/philox_cpu.cpp "$root"/csrc/host/tests/selftest.cpp \
    -lcrypto -pthread -o "$out"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0:verify_asan_link_order=0 \
    UBSAN_OPTIONS=print_stacktrace=1 "$out"
