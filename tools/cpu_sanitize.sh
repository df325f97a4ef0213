#!/bin/bash
# Sanitizer pass over the kernel sources on the CPU (GPU AddressSanitizer is not available on the pool): the lane-emulator
# build of topay_amd/csrc (tests/emu) compiled with UBSan and with ASan, running stage-1/2 solves, the feasibility gate,
# the EDT build and the collision check.  Test infrastructure only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/tests/emu
FLAGS="-O1 -g -std=c++17 -fPIC -march=x86-64-v3 -ffp-contract=off -DTOPAY_LDS= -DTOPAY_GLB= -DTOPAY_CST= -shared -x c++ -I include"
g++ $FLAGS -fsanitize=undefined -fno-sanitize-recover=undefined -o /tmp/libtopay_emu_ubsan.so ../../topay_amd/csrc/topay_hip.hip
g++ $FLAGS -fsanitize=address -fno-omit-frame-pointer -o /tmp/libtopay_emu_asan.so ../../topay_amd/csrc/topay_hip.hip
cd $ROOT
UBSAN_OPTIONS=print_stacktrace=1 python tools/cpu_sanitize_run.py /tmp/libtopay_emu_ubsan.so
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 \
  python tools/cpu_sanitize_run.py /tmp/libtopay_emu_asan.so
echo "sanitizer pass clean"
