#!/bin/bash
# Builds a variant of the library for an A/B on one box: tools/ab_lib.sh <name> [extra hipcc flags...]
#   -> tools/libs/libtopay_<name>.so, with -DTOPAY_EXPERIMENTS (the tuning switches read from the environment and the A/B-only
#      kernels exist in such builds only).  Use it with TOPAY_LIB=tools/libs/libtopay_<name>.so.
set -e
name=$1; shift
mkdir -p tools/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DTOPAY_EXPERIMENTS "$@" \
  -o tools/libs/libtopay_$name.so topay_amd/csrc/topay_hip.hip
python3 tools/isa_lint.py --build --flag=-DTOPAY_EXPERIMENTS $(for f in "$@"; do echo --flag=$f; done) | tail -2
