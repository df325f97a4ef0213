#!/bin/bash
for l in r3_stamps stamps stamps_la2; do echo "== $l"; TOPAY_LIB=$PWD/tools/libs/libtopay_$l.so timeout 300 python3 tools/gpu_stamps.py 2>&1 | tail -22; done
