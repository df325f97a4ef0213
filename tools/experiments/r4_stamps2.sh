#!/bin/bash
for l in r3_stamps stamps; do echo "== $l S=512"; TOPAY_LIB=$PWD/tools/libs/libtopay_$l.so timeout 300 python3 tools/gpu_stamps.py 512 2>&1 | tail -20; done
