#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 100 python tools/quick_perf.py dragon bunny 2>&1 | grep -E "ss=1"; }
run CTGPU_PACKET_LDS_LEVELS=14
for w in 6 8; do run CTGPU_PACKET_LDS_LEVELS=10 CTGPU_LIBRARY=$PWD/variants/libctgpu_w$w.so; done
run CTGPU_PACKET_LDS_LEVELS=6 CTGPU_LIBRARY=$PWD/variants/libctgpu_w8.so
