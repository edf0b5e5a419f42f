#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 100 python tools/quick_perf.py dragon bunny 2>&1 | grep -E "ss=1"; }
for d in 4 6 8; do
run CTGPU_LDS_DEPTH=$d
for w in 4 6 8; do run CTGPU_LDS_DEPTH=$d CTGPU_LIBRARY=$PWD/variants/libctgpu_w$w.so; done
done
