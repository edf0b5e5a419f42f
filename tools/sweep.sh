#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 100 python tools/quick_perf.py dragon bunny 2>&1 | grep -E "ss="; }
run CTGPU_POOL=0
for t in 4 8 16 24 32 48; do run CTGPU_POOL=1 CTGPU_POOL_THRESH=$t; done
