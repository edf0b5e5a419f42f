#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 100 python tools/quick_perf.py dragon bunny 2>&1 | grep -E "ss="; }
run CTGPU_WALK_THRESH=64,32,24
run CTGPU_WALK_THRESH=48,32,24
run CTGPU_WALK_THRESH=32,32,24
run CTGPU_WALK_THRESH=24,32,24
run CTGPU_WALK_THRESH=16,32,24
