#!/bin/bash
run() { echo "== $*"; env "$@" timeout -k 10 100 python tools/quick_perf.py dragon bunny 2>&1 | grep -E "ss="; }
run CTGPU_CHUNK=0
run CTGPU_CHUNK=64
run CTGPU_CHUNK=128
run CTGPU_CHUNK=192
run CTGPU_CHUNK=64 CTGPU_WALK_THRESH=32,32,24
run CTGPU_CHUNK=64 CTGPU_WALK_THRESH=48,32,24
