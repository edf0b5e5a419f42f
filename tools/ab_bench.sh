#!/bin/bash
# The pipelined headline (bench.py, four frames in flight) under two builds of libctgpu.so, alternating on one box:
#   bash tools/ab_bench.sh cobbletrace_amd/csrc/variants/libctgpu_other.so [reps]
# (tools/ab_library.py times the walks with the GPU to themselves; this times what the bench line reports)
other=$1; reps=${2:-3}
mkdir -p gpurun_out/ab
for r in $(seq 1 $reps); do
  for which in other current; do
    if [ $which = other ]; then export CTGPU_LIBRARY=$PWD/$other CTGPU_ALLOW_FOREIGN_LIBRARY=1; else unset CTGPU_LIBRARY CTGPU_ALLOW_FOREIGN_LIBRARY; fi
    timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/ab/${which}_$r.json 2> gpurun_out/ab/${which}_$r.err || exit 1
    python - $which $r <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab/%s_%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1].ljust(8), 'ms_per_step %.4f moving %.4f latency %.4f build %s' % (d['ms_per_step'], d['ms_per_step_moving_camera'], d['ms_frame_latency'], d['build_id']['libctgpu']), flush=True)
PY
  done
done
