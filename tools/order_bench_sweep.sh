#!/bin/bash
# bench.py under CTGPU_ORDER (heavy chunks first): 1 the default (frames with few packets per wave only), 2 every frame, 3 pipelined frames too
mkdir -p gpurun_out/s3
for o in 1 2 3; do
  for inf in 1 4; do
    CTGPU_ORDER=$o timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --inflight $inf > gpurun_out/s3/order_${o}_$inf.json 2> gpurun_out/s3/order_${o}_$inf.err || exit 1
  done
done
