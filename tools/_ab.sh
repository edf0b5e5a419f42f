set -e
for i in 1 2 3; do
 for v in base new; do
  if [ $v = base ]; then export CTGPU_LIBRARY=$PWD/variants/base.so; else unset CTGPU_LIBRARY; fi
  python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$v', d['ms_per_step'], d['value'], 'alone: primary', k['ct_walk_packet<primary>']['avg_ms_alone'], 'shadow', k['ct_walk_packet<shadow>']['avg_ms_alone'])"
 done
done
unset CTGPU_LIBRARY
python tools/ss0_breakdown.py dragon bunny 2>&1 | grep -E "ss=|walk_packet"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
