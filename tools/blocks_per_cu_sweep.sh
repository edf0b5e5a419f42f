mkdir -p gpurun_out/s3
for b in 0 26 24 20 16; do
  CTGPU_BLOCKS_PER_CU=$b timeout -k 10 200 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/s3/bpc_$b.json 2> gpurun_out/s3/bpc_$b.err || exit 1
  python - $b <<'PY'
import json,sys
b=sys.argv[1]
d=json.loads(open('gpurun_out/s3/bpc_%s.json'%b).read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('blocks_per_cu',b,'ms_per_step',d['ms_per_step'],'latency',d['ms_frame_latency'],'parity',d['parity']['differing_pixels'],
      ' '.join('%s %.3f/%.3f'%(n.replace('ct_',''),v['avg_ms'],v['avg_ms_inflight']) for n,v in k.items()), flush=True)
PY
done
