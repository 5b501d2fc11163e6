for c in 5 2 3; do
  python bench.py --config $c --steps 50 --warmup 5 > gpurun_out/r05_bench_config$c.json 2> gpurun_out/r05_bench_config$c.err
  python - <<PY
import json
o=json.load(open('gpurun_out/r05_bench_config$c.json'))
print('config $c', o['ms_per_step'], o['roofline']['kernel'], o['roofline']['frac'], o['roofline']['avg_launch_us'])
print({k:v for k,v in o['extras'].items() if not isinstance(v,dict)})
for k,v in o['roofline']['per_kernel'].items():
    if 'frac' in v: print('   ', k, round(v['avg_launch_us'],1), round(v['frac'],3))
PY
done
