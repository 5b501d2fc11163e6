export CMBL_PARITY_LOG=$PWD/gpurun_out/r05_parity.log
rm -f $CMBL_PARITY_LOG
python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_1.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/r05_gputest_1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_line_k20.json 2> gpurun_out/r05_bench_err.log
echo "bench rc=$?"
python - <<'PY'
import json
o=json.load(open('gpurun_out/r05_bench_line_k20.json'))
print(o['value'], o['ms_per_step'], o['roofline']['frac'], o['roofline']['avg_launch_us'])
print(json.dumps(o['extras'].get('reference_exact'), ensure_ascii=False)[:1500])
print(o.get('parity_at_config',{}).get('gf_rel_l2'), o['cpu_baseline'])
PY
