#!/bin/bash
# Round 5's reduced profile collection (the kernels of the 1024^2 / 2048^2 workloads are unchanged from round 4 except for the quarter twiddle
# table of the 2048-point double-precision rows; what changed is the launch geometry at <= 512^2, the native quadratic estimator and the
# bench line's reference-exact leg):   gpurun --timeout 2400 -- 'bash tools/run_profiles_r05.sh'
tag=r05
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 600 bash tools/run_traffic.sh $tag 1024QU_f32 --nside 1024 --pol P --dtype f32 --nrk 7
timeout 600 bash tools/run_traffic.sh $tag 512QU_f32 --nside 512 --pol P --dtype f32 --nrk 7
for f in $out/traffic_*.json; do cp $f profiles/${tag}_$(basename $f); done
CMBL_SLICE_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_bench -o b -- python bench.py --steps 50 --warmup 5 --no-ramp --no-cpu-baseline --no-roofline --no-extras > $out/trace_bench.log 2>&1
s=$(find $out/trace_bench -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s $out/kernel_stats_1024QU_f32_50steps.csv && cp $s profiles/${tag}_kernel_stats_1024QU_f32_50steps.csv
rm -rf $out/trace_bench
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_line_k20.json 2>> $out/bench.err
B="python bench.py --no-cpu-baseline"
for c in 2 3 5; do timeout 900 $B --config $c --steps 50 > $out/bench_config$c.json 2>> $out/bench.err; done
timeout 600 $B --nbatch 8 --steps 30 > $out/bench_nbatch8.json 2>> $out/bench.err
timeout 600 python tools/gpu_configs.py > $out/configs_table.txt 2>&1
ls $out
python - <<'PY'
import json
for n in ("bench_line", "bench_line_k20", "bench_config2", "bench_config3", "bench_config5", "bench_nbatch8"):
    try:
        o = json.load(open("gpurun_out/r05/%s.json" % n))
        print(n, round(o["value"], 2), round(o["ms_per_step"], 3), o.get("roofline", {}).get("kernel"), round(o.get("roofline", {}).get("frac", 0), 3))
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -3 $out/bench.err
