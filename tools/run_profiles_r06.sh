#!/bin/bash
# Round 6's profile collection, ONE call with the final build:   gpurun --timeout 3000 -- 'bash tools/run_profiles_r06.sh'   then   bash tools/collect_profiles_r06.sh  (gpurun returns gpurun_out/ only)
# (the 1024^2 / 2048^2 kernels are round 4-5's; what changed: the any-size path -- merged x passes, launch-size dependent group shapes, 1920 --, the
#  one-launch small-map flows, the bench line's new extras)
tag=r06
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
# headline: kernel trace over 50 warm steps (the roofline.rocprofv3 cross-check of bench.py) and the two counter passes
CMBL_SLICE_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_bench -o b -- python bench.py --steps 50 --warmup 5 --no-ramp --no-cpu-baseline --no-roofline --no-extras > $out/trace_bench.log 2>&1
s=$(find $out/trace_bench -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s $out/kernel_stats_1024QU_f32_50steps.csv
rm -rf $out/trace_bench
timeout 600 bash tools/run_traffic.sh $tag 1024QU_f32 --nside 1024 --pol P --dtype f32 --nrk 7
# bench lines: the default (as the driver runs it: K = 20, and K = 200), the other BASELINE configurations WITH their extras, 8 chains per GPU
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_line_k20.json 2>> $out/bench.err
B="python bench.py --no-cpu-baseline"
for c in 2 3 5; do timeout 900 $B --config $c --steps 50 > $out/bench_config$c.json 2>> $out/bench.err; done
timeout 600 $B --nbatch 8 --steps 30 > $out/bench_nbatch8.json 2>> $out/bench.err
timeout 600 python tools/gpu_configs.py > $out/configs_table.txt 2>&1
# any-size path: the times table, the kernel statistics and SQ counters of the 768^2 step
timeout 1500 bash tools/run_anysize_times.sh > $out/anysize_times.txt 2>&1
{ echo "## 1920^2 (15 * 2^7, in CMBL_CT_LIST since round 6) and 960 x 1920 with / without the compile-time plans"; python tools/gpu_time.py 1920 P f32 2>&1 | grep -v amdgpu; NT=5 ROUNDS=1 python tools/gpu_opt_ab.py gen_ct 0,1 960x1920 P f32 7 2>&1 | grep MIN; } >> $out/anysize_times.txt 2>&1
for n in 768 1000 1536; do
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/kt$n -o p -- python tools/gpu_step_loop.py $n P f32 20 > $out/kt$n.log 2>&1
  f=$(find $out/kt$n -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_${n}QU_f32_anysize.csv
  rm -rf $out/kt$n
done
timeout 900 bash tools/run_pmc_sq_anysize.sh ${tag}any > $out/pmc_sq_anysize_768.txt 2>&1
rm -rf gpurun_out/${tag}any/pmc_sq1 gpurun_out/${tag}any/pmc_sq2
# small maps
timeout 900 bash tools/run_small_ab.sh; cp gpurun_out/r06_small_ab.txt $out/small_ab_final.txt
python - <<'PY'
import json
for n in ("bench_line", "bench_line_k20", "bench_config2", "bench_config3", "bench_config5", "bench_nbatch8"):
    try:
        o = json.load(open("gpurun_out/r06/%s.json" % n))
        print(n, round(o["value"], 2), round(o["ms_per_step"], 3), o.get("roofline", {}).get("kernel"), round(o.get("roofline", {}).get("frac", 0), 3))
    except Exception as e:
        print(n, "FAILED", e)
PY
tail -3 $out/bench.err
