#!/bin/bash
# Collect the round's bench lines and rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/run_profiles.sh r04 [quick]'
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/ afterwards (profiles/README.md).
#   1. counter calibration (known-size copies)            -> counter_calibration.json
#   2. measured traffic + kernel statistics per workload  -> traffic_<workload>.json, kernel_stats_<workload>.csv
#      (headline 1024² QU fp32; BASELINE configs 2 / 3 / 5; 8 chains per GPU; one Wiener-CG iteration at 1024² QU and T+QU)
#   3. the bench lines (they read the traffic files of step 2 when those have been copied to profiles/ -- run twice, or copy first)
tag=${1:-r04}
quick=$2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
bash tools/run_calibration.sh $tag > $out/calibration.log 2>&1
bash tools/run_traffic.sh $tag 1024QU_f32 --nside 1024 --pol P --dtype f32 --nrk 7
bash tools/run_traffic.sh $tag 512QU_f32 --nside 512 --pol P --dtype f32 --nrk 7
bash tools/run_traffic.sh $tag 1024IQU_f32 --nside 1024 --pol IP --dtype f32 --nrk 7
bash tools/run_traffic.sh $tag 2048QU_f64 --nside 2048 --pol P --dtype f64 --nrk 10
bash tools/run_traffic.sh $tag 1024QU_f32_B8 --nside 1024 --pol P --dtype f32 --nrk 7 --nbatch 8
bash tools/run_traffic.sh $tag cg_1024QU_f32 --only cg --steps 19 --nside 1024 --pol P
bash tools/run_traffic.sh $tag cg_1024IQU_f32 --only cg --steps 19 --nside 1024 --pol IP
# make the fresh traffic files visible to bench.py on this box
for f in $out/traffic_*.json; do cp $f profiles/${tag}_$(basename $f); done
cp $out/counter_calibration.json profiles/${tag}_counter_calibration.json 2>/dev/null
B="python bench.py --no-cpu-baseline"
python bench.py > $out/bench_line.json 2> $out/bench.err
# kernel statistics of the headline workload over 50 warm steps, one launch over all pol slices: the per-kernel means bench.py's roofline
# leg measures with the kernels' own timestamps (kernel_stats_1024QU_f32.csv, from the 8-step counter run, includes the cold first steps)
CMBL_SLICE_STREAMS=1 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_bench -o b -- python bench.py --steps 50 --warmup 5 --no-ramp --no-cpu-baseline --no-roofline --no-extras > $out/trace_bench.log 2>&1
s=$(find $out/trace_bench -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s $out/kernel_stats_1024QU_f32_50steps.csv
rm -rf $out/trace_bench
if [ -z "$quick" ]; then
  for c in 2 3 5; do $B --config $c --steps 50 > $out/bench_config$c.json 2>> $out/bench.err; done
  $B --nbatch 8 --steps 30 > $out/bench_nbatch8.json 2>> $out/bench.err
  # default mode of the timed region (one launch chain per pol slice) for comparison with kernel_stats_1024QU_f32.csv (one launch over all slices)
  rocprofv3 --kernel-trace --stats -f csv -d $out/trace_streams -o b -- python bench.py --steps 10 --warmup 2 --no-ramp --no-cpu-baseline --no-roofline --no-extras > $out/trace_streams.log 2>&1
  s=$(find $out/trace_streams -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s $out/kernel_stats_1024QU_f32_slice_streams.csv
  rm -rf $out/trace_streams
fi
ls $out
head -c 1500 $out/bench_line.json; echo
tail -5 $out/bench.err
