#!/bin/bash
# Collect the round's bench lines and rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/run_profiles.sh r02'
# Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/ afterwards (tools/README.md).
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline"
# 1. the bench lines: headline (with cpu baseline), BASELINE configs 2/3/5, 8 chains per GPU as batch slots
python bench.py > $out/bench_line.json 2> $out/bench.err
for c in 2 3 5; do $B --config $c --steps 50 > $out/bench_config$c.json 2>> $out/bench.err; done
$B --nbatch 8 --steps 30 --no-roofline > $out/bench_nbatch8.json 2>> $out/bench.err
$B --pol IP --steps 50 > $out/bench_1024IQU.json 2>> $out/bench.err
# 2. kernel trace + stats of the headline command: one launch over all pol slices (the mode the roofline leg measures), and the default
P="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline"
CMBL_SLICE_STREAMS=1 rocprofv3 --kernel-trace --stats -f csv -d $out/trace_single -o b -- $P > $out/trace_single.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $out/trace_streams -o b -- $P > $out/trace_streams.log 2>&1
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (no other tracing domain alongside the counters)
P4="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline"
CMBL_SLICE_STREAMS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/pmc_fetch -o p -- $P4 > $out/pmc_fetch.log 2>&1
CMBL_SLICE_STREAMS=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/pmc_write -o p -- $P4 > $out/pmc_write.log 2>&1
f=$(find $out/pmc_fetch -name '*counter_collection.csv' | head -1); w=$(find $out/pmc_write -name '*counter_collection.csv' | head -1)
python tools/make_traffic_json.py $f $w $out/traffic_1024QU_f32.json 1024 2 1 f32 5 > $out/traffic.log 2>&1
find $out -name '*kernel_stats.csv' -o -name 'traffic*.json' -o -name 'bench_*.json' | sort
tail -3 $out/traffic.log
