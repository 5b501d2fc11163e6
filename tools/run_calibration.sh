#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known-size streaming kernels (tools/micro/calib_copy.hip):  bash tools/run_calibration.sh <tag>
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/calib_copy.hip -o /tmp/calib_copy || exit 1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/cal_f -o c -- /tmp/calib_copy > $out/cal_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/cal_w -o c -- /tmp/calib_copy > $out/cal_w.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/cal_t -o c -- /tmp/calib_copy > $out/cal_t.log 2>&1
f=$(find $out/cal_f -name '*counter_collection.csv' | head -1); w=$(find $out/cal_w -name '*counter_collection.csv' | head -1)
python tools/make_calib_json.py "$f" "$w" $out/counter_calibration.json | tee $out/counter_calibration.txt
s=$(find $out/cal_t -name '*kernel_stats.csv' | head -1); [ -n "$s" ] && cp $s $out/calib_copy_kernel_stats.csv
rm -rf $out/cal_f $out/cal_w $out/cal_t
