#!/bin/bash
# Measured L2<->fabric traffic of one bench workload (separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only alongside):
#   bash tools/run_traffic.sh <tag> <name> <bench.py workload flags...>
#   e.g. bash tools/run_traffic.sh r03 1024QU_f32 --nside 1024 --pol P --dtype f32 --nrk 7 --nbatch 1
#        bash tools/run_traffic.sh r03 cg_1024QU_f32 --only cg --nside 1024 --pol P        (unit = one Wiener-CG iteration)
# Output: gpurun_out/<tag>/traffic_<name>.json (+ the kernel statistics of the same command), built by tools/make_traffic_json.py with
# the calibration factors of gpurun_out/<tag>/counter_calibration.json when that file exists (tools/run_calibration.sh).
tag=$1; name=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
steps=7; warm=1
nside=1024; pol=P; dtype=f32; nrk=7; nb=1; unit="∇lnP evaluation"
args=("$@")
for ((i = 0; i < ${#args[@]}; i++)); do
  case ${args[i]} in
    --nside) nside=${args[i+1]};; --pol) pol=${args[i+1]};; --dtype) dtype=${args[i+1]};; --nrk) nrk=${args[i+1]};; --nbatch) nb=${args[i+1]};;
    --steps) steps=${args[i+1]};;
    --only) unit="Wiener-CG iteration (setup launches included)";;
  esac
done
nunits=$((steps + warm))
cmd="python bench.py --warmup $warm --no-ramp --no-cpu-baseline --no-roofline --no-extras --steps $steps $*"
case $pol in I) np=1;; P) np=2;; IP) np=3;; esac
CMBL_SLICE_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $out/tr_$name -o b -- $cmd > $out/tr_$name.log 2>&1
CMBL_SLICE_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $out/pf_$name -o p -- $cmd > $out/pf_$name.log 2>&1
CMBL_SLICE_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $out/pw_$name -o p -- $cmd > $out/pw_$name.log 2>&1
f=$(find $out/pf_$name -name '*counter_collection.csv' | head -1); w=$(find $out/pw_$name -name '*counter_collection.csv' | head -1)
s=$(find $out/tr_$name -name '*kernel_stats.csv' | head -1)
[ -n "$s" ] && cp $s $out/kernel_stats_$name.csv
cal=$out/counter_calibration.json; [ -f $cal ] || cal=$(ls profiles/r*_counter_calibration.json 2>/dev/null | sort | tail -1); [ -n "$cal" ] && [ -f $cal ] || cal=-
python tools/make_traffic_json.py "$f" "$w" $out/traffic_$name.json $nside $np $nb $dtype $nrk $nunits $cal "$unit" > $out/traffic_$name.log 2>&1
tail -4 $out/traffic_$name.log
# keep the merge-back small: the raw per-dispatch csv files are tens of MB
rm -rf $out/pf_$name $out/pw_$name $out/tr_$name
