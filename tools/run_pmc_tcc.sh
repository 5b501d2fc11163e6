#!/bin/bash
# L2 <-> fabric (EA) queueing counters per kernel: requests, queue level (=> mean cycles a request is outstanding), stalls for lack of DRAM
# credits.  bash tools/run_pmc_tcc.sh <tag> <name> <bench.py workload flags...>   (through gpurun; separate passes, kernel trace only alongside)
tag=$1; name=$2; shift 2
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null; cd - > /dev/null; export TMPDIR=/tmp
cmd="python bench.py --warmup 1 --steps 3 --no-ramp --no-cpu-baseline --no-roofline --no-extras $*"
CMBL_SLICE_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum -f csv -d $out/tcc1_$name -o p -- $cmd > $out/tcc1_$name.log 2>&1
CMBL_SLICE_STREAMS=1 timeout 900 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum -f csv -d $out/tcc2_$name -o p -- $cmd > $out/tcc2_$name.log 2>&1
for d in tcc1_$name tcc2_$name; do f=$(find $out/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | head -9 || tail -3 $out/$d.log; done > $out/pmc_tcc_$name.txt
cut -c1-200 $out/pmc_tcc_$name.txt
rm -rf $out/tcc1_$name $out/tcc2_$name
