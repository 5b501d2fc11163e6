#!/bin/bash
# SQ counters of the bench step (two passes of <= 8 counters), summarised per kernel: bash tools/run_pmc_sq.sh [tag]  (through gpurun)
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
P4="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
CMBL_SLICE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $out/pmc_sq1 -o p -- $P4 > $out/pmc_sq1.log 2>&1
CMBL_SLICE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $out/pmc_sq2 -o p -- $P4 > $out/pmc_sq2.log 2>&1
for d in pmc_sq1 pmc_sq2; do f=$(find $out/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f | head -12; done > $out/pmc_sq_summary.txt
cat $out/pmc_sq_summary.txt | cut -c1-230
