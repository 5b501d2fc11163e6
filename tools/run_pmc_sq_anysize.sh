#!/bin/bash
# SQ counters of the any-size step at 768^2 QU fp32 (two passes of <= 8 counters), summarised per kernel: bash tools/run_pmc_sq_anysize.sh [tag]
tag=${1:-r05any}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
P4="python tools/gpu_step_loop.py 768 P f32 4"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -f csv -d $out/pmc_sq1 -o p -- $P4 > $out/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -f csv -d $out/pmc_sq2 -o p -- $P4 > $out/pmc_sq2.log 2>&1
for d in pmc_sq1 pmc_sq2; do f=$(find $out/$d -name '*counter_collection.csv' | head -1); python tools/pmc_summary.py $f | head -8; done > $out/pmc_sq_summary.txt
cat $out/pmc_sq_summary.txt | cut -c1-260
