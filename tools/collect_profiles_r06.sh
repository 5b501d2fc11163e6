#!/bin/bash
# copy what tools/run_profiles_r06.sh left under gpurun_out/r06/ into profiles/ (tracked)
cd "$(dirname "$0")/.." || exit 1
o=gpurun_out/r06
for n in bench_line bench_line_k20 bench_config2 bench_config3 bench_config5 bench_nbatch8; do [ -s $o/$n.json ] && cp $o/$n.json profiles/r06_$n.json; done
for f in $o/traffic_*.json; do [ -s $f ] && cp $f profiles/r06_$(basename $f); done
for n in kernel_stats_1024QU_f32_50steps.csv kernel_stats_768QU_f32_anysize.csv kernel_stats_1000QU_f32_anysize.csv kernel_stats_1536QU_f32_anysize.csv configs_table.txt anysize_times.txt pmc_sq_anysize_768.txt; do [ -s $o/$n ] && cp $o/$n profiles/r06_$n; done
[ -s $o/small_ab_final.txt ] && cp $o/small_ab_final.txt profiles/r06_small_flow_table.txt
ls -la profiles/r06_* | wc -l
