#!/bin/bash
# Interleaved A/B of two library builds (the in-tree one and an experiment under cmblensing.jl_amd/_dev/, selected with CMBL_LIB) on
# the bench workloads: edit the `lib` list / configurations as needed.  Used for profiles/r03_ab_colblock_rejected.txt.
for rep in 1 2; do
for lib in cmblensing.jl_amd/libcmblens_hip.so cmblensing.jl_amd/_dev/lib_cb.so; do
  for cfg in "--config 3" "--config 5 --steps 20" "--nbatch 8 --steps 20" "" "--nbatch 2"; do
    CMBL_LIB=$lib python bench.py --no-cpu-baseline --no-roofline --no-extras --steps 50 --warmup 3 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $lib) [$cfg]', round(d['value'],1), 'evals/s', round(d['ms_per_step'],3), 'ms/step', d['logpdf'][0])"
  done
done
done
