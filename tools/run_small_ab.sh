#!/bin/bash
# profiles/r06_ab_small_flow.txt: the one-launch small-map flows (option small_flow) against the two-launches-per-stage path, B = 1 and B = 64
{ for cfg in "128 P f32" "64x128 P f32" "128x64 P f32" "64 P f32" "32x128 P f32" "32x64 P f32" "32 P f32" "64 P f64" "32x64 P f64" "32 P f64"; do echo "# $cfg B=1"; NT=20 ROUNDS=2 python tools/gpu_opt_ab.py small_flow 0,1 $cfg 7 2>&1 | grep "MIN"; done
  for cfg in "128 P f32" "64x128 P f32" "64 P f32" "32 P f32" "64 P f64" "32 P f64"; do echo "# $cfg B=64"; NBATCH=64 NT=10 ROUNDS=2 python tools/gpu_opt_ab.py small_flow 0,1 $cfg 7 2>&1 | grep "MIN"; done; } > gpurun_out/r06_small_ab.txt 2>&1
