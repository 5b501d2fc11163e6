#!/bin/bash
# profiles/rNN_anysize_times.txt: the any-size path with and without the compile-time-plan transforms / slice streams (tools/gpu_time.py)
for cfg in "360 P" "768 P" "1000 P" "1536 P" "768 IP"; do
  for ct in 0 1; do
    echo "## CMBL_GEN_CT=$ct  (0 = run-time-planned transforms of rounds 2-4; 1 = compile-time plans, kernels_ct.hpp)"
    CMBL_GEN_CT=$ct python tools/gpu_time.py $cfg 2>&1 | grep -v amdgpu
  done
done
echo "## rectangular 1280 x 640 (5 * 2^k): tools/gpu_opt_ab.py gen_ct 0,1"
NT=5 ROUNDS=2 python tools/gpu_opt_ab.py gen_ct 0,1 640 P f32 7 2>&1 | grep MIN
echo "## slice streams (option gen_slice_streams; forced for every size with CMBL_GEN_STREAMS_MIN_PIX=0): tools/gpu_opt_ab.py gen_slice_streams 0,1"
for cfg in "96 P" "360 P" "768 P" "768 IP" "1000 P" "1000 IP" "1536 P"; do
  echo "# $cfg"; CMBL_GEN_STREAMS_MIN_PIX=0 NT=5 ROUNDS=2 python tools/gpu_opt_ab.py gen_slice_streams 0,1 $cfg f32 7 2>&1 | grep MIN
done
