export CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_occ.so
for c in "512 P" "256 P" "256 I" "128 P" "512 IP" "256 IP" "128 I"; do
  set -- $c
  ROUNDS=2 python tools/gpu_occ_ab.py $1 $2 > gpurun_out/r05_occ3_$1$2.txt 2>&1
  grep MIN gpurun_out/r05_occ3_$1$2.txt | sed "s/^/$1$2 /"
done
