CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_f64rows1.so python -m pytest tests/test_gpu_headline_parity.py -q -k "2048_fp64_n10" 2>&1 | tail -3
for r in 1 2; do
for v in 0 1; do
  echo "== F64_ROWS_2WG=$v round $r"
  CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_f64rows$v.so python tools/gpu_time.py 2048 P f64 2>&1 | grep -E "L\*f|L'\*g|∇L|∇lnP"
done; done
