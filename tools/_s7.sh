export CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_qe.so
python -m pytest tests/test_gpu_drivers.py -x -q -k "quadratic_estimate" 2>&1 | tail -12
