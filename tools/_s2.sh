export CMBL_LIB=$PWD/cmblensing.jl_amd/_dev/lib_pl64.so
python tools/gpu_opt_ab.py col_pipeline 0,1 2048 P f64 10 > gpurun_out/r05_pl64_ab.txt 2>&1
tail -4 gpurun_out/r05_pl64_ab.txt
