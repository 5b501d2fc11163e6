CMBL_LIB=cmblensing.jl_amd/_dev/lib_colpl.so NT=5 ROUNDS=3 python tools/gpu_opt_ab.py col_pipeline 0,1,2 2048 P f64 10 2>&1 | grep "MIN\|round 0" | tee gpurun_out/r06_ab_col_pipeline.txt
