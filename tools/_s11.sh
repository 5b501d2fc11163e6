bash tools/run_pmc_tcc.sh r05 2048QU_f64 --config 5
bash tools/run_pmc_tcc.sh r05 1024QU_f32
