python -m pytest tests/test_gpu_small.py -q -x 2>&1 | tail -5
bash tools/run_small_ab.sh; cat gpurun_out/r06_small_ab.txt
