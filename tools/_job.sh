export TMPDIR=/tmp
mkdir -p gpurun_out/r06any
for n in 1536 1000; do
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r06any/kt$n -o p -- python tools/gpu_step_loop.py $n P f32 10 > gpurun_out/r06any/kt$n.log 2>&1
f=$(find gpurun_out/r06any/kt$n -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f gpurun_out/r06_kernel_stats_${n}QU_f32_anysize.csv; echo "== $n"; head -8 $f | cut -c1-140; fi
rm -rf gpurun_out/r06any/kt$n
done < /dev/null
