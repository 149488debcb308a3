mkdir -p gpurun_out/r4
export TMPDIR=/tmp

timeout -k 10 600 python -m pytest tests/test_edgeconv_gpu.py tests/test_model_gpu.py -m gpu -q -x  > gpurun_out/r4/t21.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r4/t21.log
[ $rc -eq 0 ] || exit 1
B="--no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg"
rocprofv3 --kernel-trace --stats -d gpurun_out/r4/p21 --output-format csv -- python3 bench.py --steps 20 --warmup 3 $B > gpurun_out/r4/p21.log 2>&1
cp $(find gpurun_out/r4/p21 -name "*kernel_stats.csv" | head -1) gpurun_out/r4_21_block_kernel_stats.csv
python bench.py $B > gpurun_out/r4/b21d.json 2>/dev/null
echo done
