mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for T in 768 384 192; do
  export DGCNN_WGRAD_TARGET=$T
  bash scripts/prof_train.sh r4_18_b4_t$T 4 > gpurun_out/r4/prof18_$T.log 2>&1
  python bench.py --workload train --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4/train_b4_18_$T.json 2>/dev/null
done
export DGCNN_WGRAD_TARGET=384
bash scripts/prof_train.sh r4_18_b32_t384 32 > gpurun_out/r4/prof18_b32.log 2>&1
echo done
