# experiment: edge_mlp_fwd occupancy
export TMPDIR=/tmp
for sl in 32 64 96 128; do
  echo "== slots $sl"
  rm -rf gpurun_out/sl; DGCNN_MLP_SLOTS=$sl rocprofv3 --kernel-trace --stats -d gpurun_out/sl --output-format csv -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --stages 2 > gpurun_out/sl.log 2>&1
  python3 scripts/top_kernels.py gpurun_out/sl 30 | grep "edge_mlp"
done
