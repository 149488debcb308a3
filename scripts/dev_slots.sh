# experiment: streaming-pass occupancy (workgroup slots per XCD) vs kernel time
export TMPDIR=/tmp
for sl in 64 128 256 512; do
  echo "== slots $sl"
  rm -rf gpurun_out/sl; DGCNN_EDGE_SLOTS=$sl rocprofv3 --kernel-trace --stats -d gpurun_out/sl --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sl.log 2>&1
  python3 scripts/top_kernels.py gpurun_out/sl 30 | grep "edge_\|bwd_point\|wgrad"
  grep -h metric gpurun_out/sl.log | cut -c1-140
done
