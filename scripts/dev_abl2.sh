cd /root/repo
export TMPDIR=/tmp
for f in "-DRG_ABLATE_HITS"; do echo "== $f"; DGCNN_EXTRA_HIPCC_FLAGS="$f" python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
rm -rf gpurun_out/sl; rocprofv3 --kernel-trace --stats -d gpurun_out/sl --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sl.log 2>&1
python3 scripts/top_kernels.py gpurun_out/sl 30 | grep "edge_"; done
python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
