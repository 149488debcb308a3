# experiment: reverse-list segment length (single-launch edge_rgather_kernel)
export TMPDIR=/tmp
for seg in 32 48 64 96; do
  sed -i "s/^constexpr int REV_SEG = [0-9]*;/constexpr int REV_SEG = $seg;/" "ahn3-dgcnn.pytorch_amd/csrc/edge_common.h"
  python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
  echo "== REV_SEG $seg"
  rm -rf gpurun_out/sl; rocprofv3 --kernel-trace --stats -d gpurun_out/sl --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/sl.log 2>&1
  python3 scripts/top_kernels.py gpurun_out/sl 30 | grep "rgather"
done
sed -i "s/^constexpr int REV_SEG = [0-9]*;/constexpr int REV_SEG = 64;/" "ahn3-dgcnn.pytorch_amd/csrc/edge_common.h"
python "ahn3-dgcnn.pytorch_amd/build.py" > /dev/null 2>&1
