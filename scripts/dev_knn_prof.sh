#!/bin/bash
# experiment helper: per-kernel times of the kNN path for a given set of -D switches
export TMPDIR=/tmp
flags="$1"
[ -n "$flags" ] && DGCNN_EXTRA_HIPCC_FLAGS="$flags" python "ahn3-dgcnn.pytorch_amd/build.py" --force > /dev/null 2>&1
rm -rf gpurun_out/kp; rocprofv3 --kernel-trace --stats -d gpurun_out/kp --output-format csv -- python3 scripts/dev_knn_one.py 64 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
fs=sorted(glob.glob('gpurun_out/kp/*/*_kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(fs[-1])))[:6]:
    print("   %-50s avg_us %9.2f"%(r['Name'][:50], float(r['AverageNs'])/1e3))
PY
[ -n "$flags" ] && python "ahn3-dgcnn.pytorch_amd/build.py" --force > /dev/null 2>&1
true
