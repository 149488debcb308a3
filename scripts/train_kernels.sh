#!/bin/bash
# dev helper: per-kernel statistics of the configs[3] training step (rocprofv3 --stats).  usage: scripts/train_kernels.sh TAG
export TMPDIR=/tmp
OUT=gpurun_out/train_prof_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t --output-format csv -- python3 bench.py --workload train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/train.log 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/t/**/*_kernel_stats.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step ~", tot/1e6/8)
for r in rows[:45]:
    print("%-70s n %5s avg_us %9.1f  per-step ms %7.3f" % (r["Name"].replace("void ","").replace("dgcnn::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/8))
PY
grep "^{" $OUT/train.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
