#!/bin/bash
# rocprofv3 kernel statistics of scripts/full_model_bench.py   usage: scripts/prof_full_model.sh <tag> B N k [eval|train]
export TMPDIR=/tmp
O=gpurun_out/prof_fm_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/run --output-format csv -- python3 scripts/full_model_bench.py $2 $3 $4 ${5:-eval} > $O/run.log 2>&1
tail -3 $O/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/run/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print(r["Name"][:86].ljust(86), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), ("%.1f %%" % (100.0 * int(r["TotalDurationNs"]) / tot)).rjust(8))
PY
