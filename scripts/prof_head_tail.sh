#!/bin/bash
# rocprofv3 kernel statistics of scripts/bench_head_tail.py   usage: scripts/prof_head_tail.sh <tag> [B] [N]
export TMPDIR=/tmp
O=gpurun_out/prof_ht_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/run --output-format csv -- python3 scripts/bench_head_tail.py ${2:-32} ${3:-7000} 10 > $O/run.log 2>&1
tail -1 $O/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$O/run/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9))
PY
