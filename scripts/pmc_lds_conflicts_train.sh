#!/bin/bash
# LDS bank-conflict share per kernel of the training step (one PMC pass)   usage: scripts/pmc_lds_conflicts_train.sh <tag> [B]
export TMPDIR=/tmp
O=gpurun_out/prof_ldst_$1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/run --output-format csv -- python3 bench.py --workload train --batch ${2:-32} --steps 3 --warmup 1 --no-cpu-baseline > $O/run.log 2>&1
python3 - <<PY
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/run/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in per.items():
        acc[k][c].append(v)
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_IDX_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    if m.get("SQ_LDS_IDX_ACTIVE", 0) < 1e5 or "at::" in k:
        continue
    name = k.replace("void ", "").replace("dgcnn::", "").split("(")[0][:48]
    print("  %-48s n %3d lds_active %.3g conflict %.3g (%.0f %%) insts %.3g busy %.3g" % (name, len(cs["SQ_LDS_IDX_ACTIVE"]), m["SQ_LDS_IDX_ACTIVE"], m.get("SQ_LDS_BANK_CONFLICT", 0), 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], m.get("SQ_INSTS_LDS", 0), m.get("SQ_BUSY_CYCLES", 0)))
PY
