#!/bin/bash
# LDS bank-conflict share per kernel of the block workloads (one PMC pass each)   usage: scripts/pmc_lds_conflicts.sh <tag>
export TMPDIR=/tmp
O=gpurun_out/prof_lds_$1; rm -rf $O; mkdir -p $O
B="--no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg --steps 4 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/one --output-format csv -- python3 bench.py $B > $O/one.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d $O/two --output-format csv -- python3 bench.py $B --stages 2 > $O/two.log 2>&1
python3 - <<PY
import collections, csv, glob
for sub in ("one", "two"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
    print(sub)
    for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_IDX_ACTIVE", [0])) / max(1, len(kv[1].get("SQ_LDS_IDX_ACTIVE", [0])))):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        if m.get("SQ_LDS_IDX_ACTIVE", 0) < 1e5 or "at::" in k:
            continue
        name = k.replace("void ", "").replace("dgcnn::", "").split("(")[0][:40]
        print("  %-40s lds_active %.3g conflict %.3g (%.0f %%) insts %.3g busy %.3g" % (name, m["SQ_LDS_IDX_ACTIVE"], m.get("SQ_LDS_BANK_CONFLICT", 0), 100 * m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"], m.get("SQ_INSTS_LDS", 0), m.get("SQ_BUSY_CYCLES", 0)))
PY
