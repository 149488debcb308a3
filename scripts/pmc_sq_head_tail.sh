#!/bin/bash
# SQ counters (one PMC pass, --kernel-trace only) of the head's tail   usage: scripts/pmc_sq_head_tail.sh <tag> [B] [N]
export TMPDIR=/tmp
O=gpurun_out/prof_ht_sq_$1; rm -rf $O; mkdir -p $O
B=${2:-32}; N=${3:-7000}
rocprofv3 --kernel-trace --stats -d $O/run --output-format csv -- python3 scripts/bench_head_tail.py $B $N 6 > $O/run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $O/sq --output-format csv -- python3 scripts/bench_head_tail.py $B $N 3 > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/sq2 --output-format csv -- python3 scripts/bench_head_tail.py $B $N 3 > $O/sq2.log 2>&1
python3 - <<PY
import collections, csv, glob, json
def per_kernel(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
a, b = per_kernel("sq"), per_kernel("sq2")
dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(glob.glob("$O/run/*/*kernel_stats.csv")[0]))}
out = {}
for k in sorted(a, key=lambda k: -dur.get(k, 0)):
    if "hf::h" not in k:
        continue
    d = dur[k]; den = d * 1e-9 * 2.4e9 * 1024
    cs = dict(a[k]); cs.update(b.get(k, {}))
    name = k.replace("void ", "").replace("dgcnn::hf::", "").split("(")[0]
    e = {"us": d / 1e3, "mfma_pipe": cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / den, "valu_issue": 4 * cs.get("SQ_INSTS_VALU", 0) / den}
    e.update({c: v for c, v in cs.items()})
    out[name] = e
    print(name, "%.0f us mfma %.2f valu %.2f" % (d / 1e3, e["mfma_pipe"], e["valu_issue"]), {c: "%.3g" % v for c, v in cs.items()})
json.dump(out, open("$O/sq.json", "w"), indent=1)
PY
