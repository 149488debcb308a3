#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one PMC pass each, --kernel-trace only) of the head's tail at a training step's
# shape, next to the kernels' times.   usage: scripts/pmc_head_tail.sh <tag> [B] [N]   -> gpurun_out/prof_ht_<tag>/traffic.json
export TMPDIR=/tmp
O=gpurun_out/prof_ht_$1; rm -rf $O; mkdir -p $O
B=${2:-32}; N=${3:-7000}
rocprofv3 --kernel-trace --stats -d $O/run --output-format csv -- python3 scripts/bench_head_tail.py $B $N 10 > $O/run.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 scripts/bench_head_tail.py $B $N 3 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 scripts/bench_head_tail.py $B $N 3 > $O/write.log 2>&1
python3 - <<PY
import collections, csv, glob, json
def per_kernel(sub, name):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
        for (_, k), v in per.items():
            acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(glob.glob("$O/run/*/*kernel_stats.csv")[0]))}
doc = {"command": "scripts/pmc_head_tail.sh $1 $B $N: rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) -- "
                  "python3 scripts/bench_head_tail.py $B $N 3; durations from a --stats run of the same script",
       "correction": "hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024 (MI355X_MICROARCH.md, HBM section: gfx950's "
                     "FETCH_SIZE reports half of the bytes of wide streaming reads)", "kernels": {}}
for k in sorted(set(fetch) | set(write), key=lambda k: -dur.get(k, 0)):
    if "dgcnn" not in k or dur.get(k, 0) < 20000:
        continue
    b = 2.0 * fetch.get(k, 0.0) * 1024 + write.get(k, 0.0) * 1024
    name = k.replace("void ", "").replace("dgcnn::hf::", "").replace("dgcnn::", "").split("(")[0]
    doc["kernels"][name] = {"us": dur[k] / 1e3, "hbm_MB": b / 1e6, "fetch_MB_x2": 2.0 * fetch.get(k, 0.0) * 1024 / 1e6,
                            "write_MB": write.get(k, 0.0) * 1024 / 1e6, "TBps": b / dur[k] / 1e3}
    print("%-46s %8.1f us %9.1f MB  %5.2f TB/s" % (name[:46], dur[k] / 1e3, b / 1e6, b / dur[k] / 1e3))
json.dump(doc, open("$O/traffic.json", "w"), indent=1)
PY
