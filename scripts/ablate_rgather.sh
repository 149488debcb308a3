#!/bin/bash
# Where edge_rgather's time goes: the block step's kernel statistics with the reverse gather
# (a) as built, (b) without its hit-record reads, (c) without its TQ row gathers -- ablation builds (results invalid, times
# only), made on the build host by:   bash scripts/ablate_rgather.sh build     and run on the GPU box by:   ... run <tag>
set -e
HERE="ahn3-dgcnn.pytorch_amd"
if [ "$1" = "build" ]; then
    for v in NOREC NOTQ; do
        DGCNN_EXTRA_HIPCC_FLAGS="-DDGCNN_ABL_RGATHER_$v" python "$HERE/build.py" --force > /dev/null
        cp "$HERE/libdgcnn_hip.so" "$HERE/libdgcnn_hip_abl_$v.so"
    done
    python "$HERE/build.py" --force > /dev/null
    exit 0
fi
export TMPDIR=/tmp
TAG=$2
OUT=gpurun_out/abl_$TAG
rm -rf $OUT; mkdir -p $OUT
B="--steps 20 --warmup 3 --no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg"
for v in BASE NOREC NOTQ; do
    if [ $v = BASE ]; then unset DGCNN_HIP_LIB; else export DGCNN_HIP_LIB=$PWD/$HERE/libdgcnn_hip_abl_$v.so; fi
    rocprofv3 --kernel-trace --stats -d $OUT/$v --output-format csv -- python3 bench.py $B > $OUT/$v.log 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${v}_fetch --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg > $OUT/${v}_fetch.log 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob("$OUT/$v/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "edge_rgather" in r["Name"] or "edge_point_bwd" in r["Name"]:
        print("$v", r["Name"][:40], "avg us", float(r["AverageNs"]) / 1000)
tot, n = 0.0, 0
for g in glob.glob("$OUT/${v}_fetch/*/*counter_collection.csv"):
    per = {}
    for r in csv.DictReader(open(g)):
        if r["Counter_Name"] == "FETCH_SIZE" and "edge_rgather" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    tot += sum(per.values()); n += len(per)
if n:
    print("$v", "edge_rgather FETCH_SIZE per launch: %.1f MB (x2 on gfx950 for wide streams: %.1f MB)" % (tot / n / 1024, 2 * tot / n / 1024))
PY
done
