"""Turns the rocprofv3 output of scripts/profile_round.sh into the small files kept under profiles/:
  <tag>_kernel_stats.csv, <tag>_two_stage_kernel_stats.csv, <tag>_train_step_kernel_stats.csv   (kernel, calls, total, average ns)
  <tag>_bench_line.json                      the bench.py line of the profiled block run
  hbm_traffic.json                           FETCH_SIZE / WRITE_SIZE per launch, corrected as MI355X_MICROARCH.md prescribes
  mfma_busy.json                             SQ counters per launch of the large kernels (MFMA-busy, VALU-active, ...)
usage: profile_summaries.py gpurun_out/prof_<tag> <tag>   (writes into gpurun_out/prof_<tag>/summary/; copy to profiles/)"""
import collections
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
out = os.path.join(root, "summary")
os.makedirs(out, exist_ok=True)


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


for sub, name in (("block", "%s_kernel_stats.csv"), ("two", "%s_two_stage_kernel_stats.csv"), ("train", "%s_train_step_kernel_stats.csv")):
    f = newest(os.path.join(root, sub, "**", "*_kernel_stats.csv"))
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, name % tag), "w") as g:
            w = csv.writer(g)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"]])
for sub, name in (("block", "%s_bench_line.json"), ("two", "%s_two_stage_bench_line.json"), ("train", "%s_train_bench_line.json")):
    log = os.path.join(root, sub + ".log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                open(os.path.join(out, name % tag), "w").write(line)


def per_kernel(sub, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in counters:
                per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def short(k):
    k = k.replace("void ", "").replace("dgcnn::", "")
    for cut in ("(", "<"):
        if cut in k:
            k = k[:k.index(cut)]
    if k.startswith("_ZN5dgcnn"):
        import re
        m = re.match(r"_ZN5dgcnn(\d+)", k)
        n = int(m.group(1))
        k = k[len(m.group(0)):len(m.group(0)) + n]
    return k


REGIONS = {
    "knn_bound16s_kernel": ["knn_bound16s_kernel", "knn_bound_merge_kernel"],
    "knn_filter16_kernel": ["knn_filter16_kernel"],
    "knn_refine_kernel": ["knn_refine_kernel"],
    "edge_reduce_kernel": ["edge_reduce_kernel"],
    "edge_rgather_kernel": ["edge_rgather_kernel"],
}
fetch = {k: v.get("FETCH_SIZE", 0.0) for k, v in per_kernel("fetch", {"FETCH_SIZE"}).items()}
write = {k: v.get("WRITE_SIZE", 0.0) for k, v in per_kernel("write", {"WRITE_SIZE"}).items()}
if fetch and write:
    doc = {"command": "scripts/profile_round.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) "
                      "-- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg" % tag,
           "correction": "hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024 (MI355X_MICROARCH.md section HBM: on "
                         "gfx950 FETCH_SIZE reports half of the bytes of a wide streaming read; calibrated for 16-byte-per-lane "
                         "streams, so the read side of the dword-per-lane row gathers is an upper bound)",
           "hbm_bytes_per_launch": {}, "FETCH_SIZE_KB": {}, "WRITE_SIZE_KB": {}, "all_kernels_hbm_bytes_per_launch": {}}
    for region, pats in REGIONS.items():
        f = sum(v for k, v in fetch.items() if any(p in k for p in pats))
        w = sum(v for k, v in write.items() if any(p in k for p in pats))
        doc["FETCH_SIZE_KB"][region], doc["WRITE_SIZE_KB"][region] = f, w
        doc["hbm_bytes_per_launch"][region] = 2.0 * f * 1024.0 + w * 1024.0
    for k in sorted(set(fetch) | set(write)):
        doc["all_kernels_hbm_bytes_per_launch"][short(k)] = 2.0 * fetch.get(k, 0.0) * 1024.0 + write.get(k, 0.0) * 1024.0
    json.dump(doc, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)

names = {"SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_MFMA",
         "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY"}
CLOCK_HZ, SIMDS = 2.4e9, 1024   # MI355X: 256 CUs x 4 SIMDs at up to 2.4 GHz (MI355X_MICROARCH.md)


def durations(sub):
    f = newest(os.path.join(root, sub, "**", "*_kernel_stats.csv"))
    return {short(r["Name"]): float(r["AverageNs"]) for r in csv.DictReader(open(f))} if f else {}


doc = {"command": "scripts/profile_round.sh %s: rocprofv3 --kernel-trace --pmc %s -- python3 bench.py --steps 4 --warmup 2 "
                  "--no-cpu-baseline --no-train-leg [--stages 2]; durations from the --stats run of the same script" % (
                      tag, " ".join(sorted(names))),
       "note": "per launch.  SQ_VALU_MFMA_BUSY_CYCLES is in SIMD-cycles summed over the chip (it equals SQ_INSTS_MFMA x the "
               "instruction's cycle count: 32 for v_mfma_f32_32x32x16_f16, 64 for v_mfma_f32_32x32x2_f32), so "
               "mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); valu_issue_util = 4 x "
               "SQ_INSTS_VALU / the same denominator (a wave64 VALU instruction occupies its SIMD for 4 cycles).  Where the "
               "two add up to most of the time and valu_issue_util is the larger, the kernel is bound by its VALU epilogue, "
               "not by the matrix product (DESIGN.md section 5).",
       "one_stage": {}, "two_stage": {}}
for sub, key, dsub in (("sq", "one_stage", "block"), ("sq2", "two_stage", "two")):
    dur = durations(dsub)
    for k, cs in per_kernel(sub, names).items():
        if cs.get("SQ_BUSY_CYCLES", 0) < 1e5 or "at::" in k:
            continue
        e = dict(cs)
        d = dur.get(short(k))
        if d:
            denom = d * 1e-9 * CLOCK_HZ * SIMDS
            e["duration_us"] = d / 1e3
            e["mfma_pipe_util"] = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / denom
            e["valu_issue_util"] = 4.0 * cs.get("SQ_INSTS_VALU", 0.0) / denom
        doc[key][short(k)] = e
json.dump(doc, open(os.path.join(out, "mfma_busy.json"), "w"), indent=1)
print(sorted(os.listdir(out)))
