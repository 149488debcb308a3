"""HBM traffic of the profiled kernels from the two PMC passes of scripts/profile_bench.sh.

    python scripts/traffic_json.py gpurun_out/prof_<tag> profiles/hbm_traffic.json

Corrections follow MI355X_MICROARCH.md (section HBM): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; on gfx950
FETCH_SIZE counts half of the fetched bytes (doubled here; calibrated for 16-byte-per-lane streams, so the read side
of the dword-per-lane gathers is approximate); WRITE_SIZE is taken as is.  A region of bench.py's `kernels` list may
span two launches (listed below): their per-launch means are added.
"""
import collections
import csv
import glob
import json
import sys

REGIONS = {
    "knn_bound16s_kernel": ["knn_bound16s_kernel", "knn_bound_merge_kernel"],
    "knn_filter16_kernel": ["knn_filter16_kernel"],
    "knn_refine_kernel": ["knn_refine_kernel"],
    "edge_reduce_kernel": ["edge_reduce_kernel"],
    "edge_rgather_kernel": ["edge_rgather_kernel"],   # both launches (segment 0 of every point, extra segments)
}


def per_kernel(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
        for (_, k), v in per.items():
            acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(root + "/fetch", "FETCH_SIZE"), per_kernel(root + "/write", "WRITE_SIZE")
    doc = {
        "command": "scripts/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) "
                   "-- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline",
        "correction": "hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024 (MI355X_MICROARCH.md section HBM)",
        "hbm_bytes_per_launch": {}, "FETCH_SIZE_KB": {}, "WRITE_SIZE_KB": {},
    }
    for region, pats in REGIONS.items():
        f = sum(v for k, v in fetch.items() if any(p in k for p in pats))
        w = sum(v for k, v in write.items() if any(p in k for p in pats))
        doc["FETCH_SIZE_KB"][region], doc["WRITE_SIZE_KB"][region] = f, w
        doc["hbm_bytes_per_launch"][region] = 2.0 * f * 1024.0 + w * 1024.0
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
