"""HBM traffic of the dominant kernel from the two PMC passes of scripts/profile_bench.sh.

    python scripts/traffic_json.py gpurun_out/prof_<tag> edge_rgather profiles/rgather_traffic.json

Corrections follow MI355X_MICROARCH.md (section HBM): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; on gfx950
FETCH_SIZE counts half of the fetched bytes (doubled here); WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import sys


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
    root, pat, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = per_kernel(root + "/fetch", "FETCH_SIZE"), per_kernel(root + "/write", "WRITE_SIZE")
    name = [k for k in fetch if pat in k][0]
    short = lambda k: k.replace("void dgcnn::", "").replace("dgcnn::", "").split("(")[0][:48]
    doc = {
        "kernel": short(name),
        "command": "scripts/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, WRITE_SIZE) "
                   "-- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline",
        "FETCH_SIZE_KB_per_launch": fetch[name],
        "WRITE_SIZE_KB_per_launch": write.get(name, 0.0),
        "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports half of the fetched bytes (doubled "
                      "here; this kernel reads dword-per-lane rows, not the calibrated 16-B pattern, so the read side "
                      "is approximate); WRITE_SIZE as reported",
        "hbm_bytes_per_launch": 2.0 * fetch[name] * 1024.0 + write.get(name, 0.0) * 1024.0,
        "all_kernels_KB_per_launch": {short(k): {"FETCH_SIZE": fetch[k], "WRITE_SIZE": write.get(k, 0.0)}
                                      for k in sorted(fetch, key=lambda k: -fetch[k])[:16]},
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("kernel", "hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main()
