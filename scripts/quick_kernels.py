"""dev helper: run bench.py (block workload, no CPU baseline / train leg) under rocprofv3 --stats and print the per-kernel
averages; used by the ablation runs recorded in DESIGN.md.  usage: quick_kernels.py TAG [bench args...]"""
import csv, glob, os, subprocess, sys
tag = sys.argv[1]
out = "gpurun_out/qk_" + tag
subprocess.run("rm -rf %s; mkdir -p %s" % (out, out), shell=True)
cmd = ["rocprofv3", "--kernel-trace", "--stats", "-d", out, "--output-format", "csv", "--", sys.executable, "bench.py",
       "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-train-leg"] + sys.argv[2:]
env = dict(os.environ, TMPDIR="/tmp")
with open(out + "/log.txt", "w") as f:
    subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env)
fs = sorted(glob.glob(out + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
tot = 0.0
print("== %s" % tag)
for r in rows:
    calls = int(r["Calls"])
    if calls < 20:
        continue
    us = float(r["AverageNs"]) / 1e3
    tot += us * calls / 43.0
    if us >= 3.0:
        print("   %-58s n %4d avg_us %8.2f" % (r["Name"].replace("void ", "").replace("dgcnn::", "")[:58], calls, us))
print("   kernel time per step ~ %.1f us, launches per step ~ %.1f" % (tot, sum(int(r["Calls"]) for r in rows if int(r["Calls"]) >= 20) / 43.0))
for line in open(out + "/log.txt"):
    if line.startswith("{"):
        import json
        d = json.loads(line)
        print("   ms_per_step %.4f median %.4f" % (d["ms_per_step"], d["ms_per_step_median"]))
