"""dev helper: top kernels of the newest rocprofv3 kernel_stats.csv under a directory."""
import csv, glob, os, sys
fs = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in list(csv.DictReader(open(fs[-1])))[:n]:
    print("   %-60s n %4s avg_us %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
