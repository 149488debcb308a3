"""per-step kernel table of a rocprofv3 kernel_stats.csv of the training step (steps = calls of adam_kernel), optionally
against a second file scaled by a factor:  python scripts/step_table.py a.csv [b.csv scale]"""
import csv
import re
import sys


def load(path):
    rows = list(csv.reader(open(path)))[1:]
    steps = [int(x[1]) for x in rows if "adam_kernel" in x[0]]
    steps = steps[0] if steps else 1
    d, calls = {}, {}
    for x in rows:
        name = x[0]
        if name.startswith("_ZN5dgcnn"):
            short = re.sub(r"_ZN5dgcnn\d+", "", name)[:34]
        else:
            m = re.search(r"(?:dgcnn::(?:hf::)?)?(\w+)(<[^>]*>)?\(", name)
            short = (m.group(1) + (m.group(2) or "")) if m else name[:40]
            if "at::native" in name or "Cijk" in name or "rocclr" in name:
                short = "T:" + name[:60]
        d[short] = d.get(short, 0) + int(x[2]) / steps / 1000
        calls[short] = calls.get(short, 0) + int(x[1]) / steps
    return d, calls, steps


a, ca, sa = load(sys.argv[1])
b = None
if len(sys.argv) > 3:
    b, cb, sb = load(sys.argv[2])
    scale = float(sys.argv[3])
print("steps %d  kernel us/step %.1f  launches/step %.1f" % (sa, sum(a.values()), sum(ca.values())))
for k in sorted(a, key=lambda k: -a[k]):
    if b is not None:
        print("%8.1f %5.1f  ref %8.1f  %s" % (a[k], ca[k], b.get(k, 0) * scale, k))
    else:
        print("%8.1f %5.1f  %s" % (a[k], ca[k], k))
