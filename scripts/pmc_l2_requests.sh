# L1 -> L2 request counts per kernel of the headline workload (one PMC pass, --kernel-trace only): TCP_TCC_READ_REQ, TCC_REQ / HIT / MISS
# in millions per launch.  usage: bash scripts/pmc_l2_requests.sh
export TMPDIR=/tmp
OUT=gpurun_out/pmc_l2
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "TCP_TCC_READ_REQ_sum\|TCC_REQ_sum\|TCC_HIT_sum\|TCC_MISS_sum\|TCP_TOTAL_CACHE_ACCESSES_sum\|TCP_TCC_READ_REQ_LATENCY_sum" | sort -u > $OUT/avail.txt
cat $OUT/avail.txt
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/a --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train-leg --no-model-legs --no-graph-leg > $OUT/a.log 2>&1
python3 - <<'PY'
import csv,glob,collections
fs=glob.glob('gpurun_out/pmc_l2/a/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='TCC_REQ_sum': cnt[k]+=1
for k,v in acc.items():
    n=max(cnt[k],1)
    print(k, {c: round(x/n/1e6,3) for c,x in v.items()}, "launches",n)
PY
