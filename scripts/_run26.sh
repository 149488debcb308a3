mkdir -p gpurun_out/r4
export TMPDIR=/tmp
timeout -k 10 500 python scripts/fuzz_edgeconv.py 4 100 > gpurun_out/r4/fuzz26.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r4/fuzz26.log
bash scripts/profile_round.sh r4_99 > gpurun_out/r4/profround.log 2>&1; echo "prof rc=$?"
echo done
