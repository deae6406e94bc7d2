export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
export AZD_LIB=azdopt_amd/libazdopt_amd_poll96.so
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-120
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline > $O/pmc_fetch.log 2>&1
python - <<'PY'
import csv, glob
t=n=0
for f in glob.glob("gpurun_out/r2n/pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE" and "k_pool" in r["Kernel_Name"]:
            t+=float(r["Counter_Value"]); n+=1
print("poll sleep 96: FETCH per call %.2f MB over %d launches" % (t*1024/(800*n)/1e6, n))
PY
