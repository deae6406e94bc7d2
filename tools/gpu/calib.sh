export TMPDIR=/tmp
mkdir -p gpurun_out/final_r02
cd /tmp && rm -rf calib && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/calib -- $GRAFT_REPO_ROOT/tools/probes/gather_calib > /tmp/calib.log 2>&1
cd $GRAFT_REPO_ROOT
cat /tmp/calib.log | tail -3
python - <<'PY'
import csv, glob
rows = {}
for f in glob.glob("/tmp/calib/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows[r["Kernel_Name"].split("(")[0]] = float(r["Counter_Value"])
out = open("gpurun_out/final_r02/gather_calib.txt", "w")
for k, v in rows.items():
    line = "%-12s FETCH_SIZE %.0f KB = %.3f x the 1 GiB requested" % (k, v, v * 1024 / (1 << 30))
    print(line); out.write(line + "\n")
PY
