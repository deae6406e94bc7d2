export TMPDIR=/tmp
O=gpurun_out/final_r02
mkdir -p $O
rm -rf $O/pmc_fetch $O/pmc_write
timeout -k 10 300 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-120
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline > $O/pmc_write.log 2>&1
echo done
