export TMPDIR=/tmp
O=gpurun_out/final_r02
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof.log 2>&1
ls -t $O/prof/*/*kernel_stats.csv | head -1 | xargs grep k_pool | cut -c150-260
