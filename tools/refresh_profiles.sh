set -e
export TMPDIR=/tmp
O=gpurun_out/final2
rm -rf $O
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
timeout -k 10 200 python bench.py --no-cpu-baseline --barrier-step > $O/bench_barrier.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --mlp-dtype bf16 > $O/bench_bf16.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --workload r333 > $O/bench_r333.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --workload r333 --mlp-dtype bf16 > $O/bench_r333_bf16.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --workload r44 > $O/bench_r44.log 2>&1
echo benches done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline > $O/pmc_write.log 2>&1
echo pmc done
timeout -k 10 200 python tools/phase_profile.py 4096 800 mlp async > $O/phase_async.txt 2>&1
timeout -k 10 200 python tools/phase_profile.py 4096 800 mlp async bf16 > $O/phase_async_bf16.txt 2>&1
timeout -k 10 200 python tools/agent_balance.py async > $O/balance_async.txt 2>&1
timeout -k 10 200 python tools/slow_agents.py > $O/slow_agents.txt 2>&1
timeout -k 10 120 ./tools/probes/tile_probe > $O/tile_probe.txt 2>&1
echo all done
