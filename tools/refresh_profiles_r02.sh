# Round-2 profile refresh (run on the GPU box through gpurun; tools/update_profiles_r02.py copies the summaries to profiles/)
export TMPDIR=/tmp
O=gpurun_out/final_r02
rm -rf $O
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-160
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver20.log 2>&1
timeout -k 10 200 python bench.py --no-cpu-baseline --step async > $O/bench_B_async.log 2>&1
for cfg in A C D; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > $O/bench_$cfg.log 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg --step async > $O/bench_${cfg}_async.log 2>&1
done
timeout -k 10 300 python bench.py --no-cpu-baseline --agents 8192 > $O/bench_B8192.log 2>&1
timeout -k 10 300 python bench.py --no-cpu-baseline --agents 8192 --step async > $O/bench_B8192_async.log 2>&1
timeout -k 10 400 python bench.py --config E --steps 400 --warmup 50 > $O/bench_E.log 2>&1
echo benches done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_E -- python3 bench.py --no-cpu-baseline --config E --steps 200 --warmup 20 > $O/prof_E.log 2>&1
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline > $O/pmc_write.log 2>&1
echo tcc done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 bench.py --no-cpu-baseline --steps 800 --warmup 800 > $O/pmc_sq1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 bench.py --no-cpu-baseline --steps 800 --warmup 800 > $O/pmc_sq2.log 2>&1
echo sq done
{
for ev in 16 48 88; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 120 python tools/pool_probe.py 4096 400; done
AZD_POOL_EVAL_WGS=80 timeout -k 10 120 python tools/pool_probe.py 8192 400
AZD_POOL_EVAL_WGS=56 timeout -k 10 120 python tools/pool_probe.py 8192 400 bf16
AZD_STEP_FORM=async timeout -k 10 120 python tools/pool_probe.py 4096 400
for B in 4096 8192 16384; do timeout -k 10 120 python tools/pool_probe_hash.py $B 400; AZD_STEP_FORM=async timeout -k 10 120 python tools/pool_probe_hash.py $B 400; done
} > $O/pool_probe.txt 2>&1
{
for dt in bf16 f32; do
timeout -k 10 100 python tools/time_gemm.py $dt 8192
timeout -k 10 100 python tools/time_gemm.py $dt 65536 304,256,256,256,152
timeout -k 10 100 python tools/time_gemm.py $dt 8192 4096,4096,4096
done
} > $O/gemm.txt 2>&1
echo all done
