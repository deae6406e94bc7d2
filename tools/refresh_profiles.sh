#!/bin/bash
# Profile refresh of a round (run on the GPU box through gpurun; tools/update_profiles.py copies the summaries to profiles/):
#   tools/refresh_profiles.sh r05 [part ...]     parts: bench stats tcc tcc20 tccx l2 sq insts phases gemm split curve examples  (default: all)
export TMPDIR=/tmp
R=${1:-r05}; shift
PARTS=${*:-bench stats tcc tcc20 tccx l2 sq insts phases gemm split curve examples}
O=gpurun_out/final_$R
mkdir -p $O
has() { [[ " $PARTS " == *" $1 "* ]]; }
# the diagnostic libraries of the insts / phases parts are built on the build machine and travel with the snapshot:
#   make -C azdopt_amd/csrc -j8 PROFILE=1                                        -> azdopt_amd/libazdopt_amd_prof.so
#   make -C azdopt_amd/csrc -j8 VARIANT=phases EXTRA=-DAZD_WAVE_PHASES=1         -> azdopt_amd/libazdopt_amd_phases.so
for part in insts phases; do
  if has $part && { [ ! -f azdopt_amd/libazdopt_amd_prof.so ] || { [ $part = phases ] && [ ! -f azdopt_amd/libazdopt_amd_phases.so ]; }; }; then
    echo "refresh_profiles: part '$part' needs the diagnostic libraries (see the head of this script): skipped"; PARTS=${PARTS//$part/}
  fi
done
if has bench; then
  timeout -k 10 300 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-140
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver20.log 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline --step async > $O/bench_B_async.log 2>&1
  for cfg in A C D; do timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg > $O/bench_$cfg.log 2>&1; done
  AZD_POOL_EVAL_GROUP=0 timeout -k 10 300 python bench.py --no-cpu-baseline --config A > $O/bench_A_classic.log 2>&1
  timeout -k 10 300 python bench.py --no-cpu-baseline --agents 8192 > $O/bench_B8192.log 2>&1
  timeout -k 10 400 python bench.py --config E --steps 400 --warmup 50 > $O/bench_E.log 2>&1
  AZD_DENSE_NO_POOL=1 timeout -k 10 400 python bench.py --config E --steps 400 --warmup 50 --no-cpu-baseline > $O/bench_E_per_call.log 2>&1
  timeout -k 10 400 python bench.py --config E --no-cpu-baseline > $O/bench_E_epochs.log 2>&1
  timeout -k 10 400 python bench.py --config E612 --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_E612.log 2>&1
  timeout -k 10 400 python bench.py --config E612 --no-cpu-baseline > $O/bench_E612_epochs.log 2>&1
  echo benches done
fi
if has stats; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/prof.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof20 -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/prof20.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_E -- python3 bench.py --no-cpu-baseline --config E --steps 200 --warmup 20 > $O/prof_E.log 2>&1
  echo stats done
fi
if has tcc; then
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline > $O/pmc_fetch.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline > $O/pmc_write.log 2>&1
  echo tcc done
fi
if has tcc20; then  # the same two counters over the DRIVER's window: launches of 20 calls (bench.py --steps 20 --warmup 5)
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch20 -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/pmc_fetch20.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write20 -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/pmc_write20.log 2>&1
  echo tcc20 done
fi
if has tccx; then  # the same two counters for the other BASELINE configurations' dominant kernels (bench.py --config A / C / D / E: whole epochs)
  for cfg in A C D; do  # (E: the counter mode serialises dispatches; the dense pool step is two concurrent kernels)
    timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$cfg -- python3 bench.py --no-cpu-baseline --config $cfg > $O/pmc_fetch_$cfg.log 2>&1
    timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$cfg -- python3 bench.py --no-cpu-baseline --config $cfg > $O/pmc_write_$cfg.log 2>&1
  done
  echo tccx done
fi
if has l2; then  # where the tree's records are served from: L2 hits / misses of k_pool (all XCDs summed), one pass
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_l2 -- python3 bench.py --no-cpu-baseline --steps 800 --warmup 800 > $O/pmc_l2.log 2>&1
  echo l2 done
fi
if has sq; then
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 bench.py --no-cpu-baseline --steps 800 --warmup 800 > $O/pmc_sq1.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 bench.py --no-cpu-baseline --steps 800 --warmup 800 > $O/pmc_sq2.log 2>&1
  echo sq done
fi
if has insts; then  # the instruction table (profiles/<round>_pool_insts.txt): phase stamps of the diagnostic build + SQ_INSTS_* per kernel of the pool step and of the launch-per-phase form
  { echo "## phase stamps, diagnostic build (tools/pool_cycle.py; make PROFILE=1)"; AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 4096 800
    for form in pool per_call; do n=400; [ $form = per_call ] && n=100
      echo "## rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 -- python3 tools/inst_run.py $form 4096 $n"
      timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/pmc_i_$form -- python3 tools/inst_run.py $form 4096 $n 2>/dev/null | grep -E "^form|^EXPANSIONS"
      python tools/pmc_kernels.py $O/pmc_i_$form | grep -E "k_pool|k_rollout|k_add_actions|k_gemm|k_argmin" ; rm -rf $O/pmc_i_$form; done; } > $O/insts.txt 2>&1
  echo insts done
fi
if has phases; then  # a searcher wave's time by phase at the product build's rate (AZD_WAVE_PHASES build) and the driver's window by agent (PROFILE build)
  { for cfg in "4096 800 f32" "8192 800 bf16" "8192 800 f32"; do AZD_LIB=azdopt_amd/libazdopt_amd_phases.so timeout -k 10 200 python tools/wave_phases.py $cfg; done; } > $O/wave_phases.txt 2>&1
  AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/window_tail.py 4096 20 390 > $O/window_tail.txt 2>&1
  echo phases done
fi
if has gemm; then
  { timeout -k 10 200 python tools/time_gemm16.py; echo "# zero-filled operands (they read higher; for comparison with figures quoted that way):"; AZD_GEMM_ZEROS=1 timeout -k 10 200 python tools/time_gemm16.py 8192 4096 4096; for dt in bf16 f32; do timeout -k 10 100 python tools/time_gemm.py $dt 8192; timeout -k 10 100 python tools/time_gemm.py $dt 65536 304,256,256,256,152; timeout -k 10 100 python tools/time_gemm.py $dt 8192 4096,4096,4096; done; AZD_GEMM_OLD=1 timeout -k 10 100 python tools/time_gemm.py bf16 8192;
    echo "# config E's whole forward, hidden layers fused (k_hidden2_fused) and layer by layer:"; timeout -k 10 100 python tools/time_forward16.py; AZD_MLP_FUSE_HIDDEN=0 timeout -k 10 100 python tools/time_forward16.py; } > $O/gemm.txt 2>&1
  echo gemm done
fi
if has split; then
  { timeout -k 10 300 python tools/split_util.py B 64 76 88 100 112 124; timeout -k 10 300 python tools/split_util.py X 64 76 88 100 112 124; timeout -k 10 200 python tools/split_util.py C 28 34 40 52 64
    timeout -k 10 200 python tools/split_util.py D 100 106 112 118 126; timeout -k 10 200 python tools/split_util.py B8 64 72 80 96; timeout -k 10 200 python tools/split_util.py A 33 64 129
    for c in B X C D B8 A; do timeout -k 10 300 python tools/split_feedback.py $c 10; done; } > $O/pool_split.txt 2>&1
  echo split done
fi
if has curve; then
  { timeout -k 10 300 python tools/launch_curve.py B 1 5 20 800 w1 w5;     timeout -k 10 300 python tools/launch_curve.py A 1 5 20 800 w1; } > $O/curve.txt 2>&1
  echo curve done
fi
if has examples; then
  g++ -O2 -std=c++17 -Iinclude examples/c21_tree.cpp -o $O/c21_tree -Lazdopt_amd -lazdopt_amd -Wl,-rpath,$PWD/azdopt_amd
  TIMEFORMAT="%R s wall"
  { for stride in 1 800; do echo "512 agents, 512-1024-512, stride $stride:"; time $O/c21_tree 3 800 512 $stride 0 2>&1 | tail -1; done
    for stride in 1 800; do echo "4096 agents, 3 x 256, stride $stride:"; time $O/c21_tree 3 800 4096 $stride 0 256 256 256 2>&1 | tail -1; done; } > $O/examples.txt 2>&1
  echo examples done
fi
echo all done
