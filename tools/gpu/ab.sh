#!/bin/bash
# A/B of library builds on the GPU box: tools/gpu/ab.sh OUTDIR lib1 lib2 ...   (names under azdopt_amd/, e.g. libazdopt_amd_base.so)
# per library: the default bench line (whole epochs, config B) and the driver's 20-call window; *prof* libraries run tools/pool_cycle.py instead
O=gpurun_out/$1; shift
mkdir -p $O
for lib in "$@"; do
  export AZD_LIB=$PWD/azdopt_amd/$lib
  if [[ $lib == *prof* ]]; then
    timeout -k 10 200 python tools/pool_cycle.py 4096 800 > $O/cycle_$lib.txt 2>&1
  else
    timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_$lib.log 2>&1
    timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench20_$lib.log 2>&1
  fi
  echo "$lib done"
done
unset AZD_LIB
