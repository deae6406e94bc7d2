#!/bin/bash
# tools/gpu/ab3.sh libA libB [reps]: alternating whole-epoch bench runs of two library builds on the same box
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do for lib in $A $B; do
  AZD_LIB=$PWD/azdopt_amd/$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', round(d['value']/1e6,2), round(d['roofline']['avg_launch_ms'],2), d.get('pool_split'))"
done; done
