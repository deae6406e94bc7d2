export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
for lib in libazdopt_amd.so libazdopt_amd_late.so; do
for ag in 512 1024 2048 4096; do
AZD_LIB=azdopt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --agents $ag > $O/e_$ag.log 2>&1; echo "$lib $ag $(tail -1 $O/e_$ag.log | cut -c40-70)"
done
AZD_LIB=azdopt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --config A > $O/e_A.log 2>&1; echo "$lib A $(tail -1 $O/e_A.log | cut -c40-70)"
AZD_LIB=azdopt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --config C > $O/e_C.log 2>&1; echo "$lib C $(tail -1 $O/e_C.log | cut -c40-70)"
AZD_LIB=azdopt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --config D > $O/e_D.log 2>&1; echo "$lib D $(tail -1 $O/e_D.log | cut -c40-70)"
AZD_LIB=azdopt_amd/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/e_20.log 2>&1; echo "$lib 20 $(tail -1 $O/e_20.log | cut -c40-70)"
done
