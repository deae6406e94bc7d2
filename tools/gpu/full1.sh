export TMPDIR=/tmp
O=gpurun_out/full1
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for cfg in B C D; do timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg > $O/b_$cfg.log 2>&1; tail -1 $O/b_$cfg.log | cut -c1-120; done
timeout -k 10 200 python bench.py --no-cpu-baseline --agents 8192 > $O/b_8192.log 2>&1; tail -1 $O/b_8192.log | cut -c1-120
