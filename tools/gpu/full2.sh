export TMPDIR=/tmp
O=gpurun_out/full2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for cfg in A B C D; do timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg > $O/b_$cfg.log 2>&1; echo "$cfg $(tail -1 $O/b_$cfg.log | cut -c40-70)"; done
for ag in 1024 2048 8192; do timeout -k 10 200 python bench.py --no-cpu-baseline --agents $ag > $O/b_$ag.log 2>&1; echo "$ag $(tail -1 $O/b_$ag.log | cut -c40-70)"; done
