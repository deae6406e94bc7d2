export TMPDIR=/tmp
O=gpurun_out/final_check
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 400 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench20.log 2>&1; tail -1 $O/bench20.log | cut -c1-200
