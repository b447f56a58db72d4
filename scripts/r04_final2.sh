#!/bin/bash
# the exact final tree: whole GPU suite + the driver's bench command
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r04z_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04z_bench_stdout.log 2>$OUT/r04z_bench_stderr.log; tail -n 1 $OUT/r04z_bench_stdout.log > $OUT/r04z_bench.log; tail -c 200 $OUT/r04z_bench.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
