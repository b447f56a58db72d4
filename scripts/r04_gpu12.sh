#!/bin/bash
# round 4, GPU call M: narrow / wide slot tiles of the reducer (parity + A/B against the previous build), order of RCCL's banner vs the JSON line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
C=$R/evergreen_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sparse_keys.py tests/test_pool_delta.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r04m_pytest.log
for rep in 1 2; do for l in libevg_ref.so libevg_sched.so; do
  echo "$l" | tee -a $OUT/r04m_ab.log
  EVG_SCHED_LIB=$C/$l timeout 300 python scripts/ab_tiled.py c5 2>&1 | tail -1 | tee -a $OUT/r04m_ab.log
  EVG_SCHED_LIB=$C/$l timeout 300 python scripts/ab_tiled.py skew 2>&1 | tail -1 | tee -a $OUT/r04m_ab.log
done; done
for l in libevg_ref.so libevg_sched.so; do
  EVG_SCHED_LIB=$C/$l timeout 600 python scripts/bench_config5.py 10000000 512 --steps 10 2>&1 | tail -1 | tee -a $OUT/r04m_ab.log
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-config5 --no-cpu-baseline > $OUT/r04m_bench_stdout.log 2>/dev/null; tail -n 3 $OUT/r04m_bench_stdout.log | cut -c1-120
