#!/bin/bash
# round 4, GPU call I: delta + multi + pool tests on the rebuilt library, the bench line (native-thread per_distro_calls, delta tick),
# per-phase counters of the headline kernel (ablation builds)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
timeout 600 python -m pytest tests/test_pool_delta.py tests/test_sparse_keys.py tests/test_gpu_multi_abi.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r04i_pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-config5 2>&1 | grep '^{"metric"' > $OUT/r04i_bench.log; tail -c 300 $OUT/r04i_bench.log
timeout 1200 bash scripts/ablate_valu.sh 2>&1 | tee $OUT/r04i_phase_counters.txt
