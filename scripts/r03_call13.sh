#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or config3 or alloc or fuzz or config5_share" 2>&1 | tail -2
bash scripts/ab.sh 2>&1 | grep -E "^A|^B"
bash scripts/kstats.sh 2>&1 | grep allocate
