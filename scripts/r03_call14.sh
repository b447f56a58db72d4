#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config5_full" 2>&1 | tail -2
bash scripts/ab.sh 2>&1 | grep -E "^A|^B"
