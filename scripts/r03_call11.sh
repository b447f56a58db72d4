#!/bin/bash
# A/B of the batched dep-finished gather (phase G) + the parity tests that cover it
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3 or golden or fuzz or ragged or extreme or ten_dep" 2>&1 | tail -4
bash scripts/ab.sh 2>&1 | tail -6
