#!/bin/bash
# A/B: coalesced key loads / stores in the merge passes and the tile sort (base = the previous build; mode 32 = own-keys stores)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for i in 1 2; do
  EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_sched_base.so python scripts/ab_tiled.py c5 | sed 's/^/base      /'
  python scripts/ab_tiled.py c5 | sed 's/^/new       /'
  EVG_TILED_MODE=32 python scripts/ab_tiled.py c5 | sed 's/^/new,mode32 /'
done
python scripts/ab_tiled.py skew | sed 's/^/new       /'
EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_sched_base.so python scripts/ab_tiled.py skew | sed 's/^/base      /'
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or skew or large or tiled or hint or random" 2>&1 | tail -3
