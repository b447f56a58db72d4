#!/bin/bash
# merge-path rounds in the large-distro tile sort and merge passes: parity, then A/B (EVG_TILED_MODE=64 = the networks)
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5_share or skew or large or tiled or hint or random or ragged" 2>&1 | tail -3
for i in 1 2; do
  python scripts/ab_tiled.py c5 | sed 's/^/merge path /'
  EVG_TILED_MODE=64 python scripts/ab_tiled.py c5 | sed 's/^/networks   /'
done
python scripts/ab_tiled.py skew | sed 's/^/merge path /'
EVG_TILED_MODE=64 python scripts/ab_tiled.py skew | sed 's/^/networks   /'
