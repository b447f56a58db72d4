#!/bin/bash
# every EVG_TILED_MODE variant through the large-distro parity tests (they are A/B switches, all bit-exact)
R=$GRAFT_REPO_ROOT; cd $R
for m in 1 2 4 8 16 32 64 127; do
  echo "mode $m: $(EVG_TILED_MODE=$m timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k 'config5_share or skew or large or tiled or hint or random' 2>&1 | tail -1)"
done
