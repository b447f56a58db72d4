#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for m in 0 128 256 0 128 256; do EVG_TILED_MODE=$m python scripts/ab_tiled.py c5 | sed "s/^/mode $m  /"; done
