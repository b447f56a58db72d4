#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export PYTHONPATH=$R
echo "== full GPU suite"
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^  File\|Extension modules" | tail -25
