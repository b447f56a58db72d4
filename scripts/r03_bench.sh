#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R
TAG=${1:-r03f}
(time timeout 1200 python bench.py --steps 50 --warmup 5) > $OUT/bench_$TAG.log 2>&1
tail -c 20000 $OUT/bench_$TAG.log | python -c "
import sys, json
txt = sys.stdin.read()
line = [l for l in txt.splitlines() if l.startswith('{')]
if not line: print(txt[-3000:]); sys.exit()
j = json.loads(line[-1])
def show(k, v, ind=0):
    if isinstance(v, dict):
        print(' ' * ind + k + ':')
        for kk, vv in v.items(): show(kk, vv, ind + 2)
    else:
        s = str(v)
        print(' ' * ind + '%s: %s' % (k, s[:150]))
for k, v in j.items(): show(k, v)
"
grep real $OUT/bench_$TAG.log
