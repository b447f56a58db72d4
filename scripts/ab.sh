#!/bin/bash
# A/B of two builds of the library on the same box: libevg_sched_base.so (A) against libevg_sched.so (B), interleaved.
R=$GRAFT_REPO_ROOT; cd $R
run() { EVG_SCHED_LIB=$1 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras --in-flight ${3:-1} | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$2', round(d['ms_per_step']*1e3,2), 'us/step  kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us', ' pipelined', round(d.get('pipelined',{}).get('ms_per_step',0)*1e3,2))"; }
for i in 1 2 3; do
  run $R/evergreen_amd/csrc/libevg_sched_base.so A $1
  run $R/evergreen_amd/csrc/libevg_sched.so B $1
done
