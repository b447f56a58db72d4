#!/bin/bash
# round 6: the fused tick with the delta's re-pack enqueued before the updates are checked / staged (status block page-locked), fewer
# launches in the re-pack, `distinct` behind the enqueue, and without wait_ns -- against the round's last commit (libevg_head.so,
# scripts/mkhead.sh) on ONE box; the delta / multi-device / deadline tests; short soaks. Everything under a timeout of its own.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
{
timeout -k 5 400 python -u -m pytest tests/test_pool_delta.py tests/test_gpu_multi_abi.py tests/test_deadlines.py -x -q -m gpu --timeout 90 2>&1 | grep -v amdgpu.ids | tail -6
for rep in 1 2; do for v in head sched; do
  echo "=== bench object, lib $v"; EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$v.so timeout -k 5 300 python scripts/bench_delta.py 3 2>&1 | grep -v amdgpu.ids | tail -3 | python -c "
import sys, json
for l in sys.stdin:
    try: o = json.loads(l)
    except Exception: print(l.rstrip()); continue
    f = o.get('fused') or {}
    print('three calls %.3f ms (delta %.3f update %.3f plan %.3f)  fused %.3f  fused lean %s  in place %s  same %s %s' % (o.get('three_calls_ms_per_tick', o['ms_per_tick']), o['apply_delta_ms'], o['update_ms'], o['plan_and_download_ms'], f.get('ms_per_tick', 0), f.get('ms_per_tick_without_wait_ns'), o.get('ms_per_tick_without_wait_ns_in_place'), o['identical_to_full_upload'], f.get('identical_to_full_upload')))
"
done; done
echo "=== laps (sched)"; EVG_TICK_TIMING=1 timeout -k 5 300 python scripts/bench_delta.py 1 2>&1 | grep "^\[tick\]" | tail -14
timeout -k 5 200 python scripts/soak_delta.py 60 91 fused 2>&1 | tail -1
timeout -k 5 200 python scripts/soak_multi_delta.py 60 92 2>&1 | tail -1
} > $OUT/r06i_tick.log 2>&1
cat $OUT/r06i_tick.log
