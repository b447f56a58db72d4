#!/bin/bash
# round 5, second GPU call: the suite file by file under timeouts (the first call lost 15 minutes to one hanging test), the bench line
# (batcher numbers in per_distro_calls), config 5 at FULL size per kernel for base / s1 / the tree with 20- and with 24-byte keys
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT; cd $R; export PYTHONPATH=$R
TAG=${1:-r05b}
bash scripts/gpu_suite.sh $TAG 150
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/${TAG}_bench.err | tail -n 1 > $OUT/${TAG}_bench.log; tail -c 400 $OUT/${TAG}_bench.log; tail -3 $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
for v in base s1 sched sched:64; do
  l=${v%%:*}; mode=${v##*:}; [ "$mode" = "$v" ] && mode=0
  rm -rf /tmp/kf
  EVG_TILED_MODE=$mode EVG_SCHED_LIB=$R/evergreen_amd/csrc/libevg_$l.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kf -o kf -- \
    python $R/scripts/bench_config5.py 10000000 512 --steps 6 > /tmp/kf.log 2>&1
  grep -E "config-5|ms" /tmp/kf.log | tail -2
  f=$(find /tmp/kf -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${TAG}_full_${l}_${mode}_kernel_stats.csv && python - "$f" "$v" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "evg::" in r["Name"]]
print("%-10s" % sys.argv[2], " ".join("%s %.1f" % (r["Name"].split("(")[0].replace("void evg::", "").replace("evg::", ""), float(r["AverageNs"]) / 1e3) for r in rows))
PY
done 2>&1 | tee $OUT/${TAG}_full_kstats.log
