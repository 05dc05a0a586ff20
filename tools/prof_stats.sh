#!/bin/bash
# kernel-trace stats of config C4 (eager) with the one-launch and the five-launch selection
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_stats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
for v in default multi; do
  if [ $v = multi ]; then export RATO_RS_PATH=multi; else unset RATO_RS_PATH; fi
  for c in C4 metric; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${c}_$v -- $B --config $c --jacobian products --graph off --no-cpu-baseline --no-scp --steps 50 --warmup 5 > /dev/null 2>&1
  f=$(find $O/kt_${c}_$v -name "*kernel_stats.csv" | head -1); echo "== $c $v"; cut -d, -f1-4 $f | head -12
  done
done
find $O -name "*_kernel_trace.csv" -delete
