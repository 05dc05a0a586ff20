#!/bin/bash
# Same-board alternating A/B of one environment knob on the drone SCP at the bench's size:
#   tools/ab_env.sh RATO_CUT_SIDE_STREAM 1 0        (three repeats of each value)
# prints cumulative / oracle / master / define seconds per run.
var=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do
  for val in "$@"; do
    env $var=$val python scp_bench.py --system drone --reduced --M 100000 --S 50 --iters 60 --seed 7 --define-only-M 0 2>/dev/null \
      | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print('$var=$val rep=$rep cumulative %.4f oracle %.4f master %.4f define %.4f cuts %d L2 %.2e' % (d['cumulative_s'], d['oracle_total_s'], d['master_total_s'], d['define_total_s'], d['cuts_total'], d['L2_error_last']))
"
  done
done | tee gpurun_out/ab_$var.txt
