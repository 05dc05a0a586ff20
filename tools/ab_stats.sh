#!/bin/bash
# Same-box alternating A/B of the selection launch structure inside the whole bench step.
# usage: tools/ab_stats.sh CONFIG [reps]
cfg=${1:-C4}; reps=${2:-4}
for i in $(seq $reps); do
  for v in default multi; do
    if [ $v = multi ]; then export RATO_RS_PATH=multi; else unset RATO_RS_PATH; fi
    python bench.py --config $cfg --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['config']['baseline_config'], 'step_ms %.5f' % d['ms_per_step'], 'kernel_ms %.5f' % d['roofline']['kernel_ms'])"
  done
done
