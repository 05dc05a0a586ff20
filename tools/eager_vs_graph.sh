#!/bin/bash
# the small configurations as eager steps (--graph off) against replayed ones (--graph on), same board
for cfg in "C2 eval" "C3 eval" "C3 linearize" "C4 linearize" "C2 linearize"; do
  set -- $cfg
  for g in on off; do
    python bench.py --config $1 --mode $2 --graph $g --steps 500 --warmup 20 --no-scp 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1 $2 graph=$g  ms_per_step %.4f  kernel_ms %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
"
  done
done
