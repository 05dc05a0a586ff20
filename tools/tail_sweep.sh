#!/bin/bash
# A/B of the last-round work-unit size of the drone row kernel (RATO_TAIL_SPLIT x RATO_TAIL_PCT), same box, alternating
for rep in 1 2; do
for cfg in "1 100" "2 100" "4 100" "4 60" "4 150" "8 100" "8 60"; do
  set -- $cfg
  for jac in products factored; do
    RATO_TAIL_SPLIT=$1 RATO_TAIL_PCT=$2 timeout 200 python bench.py --jacobian $jac --no-cpu-baseline --no-scp --steps 50 --warmup 5 ${EXTRA} 2>/dev/null | python tools/pline.py "split=$1 pct=$2 $jac"
  done
done
done
