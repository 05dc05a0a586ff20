#!/bin/bash
# A/B: one tile per workgroup vs dynamic tile queue vs static balanced grid, same box, alternating.  args: extra bench flags
for rep in 1 2 3; do
for cfg in "0 0" "1 0" "0 1"; do
  set -- $cfg
  for jac in products factored; do
    RATO_ROWS_DYNAMIC=$1 RATO_ROWS_BALANCED=$2 timeout 200 python bench.py --jacobian $jac --no-cpu-baseline --no-scp --steps 50 --warmup 5 ${EXTRA} 2>/dev/null | python tools/pline.py "dynamic=$1 balanced=$2 $jac"
  done
done
done
