#!/bin/bash
# A/B: one tile per workgroup (default) vs balanced grid (RATO_ROWS_BALANCED=1), same box, alternating
for rep in 1 2 3; do
for bal in 0 1; do
  for jac in products factored; do
    RATO_ROWS_PERSISTENT=0 RATO_ROWS_BALANCED=$bal timeout 200 python bench.py --jacobian $jac --no-cpu-baseline --no-scp --steps 50 --warmup 5 2>/dev/null | python tools/pline.py "balanced=$bal $jac"
  done
done
done
