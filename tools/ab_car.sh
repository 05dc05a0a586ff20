#!/bin/bash
for rep in 1 2 3; do for dyn in 0 1; do for M in 125000 1000000; do
  RATO_ROWS_DYNAMIC=$dyn timeout 200 python bench.py --workload driving --M $M --graph off --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python tools/pline.py "driving M=$M dynamic=$dyn"
done; done; done
