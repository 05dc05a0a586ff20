#!/bin/bash
for M in 200000 400000; do for rep in 1 2; do for dyn in 0 1; do
  RATO_ROWS_DYNAMIC=$dyn timeout 300 python bench.py --M $M --jacobian products --graph off --no-cpu-baseline --no-scp --steps 20 --warmup 3 2>/dev/null | python tools/pline.py "drone M=$M products dynamic=$dyn"
done; done; done
