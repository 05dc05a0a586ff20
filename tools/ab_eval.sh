#!/bin/bash
for rep in 1 2 3; do for dyn in 0 1; do for M in 1000000 10000000; do
  RATO_EVAL_DYNAMIC=$dyn timeout 200 python bench.py --workload drone --mode eval --M $M --graph off --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python tools/pline.py "drone eval M=$M dynamic=$dyn"
done; done; done
