#!/bin/bash
# products output at large M: dynamic tile queue forced (RATO_ROWS_DYNAMIC=2) vs the default policy (queue only up to four
# rounds of tiles), same box, alternating.  usage: tools/ab_big_products.sh
for i in 1 2; do
  for M in 400000 1000000; do
    for dyn in 1 2 0; do
      RATO_ROWS_DYNAMIC=$dyn python bench.py --jacobian products --M $M --no-cpu-baseline --no-scp --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('M=$M dynamic=$dyn  kernel_ms %.4f  frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
    done
  done
done
