#!/bin/bash
# products output: queue workgroups per CU (2 = what the LDS allows, 1 = 256 store streams), noise read / regenerated
for i in 1 2 3; do
  for spc in 2 1; do
    RATO_ROWS_SLOTS_PER_CU=$spc python bench.py --jacobian both --no-cpu-baseline --no-scp --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('slots/CU=$spc  products %.4f  factored %.4f  regenerated %.4f' % (d['roofline']['kernel_ms'], d['roofline_factored']['kernel_ms'], d['roofline_regenerated']['kernel_ms']))"
  done
done
