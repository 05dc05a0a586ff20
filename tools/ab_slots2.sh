#!/bin/bash
for M in 50000 200000 1000000; do
for i in 1 2; do
  for spc in 2 1; do
    RATO_ROWS_SLOTS_PER_CU=$spc python bench.py --jacobian products --M $M --no-cpu-baseline --no-scp --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('M=$M slots/CU=$spc  products %.4f' % (d['roofline']['kernel_ms']))"
    RATO_ROWS_SLOTS_PER_CU=$spc python bench.py --jacobian regenerated --M $M --no-cpu-baseline --no-scp --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('M=$M slots/CU=$spc  regenerated %.4f' % (d['roofline']['kernel_ms']))"
  done
done
done
