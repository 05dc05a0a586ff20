#!/bin/bash
for M in 125000 1000000; do
for i in 1 2; do
  for spc in 3 2 1; do
    for ph in "" "--philox"; do
    RATO_CAR_SLOTS_PER_CU=$spc python bench.py --config C5 --M $M $ph --graph off --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('M=$M slots/CU=$spc $ph kernel %.4f' % (d['roofline']['kernel_ms']))"
    done
  done
done
done
