#!/bin/bash
# new default (1 queue workgroup per CU for the products output at >= 1024 tiles) against 2 per CU forced, + the tail
for i in 1 2 3; do
  for v in "" 2; do
    RATO_ROWS_SLOTS_PER_CU=$v python bench.py --jacobian both --no-cpu-baseline --no-scp --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('slots/CU=${v:-default}  products %.4f  factored %.4f  regenerated %.4f' % (d['roofline']['kernel_ms'], d['roofline_factored']['kernel_ms'], d['roofline_regenerated']['kernel_ms']))"
  done
done
for t in 64 128 256; do RATO_DYN_TAIL_TILES=$t python bench.py --jacobian both --no-cpu-baseline --no-scp --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('tail tiles $t  products %.4f  regenerated %.4f' % (d['roofline']['kernel_ms'], d['roofline_regenerated']['kernel_ms']))"; done
