#!/bin/bash
for rep in 1 2; do
for cfg in "4 128" "4 64" "4 32" "4 96" "2 128" "8 64" "1 0"; do
  set -- $cfg
  RATO_DYN_TAIL_SPLIT=$1 RATO_DYN_TAIL_TILES=$2 python bench.py --jacobian both --no-cpu-baseline --no-scp --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('split $1 tiles $2  products %.4f  regenerated %.4f' % (d['roofline']['kernel_ms'], d['roofline_regenerated']['kernel_ms']))"
done
done
