#!/bin/bash
# products output, M = 1e5: launch-structure knobs re-checked on whatever box this lands on (same box, alternating)
for rep in 1 2 3; do
for cfg in "1 0 1" "4 256 1" "4 512 1" "2 1024 1" "4 128 1" "1 0 0"; do
  set -- $cfg
  RATO_ROWS_DYNAMIC=$3 RATO_DYN_TAIL_SPLIT=$1 RATO_DYN_TAIL_TILES=$2 timeout 200 python bench.py --jacobian products --no-cpu-baseline --no-scp --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('dynamic=$3 split=$1 tiles=$2  kernel_ms %.4f  sclk %.0f  %s' % (d['roofline']['kernel_ms'], d['device']['sclk_mhz_beside_hot_kernel'] or 0, (d['device']['board'] or {}).get('serial')))"
done
done
