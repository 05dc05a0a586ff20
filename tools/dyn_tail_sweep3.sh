#!/bin/bash
# finer sweep of the tail of the products queue (timings repeat to 0.1 % since the tiles are 2 MiB aligned)
for rep in 1 2; do
for cfg in "4 256" "3 256" "6 256" "8 256" "4 384" "6 384" "8 512" "3 384"; do
  set -- $cfg
  RATO_DYN_TAIL_SPLIT=$1 RATO_DYN_TAIL_TILES=$2 timeout 200 python bench.py --jacobian products --no-cpu-baseline --no-scp --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('split=$1 tiles=$2  kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done
done
