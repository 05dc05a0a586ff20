#!/bin/bash
# dynamic queue: the last units as fractions of a tile (RATO_DYN_TAIL_SPLIT x RATO_DYN_TAIL_TILES), same box, alternating
for rep in 1 2 3 4 5; do
for cfg in "1 0" "2 1024" "4 256"; do
  set -- $cfg
  RATO_DYN_TAIL_SPLIT=$1 RATO_DYN_TAIL_TILES=$2 timeout 200 python bench.py --jacobian products --no-cpu-baseline --no-scp --steps 100 --warmup 10 2>/dev/null | python tools/pline.py "tail split=$1 tiles=$2 products"
done
done
