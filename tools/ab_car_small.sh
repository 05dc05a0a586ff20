#!/bin/bash
for rep in 1 2 3; do for sp in 1 2 3 4 0; do
  v=$sp; [ $sp = 0 ] && v=-1
  RATO_CAR_SMALL_SPLIT=$v timeout 200 python bench.py --config C3 --graph off --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python tools/pline.py "C3 split=$v"
done; done
