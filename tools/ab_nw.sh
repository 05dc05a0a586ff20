#!/bin/bash
# A/B: waves per workgroup of the drone row kernel (8 = default build, 12, 16), same box, alternating, products
for rep in 1 2 3 4; do
for lib in "" nw12 nw16; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/tools/_build/librato_$lib.so
  RATO_SAA_LIB=$L timeout 200 python bench.py --jacobian products --no-cpu-baseline --no-scp --steps 100 --warmup 10 2>/dev/null | python tools/pline.py "waves=${lib:-nw8} products"
done
done
