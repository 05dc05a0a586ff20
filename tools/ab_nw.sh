#!/bin/bash
# A/B: waves per workgroup of the drone row kernel (8 = default build, 12, 16), dynamic tile queue, same box, alternating
for rep in 1 2 3; do
for lib in "" nw12 nw16; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/tools/_build/librato_$lib.so
  for jac in products factored; do
    RATO_SAA_LIB=$L timeout 200 python bench.py --jacobian $jac --no-cpu-baseline --no-scp --steps 50 --warmup 5 2>/dev/null | python tools/pline.py "waves=${lib:-nw8} $jac"
  done
done
done
