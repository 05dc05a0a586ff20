#!/bin/bash
# same-box alternating A/B of the current library against tools/_build/librato_prev.so (bench args: $@)
for rep in 1 2 3; do
for lib in "" "$GRAFT_REPO_ROOT/tools/_build/librato_prev.so"; do
  for jac in products factored; do
    RATO_SAA_LIB=$lib timeout 200 python bench.py --jacobian $jac --no-cpu-baseline --no-scp --steps 50 --warmup 5 "$@" 2>/dev/null | python tools/pline.py "lib=${lib##*/} $jac"
  done
done
done
