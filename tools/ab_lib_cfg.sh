#!/bin/bash
# same-box alternating A/B of the current library vs tools/_build/librato_prev.so on the driving workloads
for rep in 1 2 3; do for lib in "" "$GRAFT_REPO_ROOT/tools/_build/librato_prev.so"; do
  for a in "--config C3" "--config C5" "--workload driving --M 1000000"; do
  RATO_SAA_LIB=$lib timeout 200 python bench.py $a --graph off --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python tools/pline.py "$a lib=${lib##*/}"
done; done; done
