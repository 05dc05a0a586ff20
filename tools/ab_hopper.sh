#!/bin/bash
for rep in 1 2 3; do
for lib in "" "$GRAFT_REPO_ROOT/tools/_build/librato_old_hopper.so"; do
  RATO_SAA_LIB=$lib python bench.py --workload hopper --mode linearize --M 1000000 --graph off --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python tools/pline.py "hopper deriv M=1e6 lib=${lib##*/}"
  RATO_SAA_LIB=$lib python bench.py --workload hopper --mode linearize --M 50000 --graph off --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python tools/pline.py "hopper deriv M=5e4 lib=${lib##*/}"
done
done
