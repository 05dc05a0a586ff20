#!/bin/bash
# usage: sweep_cpt.sh <workload> <M> <S> <cpt:spl ...>
w=$1; M=$2; S=$3; shift 3
for v in "$@"; do
  c=${v%%:*}; l=${v##*:}
  python bench.py --workload $w --M $M --S $S --steps 10 --warmup 2 --no-cpu-baseline --cols-per-thread $c --samples-per-lane $l 2>/dev/null | python tools/pline.py "$w M=$M S=$S cpt=$c spl=$l"
done
