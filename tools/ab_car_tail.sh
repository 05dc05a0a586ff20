#!/bin/bash
# Same-box alternating A/B: driving row kernel, tail of the dynamic queue as parts.  usage: tools/ab_car_tail.sh [M] [reps]
M=${1:-125000}; reps=${2:-3}
for i in $(seq $reps); do
  for v in "1 -1" "2 -1" "4 -1" "2 768" "4 192" "3 -1"; do
    set -- $v
    RATO_CAR_TAIL_SPLIT=$1 RATO_CAR_TAIL_TILES=$2 python bench.py --config C5 --M $M --steps 100 --warmup 10 --graph off --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('split $1 tiles $2  kernel_ms %.5f step_ms %.5f' % (d['roofline']['kernel_ms'], d['ms_per_step']))"
  done
done
