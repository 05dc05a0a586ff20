#!/bin/bash
# same-box alternating A/B of the default library against an alternate build: tools/ab_lib2.sh <alt.so> [reps] [bench args...]
alt=$1; reps=${2:-3}; shift 2
for i in $(seq $reps); do
  for v in default alt; do
    if [ $v = alt ]; then export RATO_SAA_LIB=$alt; else unset RATO_SAA_LIB; fi
    python bench.py --no-cpu-baseline --no-scp --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v  products %.4f  factored %.4f' % (d['roofline']['kernel_ms'], d.get('roofline_factored', {}).get('kernel_ms', 0)))"
  done
done
