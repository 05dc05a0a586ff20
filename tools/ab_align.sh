#!/bin/bash
# Same-box alternating A/B of the packed tile alignment (A/B builds of tools/_build): products + factored kernels
reps=${1:-3}
for i in $(seq $reps); do
  for v in default nat 4m 2m512k; do
    if [ $v = default ]; then unset RATO_SAA_LIB RATO_PACKED_ALIGN_BYTES; else export RATO_SAA_LIB=tools/_build/librato_$v.so; fi
    if [ $v = 4m ]; then export RATO_PACKED_ALIGN_BYTES=4194304; else unset RATO_PACKED_ALIGN_BYTES; fi
    python bench.py --no-cpu-baseline --no-scp --steps 60 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v  products %.4f  factored %.4f  step %.4f' % (d['roofline']['kernel_ms'], d['roofline_factored']['kernel_ms'], d['ms_per_step']))"
  done
done
