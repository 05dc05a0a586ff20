#!/bin/bash
# eval kernels: noise read from HBM vs regenerated in the kernel (Philox), M sweep
for w in drone driving; do for M in 100000 1000000 10000000; do for ph in "" "--philox"; do
  timeout 200 python bench.py --workload $w --mode eval --M $M $ph --graph off --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python tools/pline.py "$w eval M=$M $ph"
done; done; done
