#!/bin/bash
# M-sweep of every kernel: prints one line per (workload, mode, M)
run() { timeout 200 python bench.py --no-cpu-baseline --no-scp --steps 20 --warmup 3 "$@" 2>/dev/null | python tools/pline.py "$*"; }
for M in 10000 100000 1000000; do run --workload drone --mode linearize --M $M; done
for M in 10000 100000 1000000 10000000; do run --workload drone --mode eval --M $M; done
for M in 10000 100000 1000000 4000000; do run --workload driving --mode linearize --M $M; done
for M in 10000 100000 1000000 10000000; do run --workload driving --mode eval --M $M; done
for M in 50000 1000000; do run --workload hopper --mode linearize --M $M; run --workload hopper --mode eval --M $M; done
run --workload drone --mode linearize --M 100000 --S 20
run --workload driving --mode linearize --M 100000 --S 20
