#!/bin/bash
# M-sweep of every kernel: prints one line per (workload, mode, M)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {
  timeout 200 python $R/bench.py --no-cpu-baseline --no-scp --no-configs --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
s = '%-46s kernel %.4f ms  %6.0f GB/s  frac %.3f  step %.4f ms  %.3e %s' % ('$*', r['kernel_ms'], r['achieved'], r['frac'], d['ms_per_step'], d['value'], d['unit'])
for k in ('factored', 'regenerated'):
    if 'roofline_' + k in d: s += '  | %s %.4f ms frac %.3f' % (k, d['roofline_' + k]['kernel_ms'], d['roofline_' + k]['frac'])
print(s)"
}
for M in 10000 100000 1000000; do run --workload drone --mode linearize --M $M; done
for M in 10000 100000 1000000 10000000; do run --workload drone --mode eval --M $M; done
for M in 10000 100000 1000000 4000000; do run --workload driving --mode linearize --M $M; done
for M in 10000 100000 1000000 10000000; do run --workload driving --mode eval --M $M; done
for M in 50000 1000000; do run --workload hopper --mode linearize --M $M; run --workload hopper --mode eval --M $M; done
run --workload drone --mode linearize --M 100000 --S 20
run --workload driving --mode linearize --M 100000 --S 20
