#!/bin/bash
# Same-board alternating A/B of library builds and/or environment knobs (run on the GPU box through gpurun).
#   tools/ab.sh [-r reps] 'name|-D flags for hipcc (or empty)|ENV=val ENV2=val2' ... -- <bench.py args>
# A variant with -D flags is built on the box into /tmp/ab_<name>.so (same flags as riskaversetrajopt_amd/_build.py) and
# selected with RATO_SAA_LIB; a variant without flags uses the in-tree library.  Every variant runs `bench.py <args>`
# once per repetition, interleaved, and prints kernel ms / step ms of the line (plus *_factored / *_regenerated if there).
reps=3
if [ "$1" = "-r" ]; then reps=$2; shift 2; fi
specs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do specs+=("$1"); shift; done
shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for spec in "${specs[@]}"; do
  IFS='|' read -r name flags envs <<< "$spec"
  if [ -n "$flags" ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast $flags -I $R/include -I $R/riskaversetrajopt_amd/csrc \
      -o /tmp/ab_$name.so $R/riskaversetrajopt_amd/csrc/*.hip -ldl || { echo "build of $name failed"; exit 1; }
  fi
done
for i in $(seq $reps); do
  for spec in "${specs[@]}"; do
    IFS='|' read -r name flags envs <<< "$spec"
    lib=""; [ -n "$flags" ] && lib=/tmp/ab_$name.so
    env RATO_SAA_LIB=$lib $envs python $R/bench.py --no-cpu-baseline --no-scp --no-configs "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
s = '%-14s kernel %.4f ms  frac %.3f  step %.4f ms' % ('$name', d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step'])
for k in ('factored', 'regenerated'):
    if 'roofline_' + k in d: s += '  | %s %.4f / %.4f' % (k, d['roofline_' + k]['kernel_ms'], d['ms_per_step_' + k])
print(s)"
  done
done
