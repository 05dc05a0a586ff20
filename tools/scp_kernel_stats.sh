#!/bin/bash
# Kernel-trace statistics of the reduced SCP loop:   tools/scp_kernel_stats.sh <tag> [scp_bench args]
# -> gpurun_out/<tag>_scp_kernel_stats.csv and a one-line-per-kernel summary on stdout.  Every step under `timeout`.
tag=${1:-scp}; shift
args=${@:---reduced --M 100000 --S 50 --iters 60}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${tag}_scp_kt
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scp_bench.py $args > $O/scp.log 2>&1 < /dev/null
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv under $O"; exit 1; }
cp "$f" $R/gpurun_out/${tag}_scp_kernel_stats.csv
timeout 60 python3 - "$f" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r'(\w+_kernel|rs_coop|rs_small|rs_\w+|sum_partials\w*|__amd_\w+)', r['Name'])
    print((m.group(1) if m else r['Name'][:44]).ljust(44), r['Calls'].rjust(6), '%9.1f us' % (float(r['AverageNs']) / 1e3), '%6s %%' % r['Percentage'])
PY
