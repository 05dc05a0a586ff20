#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fold; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
for v in fold nofold; do
  if [ $v = nofold ]; then export RATO_BENCH_NO_FOLD=1; else unset RATO_BENCH_NO_FOLD; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -- $B --config C4 --graph off --no-cpu-baseline --steps 50 --warmup 5 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob
for d in ("kt_fold","kt_nofold"):
    f=glob.glob(f"gpurun_out/prof_fold/{d}/*/*kernel_stats.csv")[0]
    print("==",d)
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-50s calls %4s avg %9.0f ns min %s max %s"%(r["Name"][:50], r["Calls"], float(r["AverageNs"]), r.get("MinNs"), r.get("MaxNs")))
PY
find $O -name "*_kernel_trace.csv" -delete
