#!/bin/bash
# hopper kernels at M = 1e6, 40 contacts: time (bench line) + SQ counters of the derivative kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hopper_$1; mkdir -p $O
python $R/bench.py --workload hopper --mode linearize --M 1000000 --graph off --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python $R/tools/pline.py "hopper linearize M=1e6"
python $R/bench.py --workload hopper --mode eval --M 1000000 --graph off --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python $R/tools/pline.py "hopper eval M=1e6"
python $R/bench.py --config C4 --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python $R/tools/pline.py "C4"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload hopper --mode linearize --M 1000000 --graph off --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc -- python3 $R/bench.py --workload hopper --mode linearize --M 1000000 --graph off --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
cd $R
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); head -3 $f | cut -c1-200
python3 tools/pmc_summary.py $O/pmc | head -6
find $O -name "*.csv" -size +1M -delete
