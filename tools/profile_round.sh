#!/bin/bash
# Profile set of a round (run on the GPU box through gpurun): usage tools/profile_round.sh <tag>   e.g. r03_g
# Writes under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
tag=${1:-r03_x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# 1. the driver's command line (default flags) -> bench line (roofline, roofline_factored / _regenerated, scp, configs, cpu_baseline)
timeout 900 $B > $O/bench_default.json 2> $O/bench_default.err
# 2. kernel-trace stats of the same workload (eager; rocprofv3 and hipGraph replay do not mix on this pool)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_metric -- $B --steps 50 --warmup 10 --no-cpu-baseline --no-scp --no-configs > $O/kt_metric.json 2> /dev/null
# 3. PMC passes, separate runs (HBM traffic of the dominant kernel)
for jac in products factored; do
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${jac}_$c -- $B --jacobian $jac --steps 5 --warmup 1 --no-cpu-baseline --no-scp --no-configs > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $O/pmc_${jac}_$c > $O/pmc_${jac}_${c}_summary.txt 2>&1
  done
done
# 4. BASELINE configs C2-C4 and C5's shard: step (hipGraph replay where the bench chooses it) and eager kernel stats
for c in C2 C3 C4 C5; do
  timeout 300 $B --config $c --no-cpu-baseline --no-scp --steps 50 --warmup 5 > $O/bench_$c.json 2> /dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$c -- $B --config $c --graph off --no-cpu-baseline --no-scp --steps 50 --warmup 5 > /dev/null 2>&1
done
# 4b. C2 / C3 in the reference's Monte-Carlo form (round 5): rollout kernel + exact selection
for c in C2 C3; do
  timeout 300 $B --config $c --mode eval --no-cpu-baseline --no-scp --steps 200 --warmup 5 > $O/bench_${c}_eval.json 2> /dev/null
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${c}_eval -- $B --config $c --mode eval --graph off --no-cpu-baseline --no-scp --steps 200 --warmup 5 > /dev/null 2>&1
done
cd $R
for d in kt_metric kt_C2 kt_C3 kt_C4 kt_C5 kt_C2_eval kt_C3_eval; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*.csv" -size +2M -delete; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
# 5. counter evidence for the other dominant kernels (write / fetch / SQ busy / SQ wait / LDS passes each)
bash $R/tools/pmc_config.sh ${tag}_pmc_C5 --config C5 > /dev/null 2>&1
bash $R/tools/pmc_config.sh ${tag}_pmc_C3 --config C3 > /dev/null 2>&1
bash $R/tools/pmc_config.sh ${tag}_pmc_C4 --config C4 > /dev/null 2>&1
bash $R/tools/pmc_config.sh ${tag}_pmc_drone_eval --workload drone --mode eval --M 10000000 > /dev/null 2>&1
bash $R/tools/pmc_config.sh ${tag}_pmc_car_eval --workload driving --mode eval --M 10000000 > /dev/null 2>&1
# 6. the SCP block kernel by kernel (what `scp.kernels` of the bench line is checked against)
bash $R/tools/scp_kernel_stats.sh ${tag} > $O/scp_kernel_summary.txt 2>&1
cp $R/gpurun_out/${tag}_scp_kernel_stats.csv $O/scp_kernel_stats.csv 2>/dev/null
ls $O | head -60
