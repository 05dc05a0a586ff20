#!/bin/bash
# Copy the judged summaries of a profile set (gpurun_out/<tag>/, written by tools/profile_round.sh <tag>) into profiles/
# under the round's naming: profiles/<tag>_*.      usage: tools/collect_profiles.sh <tag>
tag=$1; O=gpurun_out/$tag; P=profiles
for f in $O/bench_*.json $O/kt_*_kernel_stats.csv $O/pmc_*_summary.txt $O/scp_kernel_stats.csv $O/scp_kernel_summary.txt; do
  [ -s "$f" ] && cp "$f" $P/${tag}_$(basename $f)
done
[ -s $O/kt_metric.json ] && cp $O/kt_metric.json $P/${tag}_kt_metric_bench_line.json
for c in C3 C4 C5 drone_eval car_eval; do
  d=gpurun_out/${tag}_pmc_$c
  for n in write fetch sq_busy sq_wait sq_lds; do [ -s $d/$n.txt ] && cp $d/$n.txt $P/${tag}_pmc_${c}_$n.txt; done
  [ -s $d/kernel_stats.csv ] && cp $d/kernel_stats.csv $P/${tag}_pmc_${c}_kernel_stats.csv
done
ls $P | grep "^${tag}_" | wc -l
