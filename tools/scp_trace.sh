cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_e; mkdir -p $O
python3 $R/scp_bench.py --system drone --reduced --M 100000 --S 50 --iters 60 --seed 7 > $O/scp_plain.json 2> $O/scp_plain.err
tail -2 $O/scp_plain.err | head -1; tail -1 $O/scp_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/scp_bench.py --system drone --reduced --M 100000 --S 50 --iters 60 --seed 7 > $O/scp_prof.json 2>/dev/null
TRACE_DUMP=${TRACE_DUMP:-0} python3 $R/tools/trace_gaps.py $O/kt > $O/scp_trace_gaps.txt 2>&1; cat $O/scp_trace_gaps.txt
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/scp_kernel_stats.csv
find $O/kt -name "*.csv" -size +1M -delete
