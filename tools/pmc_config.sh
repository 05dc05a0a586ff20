#!/bin/bash
# Counter evidence for the dominant kernel of one bench configuration (run on the GPU box through gpurun).
#   usage: tools/pmc_config.sh <tag> <bench args...>      e.g.  tools/pmc_config.sh r03_a_C5 --config C5
# Separate rocprofv3 passes (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains beside --pmc), each
# summarised per kernel (mean per dispatch) into gpurun_out/<tag>/<pass>.txt.  Copy what is to be judged to profiles/.
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py $* --graph off --no-cpu-baseline --no-scp --no-configs --steps 20 --warmup 3"
run_pass() {   # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -- $B > $O/$name.log 2>&1
  python3 $R/tools/pmc_summary.py $O/$name > $O/$name.txt 2>&1
  find $O/$name -name "*.csv" -delete 2>/dev/null
}
run_pass write WRITE_SIZE
run_pass fetch FETCH_SIZE
run_pass sq_busy SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run_pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR
run_pass sq_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.json 2> /dev/null
f=$(find $O/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/kt
tail -n +1 $O/*.txt | head -120
