#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 summaries of ONE command.
#   tools/profile_cmd.sh TAG [--pmc] -- python3 bench.py ...
# Kernel trace + stats in one run; with --pmc also the PMC counter passes, each in its own run (a --pmc run never
# carries a trace domain).  The program itself follows `--` (no env / bash -c hop under rocprofv3).
set -e
TAG=$1; shift
PMC=0
if [ "$1" = "--pmc" ]; then PMC=1; shift; fi
[ "$1" = "--" ] && shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "$*" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- "$@" > $OUT/trace.log 2>&1
if [ $PMC = 1 ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- "$@" > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- "$@" > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- "$@" > $OUT/pmc_sq.log 2>&1
fi
python3 tools/summarize_profile.py $OUT > $OUT/summary.md
# keep what travels back small: the summary, the stats CSV and the traffic file
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
tail -n 40 $OUT/summary.md
