#!/bin/bash
# Collects HBM traffic and SQ counters of every kernel of the benchmark step, one rocprofv3 --pmc
# pass per counter set.  usage: tools/pmc_bench.sh <tag> [bench.py arguments]
# results: gpurun_out/pmcb_<tag>_<pass>.csv (per-kernel averages)
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=gpurun_out/pmcb_${tag}_$i
  rm -rf $d
  timeout -s KILL 180 rocprofv3 --pmc $set -d $d -o r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmcb_${tag}_$i.log 2>&1
  python3 tools/rocprof_summary.py counter $(find $d -name '*.db' | head -1) gpurun_out/pmcb_${tag}_$i.csv > /dev/null
  rm -rf $d
  echo "pass $i done"
done
