#!/bin/bash
# Collects SQ counters of the pruned Poisson kernels (one rocprofv3 --pmc pass per counter set)
# usage: tools/pmc_poisson.sh <tag> [grid]      results: gpurun_out/pmc_<tag>_<pass>.csv
set -e
tag=$1; n=${2:-512}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  d=gpurun_out/pmc_${tag}_$i
  rm -rf $d
  timeout -s KILL 120 rocprofv3 --pmc $set -d $d -o r -- python3 tools/poisson_only.py $n 1 > gpurun_out/pmc_${tag}_$i.log 2>&1
  python3 tools/rocprof_summary.py counter $(find $d -name '*.db' | head -1) gpurun_out/pmc_${tag}_$i.csv > /dev/null
  echo "pass $i done"
done
