#!/bin/bash
# Two counter passes over the Poisson solve at 512^3 (tools/poisson_only.py) for the vector-memory path of its kernels:
# instruction counts and FIFO-full stalls at issue (SQ), texture-addresser busy / stalled cycles and L2 write requests.
# results: gpurun_out/pmc_zstore.csv, gpurun_out/pmc_zstore2.csv.  Run on the GPU box from the repository root.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
d=gpurun_out/pmc_zstore
rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES -d $d -o r -- python3 tools/poisson_only.py 512 2 > gpurun_out/pmc_zstore.log 2>&1
python3 tools/rocprof_summary.py counter $(find $d -name '*.db' | head -1) gpurun_out/pmc_zstore.csv > /dev/null
rm -rf $d
d=gpurun_out/pmc_zstore2
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum -d $d -o r -- python3 tools/poisson_only.py 512 2 > gpurun_out/pmc_zstore2.log 2>&1
python3 tools/rocprof_summary.py counter $(find $d -name '*.db' | head -1) gpurun_out/pmc_zstore2.csv > /dev/null
rm -rf $d
echo done
