#!/bin/bash
# tools/rank_local_step.py for several (ranks, transpose chunks) pairs at 512^3; JSON lines in gpurun_out/rl_p*_c*.json
# usage: tools/rank_local_sweep.sh "8 1" "8 2" "4 1"
set -e
cd "$(dirname "$0")/.."
export SOPHT_MPI_AMD_LIB=$PWD/tests/rccl_standin/libsophtmi_standin.so SMI_STANDIN_ALONE=1 SOPHT_MI_TRANSPORT=rccl
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg
  SMI_FFT_TRANSPOSE_CHUNKS=$2 timeout -k 10 200 python tools/rank_local_step.py 512 $1 > gpurun_out/rl_p$1_c$2.json
  python3 -c "
import json
d = json.load(open('gpurun_out/rl_p$1_c$2.json'))
print('ranks', $1, 'chunks', $2, 'ms/step', round(d['ms_per_step_wall_no_profiling'], 3),
      {k: round(v['ms_per_step'], 3) for k, v in d['kernels'].items()}, flush=True)"
done
