#!/bin/bash
# Host time of one interactor() call at 4 host-staged ranks, with the mismatch state re-gathered only on ownership
# changes (default) and on every call (as the reference does).
cd "$(dirname "$0")/.."
export SOPHT_MI_TRANSPORT=host
for mode in "nofilter" "nofilter resync"; do
  echo "== $mode"
  timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29711 tests/_slab_worker.py navier_stokes_with_forcing fused 64x32x64 $mode 2>&1 | grep -E "host ms|rank . ok|Error|error" || exit 1
done
