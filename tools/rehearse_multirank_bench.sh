#!/bin/bash
# Rehearsal of bench.py's multi-rank path: N ranks share the one GPU over the host-staged transport.
cd "$(dirname "$0")/.."
export SOPHT_MI_TRANSPORT=host
for N in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29730+N)) bench.py --gpus $N --steps 3 --warmup 1 --grid 256 --rehearse-on-host-transport 2>&1 | grep -E "^\{|Error|error|Traceback" | cut -c1-600 || exit 1
done
