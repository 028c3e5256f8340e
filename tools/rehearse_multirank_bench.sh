#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a one-GPU box.  `python bench.py --gpus N` starts its own ranks;
# they share the one GPU, (a) over the library's RCCL branches with the socket stand-in for librccl and
# (b) over the host-staged transport.  The lines are marked invalid: they are no multi-GPU measurements.
cd "$(dirname "$0")/.."
for CHUNKS in 1 4; do
  SMI_FFT_TRANSPOSE_CHUNKS=$CHUNKS SOPHT_MI_TRANSPORT=rccl SOPHT_MPI_AMD_LIB=$PWD/tests/rccl_standin/libsophtmi_standin.so \
    timeout -k 10 600 python bench.py --gpus 8 --steps 3 --warmup 1 2>&1 | grep -E "^\{|Error|error|Traceback|giving up" | cut -c1-900 || exit 1
done
SOPHT_MI_TRANSPORT=host timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --grid 256 --rehearse-on-host-transport 2>&1 | grep -E "^\{|Error|error|Traceback" | cut -c1-600 || exit 1
