#!/bin/bash
# One measurement round on the GPU box: the four counter passes and the counters JSON that bench.py reads (first, so
# that the bench line of the same call can quote the traffic of this very build), the default bench line, and the
# rocprofv3 kernel trace summary.  usage: tools/measure_round.sh <version tag>; results in gpurun_out/
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=$1
bash tools/pmc_bench.sh r02$V
python3 tools/make_counters_json.py r02$V 512 gpurun_out/r02_counters_512cubed.json
cp gpurun_out/r02_counters_512cubed.json profiles/r02_counters_512cubed.json
timeout -k 10 400 python bench.py > gpurun_out/r02_bench512_default_$V.json
echo bench done
rm -rf gpurun_out/r02_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02_kt -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_bench512_under_rocprof_$V.json
python3 tools/rocprof_summary.py stats $(find gpurun_out/r02_kt -name '*.db' | head -1) gpurun_out/r02_kernel_stats_512cubed_$V.csv > /dev/null
rm -rf gpurun_out/r02_kt
echo trace done
