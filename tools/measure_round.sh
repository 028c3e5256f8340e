#!/bin/bash
# One measurement round on the GPU box: the four counter passes and the counters JSON that bench.py reads (first, so
# that the bench line of the same call can quote the traffic of this very build), the default bench line, and the
# rocprofv3 kernel trace summary.  usage: tools/measure_round.sh <round, e.g. r03> <version tag>; results in
# gpurun_out/ (copy what is to be judged into profiles/)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=$1
V=$2
bash tools/pmc_bench.sh $R$V
python3 tools/make_counters_json.py $R$V 512 gpurun_out/${R}_counters_512cubed.json
cp gpurun_out/${R}_counters_512cubed.json profiles/${R}_counters_512cubed.json
timeout -k 10 400 python bench.py > gpurun_out/${R}_bench512_default_$V.json
echo bench done
rm -rf gpurun_out/${R}_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${R}_kt -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_bench512_under_rocprof_$V.json
python3 tools/rocprof_summary.py stats $(find gpurun_out/${R}_kt -name '*.db' | head -1) gpurun_out/${R}_kernel_stats_512cubed_$V.csv > /dev/null
rm -rf gpurun_out/${R}_kt
echo trace done
