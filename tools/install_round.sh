#!/bin/bash
# Copies the results of tools/measure_round.sh <round> <tag> from gpurun_out/ into profiles/ and removes the files of
# the generation it replaces.  usage: tools/install_round.sh <round> <new tag> <old tag>
set -e
cd "$(dirname "$0")/.."
R=$1; V=$2; O=$3
for i in 1 2 3 4; do cp gpurun_out/pmcb_${R}${V}_$i.csv profiles/${R}_pmc_pass${i}_512cubed_$V.csv; done
cp gpurun_out/${R}_counters_512cubed.json profiles/${R}_counters_512cubed.json
cp gpurun_out/${R}_bench512_default_$V.json gpurun_out/${R}_bench512_under_rocprof_$V.json gpurun_out/${R}_kernel_stats_512cubed_$V.csv profiles/
if [ -n "$O" ]; then git rm -q --ignore-unmatch profiles/${R}_*_$O.json profiles/${R}_*_$O.csv; fi
grep -rl "_$O\b" DESIGN.md README.md INTEGRATION.md | xargs -r sed -i "s/\(${R}_[A-Za-z0-9_{},]*\)_$O\b/\1_$V/g"
git add profiles
