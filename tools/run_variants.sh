#!/bin/bash
# run a timing script with every build/var/libsophtmi_*.so and with the default library
#   tools/run_variants.sh poisson [n]     tools/poisson_only.py n 5
#   tools/run_variants.sh bench [n]       bench.py --grid n --no-cpu-baseline (per-kernel table)
MODE=${1:-poisson}; N=${2:-512}
for lib in default build/var/libsophtmi_*.so; do
  echo "== $lib"
  [ "$lib" = default ] && unset SOPHT_MPI_AMD_LIB || export SOPHT_MPI_AMD_LIB=$PWD/$lib
  if [ $MODE = poisson ]; then
    timeout -k 10 120 python3 tools/poisson_only.py $N 5 || exit 1
  else
    timeout -k 10 200 python3 bench.py --grid $N --no-cpu-baseline > /tmp/variant_bench.json || exit 1
    python3 tools/show_bench.py /tmp/variant_bench.json
  fi
done
