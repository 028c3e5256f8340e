#!/bin/bash
# run tools/poisson_only.py with every build/var/libsophtmi_*.so (and the default library)
N=${1:-512}
mkdir -p gpurun_out
for lib in default build/var/libsophtmi_*.so; do
  echo "== $lib"
  if [ "$lib" = default ]; then
    timeout -k 10 120 python3 tools/poisson_only.py $N 5 || exit 1
  else
    SOPHT_MPI_AMD_LIB=$PWD/$lib timeout -k 10 120 python3 tools/poisson_only.py $N 5 || exit 1
  fi
done
