#!/bin/bash
# build_variant.sh NAME FILE "-DFLAG=..."  -> build/var/libsophtmi_NAME.so with csrc/FILE.hip recompiled
# under the given flags and every other object taken from the regular build (run `make` first)
set -e
NAME=$1; FILE=$2; shift 2
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isopht-mpi_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-result -Wno-unused-value"
C=sopht-mpi_amd/csrc
mkdir -p build/var
/opt/rocm/bin/hipcc $F "$@" -c $C/$FILE.hip -o build/var/${FILE}_$NAME.o
OBJS=""
for f in smi_core smi_stencil smi_poisson smi_fft smi_ib smi_comm smi_step smi_2d smi_fftapi; do
  if [ $f = $FILE ]; then OBJS="$OBJS build/var/${FILE}_$NAME.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o build/var/libsophtmi_$NAME.so -L/opt/rocm/lib -lhipfft -lrccl -Wl,-rpath,/opt/rocm/lib
echo built build/var/libsophtmi_$NAME.so
