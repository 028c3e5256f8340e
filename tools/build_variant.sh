#!/bin/bash
# build_variant.sh NAME "-DFLAG=..."  -> build/var/libsophtmi_NAME.so (only smi_fft.hip recompiled)
set -e
NAME=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Isopht-mpi_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-result -Wno-unused-value"
C=sopht-mpi_amd/csrc
mkdir -p build/var
/opt/rocm/bin/hipcc $F "$@" -c $C/smi_fft.hip -o build/var/smi_fft_$NAME.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/smi_core.o $C/smi_stencil.o $C/smi_poisson.o build/var/smi_fft_$NAME.o $C/smi_ib.o $C/smi_comm.o $C/smi_step.o $C/smi_2d.o $C/smi_fftapi.o -o build/var/libsophtmi_$NAME.so -L/opt/rocm/lib -lhipfft -lrccl -Wl,-rpath,/opt/rocm/lib
echo built build/var/libsophtmi_$NAME.so
