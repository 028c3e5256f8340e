# Builds libsophtmi.so (HIP, gfx950 only) in-tree.  `make` cross-compiles without a GPU.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := sopht-mpi_amd/csrc
OUT     := sopht-mpi_amd/sopht_mpi_amd/lib/libsophtmi.so
SRCS    := $(CSRC)/smi_core.hip $(CSRC)/smi_stencil.hip $(CSRC)/smi_poisson.hip $(CSRC)/smi_fft.hip \
           $(CSRC)/smi_ib.hip $(CSRC)/smi_comm.hip $(CSRC)/smi_step.hip $(CSRC)/smi_2d.hip $(CSRC)/smi_fftapi.hip
OBJS    := $(SRCS:.hip=.o)
# packed-fp32 VALU forms (v_pk_*_f32) are switched off on purpose: some of their forms have given wrong results
# on this GPU pool; hipcc's host pass warns that it does not know the feature, which is harmless
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) \
            -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-result -Wno-unused-value
LDLIBS  := -L/opt/rocm/lib -lhipfft -lrccl -Wl,-rpath,/opt/rocm/lib

all: $(OUT)

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/sophtmi.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	mkdir -p $(dir $(OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@ $(LDLIBS)

clean:
	rm -f $(OBJS) $(OUT)

.PHONY: all clean
