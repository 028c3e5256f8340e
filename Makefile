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

STANDIN_DIR := tests/rccl_standin
STANDIN_RCCL := $(STANDIN_DIR)/librccl_standin.so
STANDIN_OUT := $(STANDIN_DIR)/libsophtmi_standin.so

all: $(OUT) $(STANDIN_OUT)

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/sophtmi.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	mkdir -p $(dir $(OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@ $(LDLIBS)

# Test infrastructure: the same objects linked against a stand-in for librccl that carries the few RCCL calls of
# the library between rank processes sharing one GPU (tests/rccl_standin/standin.cpp), so that the library's RCCL
# branches run on a one-GPU box.  Never loaded unless a test names it in SOPHT_MPI_AMD_LIB.
$(STANDIN_RCCL): $(STANDIN_DIR)/standin.cpp
	g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $< -o $@ \
	    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib

$(STANDIN_OUT): $(OBJS) $(STANDIN_RCCL)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@ -L/opt/rocm/lib -lhipfft \
	    -L$(STANDIN_DIR) -lrccl_standin -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,'$$ORIGIN'

standin: $(STANDIN_OUT)

clean:
	rm -f $(OBJS) $(OUT) $(STANDIN_RCCL) $(STANDIN_OUT)

.PHONY: all clean standin
