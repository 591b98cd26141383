# Builds libsnkhip.so (gfx950 only) and the C oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := snickery_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude
LIB   := snickery_amd/libsnkhip.so
OBJS  := $(CSRC)/knn_kernels.o $(CSRC)/knn16_kernels.o $(CSRC)/viterbi_kernels.o $(CSRC)/joinfast_kernels.o $(CSRC)/joinlb2_kernels.o $(CSRC)/greedy_kernels.o $(CSRC)/greedy32_kernels.o $(CSRC)/greedy_hoist_kernels.o $(CSRC)/greedy_res_kernels.o $(CSRC)/concat_kernels.o $(CSRC)/snk_api.o

all: $(LIB) oracle

# the K-NN prefilter kernels test the matrix results on the vector unit: accumulators in architected registers
$(CSRC)/knn16_kernels.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/snk_internal.h $(CSRC)/greedy_common.h $(CSRC)/greedy32_device.h include/snk.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
