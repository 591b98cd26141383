# Builds libsnkhip.so (gfx950 only) and the C oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := snickery_amd/csrc
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude
LIB   := snickery_amd/libsnkhip.so
OBJS  := $(CSRC)/knn_kernels.o $(CSRC)/knn16_kernels.o $(CSRC)/viterbi_kernels.o $(CSRC)/joinfast_kernels.o $(CSRC)/joinlb2_kernels.o $(CSRC)/greedy_kernels.o $(CSRC)/greedy32_kernels.o $(CSRC)/greedy_hoist_kernels.o $(CSRC)/greedy_res_kernels.o $(CSRC)/concat_kernels.o $(CSRC)/kmeans_kernels.o $(CSRC)/api_core.o $(CSRC)/api_knn.o $(CSRC)/api_viterbi.o $(CSRC)/api_greedy.o $(CSRC)/api_shard.o $(CSRC)/api_options.o

all: $(LIB) oracle

# the K-NN prefilter kernels test the matrix results on the vector unit: accumulators in architected registers
$(CSRC)/knn16_kernels.o: HIPFLAGS += -mllvm -amdgpu-mfma-vgpr-form
# the kernels of a batch step are bound by the SIMDs' vector-issue port, where a packed float32 instruction (v_pk_add_f32,
# v_pk_fma_f32: what the SLP vectoriser makes of adjacent scalar arithmetic) costs more than the two scalar ones it replaces
# (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'; measured on the B* step: 4.09 -> 4.23 M frames/s)
$(CSRC)/joinlb2_kernels.o $(CSRC)/joinfast_kernels.o $(CSRC)/knn16_kernels.o: HIPFLAGS += -fno-slp-vectorize

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/snk_internal.h $(CSRC)/snk_engine.h $(CSRC)/greedy_common.h $(CSRC)/greedy32_device.h include/snk.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS) $(CSRC)/libsnkhip.map
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/libsnkhip.map -o $@ $(OBJS)

oracle:
	$(MAKE) -C oracle

# ---- host orchestration under AddressSanitizer + UBSan, in the CPU container (no GPU): the library's own translation
# units compiled for the HOST only and linked against tools/fakehip (allocation bookkeeping, no device; kernels never run).
# Drives: tests/test_host_asan.py.  The sanitizer runtime is clang's shared one (preloaded into python by the test).
ASAN_DIR   := build/asan
ASAN_CXX   := /opt/rocm/lib/llvm/bin/clang++
ASAN_FLAGS := --cuda-host-only -O1 -g -fPIC -std=c++17 -ffp-contract=off -Wno-unused-value -Iinclude \
              -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libasan
ASAN_OBJS  := $(patsubst $(CSRC)/%.o,$(ASAN_DIR)/%.o,$(OBJS)) $(ASAN_DIR)/fake_hip.o
asan-host: $(ASAN_DIR)/libsnkhip_host_asan.so
$(ASAN_DIR)/%.o: $(CSRC)/%.hip $(CSRC)/snk_internal.h $(CSRC)/snk_engine.h $(CSRC)/greedy_common.h $(CSRC)/greedy32_device.h include/snk.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@
$(ASAN_DIR)/fake_hip.o: tools/fakehip/fake_hip.cpp
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@
# every host object refers to the device code object it was NOT given (__hip_fatbin_<hash>): empty stand-ins
$(ASAN_DIR)/fatbins.c: $(ASAN_OBJS)
	nm -u $(ASAN_OBJS) | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u | sed 's/.*/const char &[64] = {0};/' > $@
$(ASAN_DIR)/libsnkhip_host_asan.so: $(ASAN_OBJS) $(ASAN_DIR)/fatbins.c
	gcc -fPIC -c $(ASAN_DIR)/fatbins.c -o $(ASAN_DIR)/fatbins.o
	$(ASAN_CXX) -shared -fPIC -fsanitize=address,undefined -shared-libasan -o $@ $(ASAN_OBJS) $(ASAN_DIR)/fatbins.o -ldl -lpthread

clean:
	rm -rf $(ASAN_DIR)
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean

.PHONY: all oracle clean asan-host
