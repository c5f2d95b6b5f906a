# Builds the in-tree native artefacts (all git-ignored, all shipped to the GPU box by gpurun):
#   espflix_amd/libefx.so          the product: HIP kernels (gfx950) + C-ABI   (include/efx.h)
#   espflix_amd/gen/libefx_gen.so  synthetic MPEG-1 stream generator (workload tooling)
#   oracle/_build/libefx_oracle.so CPU restatement (TEST oracle)
#   oracle/_ref/*                  the unmodified reference + harnesses (TEST oracle; needs /root/reference)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC     = espflix_amd/csrc
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
OBJS     = $(CSRC)/efx_api.o $(CSRC)/k_demux.o $(CSRC)/k_index.o $(CSRC)/k_parse.o $(CSRC)/k_recon.o $(CSRC)/k_video.o $(CSRC)/k_sbc.o $(CSRC)/k_tsindex.o $(CSRC)/efx_tables.o

.PHONY: all lib gen oracle ref clean
all: lib gen oracle

lib: espflix_amd/libefx.so
gen: espflix_amd/gen/libefx_gen.so

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/efx_internal.h include/efx.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/efx_tables.o: $(CSRC)/efx_tables.cpp $(CSRC)/efx_internal.h $(CSRC)/mpeg1_codebook.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

espflix_amd/libefx.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

espflix_amd/gen/libefx_gen.so: espflix_amd/gen/efx_gen.cpp $(CSRC)/mpeg1_codebook.h
	g++ -std=c++17 -O2 -Wall -Wextra -fPIC -shared -pthread $< -o $@

oracle:
	$(MAKE) -C oracle port
ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(CSRC)/*.o espflix_amd/libefx.so espflix_amd/gen/libefx_gen.so
	$(MAKE) -C oracle clean
