# Builds the in-tree native artefacts (all git-ignored, all shipped to the GPU box by gpurun):
#   espflix_amd/libefx.so          the product: HIP kernels (gfx950) + C-ABI   (include/efx.h)
#   espflix_amd/gen/libefx_gen.so  synthetic MPEG-1 stream generator (workload tooling)
#   oracle/_build/libefx_oracle.so CPU restatement (TEST oracle)
#   oracle/_ref/*                  the unmodified reference + harnesses (TEST oracle; needs /root/reference)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC     = espflix_amd/csrc
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-function
OBJS     = $(CSRC)/efx_api.o $(CSRC)/k_demux.o $(CSRC)/k_index.o $(CSRC)/k_parse.o $(CSRC)/k_recon.o $(CSRC)/k_video.o $(CSRC)/k_sbc.o $(CSRC)/k_tsindex.o $(CSRC)/efx_tables.o $(CSRC)/efx_multi.o

.PHONY: all lib gen oracle ref clean dropin scale harness
# (`scale` links librccl: built by __graft_entry__.build() and by `make scale`, not by a plain `make`)
all: lib gen oracle

lib: espflix_amd/libefx.so
gen: espflix_amd/gen/libefx_gen.so

# k_recon.hip: the atomic optimizer would turn k_recon_all's one-lane dequeue into "atomic, wait, readfirstlane" on the spot --
# the wait is the point of issuing the claim an item ahead (the value is consumed at the pre-store drain)
$(CSRC)/k_recon.o: HIPFLAGS += -mllvm -amdgpu-atomic-optimizer-strategy=None
$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/efx_internal.h $(CSRC)/parse_tm.h $(CSRC)/efx_probe.h include/efx.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(CSRC)/efx_tables.o: $(CSRC)/efx_tables.cpp $(CSRC)/efx_internal.h $(CSRC)/parse_tm.h $(CSRC)/mpeg1_codebook.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@
$(CSRC)/efx_multi.o: $(CSRC)/efx_multi.cpp include/efx.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

espflix_amd/libefx.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

espflix_amd/gen/libefx_gen.so: espflix_amd/gen/efx_gen.cpp $(CSRC)/mpeg1_codebook.h
	g++ -std=c++17 -O2 -Wall -Wextra -fPIC -shared -pthread $< -o $@

# TEST: the reference's UNMODIFIED host player (src/espflix.cpp) and platform layer (src/streamer.cpp) built against
# libefx's drop-in headers instead of src/player.h / src/video.h -- player.cpp and video.cpp are not in the build.
# Only where the reference tree exists (the build container); the binaries travel to the GPU box like the .so files.
REF ?= /root/reference
DROPIN_FLAGS = -std=c++14 -O1 -w -fpermissive -pthread -Iinclude/espflix_dropin -Iinclude -I$(REF)/src
DROPIN_LIBS  = -Lespflix_amd -lefx -Wl,-rpath,'$$ORIGIN/../../espflix_amd' -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
# (a quoted #include looks in the including file's own directory first, so the reference sources are reached through
# symbolic links in a scratch directory that holds neither player.h nor video.h: those two -- and only those -- come
# from include/espflix_dropin; a maintainer simply replaces the two files in src/)
DROPIN_SRC = tests/_build/dropin_src
dropin: tests/_build/espflix_dropin tests/_build/espflix_dropin_long
$(DROPIN_SRC)/espflix.cpp:
	mkdir -p $(DROPIN_SRC)
	ln -sf $(REF)/src/espflix.cpp $(DROPIN_SRC)/espflix.cpp
	ln -sf $(REF)/src/streamer.cpp $(DROPIN_SRC)/streamer.cpp
tests/_build/espflix_dropin: $(DROPIN_SRC)/espflix.cpp tests/dropin_main.cpp include/efx_player.hpp include/espflix_dropin/player.h espflix_amd/libefx.so
	g++ $(DROPIN_FLAGS) tests/dropin_main.cpp $(DROPIN_SRC)/espflix.cpp $(DROPIN_SRC)/streamer.cpp $(DROPIN_LIBS) -o $@
# the same unmodified code playing a 1008-picture synthetic stream: src/splash.h is shadowed by a generated header
tests/_build/dropin_long/splash.h: tests/make_dropin_long.py espflix_amd/gen/libefx_gen.so
	python3 tests/make_dropin_long.py tests/_build/dropin_long
tests/_build/espflix_dropin_long: tests/_build/dropin_long/splash.h $(DROPIN_SRC)/espflix.cpp tests/dropin_main.cpp include/efx_player.hpp espflix_amd/libefx.so
	g++ -Itests/_build/dropin_long $(DROPIN_FLAGS) tests/dropin_main.cpp $(DROPIN_SRC)/espflix.cpp $(DROPIN_SRC)/streamer.cpp $(DROPIN_LIBS) -o $@

# the C++ scaling harness over the multi-device C-ABI (efx_multi_*), RCCL for the gather of the chain hashes
scale: tools/efx_scale
tools/efx_scale: tools/efx_scale.cpp include/efx.h espflix_amd/libefx.so espflix_amd/gen/libefx_gen.so
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Iinclude tools/efx_scale.cpp -Lespflix_amd -lefx -Lespflix_amd/gen -lefx_gen \
	    -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN/../espflix_amd' -Wl,-rpath,'$$ORIGIN/../espflix_amd/gen' -Wl,-rpath,/opt/rocm/lib -o $@

oracle:
	$(MAKE) -C oracle port

# TEST: the slice parser of k_parse (csrc/parse_tm.h + the tables of csrc/efx_tables.cpp) compiled for the host and checked,
# record by record, against the parse trace of the test oracle (tests/test_parse_machine.py; no GPU)
harness: tests/_build/parse_harness
tests/_build/parse_harness: tests/parse_harness.cpp $(CSRC)/parse_tm.h $(CSRC)/efx_tables.cpp $(CSRC)/efx_internal.h oracle/efx_oracle.c oracle/efx_oracle.h
	mkdir -p tests/_build
	gcc -O2 -std=c99 -c oracle/efx_oracle.c -o tests/_build/efx_oracle.o
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 -Iinclude -I$(CSRC) -Ioracle -x hip tests/parse_harness.cpp $(CSRC)/efx_tables.cpp \
	    -x none tests/_build/efx_oracle.o -o $@
ref:
	$(MAKE) -C oracle ref

clean:
	rm -f $(CSRC)/*.o espflix_amd/libefx.so espflix_amd/gen/libefx_gen.so
	$(MAKE) -C oracle clean
