# Build of the MI355X-native drprg predict hot path.
#   make            -> drprg_amd/lib/libdrprg_hip.so, drprg_amd/bin/pandora, oracle/liboracle.so
# hipcc cross-compiles gfx950 without a GPU present.
HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
ARCH     ?= gfx950
CXXFLAGS := -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Iinclude
HIPFLAGS := $(CXXFLAGS) --offload-arch=$(ARCH)

SRC  := drprg_amd/csrc
OBJD := build/obj
HOST_SRCS := prg.cpp kmergraph.cpp index.cpp fastx.cpp genotype.cpp params.cpp denovo.cpp mapper.cpp capi.cpp vcfio.cpp bcfout.cpp annotate.cpp report_json.cpp ingest.cpp pgunzip.cpp rccl_dyn.cpp pack.cpp
HIP_SRCS := sketch_probe.hip sketch_wave.hip sketch_filter.hip candidates.hip read_cluster.hip cluster.hip anchor_scan.hip packed.hip
LIB  := drprg_amd/lib/libdrprg_hip.so
# make EXPERIMENTAL=1: the kernel forms that were built, are bit-exact and lost their measurement -- the wave form of
# read_cluster, refine_kernel, the in-kernel clustering of sketch_wave_kernel, the read-by-read verification -- compiled in behind their
# environment switches, into a library of its own (build/exp/libdrprg_hip.so; DRPRG_HIP_LIB=... selects it, pytest -m "gpu and
# experimental" runs their parity cases).  The default build does not contain them.
EXPERIMENTAL ?= 0
# EXTRA_DEFS: build-time knobs for measurement builds (tools/rc_variants.sh), e.g. make OBJD=build/obj_x LIB=build/x/libdrprg_hip.so EXTRA_DEFS=-DDRPRG_RC_PER=4
EXTRA_DEFS ?=
CXXFLAGS += $(EXTRA_DEFS)
HIPFLAGS += $(EXTRA_DEFS)
ifeq ($(EXPERIMENTAL),1)
CXXFLAGS += -DDRPRG_EXPERIMENTAL=1
HIPFLAGS += -DDRPRG_EXPERIMENTAL=1
HIP_SRCS += read_cluster_wave.hip read_verify.hip
OBJD := build/obj_exp
LIB  := build/exp/libdrprg_hip.so
endif
OBJS := $(addprefix $(OBJD)/,$(HOST_SRCS:.cpp=.o)) $(addprefix $(OBJD)/,$(HIP_SRCS:.hip=.o))
BIN  := drprg_amd/bin/pandora
ORACLE := oracle/liboracle.so

DRPRG := drprg_amd/bin/drprg

ifeq ($(EXPERIMENTAL),1)
all: $(LIB)
else
all: $(LIB) $(BIN) $(DRPRG) $(ORACLE)
endif

experimental:
	$(MAKE) EXPERIMENTAL=1

$(DRPRG): $(SRC)/drprg_main.cpp $(LIB)
	@mkdir -p $(dir $@)
	$(HIPCC) $(CXXFLAGS) -x c++ $< -o $@ -Ldrprg_amd/lib -ldrprg_hip -Wl,-rpath,'$$ORIGIN/../lib'

$(OBJD)/%.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.h) include/drprg_hip.h
	@mkdir -p $(OBJD)
	$(HIPCC) $(CXXFLAGS) -x hip --offload-arch=$(ARCH) -c $< -o $@

$(OBJD)/%.o: $(SRC)/%.hip $(wildcard $(SRC)/*.h) include/drprg_hip.h
	@mkdir -p $(OBJD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -lz -ldl

$(BIN): $(SRC)/pandora_main.cpp $(LIB)
	@mkdir -p $(dir $@)
	$(HIPCC) $(CXXFLAGS) -x c++ $< -o $@ -Ldrprg_amd/lib -ldrprg_hip -Wl,-rpath,'$$ORIGIN/../lib'

$(ORACLE): oracle/oracle.c oracle/oracle_index.c oracle/oracle_params.c oracle/oracle_vcf.c oracle/oracle_index.h
	$(CC) -O2 -fPIC -shared -Wall -o $@ oracle/oracle.c oracle/oracle_index.c oracle/oracle_params.c oracle/oracle_vcf.c -lm

clean:
	rm -rf build $(LIB) $(BIN) $(ORACLE)


.PHONY: all clean experimental

# host code under AddressSanitizer + UBSan (CPU build only: GPU sanitizers are not available on the pool):
#   make asan && LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) \
#     ASAN_OPTIONS=detect_leaks=0 DRPRG_HIP_LIB=build/asan/libdrprg_hip.so python -m pytest tests -m "not gpu" -s
ASAN_FLAGS := -O1 -g -std=c++17 -fPIC -Iinclude -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer
asan:
	@mkdir -p build/asan
	for f in $(HOST_SRCS:.cpp=); do $(HIPCC) $(ASAN_FLAGS) -x hip --offload-arch=$(ARCH) -c $(SRC)/$$f.cpp -o build/asan/$$f.o || exit 1; done
	for f in $(HIP_SRCS:.hip=); do $(HIPCC) $(ASAN_FLAGS) --offload-arch=$(ARCH) -c $(SRC)/$$f.hip -o build/asan/$$f.o || exit 1; done
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -fsanitize=address,undefined -fno-gpu-sanitize -o build/asan/libdrprg_hip.so build/asan/*.o -lz -ldl

.PHONY: asan
