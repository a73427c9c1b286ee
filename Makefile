# Builds the product library (gfx950) and the CPU emulation library used by the non-GPU tests.
# Objects are compiled per source file (make -j parallelises; only changed files rebuild).
HIPCC ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
NAMES := gru_seq gemm elementwise prep gru_coop hmm gru_wide heads
SRC := $(foreach n,$(NAMES),vame_amd/csrc/$(n).hip)
HDR := vame_amd/csrc/gru_wide_loop.inc vame_amd/csrc/vame_device.h vame_amd/csrc/vame_common.h vame_amd/csrc/gru_desc.h include/vame_hip.h
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC
EMUFLAGS := -DVAME_EMU -O2 -std=c++17 -fPIC -pthread -Itests/emu -Wno-unknown-attributes

# identity of the kernel sources, compiled into the library (vame_source_id()): bench.py only quotes PMC traffic that was collected
# with a library built from the same sources (the .so bytes themselves differ between build directories)
SRC_ID := $(shell cat $(SRC) $(HDR) | sha256sum | cut -c1-64)

all: vame_amd/libvame_hip.so tests/emu/libvame_emu.so

build/hip/elementwise.o build/ab/elementwise.o build/probe/elementwise.o: HIPFLAGS += -DVAME_SRC_ID=\"$(SRC_ID)\"
build/hip/elementwise.o build/ab/elementwise.o build/probe/elementwise.o: $(SRC)

build/hip/%.o: vame_amd/csrc/%.hip $(HDR)
	@mkdir -p build/hip
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

vame_amd/libvame_hip.so: $(foreach n,$(NAMES),build/hip/$(n).o)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^

build/emu/%.o: vame_amd/csrc/%.hip $(HDR) tests/emu/hip_emu.h
	@mkdir -p build/emu
	$(HOSTCXX) $(EMUFLAGS) -c -o $@ -x c++ $<

build/emu/hip_emu.o: tests/emu/hip_emu.cpp tests/emu/hip_emu.h
	@mkdir -p build/emu
	$(HOSTCXX) $(EMUFLAGS) -c -o $@ $<

tests/emu/libvame_emu.so: $(foreach n,$(NAMES),build/emu/$(n).o) build/emu/hip_emu.o
	$(HOSTCXX) -shared -fPIC -pthread -o $@ $^

# tuning build with all GEMM variants selectable through VAME_GEMM_VAR / VAME_GEMM_EPI and the GRU ablation masks
# (tools/microbench.py gemm_ab / gemm_epi / gemm_sk / ablate)
build/ab/%.o: vame_amd/csrc/%.hip $(HDR) tools/gemm_split_variants.inc
	@mkdir -p build/ab
	$(HIPCC) $(HIPFLAGS) -DVAME_GEMM_AB -DVAME_TUNING_BUILD -c -o $@ $<

tools/libvame_hip_ab.so: $(foreach n,$(NAMES),build/ab/$(n).o)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^

ab: tools/libvame_hip_ab.so
	@echo "use: VAME_LIB=tools/libvame_hip_ab.so python tools/microbench.py 10 gemm_ab"

build/probe/%.o: vame_amd/csrc/%.hip $(HDR)
	@mkdir -p build/probe
	$(HIPCC) $(HIPFLAGS) -DVAME_PROBE -c -o $@ $<

probe: $(foreach n,$(NAMES),build/probe/$(n).o) tools/probe_gemm.hip
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o tools/libvame_hip_probe.so $(foreach n,$(NAMES),build/probe/$(n).o)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -DVAME_PROBE -Wno-unused-value -Wno-unused-result -c -o build/probe/probe_gemm_main.o tools/probe_gemm.hip
	$(HIPCC) --offload-arch=gfx950 -o tools/probe_gemm build/probe/probe_gemm_main.o build/probe/elementwise.o

clean:
	rm -rf build vame_amd/libvame_hip.so tests/emu/libvame_emu.so tools/libvame_hip_ab.so tools/libvame_hip_probe.so tools/probe_gemm
