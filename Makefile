# Builds the product library (gfx950) and the CPU emulation library used by the non-GPU tests.
HIPCC ?= /opt/rocm/bin/hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
SRC := vame_amd/csrc/gru_seq.hip vame_amd/csrc/gemm.hip vame_amd/csrc/elementwise.hip vame_amd/csrc/prep.hip vame_amd/csrc/gru_coop.hip
HDR := vame_amd/csrc/vame_device.h vame_amd/csrc/vame_common.h vame_amd/csrc/gru_desc.h include/vame_hip.h

all: vame_amd/libvame_hip.so tests/emu/libvame_emu.so

vame_amd/libvame_hip.so: $(SRC) $(HDR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $@ $(SRC)

tests/emu/libvame_emu.so: $(SRC) $(HDR) tests/emu/hip_emu.h tests/emu/hip_emu.cpp
	$(HOSTCXX) -DVAME_EMU -O2 -std=c++17 -fPIC -shared -pthread -Itests/emu -Wno-unknown-attributes \
	    -o $@ $(foreach f,$(SRC),-x c++ $(f)) -x c++ tests/emu/hip_emu.cpp

# tuning build with all GEMM variants selectable through VAME_GEMM_VAR (tools/microbench.py ab)
ab: $(SRC) $(HDR)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVAME_GEMM_AB -DVAME_TUNING_BUILD -o tools/libvame_hip_ab.so $(SRC)
	@echo "use: VAME_LIB=tools/libvame_hip_ab.so python tools/microbench.py 10 gemm_ab"

probe: $(SRC) $(HDR) tools/probe_gemm.hip
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -DVAME_PROBE -Wno-unused-value -Wno-unused-result -o tools/probe_gemm tools/probe_gemm.hip vame_amd/csrc/elementwise.hip
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVAME_PROBE -o tools/libvame_hip_probe.so $(SRC)

clean:
	rm -f vame_amd/libvame_hip.so tests/emu/libvame_emu.so tools/libvame_hip_ab.so tools/libvame_hip_probe.so tools/probe_gemm
