# Build recipes (no cmake): the gfx950 product library and the host-side kernel emulator used by tests.
HIPCC ?= hipcc
HOSTCXX ?= /opt/rocm/lib/llvm/bin/clang++
CSRC := gigagan_pytorch_amd/csrc
SRCS := $(CSRC)/gg_api.hip
HDRS := $(wildcard $(CSRC)/*.h) include/gigagan_amd.h

all: hip emu

hip: gigagan_pytorch_amd/libgigagan_amd.so
emu: tests/emu/libgigagan_amd_emu.so

gigagan_pytorch_amd/libgigagan_amd.so: $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -shared -fPIC $(SRCS) -o $@

tests/emu/libgigagan_amd_emu.so: $(SRCS) $(HDRS) tests/emu/gg_emu.cpp tests/emu/gg_device_emu.h
	$(HOSTCXX) -x c++ -std=c++17 -O2 -Wno-psabi -DGG_HOST_EMULATION -Itests/emu -I$(CSRC) -shared -fPIC $(SRCS) tests/emu/gg_emu.cpp -o $@

clean:
	rm -f gigagan_pytorch_amd/libgigagan_amd.so tests/emu/libgigagan_amd_emu.so
