# Build recipe for the HIP renderer (fray_amd/libfrayhip.so), the CPU oracle
# (oracle/libfray_oracle.so) and, when /root/reference is present, the partial reference build
# (oracle/_ref/libfray_ref.so).  __graft_entry__.build() runs `make all`.
MAKEFLAGS += -r
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
INC     := -Iinclude -Ifray_amd/csrc
# -ffp-contract=off: the reference build has no FMA contraction (x86-64 baseline); bit-exact hit
# records need the same on the device.
# -mllvm -disable-machine-licm: MachineLICM hoists the FP64 constants of the inlined polynomials (acos, sin / cos, atan2) out of the
# kernels' outer loops as VGPR pairs; the register allocator then spills them and every Horner step reloads one from scratch and waits
# for it.  Without the pass k_pt_bounce needs 146 VGPRs and no scratch (168 + 68 spilled with it): headline frame 135.8 -> 126.9 ms.
# -mllvm -wwm-regalloc=basic: the registers that hold SPILLED SGPRs (VGPR lanes written in whole-wave mode) are assigned by a pass of their own, and its
# default (greedy) allocator miscompiled round 3's Cube / CSG kernel variants: round 3's tree with SGPR spills in VGPR lanes renders 22 of 22 fuzz scenes wrong
# (10-20 % of the pixels of every path-traced frame), the same tree with this flag renders all of them right, as it does with the spills sent to memory
# (tools/repro/README.md).  Round 4's kernels no longer have what provoked it (out-of-line calls), but every big kernel here still spills SGPRs to lanes; the
# basic allocator costs nothing measurable (headline 95.1 vs 95.5 ms, forest 12.33 vs 12.34, boxed 8.27 vs 8.33, dragon Whitted 16.3 vs 16.1, bokeh 69.8 vs 68.4).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -disable-machine-licm -mllvm -wwm-regalloc=basic $(INC)
CXXFLAGS := -O2 -std=c++17 -fPIC -ffp-contract=off $(INC)

HOST_SRC := fray_amd/csrc/host_scene.cpp fray_amd/csrc/host_loaders.cpp fray_amd/csrc/host_exr.cpp fray_amd/csrc/capi_host.cpp
HOST_OBJ := $(HOST_SRC:.cpp=.o)
# render_variant.hip is compiled once per kernel flag word (render_impl<0..5, 8, 9>): eight independent translation
# units that `make -j` builds side by side
VARIANT_OBJ := $(foreach st,0 1 2 3 4 5 8 9,fray_amd/csrc/variant$(st).o)
# render_contract.hip: the path tracer's bounce / shadow kernels once more per flag word, built with fused multiply-adds (option "fp_contract")
CONTRACT_OBJ := $(foreach st,0 1 4 5 8 9,fray_amd/csrc/variantC$(st).o)
HIP_OBJ  := fray_amd/csrc/capi.o fray_amd/csrc/capi_comm.o $(VARIANT_OBJ) $(CONTRACT_OBJ)
HIP_HDR  := $(wildcard fray_amd/csrc/*.h) $(wildcard fray_amd/csrc/*.hpp) include/frayhip.h

all: fray_amd/libfrayhip.so oracle/libfray_oracle.so examples/fray_render examples/fray_render_mgpu tests/native/librccl_loopback.so ref

fray_amd/csrc/%.o: fray_amd/csrc/%.cpp $(HIP_HDR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

fray_amd/csrc/%.o: fray_amd/csrc/%.hip $(HIP_HDR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_HIPFLAGS) -c $< -o $@

# -Rpass-analysis=kernel-resource-usage: registers, spills, scratch and LDS of every kernel of the variant (kept
# next to the object; `make resources` gathers them into profiles/)
fray_amd/csrc/variant%.o: fray_amd/csrc/render_variant.hip $(HIP_HDR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_HIPFLAGS) -DFRAY_ST=$* -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> fray_amd/csrc/variant$*.resources.txt || (cat fray_amd/csrc/variant$*.resources.txt; false)

# (the last -ffp-contract on the command line wins)
fray_amd/csrc/variantC%.o: fray_amd/csrc/render_contract.hip $(HIP_HDR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA_HIPFLAGS) -ffp-contract=fast -DFRAY_ARITH=1 -DFRAY_ST=$* -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> fray_amd/csrc/variantC$*.resources.txt || (cat fray_amd/csrc/variantC$*.resources.txt; false)

fray_amd/libfrayhip.so: $(HOST_OBJ) $(HIP_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -ldl

oracle/libfray_oracle.so: oracle/fray_oracle.cpp include/frayhip.h
	$(CXX) $(CXXFLAGS) -shared -pthread -o $@ $<

# C++ host example over the C ABI (no Python, no torch)
examples/fray_render: examples/fray_render.cpp include/frayhip.h fray_amd/libfrayhip.so
	$(CXX) -O2 -std=c++17 -Iinclude $< -o $@ -Lfray_amd -lfrayhip -Wl,-rpath,'$$ORIGIN/../fray_amd' -Wl,-rpath,/opt/rocm/lib

# the multi-GPU frame from C++: one process per GPU, frayhip_gather_buckets (RCCL) as the exchange step
examples/fray_render_mgpu: examples/fray_render_mgpu.cpp include/frayhip.h fray_amd/libfrayhip.so
	$(CXX) -O2 -std=c++17 -Iinclude -I/opt/rocm/include $< -o $@ -Lfray_amd -lfrayhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../fray_amd' -Wl,-rpath,/opt/rocm/lib

# TEST INFRASTRUCTURE: a loopback stand-in for the RCCL entry points the library binds, so that frayhip_gather_buckets' world > 1 branch can
# execute on a one-GPU box (tests/test_gpu_gather_loopback.py names it in FRAYHIP_RCCL_LIBRARY; the product never loads it otherwise)
tests/native/librccl_loopback.so: tests/native/rccl_loopback.cpp
	$(CXX) -O2 -std=c++17 -fPIC -shared -I/opt/rocm/include $< -o $@ -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib

# Partial reference build: only when the reference tree is mounted (never on the GPU box).
ref:
	@if [ -d /root/reference/src ]; then $(MAKE) -C oracle -f Makefile.ref; else echo "reference tree absent: oracle/_ref not rebuilt"; fi

resources: $(VARIANT_OBJ)
	python3 tools/kernel_resources.py fray_amd/csrc/variant*.resources.txt

clean:
	rm -f fray_amd/csrc/*.o fray_amd/csrc/*.resources.txt fray_amd/libfrayhip.so oracle/libfray_oracle.so examples/fray_render examples/fray_render_mgpu tests/native/librccl_loopback.so
	rm -rf oracle/_ref

.PHONY: all ref clean resources
