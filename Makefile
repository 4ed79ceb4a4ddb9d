# Build of the MI355X path-tracing core and its host side.
#
#   make hpt      cudapathtracer_amd/libhpt.so        HIP kernels + C ABI (include/hpt.h)
#   make host     cudapathtracer_amd/libhpt_host.so   C++ host API (Scene/BVHAccel/Material/HipRenderer)
#                 build/gpu_path_tracer               the CLI
#   make count    cudapathtracer_amd/libhpt_count.so  diagnostic build of libhpt.so that counts the box and
#                                                      triangle tests the kernels execute (tools/count_tests.py)
#   make oracle   oracle/liboracle.so                 CPU restatement (test infrastructure)
#   make ref      oracle/_ref/libref.so               the compiled reference (needs /root/reference)
#
# hipcc cross-compiles for gfx950 without a GPU present.

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
# -ffp-contract=off: the device arithmetic is specified operation by operation
# (bit-exact against a CPU evaluation); contraction into FMA would change results.
# -fno-slp-vectorize: the SLP vectoriser pairs adjacent float adds/multiplies into packed-f32
# instructions (v_pk_add_f32 / v_pk_mul_f32) plus register shuffles; in these instruction mixes
# that measured slower on gfx950 than the scalar forms (tools/micro/pk_mix.hip; 7 % on the
# single-kernel render) and needs more registers.
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Wall -Wno-unused-function
CXXFLAGS ?= -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall

PKG   := cudapathtracer_amd
CSRC  := $(PKG)/csrc
HOST  := $(PKG)/host

HOST_SRCS := $(HOST)/BVH.cpp $(HOST)/Material.cpp $(HOST)/Scene.cpp $(HOST)/SceneFile.cpp $(HOST)/Triangle.cpp $(HOST)/Utility.cpp \
             $(HOST)/HipRenderer.cpp $(HOST)/host_capi.cpp
HOST_HDRS := $(wildcard $(HOST)/*.hpp) include/hpt.h include/hpt_host.h

.PHONY: all hpt host oracle ref clean count
all: hpt host oracle

hpt: $(PKG)/libhpt.so
$(PKG)/libhpt.so: $(CSRC)/hpt_core.hip $(CSRC)/hpt_bvh_build.hip $(CSRC)/hpt_device.h include/hpt.h Makefile
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/hpt_core.hip $(CSRC)/hpt_bvh_build.hip

count: $(PKG)/libhpt_count.so
$(PKG)/libhpt_count.so: $(CSRC)/hpt_core.hip $(CSRC)/hpt_bvh_build.hip $(CSRC)/hpt_device.h include/hpt.h Makefile
	$(HIPCC) $(HIPFLAGS) -DHPT_COUNT_TESTS -shared -o $@ $(CSRC)/hpt_core.hip $(CSRC)/hpt_bvh_build.hip

host: $(PKG)/libhpt_host.so build/gpu_path_tracer
$(PKG)/libhpt_host.so: $(HOST_SRCS) $(HOST_HDRS) $(PKG)/libhpt.so
	$(CXX) $(CXXFLAGS) -shared -o $@ -Iinclude -I$(HOST) $(HOST_SRCS) -L$(PKG) -lhpt -Wl,-rpath,'$$ORIGIN'

build/gpu_path_tracer: $(HOST)/gpu_path_tracer_main.cpp $(PKG)/libhpt_host.so
	mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ -Iinclude -I$(HOST) $(HOST)/gpu_path_tracer_main.cpp -L$(PKG) -lhpt_host -lhpt \
	    -Wl,-rpath,'$$ORIGIN/../$(PKG)'

oracle:
	$(MAKE) -C oracle oracle
ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(PKG)/libhpt.so $(PKG)/libhpt_count.so $(PKG)/libhpt_host.so build
	$(MAKE) -C oracle clean
