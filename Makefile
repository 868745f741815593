# Build of the MI355X-native megapath framework.
#   make host    -> luisarender_amd/lib/liblrhost.so   (scene description, flattening, BVH, image IO)
#   make hip     -> luisarender_amd/lib/liblrhip.so    (HIP megakernel path tracer, C ABI include/lrhip.h)
#   make oracle  -> oracle/liboracle.so                (CPU checker; tests/bench only)
#   make cli     -> luisarender_amd/bin/luisa-render-cli + plugin libluisa-render-integrator-megapath.so
CXX      ?= g++
HIPCC    ?= /opt/rocm/bin/hipcc
CXXFLAGS ?= -std=c++17 -O2 -fPIC -Wall -Wextra
ORACLE_FLAGS ?= -std=c++17 -O3 -march=x86-64-v3 -ffp-contract=off -fPIC -Wall -Wextra -pthread
# fp32 division/sqrt use the hardware approximations (+6 %): the estimator is a Monte-Carlo sum compared with the
# oracle under a stated tolerance, not a bit-exact integer pipeline.  -fapprox-func drops the denormal-range rescaling
# around every v_rcp / v_sqrt / v_rsq (-18 % VALU instructions in the kernel, +5 % throughput) without touching
# the denormal mode.  Denormal flushing itself was measured and REJECTED: it moved the C2 image mean by 0.8 %
# against the oracle (tools/dbg_c2.py); these flags do not (tests/test_gpu_parity.py bias checks).
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func

LIBDIR := luisarender_amd/lib
BINDIR := luisarender_amd/bin
HOSTDIR := luisarender_amd/csrc/host
HIPDIR := luisarender_amd/csrc/hip

HOST_SRC := $(HOSTDIR)/sdl.cpp $(HOSTDIR)/scene.cpp $(HOSTDIR)/mesh_io.cpp $(HOSTDIR)/image_io.cpp $(HOSTDIR)/environment.cpp \
            $(HOSTDIR)/accel.cpp $(HOSTDIR)/host_api.cpp $(HOSTDIR)/luisa_render_shim.cpp
HOST_HDR := $(wildcard $(HOSTDIR)/*.h) $(wildcard include/*.h)
HIP_SRC := $(HIPDIR)/lrhip.hip
HIP_HDR := $(wildcard $(HIPDIR)/*.h) $(wildcard include/*.h)

.PHONY: all host hip oracle cli clean hip-variant
all: host oracle hip cli

host: $(LIBDIR)/liblrhost.so
$(LIBDIR)/liblrhost.so: $(HOST_SRC) $(HOST_HDR) Makefile
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -ldl -pthread -lz

oracle: oracle/liboracle.so
oracle/liboracle.so: oracle/oracle.cpp $(wildcard oracle/*.h) include/lr_scene.h Makefile
	$(CXX) $(ORACLE_FLAGS) -shared -o $@ oracle/oracle.cpp

hip: $(LIBDIR)/liblrhip.so
$(LIBDIR)/liblrhip.so: $(HIP_SRC) $(HIP_HDR) Makefile
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC)

# experimental kernel variants for A/B runs on the GPU box: make hip-variant NAME=w4 DEFS="-DLR_MIN_WAVES=4"
hip-variant:
	@mkdir -p $(LIBDIR)/variants
	$(HIPCC) $(HIPFLAGS) $(DEFS) -shared -o $(LIBDIR)/variants/liblrhip_$(NAME).so $(HIP_SRC)

cli: $(BINDIR)/luisa-render-cli
$(BINDIR)/luisa-render-cli: $(HOSTDIR)/cli.cpp $(HOSTDIR)/plugin_megapath.cpp $(LIBDIR)/liblrhost.so $(HOST_HDR)
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) -shared -o $(BINDIR)/libluisa-render-integrator-megapath.so $(HOSTDIR)/plugin_megapath.cpp \
	    -L$(LIBDIR) -llrhost -ldl -Wl,-rpath,'$$ORIGIN/../lib'
	$(CXX) $(CXXFLAGS) -o $@ $(HOSTDIR)/cli.cpp -L$(LIBDIR) -llrhost -ldl -pthread -Wl,-rpath,'$$ORIGIN/../lib'

clean:
	rm -f $(LIBDIR)/*.so $(BINDIR)/* oracle/liboracle.so
