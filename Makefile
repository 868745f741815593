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
# -fno-slp-vectorize (round 2): the SLP vectorizer pairs scalar fp32 operations into v_pk_* instructions.  On gfx950 a wave64
# v_pk_fma_f32 issues in ~4.2 cycles against ~2.4 for a v_fma_f32 (tools/valu_peak.hip), so a pair gains little, and the packing
# costs v_mov shuffles into consecutive registers plus register pressure.  Off: C2 689 -> 774 Msamples/s (+12 %), C3 651 -> 758,
# C4 683 -> 747, C5 255 -> 276 at the A/B sizes (profiles/archive/r02f_ab_compiler_flags.txt).  The slab test's hand-written v2f FMAs stay.
# Same arithmetic per lane.  -fno-vectorize (loop vectorizer), -O2, -fno-unroll-loops, relaxed-occupancy scheduling: no change.
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func -fno-slp-vectorize

LIBDIR := luisarender_amd/lib
BINDIR := luisarender_amd/bin
HOSTDIR := luisarender_amd/csrc/host
HIPDIR := luisarender_amd/csrc/hip

HOST_SRC := $(HOSTDIR)/sdl.cpp $(HOSTDIR)/scene.cpp $(HOSTDIR)/mesh_io.cpp $(HOSTDIR)/subdiv.cpp $(HOSTDIR)/catmull_clark.cpp $(HOSTDIR)/image_io.cpp $(HOSTDIR)/image_codecs.cpp $(HOSTDIR)/environment.cpp \
            $(HOSTDIR)/accel.cpp $(HOSTDIR)/host_api.cpp $(HOSTDIR)/luisa_render_shim.cpp
HOST_HDR := $(wildcard $(HOSTDIR)/*.h) $(wildcard include/*.h)
HIP_SRC := $(HIPDIR)/lrhip.hip
HIP_HDR := $(wildcard $(HIPDIR)/*.h) $(wildcard include/*.h)

.PHONY: all host hip oracle cli clean hip-variant variant-lib ref ieee shallow
all: host oracle hip cli ieee shallow

# oracle/_ref: the reference's OWN sources compiled in place against the scalar LuisaCompute stand-in of oracle/ref_shim
# (test infrastructure: pins oracle/ to the reference; needs /root/reference, so only where the reference tree exists)
ref:
	$(MAKE) -f oracle/Makefile.ref ref

host: $(LIBDIR)/liblrhost.so
$(LIBDIR)/liblrhost.so: $(HOST_SRC) $(HOST_HDR) Makefile
	@mkdir -p $(LIBDIR)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -ldl -pthread -lz

oracle: oracle/liboracle.so oracle/liboracle_fma.so
oracle/liboracle.so: oracle/oracle.cpp $(wildcard oracle/*.h) include/lr_scene.h Makefile
	$(CXX) $(ORACLE_FLAGS) -shared -o $@ oracle/oracle.cpp
# the same oracle with fused multiply-adds allowed: TEST INFRASTRUCTURE for the one test that measures how far contraction alone moves
# the oracle's own frames (the lamp-lit fog case of the volumetric integrator, tests/test_gpu_parity.py); never a reference for anything else
oracle/liboracle_fma.so: oracle/oracle.cpp $(wildcard oracle/*.h) include/lr_scene.h Makefile
	$(CXX) $(ORACLE_FLAGS) -ffp-contract=fast -shared -o $@ oracle/oracle.cpp

# The megakernel is precompiled for a curated set of feature masks (csrc/hip/variants.h), one object per mask so
# that they build in parallel (make -j).  VARIANT_MASKS may be narrowed for experiments (a missing variant is a
# run-time error of lrhip_render, never a fallback).
VARIANT_MASKS ?= 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 60 61 62 63 124 125 126 127 636 637 638 639 252 253 254 255 256 257 258 259 \
                 1024 1025 1026 1027 1028 1029 1030 1031 1032 1033 1034 1035 1036 1037 1038 1039 \
                 3072 3073 3074 3075 3076 3077 3078 3079 3080 3081 3082 3083 3084 3085 3086 3087 \
                 4096 4097 4098 4099 4100 4101 4102 4103 4104 4105 4106 4107 4108 4109 4110 4111 4112 4113 4114 4115 4116 4117 4118 4119 \
                 5120 5121 5122 5123 5124 5125 5126 5127 5128 5129 5130 5131 5132 5133 5134 5135 \
                 7168 7169 7170 7171 7172 7173 7174 7175 7176 7177 7178 7179 7180 7181 7182 7183 \
                 8208 8209 8210 8211 8212 8213 8214 8215 12304 12305 12306 12307 12308 12309 12310 12311 \
                 20482 20483 20486 20487 20490 20491 20494 20495 20498 20499 20502 20503 28690 28691 28694 28695 \
                 21506 21507 21514 21515 23554 23555 23562 23563
# the heavy-closure kernels of wavefront mode (csrc/hip/heavy_kernel.h; mask: 1 counters, 2 generic sampler, 4 Mix / 8 Layered instead of
# Disney, 512 nested Mix / Layered)
HEAVY_MASKS ?= 0 1 2 3 4 5 6 7 8 9 10 11 516 517 518 519 520 521 522 523
OBJDIR := $(LIBDIR)/obj
VARIANT_OBJ := $(foreach m,$(VARIANT_MASKS),$(OBJDIR)/variant_$(m).o) $(foreach m,$(HEAVY_MASKS),$(OBJDIR)/heavy_$(m).o)

hip: $(LIBDIR)/liblrhip.so
# The volumetric megakernel (variants 256+) is built with IEEE arithmetic: no fp contraction, correctly rounded division / sqrt, no
# approximate functions.  MegaVPTNaive is chaotic where the reference's algorithm puts a ray origin ON a surface (after a medium
# "hit surface" event, src/media/homogeneous.cpp:64, the next shadow segment starts in the surface it just reached): whether that
# segment re-hits the surface is decided by the last bit, and one flipped decision desynchronises the path's PCG32 stream for
# good.  With the arithmetic of the reference-pinned oracle the device takes the same decisions (tests/test_ref_golden.py,
# test_gpu_parity.py::test_volumetric_megakernel); with contraction it renders the lamp-lit fog scenes 30 % darker than the
# reference's own code does (measured, round 2).  It is a feature row, not the benchmarked path.
VPT_HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -DLR_EXACT_LEAF=1
# EVERY variant is built with -mllvm -amdgpu-spill-sgpr-to-vgpr=0.  With the default (SGPRs spilled into lanes of a VGPR) the kernels that
# make REAL CALLS (out-of-line closures: every mask with Mix, Layered or the volumetric kernel) come out miscompiled or not depending on
# unrelated code, flags and the register budget: NaN / lost samples in <124> at 3 waves per SIMD in round 1, fine after round 2's changes,
# broken again without the SLP vectorizer, <124> (not <125>) non-deterministic at 4 waves after one more change -- each time
# bit-identical to the good builds with this flag.  Some scenes pass on a broken binary (the kitchen parity test did), so tests cannot
# establish that a fast build is sound; the flag removes the mechanism.  Cost: kitchen stand-in 275 -> 256 Msamples/s, <60> 350 -> 298
# (profiles/archive/r02f_ab_compiler_flags.txt).  Round 4: the LEAN variants too -- since round 3 they make a real call of their own
# (the texture callback of load_lobe, dev_math.h: LR_TEX_LAMBDA, is out of line) and spill SGPRs, i.e. they hold exactly the
# ingredients; no binary of theirs was ever caught wrong (profiles/archive/r03s_sgpr_spill_repro.txt: bit-identical with and without the
# flag), but the same was true of <124> for two rounds.  Cost, same box: C2 782.3 -> 777.0, C3 809.6 -> 803.4, C4 875.6 -> 876.2
# Msamples/s at 64 spp, films bit-identical (profiles/r04_ab_call_safe_lean.txt).
CALL_SAFE_FLAGS ?= -mllvm -amdgpu-spill-sgpr-to-vgpr=0
variant_flags = $(if $(filter 256 257 258 259,$(1)),$(VPT_HIPFLAGS),$(HIPFLAGS)) $(CALL_SAFE_FLAGS)
$(OBJDIR)/variant_%.o: $(HIPDIR)/megapath_variant.hip $(HIP_HDR) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(call variant_flags,$*) -DLR_VARIANT=$* -c -o $@ $(HIPDIR)/megapath_variant.hip
# (they make real calls -- the closure interpreters stay out of line -- so they take the same safe flag)
HEAVY_DEFS ?=
$(OBJDIR)/heavy_%.o: $(HIPDIR)/heavy_variant.hip $(HIP_HDR) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(CALL_SAFE_FLAGS) $(HEAVY_DEFS) -DLR_HVARIANT=$* -c -o $@ $(HIPDIR)/heavy_variant.hip
$(OBJDIR)/lrhip.o: $(HIP_SRC) $(HIP_HDR) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(HIP_SRC)
$(LIBDIR)/liblrhip.so: $(OBJDIR)/lrhip.o $(VARIANT_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -o $@ $^

# experimental kernel builds for A/B runs on the GPU box:
#   make hip-variant NAME=w3 DEFS="-DLR_WAVES_HEAVY=3" [VARIANT_MASKS="0 1"]
VOBJDIR := $(LIBDIR)/variants/obj_$(NAME)
hip-variant:
	@mkdir -p $(VOBJDIR)
	$(MAKE) --no-print-directory OBJDIR=$(VOBJDIR) HIPFLAGS='$(HIPFLAGS) $(DEFS)' VARIANT_MASKS='$(VARIANT_MASKS)' HEAVY_MASKS='$(HEAVY_MASKS)' \
	    LIBDIR_OUT=$(LIBDIR)/variants/liblrhip_$(NAME).so variant-lib
variant-lib: $(OBJDIR)/lrhip.o $(VARIANT_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -o $(LIBDIR_OUT) $^

# The lean kernel <0> / <1> once more with the oracle's arithmetic (no fp contraction, correctly rounded division / sqrt, exact
# functions, 1 / det in the triangle test): TEST INFRASTRUCTURE (MegaPathRenderer(lib_path=...)).  Round 3 used it to test the claim
# that fast math is what separates the device from the oracle on the C2 stand-in: only where both sides intersect identical vertices
# (tests/test_gpu_parity.py::test_what_separates_c2_from_the_oracle).
ieee: $(LIBDIR)/variants/liblrhip_ieee.so
$(LIBDIR)/variants/liblrhip_ieee.so: $(HIP_SRC) $(HIPDIR)/megapath_variant.hip $(HIP_HDR) Makefile
	$(MAKE) --no-print-directory hip-variant NAME=ieee HIPFLAGS='$(VPT_HIPFLAGS) -DLR_EXACT_LEAF=1' DEFS= VARIANT_MASKS='0 1' HEAVY_MASKS=

# The lean kernels of both schedulers once more with a FOUR-entry LDS traversal stack: every ray of every scene goes through the HBM
# overflow area of the stack (dev_trace.h: TraversalStack::push / pop beyond kStackLds, the wave-level `deep` paths of the node step,
# and the words a pool kernel parks on top of a lane's stack across the shading block).  TEST INFRASTRUCTURE: its frames must equal the
# shipped library's bit for bit (tests/test_gpu_pool.py::test_the_overflow_area_of_the_traversal_stack).
shallow: $(LIBDIR)/variants/liblrhip_shallow.so
$(LIBDIR)/variants/liblrhip_shallow.so: $(HIP_SRC) $(HIPDIR)/megapath_variant.hip $(HIP_HDR) Makefile
	$(MAKE) --no-print-directory hip-variant NAME=shallow DEFS='-DLR_STACK_LDS=4' VARIANT_MASKS='0 1 4096 4097' HEAVY_MASKS=

cli: $(BINDIR)/luisa-render-cli
$(BINDIR)/luisa-render-cli: $(HOSTDIR)/cli.cpp $(HOSTDIR)/plugin_megapath.cpp $(LIBDIR)/liblrhost.so $(HOST_HDR)
	@mkdir -p $(BINDIR)
	for impl in megapath direct normal megavptnaive; do \
	  $(CXX) $(CXXFLAGS) -DLR_PLUGIN_IMPL=\"$$impl\" -shared -o $(BINDIR)/libluisa-render-integrator-$$impl.so $(HOSTDIR)/plugin_megapath.cpp \
	    -L$(LIBDIR) -llrhost -ldl -Wl,-rpath,'$$ORIGIN/../lib' || exit 1; done
	$(CXX) $(CXXFLAGS) -o $@ $(HOSTDIR)/cli.cpp -L$(LIBDIR) -llrhost -ldl -pthread -Wl,-rpath,'$$ORIGIN/../lib'

clean:
	rm -rf $(LIBDIR)/*.so $(LIBDIR)/obj $(LIBDIR)/variants $(BINDIR)/* oracle/liboracle.so
