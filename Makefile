# Builds the C-ABI shared library (hipcc, gfx950 only), the CPU oracle and,
# where /root/reference exists, the reference's own host converters.
#   make            -> spgpu_amd/lib/libspgpu.so + oracle/liboracle.so (+ oracle/_ref)
#   make lib|oracle|ref|clean
ROCM      ?= /opt/rocm
HIPCC     ?= $(ROCM)/bin/hipcc
ARCH      ?= gfx950
CSRC      := spgpu_amd/csrc
LIBDIR    := spgpu_amd/lib
BUILD     := build/obj

HIPFLAGS  := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wall -Wno-unused-function $(EXTRA_HIPFLAGS)
CFLAGS    := -O2 -fPIC -Iinclude -I$(CSRC) -Wall

HIP_SRCS  := $(wildcard $(CSRC)/*.hip)
C_SRCS    := $(wildcard $(CSRC)/*.c)
CPP_SRCS  := $(wildcard $(CSRC)/*.cpp)
OBJS      := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.hip.o,$(HIP_SRCS)) \
             $(patsubst $(CSRC)/%.c,$(BUILD)/%.c.o,$(C_SRCS)) \
             $(patsubst $(CSRC)/%.cpp,$(BUILD)/%.cpp.o,$(CPP_SRCS))
HDRS      := $(wildcard include/spgpu/*.h) $(wildcard $(CSRC)/*.h)

.PHONY: all lib oracle ref tools clean
all: lib oracle ref tools

lib: $(LIBDIR)/libspgpu.so

$(LIBDIR)/libspgpu.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -o $@ $(OBJS)

$(BUILD)/%.hip.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c $< -o $@

# hell_spmm: the SLP vectorizer pairs fp32 sums of DIFFERENT rows into v_pk_fma_f32 and shuffles operands to feed
# them; that costs ~40 VGPRs (and scratch in the strip kernel) for flops an HBM-bound kernel does not need.
FLAGS_hell_spmm := -fno-slp-vectorize

# Host C/C++ goes through hipcc as well (plain clang for these files): one toolchain.
$(BUILD)/%.c.o: $(CSRC)/%.c $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) -x c $(CFLAGS) -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -c $< -o $@

$(BUILD)/%.cpp.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) -x c++ $(CFLAGS) -std=c++17 -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -c $< -o $@

# Plain-C callers of the C ABI (gcc, no hipcc): the reference's ctest.c / hellPerf.cpp / diaPerf.cpp flows and a CG solver.
TOOLS := tools/ctest_amd.bin tools/hellperf_amd.bin tools/diaperf_amd.bin tools/cg_amd.bin
tools: lib $(TOOLS)
tools/%.bin: tools/%.c $(HDRS) $(LIBDIR)/libspgpu.so
	gcc -O2 -std=gnu99 -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -Iinclude $< -L$(LIBDIR) -lspgpu -L$(ROCM)/lib -lamdhip64 -lm \
	    -Wl,-rpath,'$$ORIGIN/../spgpu_amd/lib' -Wl,-rpath,$(ROCM)/lib -o $@

oracle:
	$(MAKE) -C oracle liboracle.so

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf build $(LIBDIR)/libspgpu.so oracle/liboracle.so oracle/_ref tools/*.bin
