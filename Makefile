# Top-level build: everything lands in-tree so it travels to the GPU box.
#   make            all of the below
#   make hip        fast-feedback-service_amd/libffs_hip.so   (hipcc, gfx950)
#   make experiments  fast-feedback-service_amd/libffs_hip_exp.so (the same + timing experiments, tools/ only)
#   make synth      fast-feedback-service_amd/libffs_synth.so (gcc)
#   make cli        fast-feedback-service_amd/bin/spotfinder  (g++, links libffs_hip)
#   make oracle     oracle/liboracle.so (+ oracle/_ref when /root/reference exists)
PKG     := fast-feedback-service_amd
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
CXX     ?= g++
ARCH    ?= gfx950
HIPFLAGS = -std=c++20 -O3 --offload-arch=$(ARCH) -fPIC -Iinclude -I$(PKG)/csrc \
           -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function
HIP_SRCS := $(wildcard $(PKG)/csrc/*.hip)
HIP_HDRS := $(wildcard $(PKG)/csrc/*.h) $(wildcard $(PKG)/csrc/*.hpp) include/ffs_hip.h
HIP_OBJS := $(patsubst $(PKG)/csrc/%.hip,$(PKG)/csrc/obj/%.o,$(HIP_SRCS))
EXP_OBJS := $(patsubst $(PKG)/csrc/%.hip,$(PKG)/csrc/obj/%.exp.o,$(HIP_SRCS))

all: oracle synth hip cli

oracle:
	$(MAKE) -C oracle

synth: $(PKG)/libffs_synth.so
$(PKG)/libffs_synth.so: $(PKG)/host/ffs_synth.c include/ffs_synth.h
	$(CC) -std=c11 -O2 -ffp-contract=off -fPIC -shared -Iinclude -o $@ $< -lm

# one object per translation unit (ffs_internal.hpp has the map), so that `make -j` compiles them side by side
hip:
	$(MAKE) -j5 $(PKG)/libffs_hip.so
$(PKG)/csrc/obj/%.o: $(PKG)/csrc/%.hip $(HIP_HDRS)
	@mkdir -p $(PKG)/csrc/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(PKG)/libffs_hip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS)

# the same library with the timing experiments compiled in (phases switched off, kernels stopped half way: results are
# WRONG when they are used).  For tools/ only: FFS_HIP_LIB=.../libffs_hip_exp.so FFS_EXP_K1_DEBUG=1 python tools/...
experiments:
	$(MAKE) -j5 $(PKG)/libffs_hip_exp.so
$(PKG)/csrc/obj/%.exp.o: $(PKG)/csrc/%.hip $(HIP_HDRS)
	@mkdir -p $(PKG)/csrc/obj
	$(HIPCC) $(HIPFLAGS) -DFFS_EXPERIMENTS $(EXPFLAGS) -c -o $@ $<
$(PKG)/libffs_hip_exp.so: $(EXP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(EXP_OBJS)

# HDF5 (NXmx reader) is optional: used when its headers are found
HDF5_PREFIX ?= /opt/conda
ifneq ($(wildcard $(HDF5_PREFIX)/include/hdf5.h),)
H5FLAGS := -DFFS_HAVE_HDF5 -I$(HDF5_PREFIX)/include
H5LIBS  := $(HDF5_PREFIX)/lib/libhdf5.so -Wl,--enable-new-dtags,-rpath,$(HDF5_PREFIX)/lib
endif
# The reader lives in its own shared object so that only IT carries the HDF5 prefix as RUNPATH
# (a conda prefix also ships an older libstdc++ that must not shadow the system one for HIP).
H5_SRCS := $(PKG)/host/h5_reader.cc $(PKG)/host/h5_writer.cc
$(PKG)/libffs_h5.so: $(H5_SRCS) $(PKG)/host/reader.hpp $(PKG)/host/codecs.hpp
	$(CXX) -std=c++20 -O2 -fPIC -shared $(H5FLAGS) -I$(PKG)/host -o $@ $(H5_SRCS) $(H5LIBS)
CLI_SRCS := $(filter-out $(H5_SRCS),$(wildcard $(PKG)/host/*.cc))
CLI_HDRS := $(wildcard $(PKG)/host/*.hpp) $(wildcard include/*.h)
cli: $(PKG)/bin/spotfinder $(PKG)/bin/ffs_hosttool
$(PKG)/bin/ffs_hosttool: $(PKG)/tools/ffs_hosttool.cc $(PKG)/host/readers.cc $(CLI_HDRS) $(PKG)/libffs_synth.so $(PKG)/libffs_h5.so
	mkdir -p $(PKG)/bin && $(CXX) -std=c++20 -O2 -Iinclude -I$(PKG)/host -o $@ $(PKG)/tools/ffs_hosttool.cc \
	    $(PKG)/host/readers.cc -L$(PKG) -lffs_synth -lffs_h5 -Wl,-rpath,'$$ORIGIN/..'

$(PKG)/bin/spotfinder: $(CLI_SRCS) $(CLI_HDRS) $(PKG)/libffs_hip.so $(PKG)/libffs_synth.so $(PKG)/libffs_h5.so
	@if [ -n "$(CLI_SRCS)" ]; then mkdir -p $(PKG)/bin && \
	$(CXX) -std=c++20 -O2 -Iinclude -I$(PKG)/host -o $@ $(CLI_SRCS) \
	    -L$(PKG) -lffs_hip -lffs_synth -lffs_h5 -Wl,-rpath,'$$ORIGIN/..' -lpthread -ldl && \
	ln -sf spotfinder $(PKG)/bin/spotfinder32; \
	else echo "no CLI sources yet"; fi

clean:
	rm -f $(PKG)/libffs_hip.so $(PKG)/libffs_hip_exp.so $(PKG)/libffs_synth.so $(PKG)/libffs_h5.so
	rm -rf $(PKG)/csrc/obj
	rm -rf $(PKG)/bin
	$(MAKE) -C oracle clean

.PHONY: all oracle synth hip cli clean experiments
