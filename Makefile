# Builds the gfx950 engine (libvehicle_pm_gpu.so) and the CPU oracle (test infrastructure).
# The same commands are issued by __graft_entry__.build().
#   device code (csrc/*.hip)  -> hipcc --offload-arch=gfx950, one object per translation unit
#   host code   (host/*.cpp)  -> $(CXX): it sees the C ABI (include/pm/*.h) only, never a HIP header
#   link                      -> hipcc -shared (pulls in the HIP runtime)
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
PKG     := ocean-perception_amd
OBJDIR  := $(PKG)/build
MAKEFLAGS += -j8
# -ffp-contract=off: every float op is a single IEEE rounding, on device and host, so results are
# bit-identical to the CPU path (hipcc's default is fp-contract=fast).
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wall -Wextra -Wno-unused-parameter -Wno-pass-failed
CXXFLAGS ?= -O2 -std=c++17 -fPIC -ffp-contract=off -Wall -Wextra
LIB     := $(PKG)/lib/libvehicle_pm_gpu.so
HIP_UNITS  := pm_engine pm_launch pm_sweeps pm_seed pm_planes_host pm_hostpath pm_tile pm_tiled pm_imaging
HOST_UNITS := patchmatch_gpu imaging dataset jpeg
OBJS    := $(HIP_UNITS:%=$(OBJDIR)/%.o) $(HOST_UNITS:%=$(OBJDIR)/host_%.o)
HDRS    := $(wildcard include/pm/*.h) $(wildcard $(PKG)/csrc/*.hpp) $(wildcard $(PKG)/host/*.hpp)

all: $(LIB) oracle

$(OBJDIR)/%.o: $(PKG)/csrc/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -Iinclude -I$(PKG)/csrc -c -o $@ $<

$(OBJDIR)/host_%.o: $(PKG)/host/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -Iinclude -I$(PKG)/host -c -o $@ $<

$(LIB): $(OBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,-soname,libvehicle_pm_gpu.so -o $@ $(OBJS) -lz

oracle:
	$(MAKE) -C oracle

# The tuning build: the same sources with -DPM_TUNING, in which the A/B knobs of pm_tune.hpp are read from the
# environment.  Not part of `all`; tools/ load it through PM_LIB.  The shipped library reads no environment variable.
# TUNE_DEFS: extra -D switches of an experiment (e.g. -DPL_EARLY_EXIT=1); `rm -rf $(PKG)/build/tuning` between experiments
TUNE_DEFS ?=
TOBJDIR := $(PKG)/build/tuning
TLIB    := $(PKG)/lib/libvehicle_pm_gpu_tuning.so
TOBJS   := $(HIP_UNITS:%=$(TOBJDIR)/%.o) $(HOST_UNITS:%=$(OBJDIR)/host_%.o)
$(TOBJDIR)/%.o: $(PKG)/csrc/%.hip $(HDRS)
	@mkdir -p $(TOBJDIR)
	$(HIPCC) $(HIPFLAGS) -DPM_TUNING $(TUNE_DEFS) -Iinclude -I$(PKG)/csrc -c -o $@ $<
$(TLIB): $(TOBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(TOBJS) -lz
tuning: $(TLIB)

clean:
	rm -rf $(OBJDIR) $(LIB) $(TLIB); $(MAKE) -C oracle clean
.PHONY: all oracle clean tuning
