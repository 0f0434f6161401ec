# Builds everything in-tree: the HIP engine (C ABI), the host C++ layer + CLI,
# and the test-only CPU oracle.  hipcc cross-compiles for gfx950 without a GPU.
HIPCC    ?= hipcc
ARCH     ?= gfx950
PKG      := linear-programming_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
BINDIR   := $(PKG)/bin
# -ffp-contract=off: the reference build has no FMA; the pivot sequence depends on it.
HIPFLAGS := -O3 -std=c++17 -ffp-contract=off --offload-arch=$(ARCH) -fPIC -Iinclude -Wall -Wno-unused-result
CXXFLAGS := -O2 -std=c++17 -ffp-contract=off -fPIC -Iinclude -I$(CSRC)/host -Wall

all: engine host oracle

engine: $(LIBDIR)/liblpamd.so
$(LIBDIR)/liblpamd.so: $(CSRC)/lpamd_engine.hip $(wildcard $(CSRC)/*.hip.h) include/lpamd.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/lpamd_engine.hip

HOST_SRCS := $(wildcard $(CSRC)/host/*.cc)
host: engine
	@if [ -n "$(HOST_SRCS)" ]; then $(MAKE) --no-print-directory host-bins; fi

host-bins: $(BINDIR)/solver $(BINDIR)/lp_host_probe $(BINDIR)/lp_host_selftest
HOST_LIB_SRCS := $(filter-out %_main.cc,$(HOST_SRCS))
$(BINDIR)/solver: $(HOST_SRCS) $(wildcard $(CSRC)/host/*.h) $(LIBDIR)/liblpamd.so
	@mkdir -p $(BINDIR)
	g++ $(CXXFLAGS) -o $@ $(HOST_LIB_SRCS) $(CSRC)/host/solver_main.cc -L$(LIBDIR) -llpamd -Wl,-rpath,'$$ORIGIN/../lib'
$(BINDIR)/lp_host_probe: $(HOST_SRCS) $(wildcard $(CSRC)/host/*.h) $(LIBDIR)/liblpamd.so
	@mkdir -p $(BINDIR)
	g++ $(CXXFLAGS) -o $@ $(HOST_LIB_SRCS) $(CSRC)/host/host_probe_main.cc -L$(LIBDIR) -llpamd -Wl,-rpath,'$$ORIGIN/../lib'

$(BINDIR)/lp_host_selftest: $(HOST_SRCS) $(wildcard $(CSRC)/host/*.h) $(LIBDIR)/liblpamd.so
	@mkdir -p $(BINDIR)
	g++ $(CXXFLAGS) -o $@ $(HOST_LIB_SRCS) $(CSRC)/host/host_selftest_main.cc -L$(LIBDIR) -llpamd -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle
	gcc -O1 -g -shared -fPIC -o tests/native/libsegvinfo.so tests/native/segv_info.c   # test-only: what a fatal signal says (tests/conftest.py)

# experiments only: the engine with wall-clock checkpoints in the set-up kernels (scripts/setup_timeline.py)
dbg-clock:
	@mkdir -p bin_scratch
	$(HIPCC) $(HIPFLAGS) -DLPAMD_DBG_CLOCK -shared -o bin_scratch/liblpamd_dbgclock.so $(CSRC)/lpamd_engine.hip

# experiments only: two processes on one GPU trade tagged words through IPC-mapped uncached memory (the exchange inside the launches)
link-probe:
	@mkdir -p bin_scratch
	$(HIPCC) -O2 --offload-arch=gfx950 -o bin_scratch/link_pingpong scripts/link_pingpong.hip

# experiments only: the engine with index checks in the kernels (tests/conftest.py reads the counters after every test
# when LPAMD_LIBRARY points at this build)
dbg-checks:
	@mkdir -p bin_scratch
	$(HIPCC) $(HIPFLAGS) -DLPAMD_DEVICE_CHECKS -shared -o bin_scratch/liblpamd_checks.so $(CSRC)/lpamd_engine.hip

clean:
	rm -rf $(LIBDIR) $(BINDIR) build
	$(MAKE) -C oracle clean
.PHONY: all engine host host-bins oracle clean dbg-clock dbg-checks link-probe
