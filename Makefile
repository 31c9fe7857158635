# Builds libo3dr.so (HIP, gfx950) and the CPU oracle.  `python __graft_entry__.py` does the same.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := online_3d_reconstruction_amd/csrc
LIBDIR   := online_3d_reconstruction_amd/lib
# -ffp-contract=off: the reference arithmetic never fuses a multiply-add (DESIGN.md "Numerics")
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result

BINDIR   := online_3d_reconstruction_amd/bin
HOST     := online_3d_reconstruction_amd/host

all: $(LIBDIR)/libo3dr.so $(BINDIR)/pose oracle

$(LIBDIR)/libo3dr.so: $(CSRC)/o3dr_kernels.hip $(wildcard $(CSRC)/kernels/*.inc) $(CSRC)/o3dr_api.hip $(CSRC)/o3dr_device.h $(CSRC)/o3dr_profile.h include/o3dr.h include/o3dr_testing.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/o3dr_kernels.hip $(CSRC)/o3dr_api.hip -o $@

# C++ host mirror of the reference's Pose class + CLI (g++; links the C ABI only)
$(BINDIR)/pose: $(HOST)/pose_main.cpp $(HOST)/pose.cpp $(HOST)/png_io.cpp $(HOST)/ply_io.cpp $(HOST)/o3dr_host.h include/o3dr.h $(LIBDIR)/libo3dr.so
	@mkdir -p $(BINDIR)
	g++ -std=c++17 -O2 -Wall -Wextra -o $@ $(HOST)/pose_main.cpp $(HOST)/pose.cpp $(HOST)/png_io.cpp $(HOST)/ply_io.cpp \
	    -L$(LIBDIR) -lo3dr -lz -lpthread -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath-link,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) $(BINDIR) oracle/_build
.PHONY: all oracle clean
