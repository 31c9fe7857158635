# Builds libo3dr.so (HIP, gfx950) and the CPU oracle.  `python __graft_entry__.py` does the same.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := online_3d_reconstruction_amd/csrc
LIBDIR   := online_3d_reconstruction_amd/lib
# -ffp-contract=off: the reference arithmetic never fuses a multiply-add (DESIGN.md "Numerics")
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result

all: $(LIBDIR)/libo3dr.so oracle

$(LIBDIR)/libo3dr.so: $(CSRC)/o3dr_kernels.hip $(CSRC)/o3dr_api.hip $(CSRC)/o3dr_device.h $(CSRC)/o3dr_profile.h include/o3dr.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/o3dr_kernels.hip $(CSRC)/o3dr_api.hip -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIBDIR) oracle/_build
.PHONY: all oracle clean
