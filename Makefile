# Builds the C-ABI library without Python (the same command dvo_slam_amd/_build.py runs) and the oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
SRC   := dvo_slam_amd/csrc
LIB   := dvo_slam_amd/libdvo_amd.so
FLAGS := $(shell python3 dvo_slam_amd/_build.py --print-flags)
BUILD_ID := $(shell python3 dvo_slam_amd/_build.py --print-id)

all: $(LIB)

CPP := $(SRC)/dvo_kernels.hip $(SRC)/dvo_pyramid.cpp $(SRC)/dvo_tracker.cpp $(SRC)/dvo_sharded.cpp $(SRC)/dvo_probes.cpp \
       $(SRC)/dvo_validator.cpp $(SRC)/dvo_frontend.cpp $(SRC)/dvo_tum.cpp

$(LIB): $(CPP) $(SRC)/dvo_types.h $(SRC)/dvo_internal.h $(SRC)/se3.h include/dvo_amd.h include/dvo_amd_debug.h
	$(HIPCC) $(FLAGS) '-DDVO_AMD_BUILD_ID="$(BUILD_ID)"' -x hip $(CPP) -lz -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB)

.PHONY: all oracle clean
