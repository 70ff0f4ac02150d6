# Builds every native piece in-tree:
#   raytracer_amd/librtx_hip.so    HIP kernels + C ABI (include/rtx.h), gfx950
#   raytracer_amd/libvalu_peak.so  VALU issue-rate microbenchmark (diagnostic, used by bench.py)
#   raytracer_amd/librtx_host.so   C++ host side: reference-shaped API, BVH builder, flattener
#   oracle/liboracle.so (+ oracle/_ref/libref.so where /root/reference exists)   test infrastructure
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
# -ffp-contract=off: no fused multiply-add on host or device; results must be
# bit-identical to the reference's plain fp32 arithmetic
# -fno-slp-vectorize: the compiler's own pairing of fp32 operations into v_pk_* costs more in register
# shuffles than a packed instruction saves on gfx950 (4.3 cycles against 2 x 2.6): frames 3-5 % faster without
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -Wno-pass-failed -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -Iraytracer_amd/csrc
CXXFLAGS ?= -std=c++17 -O2 -fPIC -ffp-contract=off -Wall -Wextra -Iinclude -Iraytracer_amd/host -pthread

HOST_SRC := $(wildcard raytracer_amd/host/*.cpp)
HOST_HDR := $(wildcard raytracer_amd/host/*.h) include/rtx.h

.PHONY: all hip host examples oracle clean asan asan-test
all: hip host examples oracle

hip: raytracer_amd/librtx_hip.so raytracer_amd/libvalu_peak.so
# diagnostic beside the product library: what a SIMD's vector ALU issues per cycle (bench.py's VALU roofline)
raytracer_amd/libvalu_peak.so: tools/valu_peak.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -shared -o $@ tools/valu_peak.hip
raytracer_amd/librtx_hip.so: raytracer_amd/csrc/rtx_hip.hip raytracer_amd/csrc/rtx_device.hpp raytracer_amd/csrc/rtx_build.hpp include/rtx.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ raytracer_amd/csrc/rtx_hip.hip

host: raytracer_amd/librtx_host.so
raytracer_amd/librtx_host.so: $(HOST_SRC) $(HOST_HDR) raytracer_amd/librtx_hip.so
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRC) -Lraytracer_amd -lrtx_hip -lz -Wl,-rpath,'$$ORIGIN'

# headless example: OBJ in, BMP out, through the reference-shaped C++ API
examples: raytracer_amd/render_obj raytracer_amd/render_cli
raytracer_amd/render_obj: examples/render_obj.cpp raytracer_amd/librtx_host.so
	$(CXX) $(CXXFLAGS) -o $@ examples/render_obj.cpp -Lraytracer_amd -lrtx_host -lrtx_hip -Wl,-rpath,'$$ORIGIN'
# the reference's command-line renderer (TOML configuration, one bitmap per camera) on the GPU path
raytracer_amd/render_cli: examples/render_cli.cpp raytracer_amd/librtx_host.so
	$(CXX) $(CXXFLAGS) -o $@ examples/render_cli.cpp -Lraytracer_amd -lrtx_host -lrtx_hip -Wl,-rpath,'$$ORIGIN'

# oracle/_ref/dropin_check links the host library, so that one comes first
oracle: raytracer_amd/librtx_host.so
	$(MAKE) -C oracle all

# Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; the counterpart of the
# reference's ENABLE_SANITIZER_ADDRESS / _UNDEFINED_BEHAVIOR options,
# framework/cmake/Sanitizers.cmake:7-26): the host library (configuration / OBJ / MTL / PNG / JPEG /
# BMP readers, BVH builder, flattener, the reference-shaped API) and the CPU restatement, built into
# build/asan/ beside the product libraries; `make asan-test` runs the CPU test suite on them.
SAN_DIR  := build/asan
SANFLAGS := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
asan: $(SAN_DIR)/librtx_host.so $(SAN_DIR)/liboracle.so
$(SAN_DIR)/librtx_host.so: $(HOST_SRC) $(HOST_HDR) raytracer_amd/librtx_hip.so
	mkdir -p $(SAN_DIR)
	$(CXX) $(filter-out -O2,$(CXXFLAGS)) $(SANFLAGS) -shared -o $@ $(HOST_SRC) -Lraytracer_amd -lrtx_hip -lz -Wl,-rpath,'$(CURDIR)/raytracer_amd'
$(SAN_DIR)/liboracle.so: oracle/rt_oracle.cpp include/rtx.h
	mkdir -p $(SAN_DIR)
	$(CXX) -std=c++20 -fopenmp -fPIC -ffp-contract=off -Wall -Wextra $(SANFLAGS) -shared -Iinclude -o $@ oracle/rt_oracle.cpp
# (the interpreter is not a sanitizer build: the runtimes are loaded ahead of it; -s because a report
# printed while pytest holds the output would die with the process; leak checking off,
# CPython never frees what it interns)
asan-test: asan
	LD_PRELOAD="$$($(CXX) -print-file-name=libasan.so) $$($(CXX) -print-file-name=libubsan.so)$${LD_PRELOAD:+ $$LD_PRELOAD}" \
	ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
	RTX_HOST_SO=$(CURDIR)/$(SAN_DIR)/librtx_host.so RTX_ORACLE_SO=$(CURDIR)/$(SAN_DIR)/liboracle.so \
	python -m pytest tests -x -q -s -m "not gpu" -p no:cacheprovider

clean:
	rm -rf build/asan
	rm -f raytracer_amd/librtx_hip.so raytracer_amd/librtx_host.so raytracer_amd/libvalu_peak.so raytracer_amd/render_obj raytracer_amd/render_cli
	$(MAKE) -C oracle clean
