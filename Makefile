# Build of the MI355X (gfx950) celerite hot path.
#
#   make            -> celerite_amd/libcelerite_hip.so   (HIP kernels + C ABI, include/celerite_hip.h)
#                      celerite_amd/solver.<ext>.so      (pybind11 module `celerite_amd.solver`)
#                      oracle/libcelerite_ref.so         (CPU oracle; test infrastructure only)
#
# hipcc cross-compiles gfx950 code objects without a GPU.  The built .so files are
# git-ignored but travel to the GPU box with the source snapshot.
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
PYTHON     ?= python3
ARCH       ?= gfx950
HIPFLAGS   ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function
CXXFLAGS   ?= -O2 -std=c++17 -fPIC -Wall
EXT_SUFFIX := $(shell $(PYTHON) -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PY_INC     := $(shell $(PYTHON) -m pybind11 --includes)

CSRC   := celerite_amd/csrc
BUILD  := build
LIB    := celerite_amd/libcelerite_hip.so
PYMOD  := celerite_amd/solver$(EXT_SUFFIX)
ORACLE := oracle/libcelerite_ref.so

HIP_SRCS := api_misc api_solver api_batch api_grad api_kernels series_io small_kernels generic_kernels wide_kernels wide_scan32 wide64_kernels wide_prefix_scan grad_kernels grad_any_kernels wide_grad_kernels sweep_kernels wsweep_kernels huge_kernels rows_kernels bigsweep_kernels carma batch_w1 batch_w2 batch_w3 batch_w4 batch_w5 batch_w6 batch_w7 batch_w8 batch_split7 batch_split8
HIP_OBJS := $(addprefix $(BUILD)/,$(addsuffix .o,$(HIP_SRCS)))
HDRS     := $(CSRC)/api_internal.h $(CSRC)/clr_series_io.h $(CSRC)/clr_carma.h $(CSRC)/clr_small.h $(CSRC)/clr_prefix_kernels.h $(CSRC)/clr_grad_core.h $(CSRC)/clr_grad_kernels.h $(CSRC)/clr_core.h $(CSRC)/clr_wide.h $(CSRC)/clr_batch_kernels.h $(CSRC)/clr_split_kernels.h $(CSRC)/clr_generic_kernels.h $(CSRC)/clr_group_hooks.h $(CSRC)/clr_bsolve_kernels.h $(CSRC)/clr_bdotl_kernels.h $(CSRC)/clr_bdot_kernels.h $(CSRC)/clr_options.h include/celerite_hip.h include/celerite_hip_debug.h

all: $(LIB) $(PYMOD) $(ORACLE)

$(BUILD):
	mkdir -p $(BUILD)

$(BUILD)/%.o: $(CSRC)/%.hip $(HDRS) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# Translation units whose kernels gain from LLVM's max-ILP scheduling strategy (measured per unit, A/B builds in one GPU
# call: profiles/r06zj_wide_sched_ab.txt, r06zo_ilp_other_units.txt -- the width-32 lazy summarize 9.5 -> 9.1 ms, the chunk-wise
# tangent kernels 114 -> 91 ms at width 64, the row-distributed factorisation 1.70 -> 1.60 us per sample at width 128).  The
# other units lose or do not move under it (the width-8 prefix 0.151 -> 0.167 ms, wide_correct_kernel 0.36 -> 0.48 ms, the
# width-64 summarize 41.7 -> 45.0 ms; the headline summarize, the one-launch kernel, the wide walk: neutral).
ILP_SRCS := wide_scan32 grad_kernels rows_kernels
ILP_OBJS := $(addprefix $(BUILD)/,$(addsuffix .o,$(ILP_SRCS)))
$(ILP_OBJS): $(BUILD)/%.o: $(CSRC)/%.hip $(CSRC)/wide_kernels.hip $(HDRS) | $(BUILD)
	$(HIPCC) $(HIPFLAGS) -mllvm -amdgpu-sched-strategy=max-ilp -c $< -o $@

$(BUILD)/host_helpers.o: $(CSRC)/host_helpers.cpp include/celerite_hip.h | $(BUILD)
	$(CXX) $(CXXFLAGS) -c $< -o $@

$(BUILD)/sharded.o: $(CSRC)/sharded.cpp $(CSRC)/clr_group_hooks.h include/celerite_hip.h | $(BUILD)
	$(CXX) $(CXXFLAGS) -pthread -c $< -o $@

$(LIB): $(HIP_OBJS) $(BUILD)/host_helpers.o $(BUILD)/sharded.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -pthread -o $@ $^

$(PYMOD): $(CSRC)/solver_pybind.cpp include/celerite_hip.h $(LIB)
	$(CXX) $(CXXFLAGS) $(PY_INC) -shared $< -o $@ -Lcelerite_amd -lcelerite_hip -Wl,-rpath,'$$ORIGIN'

$(ORACLE): oracle/celerite_ref.c oracle/celerite_ref_quad.c oracle/celerite_ref_loops.inc oracle/celerite_ref.h
	$(MAKE) -C oracle

clean:
	rm -rf $(BUILD) $(LIB) $(PYMOD) $(ORACLE)

.PHONY: all clean
